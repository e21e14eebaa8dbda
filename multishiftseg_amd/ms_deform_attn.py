"""Host-side mirror of the MSDeformAttn operator stack
(lib/network/mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:32-49 and
ops/modules/ms_deform_attn.py:35-125): same class names, arguments and parameter names, on top of
the HIP op. Unlike the reference module there is no ``try/except`` around the op: a failing kernel
raises instead of silently falling back to a slow PyTorch path (ms_deform_attn.py:116-121)."""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import constant_, xavier_uniform_

from . import MultiScaleDeformableAttention as MSDA
from ._lib import call, ptr
from .linear import linear


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn,
                                                                       grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None


class _PrepareFn(Function):
    """softmax over the L*P logits + sampling locations from offsets and reference points, one HIP pass each way
    (ops/modules/ms_deform_attn.py:100-109, the 2-d reference-point branch)."""

    @staticmethod
    def forward(ctx, offsets, logits, reference_points, spatial_shapes):
        N, Lq, M, L, P, _ = offsets.shape
        offsets, logits = offsets.contiguous(), logits.contiguous()
        ref = reference_points.contiguous().float()
        loc = torch.empty_like(offsets)
        attn = torch.empty((N, Lq, M, L, P), device=offsets.device, dtype=torch.float32)
        call("mss_msda_prepare_f32", ptr(offsets), ptr(logits), ptr(ref), ptr(spatial_shapes), N, Lq, M, L, P, ptr(loc), ptr(attn))
        ctx.save_for_backward(attn, spatial_shapes)
        return loc, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, gloc, gattn):
        attn, spatial_shapes = ctx.saved_tensors
        N, Lq, M, L, P = attn.shape
        goff = torch.empty((N, Lq, M, L, P, 2), device=attn.device, dtype=torch.float32)
        glog = torch.empty((N, Lq, M, L * P), device=attn.device, dtype=torch.float32)
        call("mss_msda_prepare_backward_f32", ptr(attn), ptr(gattn.contiguous()), ptr(gloc.contiguous()), ptr(spatial_shapes),
             N, Lq, M, L, P, ptr(goff), ptr(glog))
        return goff, glog, None, None


class _FusedSampleFn(Function):
    """the op fed with the raw offsets / logits of the two Linears; softmax + location arithmetic run inside
    the sampling kernel (mss_msda_forward_fused_f32), so sampling_locations / attention_weights never touch HBM in the
    forward. The backward rebuilds them with the one-pass prepare kernel (bit-identical values), runs the op's
    backward and maps the gradients back to offsets / logits."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, offsets, logits, reference_points):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        value, offsets, logits = value.contiguous(), offsets.contiguous(), logits.contiguous()
        ref = reference_points.contiguous().float()
        out = torch.empty((N, Lq, M * D), device=value.device, dtype=torch.float32)
        call("mss_msda_forward_fused_f32", ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(offsets), ptr(logits),
             ptr(ref), N, S, M, D, L, Lq, P, ptr(out))
        ctx.save_for_backward(value, spatial_shapes, level_start_index, offsets, logits, ref)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, offsets, logits, ref = ctx.saved_tensors
        N, Lq, M, L, P, _ = offsets.shape
        loc = torch.empty_like(offsets)
        attn = torch.empty((N, Lq, M, L, P), device=offsets.device, dtype=torch.float32)
        call("mss_msda_prepare_f32", ptr(offsets), ptr(logits), ptr(ref), ptr(shapes), N, Lq, M, L, P, ptr(loc), ptr(attn))
        gvalue, gloc, gattn = MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, grad_output.contiguous(), 128)
        goff = torch.empty_like(offsets)
        glog = torch.empty((N, Lq, M, L * P), device=offsets.device, dtype=torch.float32)
        call("mss_msda_prepare_backward_f32", ptr(attn), ptr(gattn.contiguous()), ptr(gloc.contiguous()), ptr(shapes),
             N, Lq, M, L, P, ptr(goff), ptr(glog))
        return gvalue, None, None, goff, glog, None


def _is_power_of_2(n):
    if (not isinstance(n, int)) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention module (ops/modules/ms_deform_attn.py:35-125)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("d_model // n_heads is not a power of 2: the float4 fast path of the HIP kernel needs "
                          "a head dimension of 16, 32 or 64")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # ms_deform_attn.py:66-80: zero offsets weight, a ring of directions scaled by the point index as bias
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid = grid.repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        host = getattr(input_spatial_shapes, "_mss_host", None)
        if host is not None:                    # level sizes known on the host: no device read (keeps the forward capturable)
            assert sum(int(h) * int(w) for h, w in host) == Len_in
        elif not (input_flatten.is_cuda and torch.cuda.is_current_stream_capturing()):
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in      # ms_deform_attn.py:93
        value = linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        offsets = linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        weights = linear(query, self.attention_weights.weight, self.attention_weights.bias).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)
        fast = (reference_points.shape[-1] == 2 and offsets.is_cuda and offsets.dtype == torch.float32
                and self.n_levels * self.n_points <= 20 and input_spatial_shapes.dtype == torch.int64
                and not reference_points.requires_grad)
        if fast and self.d_model // self.n_heads in (16, 32, 64) and value.dtype == torch.float32 \
                and os.environ.get("MSS_MSDA_FUSED", "1") != "0":
            # one kernel: softmax + locations + sampling (no [N,Lq,M,L,P,2] / [N,Lq,M,L,P] round trip through HBM)
            output = _FusedSampleFn.apply(value, input_spatial_shapes.contiguous(), input_level_start_index.contiguous(),
                                          offsets, weights, reference_points)
            return linear(output, self.output_proj.weight, self.output_proj.bias)
        if fast:
            locations, weights = _PrepareFn.apply(offsets, weights, reference_points, input_spatial_shapes.contiguous())
        else:
            weights = F.softmax(weights, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
            if reference_points.shape[-1] == 2:
                normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
                locations = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
            elif reference_points.shape[-1] == 4:
                locations = reference_points[:, :, None, :, None, :2] \
                    + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
            else:
                raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        output = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                            locations.contiguous(), weights.contiguous(), self.im2col_step)
        return linear(output, self.output_proj.weight, self.output_proj.bias)
