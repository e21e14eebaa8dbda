"""Mask2Former's pixel decoder shell around the MSDeformAttn encoder.

Host-side mirror of ``MSDeformAttnPixelDecoder`` (lib/network/mask2former/modeling/pixel_decoder/msdeformattn.py:164-358):
same constructor keywords, same parameter names (``input_proj.{i}.0/1``, ``transformer.*``, ``adapter_1`` / ``layer_1``
with their ``.norm`` children as detectron2's Conv2d wrapper names them, ``mask_features``), same
``forward_features(features) -> (mask_features, out[0], multi_scale_features)`` with NCHW maps at the boundary.
detectron2 is not a dependency: ``input_shape`` values only need ``.channels`` and ``.stride`` (``ShapeSpec`` below).

What runs (all in libmss_hip.so, NHWC inside):
  * input_proj: 1x1 conv (+bias in the GEMM epilogue) -> GroupNorm(32) whose output is written straight into the
    encoder's token buffer [N, sum(HW), 256] at the level's offset -- the reference's flatten/transpose/cat copies
    (msdeformattn.py:66-79) do not exist;
  * the encoder (msdeformattn_encoder.py) on that buffer;
  * the FPN top-down step for the stride-4 level: lateral 1x1 conv -> GroupNorm, bilinear(align_corners=False) + add in
    one kernel reading the finest encoder level in place, 3x3 conv (Winograd / implicit GEMM) -> GroupNorm+ReLU;
  * mask_features 1x1 conv; NHWC -> NCHW only for the five returned maps.
Trainable: the shell is two autograd nodes around the (already differentiable) encoder -- `_InputProjFn` (the three input
projections -> token buffer) and `_FpnFn` (FPN step + mask_features + the returned maps) -- whose backward runs on the
same kernels: 1x1 / 3x3 weight and data gradients (conv_wgrad / gemm_nt / Winograd), GroupNorm backward, the transpose of
the bilinear up-sampling, deterministic bias / affine reductions. Supported for the Mask2Former configuration (three
transformer levels, one FPN level, norm="GN"); other layouts run forward-only.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from . import kernels as K
from .msdeformattn_encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine

ShapeSpec = namedtuple("ShapeSpec", ["channels", "stride"])


class _NormConv2d(nn.Conv2d):
    """Parameter container with the layout of detectron2.layers.Conv2d: an optional ``norm`` child and an activation
    applied after it (the arithmetic is done by the HIP kernels, never by this module's forward)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation


def _get_norm(norm, channels):
    if norm in (None, ""):
        return None
    if norm == "GN":
        return nn.GroupNorm(32, channels)                 # detectron2 get_norm("GN", c)
    raise NotImplementedError(f"norm={norm!r}: the HIP path implements GroupNorm (the Mask2Former configs' 'GN') or none")


def _xavier_fill(conv):
    """fvcore c2_xavier_fill: kaiming_uniform_(a=1) weight, zero bias."""
    nn.init.kaiming_uniform_(conv.weight, a=1)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, input_shape, *, transformer_dropout, transformer_nheads, transformer_dim_feedforward,
                 transformer_enc_layers, conv_dim, mask_dim, norm=None, transformer_in_features, common_stride):
        super().__init__()
        tshape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        items = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, _ in items]                       # "res2" .. "res5"
        self.feature_strides = [v.stride for _, v in items]
        self.feature_channels = [v.channels for _, v in items]
        titems = sorted(tshape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, _ in titems]
        t_channels = [v.channels for _, v in titems]
        self.transformer_feature_strides = [v.stride for _, v in titems]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        if self.transformer_num_feature_levels > 1:
            projs = [nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in t_channels[::-1]]
        else:
            projs = [nn.Sequential(nn.Conv2d(t_channels[-1], conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim))]
        self.input_proj = nn.ModuleList(projs)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = _NormConv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        _xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral = _NormConv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=_get_norm(norm, conv_dim))
            output = _NormConv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=_get_norm(norm, conv_dim), activation=torch.relu)
            _xavier_fill(lateral)
            _xavier_fill(output)
            self.add_module(f"adapter_{idx + 1}", lateral)
            self.add_module(f"layer_{idx + 1}", output)
            lateral_convs.append(lateral)
            output_convs.append(output)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    # ---- pieces ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _bias_affine(conv):
        if conv.bias is None:
            return None
        return (K.ones(conv.bias.numel(), conv.bias.device), conv.bias.detach())

    def _conv1x1(self, x_act, conv):
        return K.conv2d(x_act, K.packed(conv.weight), out_affine=self._bias_affine(conv))

    def _norm_act(self, y, conv):
        """detectron2 Conv2d.forward after the convolution: norm, then activation."""
        if conv.norm is not None:
            return K.groupnorm(y, conv.norm, relu=conv.activation is not None)
        if conv.activation is not None:
            raise NotImplementedError("activation without a norm does not occur in the pixel decoder")
        return y

    # ---- reference API ---------------------------------------------------------------------------------------------------
    def _trainable_layout(self):
        return (self.transformer_num_feature_levels == 3 and self.num_fpn_levels == 1 and self.lateral_convs[0].norm is not None
                and self.lateral_convs[0].bias is None)

    def forward_features(self, features):
        names = self.transformer_in_features[::-1]                      # res5 -> res3 (msdeformattn.py:319)
        xs = [features[f].float() for f in names]
        if not xs[0].is_cuda:
            raise RuntimeError("MSDeformAttnPixelDecoder (multishiftseg_amd) runs on an MI355X only; there is no CPU path")
        want_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                                 any(t.requires_grad for t in features.values()))
        if not want_grad:
            with torch.no_grad():
                return self._forward_features(features, xs)
        if not self._trainable_layout():
            raise NotImplementedError("MSDeformAttnPixelDecoder (multishiftseg_amd): the backward is built for three transformer "
                                      "levels + one GroupNorm FPN level (the Mask2Former configs); call under torch.no_grad()")
        shapes = [(x.shape[2], x.shape[3]) for x in xs]
        ip = [m for proj in self.input_proj for m in (proj[0].weight, proj[0].bias, proj[1].weight, proj[1].bias)]
        tokens = _InputProjFn.apply(self, *xs, *ip)
        memory, _, _ = self.transformer.forward_tokens(tokens, None, shapes, pe_layer=self.pe_layer)
        lat, outc = self.lateral_convs[0], self.output_convs[0]
        x2 = features[self.in_features[0]].float()
        mask, m0, m1, m2 = _FpnFn.apply(self, shapes, memory, x2, lat.weight, lat.norm.weight, lat.norm.bias, outc.weight,
                                        outc.norm.weight, outc.norm.bias, self.mask_features.weight, self.mask_features.bias)
        return mask, m0, [m0, m1, m2]

    def _forward_features(self, features, xs):
        N = xs[0].shape[0]
        shapes = [(x.shape[2], x.shape[3]) for x in xs]
        starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])]).tolist()
        S, C = starts[-1], self.input_proj[0][1].num_channels
        dev = xs[0].device
        tokens = torch.empty((N, S, C), device=dev, dtype=torch.float32)
        for idx, x in enumerate(xs):
            conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
            y = self._conv1x1(K.nchw_to_act(x), conv)
            # GroupNorm output goes straight to rows [start, start + H*W) of every sample of the token buffer
            K.groupnorm(y, gn, out=tokens[0, starts[idx]:], out_sample_stride=S * C, out_ld=C)
        memory, spatial_shapes, level_start_index = self.transformer.forward_tokens(tokens, None, shapes, pe_layer=self.pe_layer)
        levels = [K.TokenLevel(memory, starts[i], *shapes[i]) for i in range(len(shapes))]
        out = list(levels)
        for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
            x = features[f].float()
            lateral, output = self.lateral_convs[idx], self.output_convs[idx]
            cur = self._norm_act(self._conv1x1(K.nchw_to_act(x), lateral), lateral)
            y = K.upsample_bilinear_add(out[-1], cur)
            y = K.conv3x3(y, output.weight) if output.bias is None else \
                K.conv2d(y, K.packed(output.weight), pad=1, out_affine=self._bias_affine(output))
            out.append(self._norm_act(y, output))
        multi_scale = [K.nhwc_to_nchw(o) for o in out[:self.maskformer_num_feature_levels]]
        last = out[-1]
        if isinstance(last, K.TokenLevel):      # no FPN level: mask_features reads the finest encoder level
            last = K.Act(K.nhwc_to_nchw(last).permute(0, 2, 3, 1).contiguous())
        mask = K.nhwc_to_nchw(self._conv1x1(last, self.mask_features))
        return mask, multi_scale[0], multi_scale

    def forward(self, features, targets=None):
        raise NotImplementedError("the reference only calls forward_features (msdeformattn.py:314)")


def _conv1x1_backward(x_act, dy, conv, need_w, need_b, need_x):
    """Gradients of y = conv1x1(x) (+bias) given dy (Act): weight [K,C,1,1], bias [K], input (Act) -- each only on request."""
    k, c = conv.weight.shape[0], conv.weight.shape[1]
    dw = K.conv2d_wgrad(x_act, dy, k, c, 1, 1) if need_w else None
    db = K.colsum(dy).sum(0) if need_b else None
    dx = K.conv2d(dy, K.packed(conv.weight, flip=True)) if need_x else None
    return dw, db, dx


class _InputProjFn(torch.autograd.Function):
    """features (res5, res4, res3; NCHW) -> 1x1 conv + bias -> GroupNorm(32) -> token buffer [N, sum(HW), C]
    (msdeformattn.py:215-219,319-323 and the flatten / cat of :66-79)."""

    @staticmethod
    def forward(ctx, dec, x0, x1, x2, *params):
        xs = (x0, x1, x2)
        N = x0.shape[0]
        shapes = [(x.shape[2], x.shape[3]) for x in xs]
        starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])]).tolist()
        S, C = starts[-1], dec.input_proj[0][1].num_channels
        tokens = torch.empty((N, S, C), device=x0.device, dtype=torch.float32)
        saved = []
        for idx, x in enumerate(xs):
            conv, gn = dec.input_proj[idx][0], dec.input_proj[idx][1]
            xa = K.nchw_to_act(x)
            y = dec._conv1x1(xa, conv)
            _, stat = K.groupnorm(y, gn, out=tokens[0, starts[idx]:], out_sample_stride=S * C, out_ld=C, want_stat=True)
            saved.append((xa, y, stat))
        ctx.dec, ctx.saved, ctx.starts, ctx.shapes = dec, saved, starts, shapes
        ctx.in_channels = [x.shape[1] for x in xs]
        return tokens

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_tokens):
        dec, starts = ctx.dec, ctx.starts
        g_tokens = g_tokens.contiguous()
        N, S, C = g_tokens.shape
        need = ctx.needs_input_grad            # (dec, x0, x1, x2, w0, b0, gw0, gb0, w1, ...)
        gx = [None, None, None]
        gp = []
        for idx in range(3):
            conv, gn = dec.input_proj[idx][0], dec.input_proj[idx][1]
            xa, y, stat = ctx.saved[idx]
            gptr = ctypes.c_void_p(g_tokens.data_ptr() + 4 * starts[idx] * C)
            dy, dgam, dbet = K.groupnorm_backward(gptr, C, S * C, y, gn, stat)
            base = 4 + 4 * idx
            dw, db, dx = _conv1x1_backward(xa, dy, conv, need[base], need[base + 1], need[1 + idx])
            if dx is not None:
                gx[idx] = K.nhwc_to_nchw(dx)[:, :ctx.in_channels[idx]].contiguous()
            gp += [dw, db, dgam if need[base + 2] else None, dbet if need[base + 3] else None]
        ctx.saved = None
        return (None, *gx, *gp)


class _FpnFn(torch.autograd.Function):
    """encoder memory + res2 -> the returned maps (msdeformattn.py:326-358): level split, lateral 1x1 conv + GN, bilinear
    (align_corners=False) + add, 3x3 conv + GN + ReLU, mask_features 1x1 conv, NHWC -> NCHW."""

    @staticmethod
    def forward(ctx, dec, shapes, memory, x2, lat_w, lat_gw, lat_gb, out_w, out_gw, out_gb, mask_w, mask_b):
        lat, outc = dec.lateral_convs[0], dec.output_convs[0]
        starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])]).tolist()
        memory = memory.contiguous()
        levels = [K.TokenLevel(memory, starts[i], *shapes[i]) for i in range(3)]
        xa = K.nchw_to_act(x2)
        lat_y = dec._conv1x1(xa, lat)
        cur, lat_stat = K.groupnorm(lat_y, lat.norm, want_stat=True)
        y = K.upsample_bilinear_add(levels[-1], cur)
        z = K.conv3x3(y, outc.weight)
        o, out_stat = K.groupnorm(z, outc.norm, relu=True, want_stat=True)
        mask = K.nhwc_to_nchw(dec._conv1x1(o, dec.mask_features))
        ms = [K.nhwc_to_nchw(l) for l in levels]
        ctx.dec, ctx.shapes, ctx.starts = dec, shapes, starts
        ctx.saved = (xa, lat_y, lat_stat, y, z, out_stat, o)
        ctx.mem_shape, ctx.in_channels = tuple(memory.shape), x2.shape[1]
        return mask, ms[0], ms[1], ms[2]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_mask, g0, g1, g2):
        dec, shapes, starts = ctx.dec, ctx.shapes, ctx.starts
        lat, outc = dec.lateral_convs[0], dec.output_convs[0]
        xa, lat_y, lat_stat, y, z, out_stat, o = ctx.saved
        need = ctx.needs_input_grad     # (dec, shapes, memory, x2, lat_w, lat_gw, lat_gb, out_w, out_gw, out_gb, mask_w, mask_b)
        N, S, C = ctx.mem_shape
        dev = g_mask.device
        g_mem = torch.empty((N, S, C), device=dev, dtype=torch.float32)
        for i, g in enumerate((g0, g1, g2)):          # the three returned levels are views of the memory rows
            gptr = ctypes.c_void_p(g_mem.data_ptr() + 4 * starts[i] * C)
            if g is None:
                g_mem[:, starts[i]:starts[i + 1]].zero_()
            else:
                K.nchw_into_rows(g, gptr, C, S * C)
        # mask_features
        dm = K.nchw_to_act(g_mask, Cp=g_mask.shape[1])
        d_mask_w, d_mask_b, do = _conv1x1_backward(o, dm, dec.mask_features, need[10], need[11], True)
        # output conv: GroupNorm + ReLU, then the 3x3 convolution
        dz, d_out_gw, d_out_gb = K.groupnorm_backward(do.ptr, do.ld, do.H * do.W * do.ld, z, outc.norm, out_stat, relu=True)
        # (tile attribution, tools/decoder_grad_err.py at 704 x 704: this gradient is 3.5e-3 relative L2 from the reference's on the
        # policy's F(6x6) tiles, 1.6e-4 with every 3x3 product on F(4x4), 2e-6 direct -- the reference's own fp32-vs-fp64 noise on it
        # is 2.1e-3. Capping only THIS product at F(4x4) changes nothing (measured): the difference enters through the forward's
        # z -> GroupNorm + ReLU mask, not through the weight-gradient transform.)
        d_out_w = K.conv3x3_wgrad(y, dz, outc.weight.shape[0], outc.weight.shape[1]) if need[7] else None
        dyy = K.conv3x3(dz, outc.weight, flip=True)
        # y = cur + up(level 2): the lateral branch gets dy itself, the finest encoder level its bilinear transpose (added
        # to the gradient that came in through the returned map)
        h2, w2 = shapes[2]
        K.upsample_bilinear_bwd(dyy, ctypes.c_void_p(g_mem.data_ptr() + 4 * starts[2] * C), C, S * C, h2, w2, accumulate=True)
        dlat, d_lat_gw, d_lat_gb = K.groupnorm_backward(dyy.ptr, dyy.ld, dyy.H * dyy.W * dyy.ld, lat_y, lat.norm, lat_stat)
        d_lat_w, _, dx2 = _conv1x1_backward(xa, dlat, lat, need[4], False, need[3])
        gx2 = K.nhwc_to_nchw(dx2)[:, :ctx.in_channels].contiguous() if dx2 is not None else None
        ctx.saved = None
        return (None, None, g_mem, gx2, d_lat_w, d_lat_gw if need[5] else None, d_lat_gb if need[6] else None, d_out_w,
                d_out_gw if need[8] else None, d_out_gb if need[9] else None, d_mask_w, d_mask_b)


class GraphedFeatures:
    """`forward_features` of a frozen pixel decoder (inference: test_m2f.py's loop) captured ONCE into a hipGraph and replayed.
    At N = 1 the eager forward is bound by the host's launch path (~80 kernel launches for 6 encoder layers, 4.4 ms at
    704x704 where the kernels need ~1.5 ms): one graph launch removes it. Fixed feature shapes; outputs are the graph's
    static tensors (copy them if they must survive the next call). Like trainer.GraphedEval it owns the packed weight copies
    the replay reads and re-captures when any parameter's (version, data_ptr) changes."""

    def __init__(self, decoder, features, warmup=2):
        self.dec = decoder
        self.static_in = {k: torch.zeros_like(v, dtype=torch.float32, device="cuda") for k, v in features.items()}
        self.captures = 0
        self._capture(warmup)

    def _signature(self):
        return [(p._version, p.data_ptr()) for p in self.dec.parameters()]

    def _capture(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # warm-up off the capture: packs weights, fills the index caches
            for _ in range(warmup):
                self.dec.forward_features(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self.dec.forward_features(self.static_in)
        self._keep = [dict(p.__dict__.get("_mss_packed", {})) for p in self.dec.parameters()]
        # ... and the level-size / start-index / valid-ratio tensors the sampler kernels read by pointer
        self._keep.append([dict(m.__dict__.get("_index_cache", {})) for m in self.dec.modules()])
        self._sig = self._signature()
        self.captures += 1

    def __call__(self, features):
        for k, v in features.items():
            if tuple(v.shape) != tuple(self.static_in[k].shape):
                raise ValueError(f"GraphedFeatures was captured for {k} {tuple(self.static_in[k].shape)}, got {tuple(v.shape)}")
        if self._signature() != self._sig:
            self._capture(1)
        for k, v in features.items():
            self.static_in[k].copy_(v)
        self.graph.replay()
        return self.static_out
