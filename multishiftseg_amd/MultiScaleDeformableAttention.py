"""Drop-in for the reference's compiled ``MultiScaleDeformableAttention`` extension module.

Same two functions, same argument order and meaning, same preconditions as
ops/src/vision.cpp:18-21 -> ms_deform_attn.h:25-66 -> cuda/ms_deform_attn_cuda.cu:25-157, backed by
mss_msda_{forward,backward}_{f32,f64} in libmss_hip.so. To let reference-style code
(`import MultiScaleDeformableAttention as MSDA`) pick it up unchanged call `install()`.
"""
import ctypes
import os
import sys

import torch

from . import _lib
from ._lib import call, ptr

_HOST_SHAPES = {}      # id(tensor) -> (tensor, version, ctypes int64 array): host copies of spatial_shapes tensors


def host_shapes(spatial_shapes):
    """HOST copy of a device spatial_shapes tensor for the binned backward (its tile geometry and launch grids depend on the level sizes).
    Callers that know the sizes attach them (`t._mss_host = [(H, W), ...]`, msdeformattn_encoder.py); otherwise one
    synchronising copy per distinct tensor object, remembered while that tensor is alive and unmodified. None while a
    hipGraph is being captured (no host copy possible): the caller then takes the kernel that reads the device tensor."""
    hint = getattr(spatial_shapes, "_mss_host", None)
    if hint is not None:
        flat = [int(v) for hw in hint for v in hw]
        return (ctypes.c_int64 * len(flat))(*flat)
    ent = _HOST_SHAPES.get(id(spatial_shapes))
    if ent is not None and ent[0] is spatial_shapes and ent[1] == spatial_shapes._version:
        return ent[2]
    if torch.cuda.is_current_stream_capturing():
        return None
    flat = spatial_shapes.detach().cpu().flatten().tolist()
    arr = (ctypes.c_int64 * len(flat))(*flat)
    if len(_HOST_SHAPES) >= 32:
        _HOST_SHAPES.clear()
    _HOST_SHAPES[id(spatial_shapes)] = (spatial_shapes, spatial_shapes._version, arr)
    return arr


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, extra=()):
    tensors = dict(value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                   sampling_loc=sampling_loc, attn_weight=attn_weight, **dict(extra))
    for name, t in tensors.items():
        if not t.is_contiguous():                       # ms_deform_attn_cuda.cu:33-37,98-103
            raise RuntimeError(f"{name} tensor has to be contiguous")
    for name, t in tensors.items():
        if not t.is_cuda:                               # ms_deform_attn.h:43,65 / .cu:39-43
            raise RuntimeError("Not implemented on the CPU" if name == "value" else f"{name} must be a CUDA tensor")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64")
    if value.dtype not in (torch.float32, torch.float64):   # AT_DISPATCH_FLOATING_TYPES, .cu:69,139
        raise RuntimeError(f"ms_deform_attn: unsupported dtype {value.dtype}")
    for name in ("sampling_loc", "attn_weight") + tuple(k for k, _ in extra):
        if tensors[name].dtype != value.dtype:
            raise RuntimeError(f"{name} dtype {tensors[name].dtype} != value dtype {value.dtype}")
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if batch and batch % step != 0:                      # .cu:55-57,122-124
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    return N, S, M, D, L, Lq, P


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    N, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    out = torch.empty((N, Lq, M * D), device=value.device, dtype=value.dtype)
    sfx = "f32" if value.dtype == torch.float32 else "f64"
    call(f"mss_msda_forward_{sfx}", ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(sampling_loc),
         ptr(attn_weight), N, S, M, D, L, Lq, P, ptr(out))
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    N, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                                  extra=(("grad_output", grad_output),))
    grad_value = torch.empty_like(value)                 # zero-filled by the callee
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    sfx = "f32" if value.dtype == torch.float32 else "f64"
    # fp32 / D = 32: grad_value on the binned owner-computes path (csrc/msda.hip, round 3) whenever a host copy of the level
    # sizes is at hand; MSS_MSDA_BWD_BINNED=0: the generic scatter-add kernel (fp64 / other head dimensions; the second formulation in the tests)
    if sfx == "f32" and D == 32 and N * Lq > 0 and os.environ.get("MSS_MSDA_BWD_BINNED", "1") != "0" \
            and value.data_ptr() % 16 == 0 and grad_output.data_ptr() % 16 == 0:
        hs = host_shapes(spatial_shapes)
        nbytes = _lib.value("mss_msda_backward_workspace_bytes", hs, N, M, D, L, Lq, P) if hs is not None else 0
        if nbytes > 0:
            ws = torch.empty(nbytes, device=value.device, dtype=torch.uint8)
            call("mss_msda_backward_binned_f32", ptr(value), ptr(spatial_shapes), ptr(level_start_index), hs, ptr(sampling_loc),
                 ptr(attn_weight), ptr(grad_output), N, S, M, D, L, Lq, P, ptr(grad_value), ptr(grad_loc), ptr(grad_attn), ptr(ws),
                 nbytes)
            return [grad_value, grad_loc, grad_attn]
    call(f"mss_msda_backward_{sfx}", ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(sampling_loc),
         ptr(attn_weight), ptr(grad_output), N, S, M, D, L, Lq, P, ptr(grad_value), ptr(grad_loc), ptr(grad_attn))
    return [grad_value, grad_loc, grad_attn]


def ms_deform_attn_backward_proj(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, grad_proj, ko):
    """The op's backward with the MODULE's backward (softmax, location arithmetic: ops/modules/ms_deform_attn.py:100-109) folded into
    its gather pass: writes d(sampling offsets) into grad_proj[..., :ko] and d(attention logits) into grad_proj[..., ko:] (one
    [N, Lq, M*3*L*P] buffer: the output gradient of the merged projection) and returns grad_value -- or None where the binned path
    does not run (the caller then uses ms_deform_attn_backward + the prepare-backward kernel)."""
    N, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, 128,
                                  extra=(("grad_output", grad_output),))
    if not (value.dtype == torch.float32 and D == 32 and N * Lq > 0 and L * P <= 20 and os.environ.get("MSS_MSDA_BWD_BINNED", "1") != "0"
            and os.environ.get("MSS_MSDA_BWD_PROJ", "1") != "0" and value.data_ptr() % 16 == 0 and grad_output.data_ptr() % 16 == 0):
        return None
    hs = host_shapes(spatial_shapes)
    nbytes = _lib.value("mss_msda_backward_workspace_bytes", hs, N, M, D, L, Lq, P) if hs is not None else 0
    if nbytes <= 0:
        return None
    ld = grad_proj.shape[-1]
    assert grad_proj.is_contiguous() and ld >= ko + M * L * P and ko == M * L * P * 2
    grad_value = torch.empty_like(value)
    ws = torch.empty(nbytes, device=value.device, dtype=torch.uint8)
    call("mss_msda_backward_binned_proj_f32", ptr(value), ptr(spatial_shapes), ptr(level_start_index), hs, ptr(sampling_loc),
         ptr(attn_weight), ptr(grad_output), N, S, M, D, L, Lq, P, ptr(grad_value), ptr(grad_proj), ld,
         ctypes.c_void_p(grad_proj.data_ptr() + 4 * ko), ld, ptr(ws), nbytes)
    return grad_value


def install():
    """Register this module under the extension's name for `import MultiScaleDeformableAttention`."""
    sys.modules["MultiScaleDeformableAttention"] = sys.modules[__name__]
