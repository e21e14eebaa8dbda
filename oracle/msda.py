"""TEST INFRASTRUCTURE ONLY -- numpy restatement of multi-scale deformable attention.

Forward follows ms_deformable_im2col_gpu_kernel + ms_deform_attn_im2col_bilinear
(lib/network/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304,
:38-89); backward follows ms_deform_attn_col2im_bilinear (:92-164) and the reduce kernels
(:306-408). The CUDA source cannot be built here (needs nvcc and THC headers), so this
restatement is pinned against the reference's own PyTorch oracle ms_deform_attn_core_pytorch
(ops/functions/ms_deform_attn_func.py:52-72) and its autograd, via tests/golden/msda_*.npz.
"""
import numpy as np


def _geometry(shapes, starts, loc, l):
    H, W = int(shapes[l][0]), int(shapes[l][1])
    x = loc[:, :, :, l, :, 0] * W - 0.5          # .cuh:290-291
    y = loc[:, :, :, l, :, 1] * H - 0.5
    inside = (y > -1) & (x > -1) & (y < H) & (x < W)   # .cuh:293
    y0 = np.floor(y).astype(np.int64)
    x0 = np.floor(x).astype(np.int64)
    lh = y - y0
    lw = x - x0
    return H, W, inside, y0, x0, lh, lw


def _corner(value_l, n_idx, m_idx, yy, xx, H, W, ok):
    """value_l [N,H*W,M,D] -> [N,Lq,M,P,D], zero where the corner is outside."""
    okc = ok & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
    pos = np.clip(yy, 0, H - 1) * W + np.clip(xx, 0, W - 1)
    v = value_l[n_idx, pos, m_idx]
    return np.where(okc[..., None], v, 0), okc, pos


def forward(value, shapes, starts, loc, attn):
    """value [N,S,M,D], shapes [L,2], starts [L], loc [N,Lq,M,L,P,2], attn [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((N, Lq, M, D), dtype=value.dtype)
    n_idx = np.arange(N)[:, None, None, None]
    m_idx = np.arange(M)[None, None, :, None]
    for l in range(L):
        H, W, inside, y0, x0, lh, lw = _geometry(shapes, starts, loc, l)
        vl = value[:, int(starts[l]):int(starts[l]) + H * W]
        hh, hw = 1 - lh, 1 - lw
        v1, _, _ = _corner(vl, n_idx, m_idx, y0, x0, H, W, inside)
        v2, _, _ = _corner(vl, n_idx, m_idx, y0, x0 + 1, H, W, inside)
        v3, _, _ = _corner(vl, n_idx, m_idx, y0 + 1, x0, H, W, inside)
        v4, _, _ = _corner(vl, n_idx, m_idx, y0 + 1, x0 + 1, H, W, inside)
        val = ((hh * hw)[..., None] * v1 + (hh * lw)[..., None] * v2 +
               (lh * hw)[..., None] * v3 + (lh * lw)[..., None] * v4)   # .cuh:85-88
        out += (attn[:, :, :, l, :, None] * val).sum(axis=3)
    return out.reshape(N, Lq, M * D)


def backward(value, shapes, starts, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn), shapes of value / loc / attn."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    g = grad_out.reshape(N, Lq, M, 1, D)
    grad_value = np.zeros_like(value)
    grad_loc = np.zeros_like(loc)
    grad_attn = np.zeros_like(attn)
    n_idx = np.arange(N)[:, None, None, None]
    m_idx = np.arange(M)[None, None, :, None]
    n_b = np.broadcast_to(n_idx, (N, Lq, M, P))
    m_b = np.broadcast_to(m_idx, (N, Lq, M, P))
    for l in range(L):
        H, W, inside, y0, x0, lh, lw = _geometry(shapes, starts, loc, l)
        s0 = int(starts[l])
        vl = value[:, s0:s0 + H * W]
        gl = grad_value[:, s0:s0 + H * W]
        hh, hw = 1 - lh, 1 - lw
        a = attn[:, :, :, l, :]
        tgv = a[..., None] * g                                           # top_grad_value, .cuh:125
        corners = ((y0, x0, hh * hw), (y0, x0 + 1, hh * lw), (y0 + 1, x0, lh * hw), (y0 + 1, x0 + 1, lh * lw))
        vs = []
        for yy, xx, w in corners:
            v, okc, pos = _corner(vl, n_idx, m_idx, yy, xx, H, W, inside)
            vs.append(v)
            contrib = np.where(okc[..., None], w[..., None] * tgv, 0)
            np.add.at(gl, (n_b, pos, m_b), contrib)                      # atomicAdd, .cuh:130-157
        v1, v2, v3, v4 = vs
        val = ((hh * hw)[..., None] * v1 + (hh * lw)[..., None] * v2 +
               (lh * hw)[..., None] * v3 + (lh * lw)[..., None] * v4)
        gw = -hh[..., None] * v1 + hh[..., None] * v2 - lh[..., None] * v3 + lh[..., None] * v4
        gh = -hw[..., None] * v1 - lw[..., None] * v2 + hw[..., None] * v3 + lw[..., None] * v4
        grad_attn[:, :, :, l, :] = np.where(inside, (g * val).sum(-1), 0)              # .cuh:161
        grad_loc[:, :, :, l, :, 0] = np.where(inside, W * (gw * tgv).sum(-1), 0)       # .cuh:162
        grad_loc[:, :, :, l, :, 1] = np.where(inside, H * (gh * tgv).sum(-1), 0)       # .cuh:163
    return grad_value, grad_loc, grad_attn


def _forward_sampled_t(v, shapes, starts, lc, at):
    """torch tensors in / out; differentiable (see backward_sampled)."""
    import torch
    from torch.nn.functional import grid_sample
    N, S, M, D = v.shape
    Lq, L, P = lc.shape[1], lc.shape[3], lc.shape[4]
    acc = torch.zeros((N * M, D, Lq), dtype=v.dtype)
    for l in range(L):
        H, W, s0 = int(shapes[l][0]), int(shapes[l][1]), int(starts[l])
        level_map = v[:, s0:s0 + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)
        grid = (2 * lc[:, :, :, l] - 1).permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        sampled = grid_sample(level_map, grid, mode="bilinear", padding_mode="zeros", align_corners=False)   # [N*M, D, Lq, P]
        weights = at[:, :, :, l].permute(0, 2, 1, 3).reshape(N * M, 1, Lq, P)
        acc = acc + (sampled * weights).sum(-1)
    return acc.view(N, M * D, Lq).permute(0, 2, 1).contiguous()


def backward_sampled(value, shapes, starts, loc, attn, grad_out):
    """(grad_value, grad_loc, grad_attn) as torch autograd derives them from the grid_sample composition -- how the
    reference's own test obtains its comparison gradients (ops/test.py:66-81 runs gradcheck on the op; its fixtures in
    tests/golden/msda_*.npz are autograd of ms_deform_attn_core_pytorch). Seconds at the BASELINE sizes, where the
    np.add.at restatement above takes minutes: used by the full-size GPU parity tests."""
    import torch
    v, lc, at = (torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True) for a in (value, loc, attn))
    out = _forward_sampled_t(v, shapes, starts, lc, at)
    out.backward(torch.from_numpy(np.ascontiguousarray(grad_out)))
    return v.grad.numpy(), lc.grad.numpy(), at.grad.numpy()


def forward_sampled(value, shapes, starts, loc, attn):
    """The same result as `forward`, on stock torch CPU ops the way the reference's CPU path gets it
    (`ms_deform_attn_core_pytorch`, ops/functions/ms_deform_attn_func.py:52-72): every level's [N*M, D, H, W] map is
    re-sampled bilinearly (zero padding, align_corners=False, grid = 2*loc - 1) at each query's P locations; here the
    level's attention weights are applied and accumulated level by level instead of stacking all L*P samples first.
    bench.py's MSDA `cpu_baseline` times this on the GPU host (threads = torch.get_num_threads()). numpy in / out."""
    import torch
    v, lc, at = (torch.from_numpy(np.ascontiguousarray(a)) for a in (value, loc, attn))
    with torch.no_grad():
        return _forward_sampled_t(v, shapes, starts, lc, at).numpy()
