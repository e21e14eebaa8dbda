"""MSDeformAttn op at the BASELINE shapes (C4: 22^2/44^2/88^2 at N = 1 and 16; C5: 32x64/64x128/128x256 at N = 1):
forward (locations + weights given), fused forward (raw offsets + logits), backward; GB/s of the COMPULSORY bytes
4(N S M D + 3 N Lq M L P + N Lq M D) and the L2 row-gather bytes (48 x 128 B per (query, head))."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
from multishiftseg_amd.ms_deform_attn import _FusedSampleFn
from tools.microbench import timeit

for N, shapes in ((1, [(22, 22), (44, 44), (88, 88)]), (16, [(22, 22), (44, 44), (88, 88)]), (1, [(32, 64), (64, 128), (128, 256)])):
    shp = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
    starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    S = int(shp.prod(1).sum())
    value = torch.randn(N, S, 8, 32, device="cuda")
    # reference points at the pixel centres of each query's own level + offsets of a few pixels, as the encoder produces
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                                indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes])[None, :, None, :].expand(N, S, 3, 2).contiguous()
    off = torch.randn(N, S, 8, 3, 4, 2, device="cuda") * 3
    lg = torch.randn(N, S, 8, 12, device="cuda")
    loc = (ref[:, :, None, :, None, :] + off / shp.flip(-1)[None, None, None, :, None, :].float()).contiguous()
    attn = torch.softmax(lg, -1).view(N, S, 8, 3, 4).contiguous()
    g = torch.randn(N, S, 256, device="cuda")
    ms_f = timeit(lambda: MSDA.ms_deform_attn_forward(value, shp, starts, loc, attn, 128), iters=20)
    with torch.no_grad():
        ms_ff = timeit(lambda: _FusedSampleFn.apply(value, shp, starts, off, lg, ref), iters=20)
    shp._mss_host = shapes
    ms_b = timeit(lambda: MSDA.ms_deform_attn_backward(value, shp, starts, loc, attn, g, 128), iters=10)
    os.environ["MSS_MSDA_BWD_BINNED"] = "0"        # the generic scatter-add kernel (memory-side atomics)
    ms_b_at = timeit(lambda: MSDA.ms_deform_attn_backward(value, shp, starts, loc, attn, g, 128), iters=10)
    os.environ.pop("MSS_MSDA_BWD_BINNED")
    # forward inputs + grad_out + grad_value (zero + read-modify-write >= 2x) + grad_loc + grad_attn
    byt_b = 4 * (N * S * 256 + 3 * N * S * 8 * 12 + N * S * 256 + 2 * N * S * 256 + 3 * N * S * 8 * 12)
    byt = 4 * (N * S * 256 + 3 * N * S * 8 * 12 + N * S * 256)
    gather = N * S * 8 * 48 * 128
    print(json.dumps(dict(kernel="msda", N=N, S=S, compulsory_MB=round(byt / 1e6, 1), fwd_ms=round(ms_f, 4), fused_fwd_ms=round(ms_ff, 4),
                          bwd_ms=round(ms_b, 4), bwd_atomic_ms=round(ms_b_at, 4), bwd_algorithmic_MB=round(byt_b / 1e6, 1),
                          bwd_GBs=round(byt_b / ms_b / 1e6, 1), bwd_frac_of_8TBs=round(byt_b / ms_b / 1e6 / 8000, 3), fwd_compulsory_GBs=round(byt / ms_f / 1e6, 1), fused_fwd_compulsory_GBs=round(byt / ms_ff / 1e6, 1),
                          fwd_frac_of_8TBs=round(byt / ms_f / 1e6 / 8000, 3), fwd_L2_gather_GBs=round(gather / ms_f / 1e6, 1))), flush=True)
