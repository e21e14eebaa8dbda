"""M2F anomaly score at BASELINE C5 (1x100 queries, 256x512 mask features -> 1024x2048): unfused chain (mask GEMM,
torch bilinear upsample to [1,100,1024,2048], full-resolution score kernel) vs the fused kernel (8f-2)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
for B in (1, 4):
    emb = torch.randn(B, 100, 256, device="cuda") * 0.2
    feat = torch.randn(B, 256, 256, 512, device="cuda")
    cls = torch.randn(B, 100, 20, device="cuda") * 2
    featn = K.Act.from_nchw(feat)
    t_gemm = timeit(lambda: K.m2f_mask_logits(emb, feat), iters=10, warm=3)
    lg = K.m2f_mask_logits(emb, feat)
    t_fused = timeit(lambda: K.m2f_score_fused(cls, lg, (1024, 2048)), iters=10, warm=3)
    nchw = lg.permute(0, 3, 1, 2).contiguous()
    t_up = timeit(lambda: torch.nn.functional.interpolate(nchw, size=(1024, 2048), mode="bilinear", align_corners=False), iters=5, warm=2)
    up = torch.nn.functional.interpolate(nchw, size=(1024, 2048), mode="bilinear", align_corners=False)
    t_score = timeit(lambda: K.m2f_score(cls, up, (1024, 2048)), iters=5, warm=2)
    px = B * 1024 * 2048
    print(json.dumps(dict(B=B, mask_gemm_incl_nhwc_copy_ms=round(t_gemm, 3), fused_score_ms=round(t_fused, 3),
                          fused_gpix_s=round(px / t_fused / 1e6, 2), fused_alg_GBs=round((B * 256 * 512 * 400 + px * 4) / t_fused / 1e6, 1),
                          unfused_upsample_ms=round(t_up, 3), unfused_score_ms=round(t_score, 3),
                          speedup_vs_unfused=round((t_up + t_score) / t_fused, 2))), flush=True)
