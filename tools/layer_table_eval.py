#!/usr/bin/env python3
"""Per-launch table of the MFMA kernels and Winograd transforms of ONE eval forward (image -> score, logits) at N x 1024 x 2048."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from multishiftseg_amd.deepv3 import DeepWV3Plus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = DeepWV3Plus(19).cuda().eval()
img = torch.randn(n, 3, 1024, 2048, device="cuda")
with torch.no_grad():
    for _ in range(2):
        m(img)
    prof = K.ConvProfile(); K.set_conv_profile(prof)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); m(img); e.record(); torch.cuda.synchronize()
    K.set_conv_profile(None)
rows = prof.per_launch()
print(f"forward {s.elapsed_time(e):.2f} ms")
agg = {}
for kind, tag, ms, tf in rows:
    d = agg.setdefault((kind, tag), [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += ms * tf
for kind in ("gemm_nt", "gemm_nt_bf16x3", "conv_igemm", "conv_igemm_bf16x3", "wino_transform"):
    sub = {k: v for k, v in agg.items() if k[0] == kind}
    t = sum(v[1] for v in sub.values()); w = sum(v[2] for v in sub.values())
    print(f"== {kind}: {t:.2f} ms, {w / t if t else 0:.1f} {'TB/s' if kind == 'wino_transform' else 'TF/s'}")
    for (k, tag), (c, ms, wt) in sorted(sub.items(), key=lambda kv: -kv[1][1]):
        print(f"  {str(tag):44s} x{c:<3d} {ms:7.3f} ms {wt / ms:7.1f}")
