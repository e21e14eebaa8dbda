import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
torch.manual_seed(0)
for (c, k, r, n, h, w) in [(16, 8, 3, 2, 20, 24), (16, 128, 3, 2, 20, 24), (32, 8, 3, 2, 20, 24), (32, 128, 3, 2, 20, 24),
                           (32, 128, 1, 2, 20, 24), (64, 128, 1, 1, 16, 16), (64, 128, 3, 2, 33, 47), (48, 48, 1, 1, 17, 19)]:
    x = torch.randn(n, c, h, w, device="cuda")
    wt = torch.randn(k, c, r, r, device="cuda") / (c * r * r) ** 0.5
    ref = torch.nn.functional.conv2d(x, wt, padding=r // 2)
    y = K.conv2d(K.Act.from_nchw(x), K.pack_weight(wt), pad=r // 2).nchw()
    err = (y - ref).abs().max().item()
    print(f"C={c} K={k} r={r} M={n*h*w}: max err {err:.3e}  finite={torch.isfinite(y).all().item()}", flush=True)
