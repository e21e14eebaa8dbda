import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_decoder import build, SHAPE
from multishiftseg_amd import kernels as K
for route in ("native", "bf16x3"):
    K.set_gemm_route(route)
    dec, g = build()
    dec = dec.cuda()
    rng = np.random.default_rng(int(g["seed"]))
    H, W = (int(v) for v in g["hw"])
    feats_np = {k: rng.standard_normal((2, c, H // s, W // s), dtype=np.float32) for k, (c, s) in SHAPE.items()}
    crng = np.random.default_rng(int(g["cot_seed"]))
    shapes = [(2, 256, 24, 40), (2, 256, 3, 5), (2, 256, 6, 10), (2, 256, 12, 20)]
    cot = [torch.from_numpy(crng.standard_normal(s, dtype=np.float32)).cuda() for s in shapes]
    for p in dec.parameters():
        p.requires_grad_(True); p.grad = None
    feats = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in feats_np.items()}
    mask, out0, ms = dec.forward_features(feats)
    loss = sum((t * c).sum() for t, c in zip((mask, *ms), cot))
    loss.backward()
    rel = lambda got, ref: float(np.sqrt(((got.astype(np.float64) - ref) ** 2).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30))
    worst = {}
    for k, p in dec.named_parameters():
        got = p.grad.cpu().numpy()
        if "g_" + k in g.files: worst[k] = rel(got, g["g_" + k])
        else:
            flat = got.reshape(got.shape[0], -1)
            worst[k] = rel(flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)], g["gsub_" + k])
    for k, t in feats.items():
        got = t.grad.cpu().numpy()
        worst["feat_" + k] = rel(got[:, ::max(1, got.shape[1] // 32)], g["gsub_feat_" + k])
    print(route, "forward err", float(np.abs(mask.detach().cpu().numpy()[:, ::4] - g["mask_sub"]).max()), float(np.abs(ms[0].detach().cpu().numpy() - g["out0"]).max()))
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:14]:
        print(f"   {v:.3e}  {k}")
