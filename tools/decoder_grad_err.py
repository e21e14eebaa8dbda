"""Pixel-decoder gradient error against the reference fixture at 704^2 per conv route (default policy, F(4x4) cap, direct):
which part of the error is the Winograd tiles', which is fp32 noise the reference has too (gnoise_* in the fixture).
usage: python tools/decoder_grad_err.py [fixture]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from multishiftseg_amd import _lib
from test_decoder import SHAPE, build

fixture = sys.argv[1] if len(sys.argv) > 1 else "m2f_decoder_704"
res = {}
for route, env in {"default": {}, "f4": {"MSS_WINO_MAX_TILE": "4"}, "direct": {"MSS_WINOGRAD": "0"}}.items():
    for k in ("MSS_WINO_MAX_TILE", "MSS_WINOGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    _lib.reset_env_cache()
    from conftest import golden
    dec, g = build(fixture, layers=int(golden(fixture)["layers"]))
    dec = dec.cuda()
    n, H, W = (int(v) for v in g["nhw"])
    rng = np.random.default_rng(int(g["seed"]))
    feats_np = {k: rng.standard_normal((n, c, H // s, W // s), dtype=np.float32) for k, (c, s) in SHAPE.items()}
    crng = np.random.default_rng(int(g["cot_seed"]))
    shapes = [(n, 256, H // 4, W // 4)] + [(n, 256, H // s, W // s) for s in (32, 16, 8)]
    cot = [torch.from_numpy(crng.standard_normal(s, dtype=np.float32)).cuda() for s in shapes]
    for p in dec.parameters():
        p.requires_grad_(True)
    feats = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in feats_np.items()}
    mask, out0, ms = dec.forward_features(feats)
    sum((t * c).sum() for t, c in zip((mask, *ms), cot)).backward()
    rel = lambda a, b: float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / (np.sqrt((b.astype(np.float64) ** 2).sum()) + 1e-30))
    out = {}
    for k, p in dec.named_parameters():
        got = p.grad.cpu().numpy()
        if "g_" + k in g.files:
            out[k] = rel(got, g["g_" + k])
        else:
            flat = got.reshape(got.shape[0], -1)
            out[k] = rel(flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)], g["gsub_" + k])
    for k, t in feats.items():
        got = t.grad.cpu().numpy()
        out["feat_" + k] = rel(got[:, ::max(1, got.shape[1] // 32), ::max(1, got.shape[2] // 16), ::max(1, got.shape[3] // 16)], g["gsub_feat_" + k])
    out["fwd_mask_sub_maxerr"] = float((mask.detach()[:, ::8, ::4, ::4].cpu() - torch.from_numpy(g["mask_sub"])).abs().max())
    res[route] = out
keys = ["feat_res2", "feat_res3", "feat_res4", "feat_res5", "layer_1.weight", "adapter_1.weight", "mask_features.weight", "fwd_mask_sub_maxerr"]
for k in keys:
    noise = float(g["gnoise_" + k]) if "gnoise_" + k in g.files else float("nan")
    print(f"{k:28s} ref fp32-vs-fp64 {noise:.2e} | " + " | ".join(f"{r} {res[r][k]:.2e}" for r in res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/decoder_grad_err_{fixture}.json", "w"), indent=1, sort_keys=True)
