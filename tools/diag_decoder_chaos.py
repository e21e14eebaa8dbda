"""How much do the pixel decoder's stored gradient slices move between fp32 evaluations that differ only in rounding? The same
backward (tests/test_decoder.py fullsize fixtures) on GEMM routes x Winograd tile caps; prints, per variant, the entries that exceed
the test's bound and a few watched ones. (Round 5: decides whether a bound violation is an accuracy problem or a flipped ReLU /
crossed bilinear cell inside a small stored slice.)"""
import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_decoder import build, SHAPE, _pooled_sens
from conftest import golden
from multishiftseg_amd import kernels as K, _lib
fx = sys.argv[1] if len(sys.argv) > 1 else "m2f_decoder_704"
g0 = golden(fx)
WATCH = ["feat_res2", "feat_res3", "transformer.encoder.layers.5.self_attn.sampling_offsets.weight", "transformer.encoder.layers.3.self_attn.sampling_offsets.bias", "layer_1.weight"]
for route in ("native", "bf16x3"):
    for env in ({}, {"MSS_WINO_MAX_TILE": "4"}, {"MSS_WINO_TILE": "2"}, {"MSS_WINOGRAD": "0"}):
        for k in ("MSS_WINO_MAX_TILE", "MSS_WINO_TILE", "MSS_WINOGRAD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        _lib.reset_env_cache()
        K.set_gemm_route(route)
        dec, g = build(fx, layers=int(g0["layers"]))
        dec = dec.cuda()
        n, H, W = (int(v) for v in g["nhw"])
        rng = np.random.default_rng(int(g["seed"]))
        feats_np = {k: rng.standard_normal((n, c, H // s, W // s), dtype=np.float32) for k, (c, s) in SHAPE.items()}
        crng = np.random.default_rng(int(g["cot_seed"]))
        shapes = [(n, 256, H // 4, W // 4)] + [(n, 256, H // s, W // s) for s in (32, 16, 8)]
        cot = [torch.from_numpy(crng.standard_normal(s, dtype=np.float32)).cuda() for s in shapes]
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        feats = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in feats_np.items()}
        mask, out0, ms = dec.forward_features(feats)
        sum((t * c).sum() for t, c in zip((mask, *ms), cot)).backward()
        rel = lambda got, ref: float(np.sqrt(((got.astype(np.float64) - ref) ** 2).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30))
        rel_either = lambda got, k32, k64: min(rel(got, g[k32]), rel(got, g[k64].astype(np.float64))) if k64 in g.files else rel(got, g[k32])
        worst = {}
        for k, p in dec.named_parameters():
            got = p.grad.cpu().numpy()
            if "g_" + k in g.files:
                worst[k] = rel(got, g["g_" + k])
            else:
                flat = got.reshape(got.shape[0], -1)
                worst[k] = rel(flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)], g["gsub_" + k])
        for k, t in feats.items():
            got = t.grad.cpu().numpy()
            worst["feat_" + k] = rel_either(got[:, ::max(1, got.shape[1] // 32), ::max(1, got.shape[2] // 16), ::max(1, got.shape[3] // 16)], "gsub_feat_" + k, "g64sub_feat_" + k)
        sens = _pooled_sens(g)
        bad = {k: (round(v, 5), round(max(1e-3, 3 * float(g["gnoise_" + k]), 2 * sens(k)), 5)) for k, v in worst.items() if v > max(1e-3, 3 * float(g["gnoise_" + k]), 2 * sens(k))}
        print(json.dumps({"route": route, "env": env, "over_bound": bad, "watch": {k.replace("transformer.encoder.layers.", "L"): float(f"{worst[k]:.2e}") for k in WATCH}}), flush=True)
        K.set_gemm_route(None)
