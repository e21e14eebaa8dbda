"""The narrow (<= 64 output channels) weight gradients of the 2x1024x2048 step in isolation: the 19-channel head over a slice of the
48-wide gradient buffer with the BatchNorm + ReLU prologue, bot_fine's 128 -> 48. MSS_WGRAD_NARROW=0 / 1 alternating.
    python tools/bench_wgrad_narrow.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K, _lib
from tools.microbench import timeit

for (n, h, w, cin, k, ld, c0, affine) in [(2, 512, 1024, 256, 19, 48, 20, True), (2, 512, 1024, 128, 48, 48, 0, True)]:
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    dy = K.Act(torch.randn(n, h, w, ld, device="cuda"), k, c0)
    aff = (torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3) if affine else None
    res = {}
    fns = {}
    for mode in ("0", "1"):
        def f(mode=mode):
            os.environ["MSS_WGRAD_NARROW"] = mode
            _lib.reset_env_cache()
            K.conv2d_wgrad(x, dy, k, cin, 1, 1, in_affine=aff, in_relu=affine)
        fns[mode] = f
    best = {"0": 1e9, "1": 1e9}
    timeit(fns["0"], iters=10, warm=5)
    for _ in range(4):
        for mode in ("0", "1"):
            best[mode] = min(best[mode], timeit(fns[mode], iters=10, warm=2))
    gb = (n * h * w * (cin + k) * 4) / 1e9
    print(json.dumps(dict(shape=[n, h, w, cin, k], lds_ms=round(best["0"], 4), narrow_ms=round(best["1"], 4),
                          lds_TBs=round(gb / best["0"], 2), narrow_TBs=round(gb / best["1"], 2))), flush=True)
