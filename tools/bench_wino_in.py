"""A/B of the Winograd INPUT transform alone on the step's shapes: two settings of one environment switch, alternating (A B A B after a
long warm-up, best round of each -- see tools/bench_gemm_variant.py for why), TB/s of the algorithmic bytes (x once + X' once).
    python tools/bench_wino_in.py MSS_WINO_INPUT_LDS 1 2"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import call, ptr
from tools.microbench import timeit
VAR, VA, VB = sys.argv[1], sys.argv[2], sys.argv[3]
SHAPES = [(2, 512, 1024, 128, 1, 4), (2, 512, 1024, 256, 1, 6), (2, 512, 1024, 304, 1, 6), (2, 256, 512, 256, 1, 4), (2, 128, 256, 512, 1, 4),
          (2, 128, 256, 512, 2, 6), (2, 128, 256, 1024, 2, 6), (2, 128, 256, 1024, 4, 6), (2, 128, 256, 4096, 12, 6), (2, 128, 256, 4096, 24, 6),
          (2, 128, 256, 4096, 36, 4), (1, 128, 256, 512, 1, 6), (16, 96, 96, 512, 1, 4)]
ONLY = int(os.environ.get("MSS_ONLY_TS", "0"))
for (n, h, w, c, dil, ts) in SHAPES:
    if ONLY and ts != ONLY:
        continue
    x = torch.randn(n, h, w, c, device="cuda")
    sc = torch.rand(c, device="cuda") + 0.5
    sh = torch.randn(c, device="cuda")
    P = (ts + 2) ** 2
    T = _lib.value("mss_wino_num_tiles", n, h, w, dil, ts)
    outs, fns = {}, {}
    for v in (VA, VB):
        xt = torch.empty(P, T, c, device="cuda")

        def f(v=v, xt=xt):
            os.environ[VAR] = v
            _lib.reset_env_cache()
            call("mss_wino_input_transform_f32", ptr(x), c, n, h, w, c, dil, ts, ptr(sc), ptr(sh), 1, ptr(xt))
        fns[v], outs[v] = f, xt
    timeit(fns[VA], iters=30, warm=10)
    best = {VA: 1e9, VB: 1e9}
    for _ in range(4):
        for v in (VA, VB):
            best[v] = min(best[v], timeit(fns[v], iters=10, warm=2))
    nbytes = 4.0 * (n * h * w * c + P * T * c)
    print(json.dumps({"shape": [n, h, w, c, dil, ts], VAR + "=" + VA: [round(best[VA], 4), round(nbytes / best[VA] / 1e9, 2)],
                      VAR + "=" + VB: [round(best[VB], 4), round(nbytes / best[VB] / 1e9, 2)], "gain": round(best[VA] / best[VB], 3),
                      "equal": bool(torch.equal(outs[VA], outs[VB]))}), flush=True)
