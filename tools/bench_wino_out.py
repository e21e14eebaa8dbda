"""F(4x4) output transform A/B (MSS_WINO_OUT4_LDS=0|1 in separate processes): ms and TB/s of algorithmic bytes at the step's shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K, _lib
from multishiftseg_amd._lib import call, ptr
from tools.microbench import timeit
for (N, H, W, C, dil, res) in ((2, 512, 1024, 128, 1, True), (2, 256, 512, 256, 1, True), (2, 128, 256, 512, 1, True), (2, 128, 256, 256, 36, False), (1, 128, 256, 512, 1, True)):
    ts = 4
    T = _lib.value("mss_wino_num_tiles", N, H, W, dil, ts)
    yt = torch.randn(36, T, C, device="cuda")
    out = K.Act.empty(N, H, W, C, "cuda")
    r = K.Act(torch.randn(N, H, W, C, device="cuda")) if res else None
    stats = torch.empty((_lib.value("mss_wino_output_stats_parts", N, H, W, C, dil, ts), 2, C), device="cuda")
    f = lambda: call("mss_wino_output_transform_f32", ptr(yt), N, H, W, C, dil, ts, r.ptr if r else None, r.ld if r else 0, out.ptr, out.ld, ptr(stats))
    ms = timeit(f, iters=20)
    by = 4.0 * (36 * T * C + N * H * W * C * (2 if res else 1))
    print(json.dumps(dict(shape=[N, H, W, C, dil], lds=os.environ.get("MSS_WINO_OUT4_LDS", "1"), ms=round(ms, 4), TBs=round(by / ms / 1e9, 2))), flush=True)
