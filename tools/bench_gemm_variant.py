"""A/B of gemm_nt variants (MSS_GEMM_VARIANT=2 shipped in r02, 3 = r03 branch-free loader + interleave, 4 = 3 with swapped MFMA
operands and 16-byte stores) on the step's products.  python tools/bench_gemm_variant.py [a,b]   (default 2,3)"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
CASES = [(1, 65536, 2048, 1024), (36, 65536, 128, 128), (36, 16384, 256, 256), (36, 4096, 512, 512), (64, 1936, 512, 1024), (64, 1936, 1024, 512), (64, 2112, 1024, 2048),
         (64, 2304, 4096, 256), (36, 5184, 4096, 256), (64, 29412, 304, 256), (64, 29412, 256, 256), (1, 65536, 2048, 4096), (1, 65536, 1024, 2048)]
VA, VB = (sys.argv[1].split(",") if len(sys.argv) > 1 else ["2", "3"])
VAR = sys.argv[2] if len(sys.argv) > 2 else "MSS_GEMM_VARIANT"        # e.g. `bench_gemm_variant.py 0,8 MSS_GEMM_GROUP_M`
for (P, T, C, Ko) in CASES:
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.randn(P, Kpad, C, device="cuda")
    xt = torch.randn(P, T, C, device="cuda")
    outs, res, fns = {}, {}, {}
    for var in (VA, VB):
        yt = torch.zeros(P, T, Ko, device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko

        def f(a=a, var=var):
            os.environ[VAR] = var
            _lib.reset_env_cache()
            call("mss_conv2d_forward_f32", ctypes.byref(a))
        fns[var], outs[var] = f, yt
    # the variants alternate (A B A B ...) after a long warm-up and the best round of each counts: the first thing timed after the
    # random fills runs on clocks that are still ramping (an order effect of 10-15 % on the sub-millisecond products)
    timeit(fns[VA], iters=30, warm=10)
    best = {VA: 1e9, VB: 1e9}
    for _ in range(4):
        for var in (VA, VB):
            best[var] = min(best[var], timeit(fns[var], iters=10, warm=2))
    for var in (VA, VB):
        res[var] = (round(best[var], 3), round(2.0 * P * T * C * Ko / best[var] / 1e9, 1))
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, **{"v" + VA: res[VA], "v" + VB: res[VB]}, gain=round(res[VA][0] / res[VB][0], 3),
                          equal=bool(torch.equal(outs[VA], outs[VB])))), flush=True)
