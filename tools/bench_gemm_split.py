"""The split-bf16 GEMM route (csrc/gemm_bf16x3.hip, MssConvArgs.w_split) against the native fp32 MFMA kernel on the step's products:
error of both against float64 on a sample of rows, and time (alternating A B A B after a long warm-up, best round of each).
python tools/bench_gemm_split.py [--affine] [--quick] [--mfma]     --mfma: the split route twice, on v_mfma_f32_16x16x32_bf16 with
concatenated planes ("split", MSS_GEMM_SPLIT_MFMA=16) and on v_mfma_f32_32x32x16_bf16 ("split32", =32), in the same alternation"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib, kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit

CASES = [(1, 65536, 2048, 1024), (36, 65536, 128, 128), (36, 16384, 256, 256), (36, 4096, 512, 512), (64, 1936, 512, 1024), (64, 1936, 1024, 512),
         (64, 2112, 1024, 2048), (64, 2304, 4096, 256), (36, 5184, 4096, 256), (64, 29412, 304, 256), (64, 29412, 256, 256), (1, 65536, 2048, 4096),
         (1, 65536, 1024, 2048), (1, 162624, 256, 1024), (1, 162624, 1024, 256), (1, 162624, 256, 288)]
AFFINE = "--affine" in sys.argv
MFMA_AB = "--mfma" in sys.argv
if "--quick" in sys.argv:
    CASES = [(3, 1000, 64, 128), (36, 4096, 512, 512), (1, 65536, 2048, 4096), (2, 777, 256, 384)]
for (P, T, C, Ko) in CASES:
    torch.manual_seed(P + T)
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.zeros(P, Kpad, C, device="cuda")
    w[:, :Ko] = torch.randn(P, Ko, C, device="cuda") / C ** 0.5
    xt = torch.randn(P, T, C, device="cuda")
    planes = K.split_planes(w, Kpad, C)
    sc = sh = None
    if AFFINE and P == 1:
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
    outs, fns = {}, {}
    routes = ["native", "split"] + (["split32"] if MFMA_AB else [])
    for route in routes:
        yt = torch.zeros(P, T, Ko, device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
        if route != "native":
            a.w_split = ptr(planes)
        if sc is not None:
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko

        def f(a=a, route=route):
            if MFMA_AB:
                os.environ["MSS_GEMM_SPLIT_MFMA"] = "32" if route == "split32" else "16"
                _lib.reset_env_cache()
            call("mss_conv2d_forward_f32", ctypes.byref(a))
        assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == (1 if route == "native" else 3), route
        fns[route], outs[route] = f, yt
    for route in routes:
        fns[route]()
    torch.cuda.synchronize()
    # float64 reference on a sample of rows of the first and last batch entry
    rows = torch.randint(0, T, (min(T, 256),), device="cuda")
    err = {}
    for route in routes:
        e = 0.0
        for b in {0, P - 1}:
            xs = xt[b, rows].double()
            if sc is not None:
                xs = torch.relu(xs * sc.double() + sh.double())
            ref = xs @ w[b, :Ko].double().T
            e = max(e, ((outs[route][b, rows].double() - ref).abs().max() / ref.abs().max()).item())
        err[route] = e
    timeit(fns["native"], iters=20, warm=10)
    best = {r: 1e9 for r in routes}
    for _ in range(4):
        for r in routes:
            best[r] = min(best[r], timeit(fns[r], iters=10, warm=2))
    flops = 2.0 * P * T * C * Ko
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, affine=sc is not None,
                          **{r: dict(ms=round(best[r], 3), tflops=round(flops / best[r] / 1e9, 1), err_vs_f64=float(f"{err[r]:.2e}")) for r in routes},
                          gain=round(best["native"] / best["split"], 3),
                          **({"gain_16_over_32": round(best["split32"] / best["split"], 3)} if MFMA_AB else {}))), flush=True)
