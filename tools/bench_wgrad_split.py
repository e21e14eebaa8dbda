"""The TN weight-gradient product on the split-bf16 route (MssConvArgs.route = 1, csrc/gemm_bf16x3.hip gemm_tn_bf16x3_kernel) against
the native fp32 MFMA kernels: error of both against float64 (sampled output rows) and time, alternating A B A B.
python tools/bench_wgrad_split.py [--quick]"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
# (P, T, C, K): ASPP through F(6x6) / F(4x4), the decoder's second convolution through F(6x6), the pixel decoder's Linears, the 1x1 ASPP branch
CASES = [(64, 2304, 4096, 256), (36, 5184, 4096, 256), (64, 29412, 256, 256), (1, 162624, 256, 256), (1, 162624, 1024, 256), (1, 162624, 256, 1024),
         (1, 65536, 4096, 256), (1, 65536, 1280, 256)]
if "--quick" in sys.argv:
    CASES = [(3, 1000, 256, 128), (1, 777, 512, 256), (2, 5000, 256, 384), (1, 70001, 256, 128)]
for (P, T, C, Ko) in CASES:
    torch.manual_seed(T)
    xt = torch.randn(P, T, C, device="cuda")
    dyt = torch.randn(P, T, Ko, device="cuda")
    res, fns = {}, {}
    for route in (0, 1):
        du = torch.full((P, Ko, C), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x = ptr(xt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.route = route
        if P > 1:
            a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        fns[route] = (lambda a=a, du=du, ws=ws, wsb=wsb: call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb))
        res[route] = du
    for r in (0, 1):
        fns[r]()
    torch.cuda.synchronize()
    rows = torch.randint(0, Ko, (min(Ko, 64),), device="cuda")
    err = {}
    for r in (0, 1):
        e = 0.0
        for b in {0, P - 1}:
            ref = dyt[b][:, rows].double().T @ xt[b].double()
            e = max(e, ((res[r][b, rows].double() - ref).abs().max() / ref.abs().max()).item())
        err[r] = e
    timeit(fns[0], iters=10, warm=5)
    best = {0: 1e9, 1: 1e9}
    for _ in range(3):
        for r in (0, 1):
            best[r] = min(best[r], timeit(fns[r], iters=6, warm=2))
    flops = 2.0 * P * T * C * Ko
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, native=dict(ms=round(best[0], 3), tflops=round(flops / best[0] / 1e9, 1), err=float(f"{err[0]:.2e}")),
                          split=dict(ms=round(best[1], 3), tflops=round(flops / best[1] / 1e9, 1), err=float(f"{err[1]:.2e}")),
                          gain=round(best[0] / best[1], 3), finite=bool(torch.isfinite(res[1]).all()))), flush=True)
