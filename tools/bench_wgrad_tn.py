"""Winograd-domain weight-gradient products of the 2x1024x2048 step in isolation (TFLOP/s of executed MFMA work)."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
# (P, T, C, K): ASPP dilation 12 / 24 through F(6x6), dilation 36 through F(4x4), the two decoder convolutions through F(6x6)
CASES = [(64, 2304, 4096, 256), (36, 5184, 4096, 256), (64, 29412, 256, 256), (64, 29412, 304, 256), (1, 162624, 256, 256), (1, 162624, 1024, 256), (1, 162624, 256, 1024),
         (1, 65536, 4096, 256)]
for (P, T, C, Ko) in CASES:
    xt = torch.randn(P, T, C, device="cuda")
    dyt = torch.randn(P, T, Ko, device="cuda")
    du = torch.empty(P, Ko, C, device="cuda")
    a = MssConvArgs()
    a.x = ptr(xt)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
    a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    if P > 1:
        a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
    ws, wsb = K._wgrad_workspace(a, C, "cuda")
    ms = timeit(lambda: call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb), iters=5, warm=2)
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, ms=round(ms, 3), tflops=round(2.0 * P * T * C * Ko / ms / 1e9, 1), ws_MB=round(wsb / 1e6, 1),
                          tn=os.environ.get("MSS_WGRAD_TN", "1"))), flush=True)
