import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from multishiftseg_amd import kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
for (P, T, C, Ko) in [(16, 576, 4096, 256), (16, 2304, 4096, 256), (16, 576, 256, 256), (36, 576, 4096, 256), (16, 1152, 4096, 256)]:
    torch.manual_seed(T)
    xt = torch.randn(P, T, C, device="cuda"); dyt = torch.randn(P, T, Ko, device="cuda")
    res = {}
    for route in (0, 1):
        du = torch.full((P, Ko, C), float("nan"), device="cuda")
        a = MssConvArgs(); a.x = ptr(xt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.route = route
        a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb)
        res[route] = du
    d = (res[0] - res[1]).abs().max().item()
    print(P, T, C, Ko, "max diff", d, "equal", torch.equal(res[0], res[1]), "finite", bool(torch.isfinite(res[1]).all()), flush=True)
