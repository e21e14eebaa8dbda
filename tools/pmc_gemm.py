"""PMC target: one batched Winograd-domain GEMM (36 x [4096 x 1024] x [1024 x 2048]); MSS_GEMM_BF16X6=0|1."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import MssConvArgs, call, ptr
P, T, C, Ko = 36, 4096, 1024, 2048
xt = torch.randn(P, T, C, device="cuda"); w = torch.randn(P, Ko, C, device="cuda"); yt = torch.empty(P, T, Ko, device="cuda")
a = MssConvArgs()
a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Ko, Ko
a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Ko * C, T * Ko
for _ in range(3):
    call("mss_conv2d_forward_f32", ctypes.byref(a))
torch.cuda.synchronize()
