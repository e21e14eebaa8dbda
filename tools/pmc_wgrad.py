"""PMC target: the 1x1 weight-gradient GEMM of ASPP (2x128x256, 4096 -> 256) and the forward GEMM of the same size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
n, h, w, cin, cout = 2, 128, 256, 4096, 256
x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
dy = K.Act(torch.randn(n, h, w, cout, device="cuda"))
pw = K.pack_weight(torch.randn(cout, cin, 1, 1, device="cuda") / 64)
for _ in range(3):
    K.conv2d_wgrad(x, dy, cout, cin, 1, 1)
    K.conv2d(x, pw)
torch.cuda.synchronize()
