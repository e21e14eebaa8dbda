import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
n, h, w, cin, cout, r, dil = 2, 128, 256, 4096, 256, 3, 12
x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
dy = K.Act(torch.randn(n, h, w, cout, device="cuda"))
for _ in range(3):
    K.conv2d_wgrad(x, dy, cout, cin, r, r, dil=dil, pad=dil)
torch.cuda.synchronize()
