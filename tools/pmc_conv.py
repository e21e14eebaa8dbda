import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
n, h, w, cin, cout, r, dil = 2, 128, 256, 512, 512, 3, 1
x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
wt = torch.randn(cout, cin, r, r, device="cuda") / (cin * r * r) ** 0.5
pw = K.pack_weight(wt)
out = K.Act.empty(n, h, w, cout, "cuda")
sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
for _ in range(5):
    K.conv2d(x, pw, dil=dil, pad=dil, in_affine=(sc, sh), in_relu=True, out=out)
torch.cuda.synchronize()
