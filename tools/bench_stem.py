import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
for n in (2, 1):
    img = torch.randn(n, 3, 1024, 2048, device="cuda")
    wt = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device="cuda") * 0.2)
    f1 = lambda: K.stem_conv_pool(img, wt)
    f2 = lambda: K.maxpool3s2(K.conv2d(K.stem_im2col(img), K.packed_stem(wt)))
    timeit(f1, iters=30, warm=10)
    b1 = min(timeit(f1, iters=20, warm=3) for _ in range(3))
    b2 = min(timeit(f2, iters=20, warm=3) for _ in range(3))
    print("N", n, "fused ms", round(b1, 4), "two-kernel ms", round(b2, 4))
