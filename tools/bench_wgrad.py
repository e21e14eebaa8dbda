import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
for (n, h, w, cin, cout, r, dil) in [(2, 128, 256, 4096, 256, 3, 12), (2, 128, 256, 4096, 256, 1, 1), (2, 512, 1024, 256, 256, 3, 1),
                                  (2, 512, 1024, 256, 19, 1, 1), (2, 512, 1024, 128, 48, 1, 1), (2, 128, 256, 1280, 256, 1, 1)]:
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    dy = K.Act(torch.randn(n, h, w, 48 if cout < 48 else cout, device="cuda"))
    if cout < 48:
        dy = dy.slice(20, cout)          # the head gradients are channel slices of the fused 48-wide buffer
    pad = dil if r == 3 else 0
    ms = timeit(lambda: K.conv2d_wgrad(x, dy, cout, cin, r, r, dil=dil, pad=pad), iters=5, warm=2)
    print(os.environ.get("MSS_WGRAD_VARIANT", "0"), (n, h, w, cin, cout, r, dil), round(ms, 3), "ms", round(2.0 * n * h * w * cin * cout * r * r / ms / 1e9, 1), "TF")
