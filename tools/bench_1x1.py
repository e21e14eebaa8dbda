"""1x1 stride-1 convolutions of the trunk/heads with the fused BN+ReLU prologue and residual epilogue (MSS_GEMM=0|1)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
AFF = os.environ.get("AFFINE", "1") == "1"
for (n, h, w, cin, cout, res, stats) in [(2, 128, 256, 2048, 4096, True, True), (2, 128, 256, 2048, 4096, True, False), (2, 128, 256, 2048, 4096, False, False),
                                         (2, 128, 256, 1024, 2048, True, True), (2, 128, 256, 2048, 1024, False, True), (2, 128, 256, 512, 1024, False, True),
                                         (2, 128, 256, 4096, 256, False, True), (2, 128, 256, 1280, 256, False, False)]:
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = K.pack_weight(torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5)
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
    r = K.Act(torch.randn(n, h, w, cout, device="cuda")) if res else None
    out = K.Act.empty(n, h, w, cout, "cuda")
    ms = timeit(lambda: K.conv2d(x, wt, in_affine=(sc, sh) if AFF else None, in_relu=AFF, res=r, out=out, want_stats=stats), iters=5, warm=2)
    print(json.dumps(dict(shape=[n, h, w, cin, cout], res=res, stats=stats, ms=round(ms, 3), tflops=round(2.0 * n * h * w * cin * cout / ms / 1e9, 1),
                          gemm=os.environ.get("MSS_GEMM", "1"), affine=AFF)), flush=True)
