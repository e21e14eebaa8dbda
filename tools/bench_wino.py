import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
for (n, h, w, cin, cout, dil) in [(2, 128, 256, 512, 512, 1), (2, 128, 256, 512, 1024, 2), (2, 128, 256, 1024, 512, 2), (2, 128, 256, 1024, 2048, 4),
                                  (2, 256, 512, 256, 256, 1), (2, 512, 1024, 128, 128, 1), (2, 512, 1024, 256, 256, 1), (1, 64, 128, 512, 512, 1), (1, 64, 128, 1024, 2048, 4)]:
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
    res = K.Act(torch.randn(n, h, w, cout, device="cuda"))
    out = K.Act.empty(n, h, w, cout, "cuda")
    pw, ww = K.pack_weight(wt), K.pack_weight_wino(wt)
    md = timeit(lambda: K.conv2d(x, pw, dil=dil, pad=dil, in_affine=(sc, sh), in_relu=True, res=res, out=out), iters=5, warm=2)
    mw = timeit(lambda: K.conv2d_winograd(x, ww, dil=dil, in_affine=(sc, sh), in_relu=True, res=res, out=out), iters=5, warm=2)
    fl = 2.0 * n * h * w * cin * cout * 9
    print(json.dumps(dict(shape=[n, h, w, cin, cout, dil], direct_ms=round(md, 3), wino_ms=round(mw, 3), direct_tf=round(fl / md / 1e9, 1),
                          wino_tf_algorithmic=round(fl / mw / 1e9, 1), speedup=round(md / mw, 3))), flush=True)
