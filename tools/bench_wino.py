"""Direct implicit GEMM vs Winograd F(2x2,3x3) vs F(4x4,3x3) on the network's channel-heavy 3x3 shapes
(time, dense-algorithmic TFLOP/s, and max error of each Winograd form against the direct kernel)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K
from tools.microbench import timeit
SHAPES = [(2, 128, 256, 512, 512, 1), (2, 128, 256, 512, 1024, 2), (2, 128, 256, 1024, 512, 2), (2, 128, 256, 1024, 2048, 4),
          (2, 256, 512, 256, 256, 1), (2, 256, 512, 304, 256, 1), (2, 512, 1024, 128, 128, 1),
          (2, 128, 256, 4096, 256, 12), (2, 128, 256, 4096, 256, 24), (2, 128, 256, 4096, 256, 36),
          (1, 64, 128, 512, 512, 1), (1, 64, 128, 1024, 2048, 4)]
for (n, h, w, cin, cout, dil) in SHAPES:
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
    res = K.Act(torch.randn(n, h, w, cout, device="cuda"))
    out = K.Act.empty(n, h, w, cout, "cuda")
    pw = K.pack_weight(wt)
    run_d = lambda: K.conv2d(x, pw, dil=dil, pad=dil, in_affine=(sc, sh), in_relu=True, res=res, out=out)
    md = timeit(run_d, iters=5, warm=2)
    ref = out.buf.clone()
    fl = 2.0 * n * h * w * cin * cout * 9
    row = dict(shape=[n, h, w, cin, cout, dil], direct_ms=round(md, 3), direct_tf=round(fl / md / 1e9, 1), policy_tile=K.wino_tile(h, w, dil))
    for ts in (2, 4):
        ww = K.pack_weight_wino(wt, tile=ts)
        run_w = lambda: K.conv2d_winograd(x, ww, dil=dil, in_affine=(sc, sh), in_relu=True, res=res, out=out)
        mw = timeit(run_w, iters=5, warm=2)
        err = ((out.buf - ref).abs().max() / ref.abs().max()).item()
        row[f"f{ts}_ms"] = round(mw, 3); row[f"f{ts}_tf_alg"] = round(fl / mw / 1e9, 1); row[f"f{ts}_speedup"] = round(md / mw, 3)
        row[f"f{ts}_relerr"] = float(f"{err:.2e}")
        del ww
    print(json.dumps(row), flush=True)
