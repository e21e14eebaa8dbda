"""A/B of gemm_nt variants (MSS_GEMM_VARIANT=2 shipped in r02, 3 = r03 branch-free loader + interleave) on the step's products."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
CASES = [(36, 65536, 128, 128), (36, 16384, 256, 256), (36, 4096, 512, 512), (64, 1936, 512, 1024), (64, 1936, 1024, 512), (64, 2112, 1024, 2048),
         (64, 2304, 4096, 256), (36, 5184, 4096, 256), (64, 29412, 304, 256), (64, 29412, 256, 256), (1, 65536, 2048, 4096), (1, 65536, 1024, 2048)]
for (P, T, C, Ko) in CASES:
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.randn(P, Kpad, C, device="cuda")
    xt = torch.randn(P, T, C, device="cuda")
    outs, res = {}, {}
    for var in ("2", "3"):
        os.environ["MSS_GEMM_VARIANT"] = var
        yt = torch.zeros(P, T, Ko, device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko
        f = lambda: call("mss_conv2d_forward_f32", ctypes.byref(a))
        ms = timeit(f, iters=10, warm=3)
        outs[var] = yt
        res[var] = (round(ms, 3), round(2.0 * P * T * C * Ko / ms / 1e9, 1))
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, v2=res["2"], v3=res["3"], gain=round(res["2"][0] / res["3"][0], 3),
                          equal=bool(torch.equal(outs["2"], outs["3"])))), flush=True)
