"""The batched 1x1 GEMMs of the Winograd path in isolation: [P][T][C] x [P][K][C]^T -> [P][T][K] (TFLOP/s, GB/s)."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K, _lib
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
# default: the F(4x4) products of the 2x1024x2048 step; or `bench_bgemm.py P,T,C,K [P,T,C,K ...]`
CASES = [(36, T, C, Ko) for (T, C, Ko) in [(65536, 128, 128), (16384, 256, 256), (65536, 256, 256), (65536, 304, 256), (4096, 512, 512),
                                            (4096, 512, 1024), (4096, 1024, 2048), (5184, 4096, 256)]]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
NBUF = int(os.environ.get("MSS_BENCH_NBUF", "1"))      # > 1: rotate over that many distinct X' / Y' buffers (cold caches / TLB, as in the step)
for (P, T, C, Ko) in CASES:
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.randn(P, Kpad, C, device="cuda")
    sets = []
    for _ in range(NBUF):
        xt = torch.randn(P, T, C, device="cuda")
        yt = torch.empty(P, T, Ko, device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko
        sets.append((a, xt, yt))
    it = [0]

    def run():
        a = sets[it[0] % NBUF][0]
        it[0] += 1
        call("mss_conv2d_forward_f32", ctypes.byref(a))
    ms = timeit(run, iters=10, warm=3)
    fl = 2.0 * P * T * C * Ko
    by = 4.0 * P * T * (C + Ko)
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, nbuf=NBUF, ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1), GBs=round(by / ms / 1e6, 1),
                          bk="policy")), flush=True)
    del sets
