"""A/B of the wave-private GEMM experiment (MSS_GEMM_WP=1) against the shipped persistent kernel on batched products; checks equality."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import MssConvArgs, call, ptr
from tools.microbench import timeit
CASES = [(64, 2112, 1024, 2048), (36, 4096, 512, 512), (1, 65536, 2048, 4096)]
for (P, T, C, Ko) in CASES:
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.randn(P, Kpad, C, device="cuda")
    xt = torch.randn(P, T, C, device="cuda")
    outs = {}
    res = {}
    for wp in ("0", "1", "2", "3", "4"):
        os.environ["MSS_GEMM_WP"] = wp
        os.environ["MSS_GEMM_BN"] = "256"
        os.environ["MSS_GEMM_VARIANT"] = "2"        # "shipped" here = the round-2 loop the ablation kernel was derived from
        _lib.reset_env_cache()
        yt = torch.zeros(P, T, Ko, device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko
        f = lambda: call("mss_conv2d_forward_f32", ctypes.byref(a))
        ms = timeit(f, iters=10, warm=3)
        outs[wp] = yt
        res[wp] = (round(ms, 3), round(2.0 * P * T * C * Ko / ms / 1e9, 1))
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, shipped=res["0"], wave_private=res["1"], wp_no_loads=res["2"], wp_mfma_only=res["3"], wp_mfma_only_zero_operands=res["4"], equal=bool(torch.equal(outs["0"], outs["1"])),
                          maxdiff=float((outs["0"] - outs["1"]).abs().max()))), flush=True)
