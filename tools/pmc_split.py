"""PMC / timing target for the split-bf16 GEMM route: launches of the given products P,T,C,K through MssConvArgs.w_split
(`--native`: the fp32 MFMA kernel on the same operands).  python tools/pmc_split.py [--native] [--launches N] P,T,C,K ..."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib, kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
args = [a for a in sys.argv[1:] if not a.startswith("--")]
native = "--native" in sys.argv
tn = os.environ.get("PMC_SPLIT_TN") == "1"            # the TN weight-gradient product dU[P][K][C] = dY'^T X' instead of the NT product
n = int(sys.argv[sys.argv.index("--launches") + 1]) if "--launches" in sys.argv else 5
if "--launches" in sys.argv:
    args.remove(str(n))
for spec in args or ["1,65536,2048,4096"]:
    P, T, C, Ko = (int(v) for v in spec.split(","))
    Kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.zeros(P, Kpad, C, device="cuda")
    w[:, :Ko] = torch.randn(P, Ko, C, device="cuda") / C ** 0.5
    xt = torch.randn(P, T, C, device="cuda")
    yt = torch.empty(P, T, Ko, device="cuda")
    planes = K.split_planes(w, Kpad, C)
    a = MssConvArgs()
    a.x, a.w, a.y = ptr(xt), ptr(w), ptr(yt)
    if not native:
        a.w_split = ptr(planes)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
    a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko
    if tn:
        dyt = torch.randn(P, T, Ko, device="cuda")
        du = torch.empty(P, Ko, C, device="cuda")
        a = MssConvArgs()
        a.x = ptr(xt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.route = 0 if native else 1
        if P > 1:
            a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        launch = lambda: call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb)
    else:
        launch = lambda: call("mss_conv2d_forward_f32", ctypes.byref(a))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        launch()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(json.dumps(dict(P=P, T=T, C=C, K=Ko, route="native" if native else "bf16x3", ms=round(ms, 4), tflops=round(2.0 * P * T * C * Ko / ms / 1e9, 1))), flush=True)
