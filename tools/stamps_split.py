"""DIAGNOSTIC (needs the -DMSS_SPLIT_STAMPS build of gemm_bf16x3.hip as MSS_LIB): where the issue time of gemm_nt_bf16x3_kernel's K-step goes.
Per wave the kernel sums s_memtime deltas over eight stretches of every K-step: top (fragment reads issued, waits included), the six
fenced segments of 8 MFMAs, and the barrier. Prints the mean cycles per K-step and stretch over all waves.
usage: make -C multishiftseg_amd/csrc stamps; MSS_LIB=multishiftseg_amd/csrc/build/libmss_hip_stamps.so python tools/stamps_split.py P,T,C,K
--dump also prints every workgroup's lifetime. s_memtime counts shader cycles, s_memrealtime 100 MHz (tools/clock_check.py), so the
quotient per wave is the clock it ran at. Round 5 result (profiles/r05/stamps_split.txt, dynamic_tiles.md): 1.6 GHz inside the kernel;
static tile walk = bimodal lifetimes (3.3 / 4.85 ms), ticket order = 4.26 - 4.65 ms and 3724 cycles per K-step and wave with two waves
per SIMD, i.e. the pipe busy 2 x 1536 / 3724 = 82 % of that clock."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import _lib, kernels as K
from multishiftseg_amd._lib import MssConvArgs, call, ptr
P, T, C, Ko = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,65536,2048,4096").split(","))
Kpad = _lib.value("mss_conv2d_kpad", Ko)
w = torch.randn(P, Kpad, C, device="cuda") / C ** 0.5
xt = torch.randn(P, T, C, device="cuda"); yt = torch.empty(P, T, Ko, device="cuda")
planes = K.split_planes(w, Kpad, C)
a = MssConvArgs()
a.x, a.w, a.y, a.w_split = ptr(xt), ptr(w), ptr(yt), ptr(planes)
a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, Kpad, Ko
a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, Kpad * C, T * Ko
for _ in range(5):
    call("mss_conv2d_forward_f32", ctypes.byref(a))
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    call("mss_conv2d_forward_f32", ctypes.byref(a))
ev1.record(); torch.cuda.synchronize()
kernel_ms = ev0.elapsed_time(ev1) / 10
lib = _lib.load()
n = 512 * 4 * 12
buf = (ctypes.c_ulonglong * n)()
rc = lib.mss_debug_read_stamps(buf, n)
arr = np.array(buf[:], dtype=np.float64).reshape(-1, 12)
arr = arr[arr[:, 8] > 0]
steps = arr[:, 8:9]
per = arr[:, :8] / steps
names16 = ["top (12 fragment reads)", "first touch of the A registers", "group 1 (DMA, row 0 split)", "group 2 (A loads, row 1 split)", "group 3 (bookkeeping)",
           "wait for the DMAs + LDS writes", "-", "barrier"]
names = names16 if os.environ.get("MSS_GEMM_SPLIT_MFMA", "16") != "32" else ["top (10 fragment reads)", "seg1 B LDS writes", "seg2 B loads + b_mid", "seg3 split row 0", "seg4 split row 1 + A loads + b_lo", "seg5 bookkeeping", "seg6", "barrier"]
tot = per.sum(1).mean()
clock = float(np.median(arr[:, 9] / arr[:, 10])) * 0.1          # s_memtime ticks per s_memrealtime tick (100 MHz) -> GHz
wg_ms = float(np.median(arr[:, 10])) / 1e5                      # a wave's lifetime in 100 MHz ticks -> ms
print(json.dumps({"product": [P, T, C, Ko], "kernel_ms": round(kernel_ms, 4), "tflops": round(2.0 * P * T * C * Ko / kernel_ms / 1e9, 1),
                  "wave_lifetime_ms": round(wg_ms, 4), "in_kernel_clock_GHz": round(clock, 3), "waves": int(arr.shape[0]), "steps_per_wave": float(steps.mean()), "cycles_per_K_step": round(tot, 1),
                  "note": "s_memtime ticks = shader cycles of ISSUE time; 48 MFMAs x 32 cycles = 1536 pipe cycles per wave and step, two waves share a SIMD's pipe"}))
life = arr[:, 10] / 1e5
print("  wave lifetime ms: min %.3f p10 %.3f median %.3f p90 %.3f p99 %.3f max %.3f  (kernel %.3f ms)" % (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), kernel_ms))
clk = arr[:, 9] / arr[:, 10] * 0.1
print("  in-kernel clock GHz per wave: min %.3f p10 %.3f median %.3f p90 %.3f max %.3f" % (clk.min(), np.percentile(clk, 10), np.median(clk), np.percentile(clk, 90), clk.max()))
wg = arr.reshape(-1, 4, 12)[:, 0, :]
xcd = np.arange(wg.shape[0]) % 8
print("  median lifetime by XCD (block id % 8): " + " ".join("%.3f" % np.median(wg[xcd == x, 10] / 1e5) for x in range(8)))
if "--dump" in sys.argv:
    lt = wg[:, 10] / 1e5
    print("  lifetime by block id (ms), 16 per row:")
    for r0 in range(0, min(len(lt), 128), 16):
        print("   ", " ".join("%.2f" % v for v in lt[r0:r0 + 16]))
    st = wg[:, 8]
    print("  K-steps by block id:", sorted(set(int(v) for v in st)))
for i, nm in enumerate(names):
    print(f"  {nm:36s} mean {per[:, i].mean():8.1f}  p10 {np.percentile(per[:, i], 10):8.1f}  p90 {np.percentile(per[:, i], 90):8.1f}   {100 * per[:, i].mean() / tot:5.1f} %")
