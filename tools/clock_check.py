import sys, time
sys.path.insert(0, "/root/repo")
import torch
from multishiftseg_amd._lib import call, ptr
out = torch.zeros(2, dtype=torch.int64, device="cuda")
for ticks in (1_000_000, 5_000_000):
    call("mss_peak_clock", ptr(out), 1000); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); call("mss_peak_clock", ptr(out), ticks); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e); o = out.tolist()
    print("ticks", ticks, "wall ms", round(ms, 3), "realtime MHz", round(o[0] / ms / 1e3, 2), "memtime MHz", round(o[1] / ms / 1e3, 2))
