"""Does a working set that fits the 256 MB Infinity Cache stream faster than HBM? float4 copy (mss_peak_stream_f32, variant 3) of
src -> dst for growing sizes, repeated back to back (so src AND dst of the previous repetition are the cache's most recent lines)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd._lib import call, ptr
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda").normal_(); b = torch.empty(n, device="cuda")
    for variant in (3,):
        for _ in range(5):
            call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, 2048 // mb)
        s.record()
        for _ in range(reps):
            call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print(json.dumps(dict(src_MB=mb, working_set_MB=2 * mb, us=round(ms * 1e3, 1), GBs=round(8.0 * n / ms / 1e6, 1))), flush=True)
