"""Eval forward (image -> OOD score + logits) of one 1x3x1024x2048 image, 6 times: run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd.deepv3 import DeepWV3Plus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = DeepWV3Plus(19).cuda().eval()            # random-init weights: timing only
img = torch.randn(n, 3, 1024, 2048, device="cuda")
with torch.no_grad():
    for _ in range(2):
        m(img)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(6):
        m(img)
    e.record()
    torch.cuda.synchronize()
print("ms per forward", s.elapsed_time(e) / 6, "Mpix/s", n * 1024 * 2048 / (s.elapsed_time(e) / 6) / 1e3)
