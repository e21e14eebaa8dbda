"""OODMeter.update x 16 + compute at 1024x2048, for `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import metric as M
g = torch.Generator(device="cuda").manual_seed(0)
batches = []
for _ in range(16):
    lab = (torch.rand(1, 1024, 2048, device="cuda", generator=g) < 0.03).long()
    lab[torch.rand(1, 1024, 2048, device="cuda", generator=g) < 0.05] = 255
    batches.append((torch.randn(1, 1024, 2048, device="cuda", generator=g) + 1.2 * (lab == 1), lab))
for rep in range(3):
    m = M.OODMeter()
    for s, l in batches:
        m.update(s, l)
    m.compute()
torch.cuda.synchronize()
