#!/usr/bin/env python3
"""Per-launch table of the MFMA kernels in one training step (HIP events): which layers sit below the kernel's average."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import kernels as K, synth
from multishiftseg_amd.deepv3 import DeepWV3Plus
from multishiftseg_amd.loss import RelContrastiveLoss
from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
PAIRS = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # (orig, aug) pairs: `layer_table.py 768 768 8` = the c2 workload
model = DeepWV3Plus(19)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.deepwv3plus_params(0).items()})
model = model.cuda(); model.uncertainty_func_init()
step = TrainStep(model, RelContrastiveLoss(LOSS_PARAMS, pairing="device"), stage=2)
img = torch.randn(2 * PAIRS, 3, H, W, device="cuda")
tgt = torch.from_numpy(synth.synth_targets(1, PAIRS, H, W)).cuda()
step(img, tgt.clone())
prof = K.ConvProfile(); K.set_conv_profile(prof)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); step(img, tgt.clone()); e.record(); torch.cuda.synchronize()
K.set_conv_profile(None)
rows = prof.per_launch()
tot = sum(r[2] for r in rows if r[0] != "wino_transform")     # (the transforms are inside their layer's conv_winograd row)
print(f"step {s.elapsed_time(e):.1f} ms; MFMA kernels {tot:.1f} ms in {len(rows)} launches")
print(f"{'kind':11s} {'N,H,W,C,K,R,stride,dil':34s} {'ms':>8s} {'TF/s':>7s} {'%step':>6s}   (wino_transform rows: (which, N, H, W, channels, dil, tile), TB/s of algorithmic bytes)")
for kind, tag, ms, tf in rows:
    print(f"{kind:11s} {str(tag):34s} {ms:8.3f} {tf:7.1f} {100*ms/s.elapsed_time(e):6.2f}")
