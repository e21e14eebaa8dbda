"""Soak run of the training step: K steps of stage 2 at 2x1024x2048 on synthetic data -- loss finite and moving, step time stable,
no growth of allocated memory (python tools/soak_train.py [steps])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import synth
from multishiftseg_amd.deepv3 import DeepWV3Plus
from multishiftseg_amd.loss import RelContrastiveLoss
from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
model = DeepWV3Plus(19)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.deepwv3plus_params(0).items()})
model = model.cuda(); model.uncertainty_func_init()
step = TrainStep(model, RelContrastiveLoss(LOSS_PARAMS, pairing="device"), stage=2)
g = torch.Generator(device="cuda").manual_seed(0)
tgt0 = torch.from_numpy(synth.synth_targets(1, 1, 1024, 2048)).cuda()
losses, mem, times = [], [], []
for i in range(steps):
    img = torch.randn(2, 3, 1024, 2048, device="cuda", generator=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = step(img, tgt0.clone())
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    if i % max(1, steps // 10) == 0 or i == steps - 1:
        losses.append(round(float(loss), 4)); mem.append(round(torch.cuda.memory_allocated() / 2**30, 3))
print("losses", losses)
print("allocated GiB", mem, "peak", round(torch.cuda.max_memory_allocated() / 2**30, 2))
t = np.array(times[5:]) * 1e3
print(f"ms/step median {np.median(t):.2f} p5 {np.percentile(t, 5):.2f} p95 {np.percentile(t, 95):.2f} max {t.max():.2f}")
assert all(np.isfinite(losses)) and mem[-1] <= mem[1] + 0.05
