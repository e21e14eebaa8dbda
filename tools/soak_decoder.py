"""Soak run of the pixel decoder's training loop: K forward + backward + SGD steps at C4 (16 x 704^2) -- finite gradients, stable step
time, no growth of allocated memory (python tools/soak_decoder.py [steps])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd.msdeformattn_decoder import MSDeformAttnPixelDecoder, ShapeSpec
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
shape = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
torch.manual_seed(0)
dec = MSDeformAttnPixelDecoder({k: ShapeSpec(*v) for k, v in shape.items()}, transformer_dropout=0.0, transformer_nheads=8,
                               transformer_dim_feedforward=1024, transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                               transformer_in_features=["res3", "res4", "res5"], common_stride=4).cuda()
opt = torch.optim.SGD(dec.parameters(), lr=1e-4)
g = torch.Generator(device="cuda").manual_seed(1)
vals, mem, times = [], [], []
for i in range(steps):
    feats = {k: torch.randn(16, c, 704 // s, 704 // s, device="cuda", generator=g) for k, (c, s) in shape.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    mask, out0, ms = dec.forward_features(feats)
    loss = mask.square().mean() + sum(m.square().mean() for m in ms)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    if i % max(1, steps // 6) == 0 or i == steps - 1:
        gn = sum(float(p.grad.double().square().sum()) for p in dec.parameters() if p.grad is not None) ** 0.5
        vals.append((round(float(loss), 5), round(gn, 4))); mem.append(round(torch.cuda.memory_allocated() / 2**30, 3))
print("(loss, grad norm)", vals)
print("allocated GiB", mem, "peak", round(torch.cuda.max_memory_allocated() / 2**30, 2))
t = np.array(times[3:]) * 1e3
print(f"ms/iteration median {np.median(t):.2f} p5 {np.percentile(t, 5):.2f} p95 {np.percentile(t, 95):.2f}")
assert all(np.isfinite(v[0]) and np.isfinite(v[1]) for v in vals) and mem[-1] <= mem[1] + 0.05
