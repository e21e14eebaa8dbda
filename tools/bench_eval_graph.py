"""Eval forward (OOD-score path) at 1x1024x2048: eager launches vs one hipGraph replay."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import synth
from multishiftseg_amd.deepv3 import DeepWV3Plus
from multishiftseg_amd.trainer import GraphedEval, ood_scores

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
model = DeepWV3Plus(19)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.deepwv3plus_params(0).items()})
model = model.cuda().eval()
img = torch.randn(1, 3, H, W, device="cuda")


def wall(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = wall(lambda: ood_scores(model, img))
ge = GraphedEval(model, img.shape)
s0, l0 = ood_scores(model, img)
s1, l1 = ge(img)
assert torch.equal(s0, s1) and torch.equal(l0, l1)
graph = wall(lambda: ge(img))
print(json.dumps(dict(image=f"1x3x{H}x{W}", eager_ms=round(eager, 2), graph_ms=round(graph, 2),
                      eager_mpix_s=round(H * W / eager / 1e3, 2), graph_mpix_s=round(H * W / graph / 1e3, 2))))
