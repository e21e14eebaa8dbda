"""Debug: one stage-2 TrainStep of the small fixture on both GEMM routes; where do they part?"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import golden
from multishiftseg_amd import kernels as K, synth
from multishiftseg_amd.deepv3 import DeepWV3Plus
from multishiftseg_amd.loss import RelContrastiveLoss
from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep
fixture = sys.argv[1] if len(sys.argv) > 1 else "deepwv3plus_train_step"
g = golden(fixture)
pairs, h, w = (int(v) for v in g["shape"])
params = synth.deepwv3plus_params(0)
res = {}
for route in ("native", "bf16x3"):
    K.set_gemm_route(route)
    if len(sys.argv) > 2:
        os.environ["MSS_CONV_SPLIT"] = sys.argv[2]
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, strict=True)
    m = m.cuda(); m.uncertainty_func_init()
    step = TrainStep(m, RelContrastiveLoss(LOSS_PARAMS), stage=2)
    step.keep_outputs = True
    pre = "stage2_"
    m.dropout_masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, h, w)).cuda()
    target = torch.from_numpy(g["target"].astype(np.int64)).cuda()
    perms = [torch.from_numpy(g[pre + f"perm{i}"].astype(np.int64)) for i in range(3)]
    prof = K.ConvProfile()
    loss = step(img, target, perms=perms)
    K.set_conv_profile(None)
    score, logit = step.last_outputs
    res[route] = (loss.item(), score.detach().clone(), logit.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                  [(k, t) for (k, _, _, _), t in zip(prof.records, prof.tags)])
    K.set_gemm_route(None)
a, b = res["native"], res["bf16x3"]
print("loss", a[0], b[0], "logit diff", (a[2] - b[2]).abs().max().item(), "score diff", (a[1] - b[1]).abs().max().item())
for n in a[3]:
    d = (a[3][n] - b[3][n]).double().norm().item() / (a[3][n].double().norm().item() + 1e-30)
    print(f"  {n:40s} rel diff {d:.3e}  finite {bool(torch.isfinite(b[3][n]).all())}")
print([x for x in b[4] if "wgrad" in x[0]])
# mimic the test's checks against the golden on the bf16x3 run
pre = "stage2_"
for route in ("native", "bf16x3"):
    grads = res[route][3]
    for k in [k for k in g.files if k.startswith(pre + "grad_l2_")]:
        name = k[len(pre) + 8:]
        gr = grads[name]
        got = gr.double().norm().item()
        flat = gr.cpu().numpy().reshape(gr.shape[0], -1)
        sub = flat[:, ::max(1, flat.shape[1] // 64)][:, :64]
        ref = g[pre + "grad_sub_" + name]
        rel = float(np.sqrt(((sub - ref).astype(np.float64) ** 2).sum()) / np.sqrt((ref.astype(np.float64) ** 2).sum()))
        print(route, name, "norm ratio", got / float(g[k]), "sub rel", rel, "sub finite", bool(np.isfinite(sub).all()), "absmax", float(np.abs(sub).max()), float(np.abs(ref).max()))
