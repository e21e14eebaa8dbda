"""Stage-2 golden train step under the three 3x3 paths (direct / F(2x2) / F(4x4)): per-parameter gradient
error against the reference's CPU autograd (tests/golden/deepwv3plus_train_step.npz)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import synth
from multishiftseg_amd.deepv3 import DeepWV3Plus
from multishiftseg_amd.loss import RelContrastiveLoss
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(G, "deepwv3plus_train_step.npz"))
params = synth.deepwv3plus_params(0)
pairs, h, w = (int(v) for v in g["shape"])
print("shape", pairs, h, w)
pre = "stage2_"
for mode in (("0", None), ("1", "2"), ("1", None)):      # direct / F(2x2) everywhere / tile policy
    os.environ["MSS_WINOGRAD"] = mode[0]
    os.environ.pop("MSS_WINO_TILE", None)
    if mode[1]: os.environ["MSS_WINO_TILE"] = mode[1]
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, strict=True)
    m = m.cuda(); m.uncertainty_func_init()
    for n, p in m.named_parameters():
        p.requires_grad = any(s in n for s in ["aspp", "bot_fine", "bot_aspp", "ood_head"])
    m.train()
    m.dropout_masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, h, w)).cuda()
    target = torch.from_numpy(g["target"].astype(np.int64)).cuda()
    crit = RelContrastiveLoss({"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
                               "inoutaug_contras_margins_tri": [10, 5, 5]})
    perms = [torch.from_numpy(g[pre + f"perm{i}"].astype(np.int64)) for i in range(3)]
    score, logit = m(img)
    loss = crit(logit, score, target, perms=perms).mean()
    loss.backward()
    out = {"mode": mode, "score_err": float(np.abs(score.detach().cpu().numpy() - g[pre + "score"]).max()),
           "logit_err": float(np.abs(logit.detach().cpu().numpy()[:, :, ::4, ::4] - g[pre + "logit_sub"]).max()),
           "loss": loss.item(), "ref_loss": float(g[pre + "loss"])}
    pd = dict(m.named_parameters())
    rows = {}
    for k in g.files:
        if k.startswith(pre + "grad_") and not k.startswith(pre + "grad_sub_") and not k.startswith(pre + "grad_l2_"):
            name = k[len(pre) + 5:]; got = pd[name].grad.cpu().numpy(); ref = g[k]
        elif k.startswith(pre + "grad_sub_"):
            name = k[len(pre) + 9:]
            flat = pd[name].grad.cpu().numpy().reshape(pd[name].shape[0], -1)
            got = flat[:, ::max(1, flat.shape[1] // 64)][:, :64]; ref = g[k]
        else:
            continue
        err = np.abs(got - ref)
        rows[name] = (float(np.sqrt((err.astype(np.float64) ** 2).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30)),
                      float(err.max() / (np.abs(ref).max() + 1e-12)))
    out["worst"] = sorted(((v[0], v[1], n) for n, v in rows.items()), reverse=True)[:6]
    print(json.dumps(out), flush=True)
