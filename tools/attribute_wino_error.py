#!/usr/bin/env python3
"""Per-layer attribution of the Winograd rounding error at the headline configuration (VERDICT r02, next #1c).

Train-mode forward of DeepWV3Plus at 2x3x1024x2048 (the per-GPU batch of BASELINE config 3) with the Dropout2d masks
of the reference fixture tests/golden/deepwv3plus_train_step_2x1024x2048.npz. Routes:

  direct      every 3x3 layer on the direct implicit GEMM (no Winograd)
  policy      kernels.wino_tile as shipped
  only[i]=m   every layer direct except forward 3x3 layer i, which runs F(m x m, 3x3), m in (policy tile, 4)

Each route is compared (a) with the direct route over ALL logits / scores on the device and (b) with the reference's own
outputs (strided slices + the full argmax label map of the fixture). Writes gpurun_out/wino_attribution.{json,txt}.

    python tools/attribute_wino_error.py [--fixture deepwv3plus_train_step_2x1024x2048] [--eval]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multishiftseg_amd import kernels as K, synth  # noqa: E402
from multishiftseg_amd.deepv3 import DeepWV3Plus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="deepwv3plus_train_step_2x1024x2048")
    ap.add_argument("--tiles", default="policy,4")
    ap.add_argument("--totals-only", action="store_true", help="only the direct / policy / policy<=4 rows")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", args.fixture + ".npz"))
    pairs, h, w = (int(v) for v in g["shape"])
    pre = "stage2_"
    ss, ls = int(g["score_stride"]), int(g["logit_stride"])
    params = synth.deepwv3plus_params(0)
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()})
    m = m.cuda()
    m.uncertainty_func_init()
    saved = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.train()
    m.dropout_masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, h, w)).cuda()
    ref_logit = torch.from_numpy(g[pre + "logit_sub"]).cuda()
    ref_score = torch.from_numpy(g[pre + "score"]).cuda()
    ref_label = torch.from_numpy(g[pre + "label"]).cuda()

    layers = []

    def run(choose):
        """choose(index, info) -> None | 0 | tile."""
        m.load_state_dict(saved)
        idx = [0]
        seen = []

        def hook(info):
            i = idx[0]
            idx[0] += 1
            seen.append(info)
            return choose(i, info)
        K.set_tile_hook(hook)
        try:
            with torch.no_grad():
                score, logit = m(img)
        finally:
            K.set_tile_hook(None)
        torch.cuda.synchronize()
        return score, logit, seen

    def compare(score, logit, base):
        r = {"vs_reference": {"max_abs_logit_err_sub": float((logit[:, :, ::ls, ::ls] - ref_logit).abs().max()),
                              "max_abs_score_err_sub": float((score[:, ::ss, ::ss] - ref_score).abs().max()),
                              "argmax_flips_all_pixels": int((logit.argmax(1) != ref_label).sum())}}
        if base is not None:
            d = (logit - base[1]).abs()
            r["vs_direct"] = {"max_abs_logit_diff": float(d.max()), "rms_logit_diff": float(d.double().pow(2).mean().sqrt()),
                              "max_abs_score_diff": float((score - base[0]).abs().max()),
                              "argmax_flips": int((logit.argmax(1) != base[1].argmax(1)).sum())}
        return r

    s0, l0, seen = run(lambda i, info: 0)
    layers = seen
    base = (s0, l0)
    report = {"fixture": args.fixture, "shape": [2 * pairs, h, w], "pixels": 2 * pairs * h * w,
              "direct": compare(s0, l0, None), "layers": []}
    s1, l1, _ = run(lambda i, info: None)
    report["policy"] = compare(s1, l1, base)
    s4, l4, _ = run(lambda i, info: min(info["policy_tile"], 4))
    report["policy_max_tile_4"] = compare(s4, l4, base)
    del s1, l1, s4, l4
    lines = [f"{args.fixture}: train-mode forward {2 * pairs}x3x{h}x{w}; logit/score differences in absolute units",
             f"direct   vs reference: {report['direct']['vs_reference']}",
             f"policy   vs reference: {report['policy']['vs_reference']}  vs direct: {report['policy']['vs_direct']}",
             f"policy<=4 vs reference: {report['policy_max_tile_4']['vs_reference']}  vs direct: {report['policy_max_tile_4']['vs_direct']}",
             "", f"{'#':>2} {'HxW':>10} {'dil':>3} {'Cin':>5} {'Cout':>5} {'tile':>4} | {'max|dlogit|':>11} {'rms':>9} {'flips':>6} | F(4x4): {'max':>9} {'rms':>9}"]
    for i, info in enumerate([] if args.totals_only else layers):
        ent = dict(info, index=i)
        if not K.use_winograd(info["c_in"], info["k_out"], info["stride"], None, info["policy_tile"]):
            ent["winograd"] = False
            report["layers"].append(ent)
            continue
        for tag in args.tiles.split(","):
            tile = info["policy_tile"] if tag == "policy" else int(tag)
            if tag != "policy" and tile >= info["policy_tile"]:
                continue
            s, l, _ = run(lambda j, inf, i=i, tile=tile: tile if j == i else 0)
            ent[f"only_this_layer_F{tile}"] = compare(s, l, base)["vs_direct"]
        report["layers"].append(ent)
        a = ent.get(f"only_this_layer_F{info['policy_tile']}", {})
        b = ent.get("only_this_layer_F4", {})
        lines.append(f"{i:>2} {info['H']:>4}x{info['W']:<5} {info['dil']:>3} {info['c_in']:>5} {info['k_out']:>5} {info['policy_tile']:>4} | "
                     f"{a.get('max_abs_logit_diff', 0):11.3e} {a.get('rms_logit_diff', 0):9.2e} {a.get('argmax_flips', 0):>6} | "
                     f"         {b.get('max_abs_logit_diff', 0):9.2e} {b.get('rms_logit_diff', 0):9.2e}")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    report["policy_tiles"] = [info["policy_tile"] for info in layers]
    lines.insert(1, f"MSS_WINO_ACCURACY={os.environ.get('MSS_WINO_ACCURACY', 'strict')}; policy tiles per forward 3x3 layer: {report['policy_tiles']}")
    json.dump(report, open(os.path.join(out, f"wino_attribution{args.tag}.json"), "w"), indent=1)
    open(os.path.join(out, f"wino_attribution{args.tag}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
