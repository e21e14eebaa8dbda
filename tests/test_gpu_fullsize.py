"""DeepLab parity at the BASELINE sizes and on the routes bench.py really takes.

  * C1 (1x3x512x1024 eval) against the stock-torch CPU restatement (oracle/deepv3_torch.py, pinned to the reference's
    own outputs by tests/test_oracle_golden.py);
  * reference-generated fixtures at 592x600 -- a size where the Winograd policy picks F(4x4,3x3) for all three ASPP rates
    by itself and nothing divides evenly -- eval forward and one stage-2 optimizer step (loss, gradients incl. the
    Winograd-domain weight gradient with the X' kept from the forward, running statistics);
  * at 2x1024x2048 (C3, what bench.py times) and 16x700x700 (C2): three dispatch routes -- Winograd + persistent GEMM
    (default), direct implicit GEMM + persistent GEMM for 1x1 (MSS_WINOGRAD=0), direct implicit GEMM for everything
    (MSS_WINOGRAD=0 MSS_GEMM=0) -- must agree on logits, scores, loss and every stage-2 gradient. They are TWO independent
    algorithms for the 3x3 layers (Winograd vs direct); the two direct routes differ only in which kernel carries the 1x1
    layers, and both kernels accumulate K in the same order on the same MFMA instruction, so they agree to the bit
    (profiles/r03/fullsize_parity.json: 0.0) -- that comparison checks dispatch and tiling, not arithmetic. Two identical
    steps must give bit-identical gradients (deterministic weight gradient);
  * fused loss and the OOD-score tail at 16x19x700x700 against the numpy oracle.
Every number that decides a bound is also written to gpurun_out/fullsize_parity.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from multishiftseg_amd import _lib

pytestmark = pytest.mark.gpu

LOSS_PARAMS = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
               "inoutaug_contras_margins_tri": [10, 5, 5]}
STAGE2 = ["aspp", "bot_fine", "bot_aspp", "ood_head"]
ROUTES = {"winograd": {}, "direct3x3": {"MSS_WINOGRAD": "0"},
          "igemm_only": {"MSS_WINOGRAD": "0", "MSS_GEMM": "0", "MSS_STEM_IM2COL": "0"},
          # r04 (direct3x3 and igemm_only share K-order and MFMA instruction, i.e. the same bits): a THIRD 3x3
          # algorithm -- every Winograd layer forced to F(2x2) (its own transform matrices, 16 products per tile, other GEMM shapes)
          "winograd_f2": {"MSS_WINO_TILE": "2"}}
# "winograd" is the policy's own mix (F(6x6) wherever it saves >= 5 % over F(4x4), else F(4x4) / F(2x2)); winograd_f4 keeps
# the policy off the 6x6 tiles so that the F(4x4) kernels stay pinned by the same fixtures. The experimental
# fp32-on-bf16-matrix-cores GEMM has to pass the same reference fixtures to be reported at all
# winograd_fast is the round-2 policy (no accuracy caps: F(6x6) wherever it is cheapest), still a supported switch
ROUTES_X = dict(ROUTES, winograd_f4={"MSS_WINO_MAX_TILE": "4"}, winograd_fast={"MSS_WINO_ACCURACY": "fast"},
                bf16x3={"MSS_GEMM_SPLIT": "1"})       # the split-bf16 GEMM route (csrc/gemm_bf16x3.hip): same fixtures, same bounds
_report = {}


def _note(key, value):
    _report[key] = value
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fullsize_parity.json"), "w") as f:
            json.dump(_report, f, indent=1, sort_keys=True)
    except OSError:
        pass


class _Env:
    def __init__(self, env):
        self.env = env

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("MSS_WINOGRAD", "MSS_GEMM", "MSS_WINO_TILE", "MSS_WINO_MAX_TILE", "MSS_STEM_IM2COL", "MSS_GEMM_SPLIT",
                                                          "MSS_WINO_ACCURACY")}
        for k in self.old:
            os.environ.pop(k, None)
        os.environ.update(self.env)
        _lib.reset_env_cache()

    def __exit__(self, *a):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        _lib.reset_env_cache()


def _new_model(deeplab_params):
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in deeplab_params.items()}, strict=True)
    return m.cuda()


@pytest.fixture(scope="module")
def model(deeplab_params):
    return _new_model(deeplab_params)


def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


# ------------------------------------------------------------------------------------------------ eval
def test_c1_eval_512x1024_vs_torch_oracle(model, deeplab_params, gemm_route):
    """BASELINE config 1: logits and OOD scores within 1e-3, argmax bit-exact where the oracle's top-2 margin > 1e-3."""
    from multishiftseg_amd import kernels as K, synth
    from oracle import deepv3_torch
    img = synth.synth_image(21, 1, 512, 1024)
    assert [K.wino_tile(64, 128, r) for r in (1, 2, 4, 12, 24, 36)] == [6, 6, 4, 6, 6, 2]      # what the policy does at C1
    model.eval()
    with torch.no_grad():
        score, logit = model(torch.from_numpy(img).cuda())
    rs, rl = deepv3_torch.forward(deeplab_params, img)
    logit, score = logit.cpu().numpy(), score.cpu().numpy()
    e_l, e_s = float(np.abs(logit - rl).max()), float(np.abs(score - rs).max())
    _note(f"c1_eval[{gemm_route}]", {"max_abs_logit_err": e_l, "max_abs_score_err": e_s})
    assert e_l < 1e-3 and e_s < 1e-3
    top2 = np.sort(rl, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(logit.argmax(1)[clear], rl.argmax(1)[clear])


@pytest.mark.parametrize("route", list(ROUTES_X))
def test_eval_golden_592x600(model, route):
    """Outputs of the reference model itself at a size where F(4x4) is the policy's own choice for dil 12/24/36 and
    F(6x6) for the undilated / dilation-2 layers."""
    from multishiftseg_amd import kernels as K, synth
    g = golden("deepwv3plus_eval_1x592x600")
    n, h, w = (int(v) for v in g["shape"])
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), n, h, w)).cuda()
    model.eval()
    with _Env(ROUTES_X[route]):
        if route == "winograd":
            assert [K.wino_tile(74, 75, r) for r in (1, 2, 4, 12, 24, 36)] == [6, 6, 4, 4, 4, 4]
        with torch.no_grad():
            score, logit = model(img)
    logit, score = logit.cpu().numpy(), score.cpu().numpy()
    e_l = float(np.abs(logit[:, :, ::4, ::4] - g["logit_sub"]).max())
    e_s = float(np.abs(score[:, ::2, ::2] - g["score_sub"]).max())
    _note(f"eval_592x600[{route}]", {"max_abs_logit_err": e_l, "max_abs_score_err": e_s})
    assert e_l < 1e-3 and e_s < 1e-3
    assert np.abs(logit[:, :, h // 3] - g["logit_row"]).max() < 1e-3
    assert np.abs(score[:, h // 3] - g["score_row"]).max() < 1e-3
    np.testing.assert_allclose(np.abs(logit.astype(np.float64)).sum(), float(g["logit_abs_sum"]), rtol=1e-5)
    np.testing.assert_allclose(np.abs(score.astype(np.float64)).sum(), float(g["score_abs_sum"]), rtol=1e-5)
    clear = np.unpackbits(g["clear_bits"])[:n * h * w].reshape(n, h, w).astype(bool)
    np.testing.assert_array_equal(logit.argmax(1)[clear], g["label"][clear])


# ------------------------------------------------------------------------------------------- train step
def _grad_close(got, ref, name, noise, sens, lo=2e-3):
    """rel-L2 <= max(lo, 2*noise, 3*sens): `sens` is what the REFERENCE's own gradient moves by when the trunk outputs
    are jittered by one Winograd layer's rounding (4e-6 relative), `noise` its fp32-vs-fp64 distance
    (tools/gen_golden.py); no fp32 implementation can be asked to be tighter than those."""
    err = np.abs(got.astype(np.float64) - ref)
    rel = float(np.sqrt((err ** 2).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30))
    bound = max(lo, 2 * noise, 3 * sens)
    return rel, bound


def _flip_report(logit, g, pre, n, h, w):
    """argmax flips against the reference's label map over ALL pixels, split by how close to a tie the reference was."""
    ref = torch.from_numpy(g[pre + "label"]).to(logit.device)
    flip = (logit.argmax(1).to(torch.uint8) != ref).cpu().numpy()
    out = {"pixels": int(flip.size), "flips_all_pixels": int(flip.sum())}
    for tag in ("1e3", "1e4", "1e5"):
        clear = np.unpackbits(g[pre + "clear_bits_" + tag])[:n * h * w].reshape(n, h, w).astype(bool)
        out[f"flips_where_ref_margin_gt_{tag}"] = int((flip & clear).sum())
        out[f"ref_pixels_with_margin_gt_{tag}"] = int(clear.sum())
    return out


def _train_step_vs_golden(deeplab_params, fixture, route, tile_check, out_bound):
    from multishiftseg_amd import kernels as K, synth
    from multishiftseg_amd.loss import RelContrastiveLoss
    g = golden(fixture)
    pairs, h, w = (int(v) for v in g["shape"])
    ss, ls = int(g["score_stride"]), int(g["logit_stride"])
    pre = "stage2_"
    m = _new_model(deeplab_params)
    m.uncertainty_func_init()
    from multishiftseg_amd.trainer import TrainStep
    crit = RelContrastiveLoss(LOSS_PARAMS)
    step = TrainStep(m, crit, stage=2)      # the product's step: fused loss route + HIP Adam (optim.py), as bench.py times it
    step.keep_outputs = True
    m.dropout_masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, h, w)).cuda()
    target = torch.from_numpy(g["target"].astype(np.int64)).cuda()
    perms = [torch.from_numpy(g[pre + f"perm{i}"].astype(np.int64)) for i in range(3)]
    with _Env(ROUTES_X[route]):
        if route == "winograd":
            h8, w8 = -(-h // 8), -(-w // 8)
            assert [K.wino_tile(h8, w8, r) for r in (1, 2, 4, 12, 24, 36)] == tile_check
        loss = step(img, target, perms=perms)
        score, logit = step.last_outputs
    e_s = float(np.abs(score.detach().cpu().numpy()[:, ::ss, ::ss] - g[pre + "score"]).max())
    e_l = float(np.abs(logit.detach().cpu().numpy()[:, :, ::ls, ::ls] - g[pre + "logit_sub"]).max())
    rep = {"score_err": e_s, "logit_err": e_l, "grads": {}}
    if pre + "label" in g.files:
        rep["argmax"] = _flip_report(logit.detach(), g, pre, 2 * pairs, h, w)
    np.testing.assert_allclose(logit.detach().double().abs().sum().item(), float(g[pre + "logit_abs_sum"]), rtol=1e-5)
    np.testing.assert_allclose(score.detach().double().abs().sum().item(), float(g[pre + "score_abs_sum"]), rtol=1e-5)
    mism = float((target.cpu().numpy().astype(np.uint8) != g[pre + "target_mut"]).mean())
    rep["target_mismatch"] = mism
    rep["loss"], rep["loss_ref"] = loss.item(), float(g[pre + "loss"])
    sd = m.state_dict()
    pd = dict(m.named_parameters())
    bad = []
    for k in [k for k in g.files if k.startswith(pre + "grad_") and not k.startswith((pre + "grad_sub_", pre + "grad_l2_"))]:
        name = k[len(pre) + 5:]
        sens = float(g[pre + "gradsens_" + name]) if pre + "gradsens_" + name in g.files else 0.0
        rel, bound = _grad_close(pd[name].grad.cpu().numpy(), g[k], name, 0.0, sens)
        rep["grads"][name] = {"rel_l2": rel, "bound": bound, "gradsens": sens}
        if rel > bound:
            bad.append((name, rel, bound))
    for k in [k for k in g.files if k.startswith(pre + "grad_l2_")]:
        name = k[len(pre) + 8:]
        sens = float(g[pre + "gradsens_" + name]) if pre + "gradsens_" + name in g.files else 0.0
        flat = pd[name].grad.cpu().numpy().reshape(pd[name].shape[0], -1)
        sub = flat[:, ::max(1, flat.shape[1] // 64)][:, :64]
        rel, bound = _grad_close(sub, g[pre + "grad_sub_" + name], name, 0.0, sens)
        l2 = pd[name].grad.double().norm().item()
        rep["grads"][name] = {"rel_l2_slice": rel, "bound": bound, "gradsens": sens, "l2": l2, "l2_ref": float(g[k])}
        if rel > bound or abs(l2 / float(g[k]) - 1) > max(2e-3, 3 * sens):
            bad.append((name, rel, bound, l2, float(g[k])))
    _note(f"{fixture}[{route}]", rep)
    assert e_s < out_bound and e_l < out_bound, (e_s, e_l)
    np.testing.assert_allclose(loss.item(), float(g[pre + "loss"]), rtol=1e-4)
    assert mism < 1e-4
    for k in [k for k in g.files if k.startswith(pre + "rs_")]:
        np.testing.assert_allclose(sd[k[len(pre) + 3:]].cpu().numpy(), g[k], rtol=1e-3, atol=1e-4, err_msg=k)
    assert not bad, bad
    return rep


@pytest.mark.parametrize("route", list(ROUTES_X))
def test_train_step_golden_2x592x600(deeplab_params, route):
    """One stage-2 optimizer step of the reference on a (1+1)x3x592x600 batch -- the per-GPU batch shape of C3 -- with
    its Dropout2d masks and loss permutations injected. In the default route every ASPP layer runs F(4x4) and its weight
    gradient consumes the X' kept by the forward."""
    _train_step_vs_golden(deeplab_params, "deepwv3plus_train_step_2x592x600", route, [6, 6, 4, 4, 4, 4], 1e-3)


# ---------------------------------------------------------------------- the headline configuration vs the REFERENCE
@pytest.mark.parametrize("route", ["winograd", "bf16x3"])
def test_eval_golden_c3_1x1024x2048(model, route):
    """BASELINE's metric resolution, eval mode (test_deeplab.py:86-96): outputs of the reference model itself
    (tools/gen_golden.py deeplab_c3) against the DEFAULT route. Bound 5e-4 = half the 1e-3 bar; argmax flips counted over
    ALL pixels, none allowed where the reference's own top-2 margin exceeds the bar."""
    from multishiftseg_amd import synth
    g = golden("deepwv3plus_eval_1x1024x2048")
    n, h, w = (int(v) for v in g["shape"])
    ss, ls = int(g["score_stride"]), int(g["logit_stride"])
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), n, h, w)).cuda()
    model.eval()
    with _Env(ROUTES_X[route]):
        with torch.no_grad():
            score, logit = model(img)
    e_l = float((logit[:, :, ::ls, ::ls].cpu() - torch.from_numpy(g["logit_sub"])).abs().max())
    e_s = float((score[:, ::ss, ::ss].cpu() - torch.from_numpy(g["score_sub"])).abs().max())
    rep = {"max_abs_logit_err": e_l, "max_abs_score_err": e_s, "argmax": _flip_report(logit, g, "", n, h, w)}
    _note("eval_c3_1x1024x2048[" + ("default" if route == "winograd" else route) + "]", rep)
    assert e_l < 5e-4 and e_s < 5e-4, rep
    assert float((logit[:, :, h // 3].cpu() - torch.from_numpy(g["logit_row"])).abs().max()) < 5e-4
    assert float((score[:, h // 3].cpu() - torch.from_numpy(g["score_row"])).abs().max()) < 5e-4
    np.testing.assert_allclose(logit.double().abs().sum().item(), float(g["logit_abs_sum"]), rtol=1e-5)
    np.testing.assert_allclose(score.double().abs().sum().item(), float(g["score_abs_sum"]), rtol=1e-5)
    assert rep["argmax"]["flips_where_ref_margin_gt_1e3"] == 0, rep


C3_FLIPS_MAX = {"winograd": 460, "bf16x3": 400}


@pytest.mark.parametrize("route", ["winograd", "bf16x3"])
def test_train_step_golden_c3_2x1024x2048(deeplab_params, route):
    """THE headline configuration (BASELINE config 3, per GPU: one orig+aug pair of 1024x2048, stage 2, train-mode BN /
    Dropout2d): one optimizer step of the reference itself (train_deeplab.py:189-204 with lib/loss.py:34-156; fixture from
    tools/gen_golden.py train_c3) against the DEFAULT route bench.py times. Logits and scores within 5e-4 of the reference
    (half the bar), loss, target mutation, running statistics, every stage-2 gradient; argmax flips over ALL 4.2 M pixels
    are counted and none is allowed where the reference's top-2 margin exceeds 1e-3."""
    rep = _train_step_vs_golden(deeplab_params, "deepwv3plus_train_step_2x1024x2048", route, [6, 6, 6, 6, 6, 4], 5e-4)
    assert rep["argmax"]["flips_where_ref_margin_gt_1e3"] == 0, rep["argmax"]
    # flips can only sit on pixels the reference itself decided by less than the fp32 noise of ANY implementation:
    # the direct-kernel route flips 118 of 4.2 M (profiles/r03/wino_attribution_*.txt)
    # <= 2 x what was observed on this fixture (native 230, split 198 of 4 194 304; profiles/r05/fullsize_parity.json) -- VERDICT r04 #6 / r05 #1d
    assert rep["argmax"]["flips_all_pixels"] <= C3_FLIPS_MAX[route], rep["argmax"]


def _stage2_step(m, img, target, masks, seed):
    """One stage-2 forward/backward (no optimizer step) -> (score, logit, loss, {name: grad})."""
    from multishiftseg_amd.loss import RelContrastiveLoss
    for n, p in m.named_parameters():
        p.requires_grad = any(s in n for s in STAGE2)
        p.grad = None
    m.train()
    m.dropout_masks = masks
    crit = RelContrastiveLoss(LOSS_PARAMS, pairing="device", seed=seed)
    score, logit = m(img)
    tgt = target.clone()
    loss = crit(logit, score, tgt).mean()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    return score.detach(), logit.detach(), float(loss), grads, tgt


@pytest.mark.parametrize("tag,pairs,h,w", [("c3_2x1024x2048", 1, 1024, 2048), ("c2_16x700x700", 8, 700, 700),
                                            ("c2_16x768x768", 8, 768, 768)])
def test_three_routes_agree_at_bench_size(deeplab_params, tag, pairs, h, w):
    """The configurations bench.py measures (c2 both as exps/DeepLab.yaml crops it, 700^2, and as BASELINE words it,
    768^2). Same weights, inputs, Dropout2d masks and device-side pair sampling on the
    routes (THREE independent 3x3 algorithms since r04: the policy's Winograd mix, F(2x2) forced everywhere, and the direct
    implicit GEMM, the last one on two launch paths that produce the same bits); running statistics are restored between runs
    so every route sees the same BatchNorm buffers."""
    from multishiftseg_amd import kernels as K, synth
    m = _new_model(deeplab_params)
    m.uncertainty_func_init()
    saved = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    n = 2 * pairs
    img = torch.randn((n, 3, h, w), device="cuda", generator=gen)
    target = torch.from_numpy(synth.synth_targets(5, pairs, h, w)).cuda()
    rng = np.random.default_rng(9)
    masks = {"mod6": torch.from_numpy(((rng.random((n, 1024)) >= 0.3) / 0.7).astype(np.float32)),
             "mod7": torch.from_numpy(((rng.random((n, 2048)) >= 0.5) / 0.5).astype(np.float32))}
    h8, w8 = -(-h // 8), -(-w // 8)
    tiles = [K.wino_tile(h8, w8, r) for r in (1, 2, 4, 12, 24, 36)]
    # the whole point: the benchmark's own tile mix -- F(6x6) in the trunk, F(6x6) / F(4x4) on the ASPP layers
    assert tiles == ([6, 6, 6, 6, 6, 4] if tag.startswith("c3") else [6, 6, 6, 4, 4, 4]), tiles
    gs = golden("deepwv3plus_train_step_2x592x600")
    sens = {k[len("stage2_gradsens_"):]: float(gs[k]) for k in gs.files if k.startswith("stage2_gradsens_")}
    out = {}
    # + the split-bf16 GEMM route as a whole step (VERDICT r05 next #1b): at 16 x 700^2 / 768^2 it needs the 64-bit per-entry bases
    # (X' beyond 4 GB) and the ROWAFF prologue (Dropout2d tiles straddling images) that only kernel-level tests covered before
    for route, env in dict(ROUTES, bf16x3=ROUTES_X["bf16x3"]).items():
        m.load_state_dict(saved)
        with _Env(env):
            out[route] = _stage2_step(m, img, target, masks, seed=4242)
        torch.cuda.synchronize()
    # bit-reproducibility of a step (deterministic weight gradients: no atomics anywhere on this path)
    m.load_state_dict(saved)
    again = _stage2_step(m, img, target, masks, seed=4242)
    nondet = [k for k, gr in out["winograd"][3].items() if not torch.equal(gr, again[3][k])]
    if not torch.equal(out["winograd"][0], again[0]) or not torch.equal(out["winograd"][1], again[1]):
        nondet.insert(0, "forward outputs (score/logit)")
    ref = out["igemm_only"]
    rep = {"wino_tiles": tiles, "nondeterministic_grads": nondet}
    bad = []
    for route in ("winograd", "direct3x3", "winograd_f2", "bf16x3"):
        s, l, loss, grads, tgt = out[route]
        e_s = float((s - ref[0]).abs().max())
        e_l = float((l - ref[1]).abs().max())
        flips = float((l.argmax(1) != ref[1].argmax(1)).float().mean())
        tm = float((tgt != ref[4]).float().mean())
        r = {"max_abs_score_diff": e_s, "max_abs_logit_diff": e_l, "argmax_flip_fraction": flips, "loss": loss,
             "loss_ref": ref[2], "target_mutation_mismatch": tm, "grad_rel_l2": {}}
        for k, gr in grads.items():
            r["grad_rel_l2"][k] = _rel_l2(gr, ref[3][k])
        rep[route] = r
        if e_s > 5e-4 or e_l > 5e-4:          # half the 1e-3 bar, between any two routes
            bad.append((route, "outputs", e_s, e_l))
        if abs(loss / ref[2] - 1) > 1e-4:
            bad.append((route, "loss", loss, ref[2]))
        for k, v in r["grad_rel_l2"].items():
            # image-pooling branch at 2 images per GPU: BatchNorm over 2 samples is sign(x0 - x1) -- its input gradient
            # is O(eps) and pure rounding noise in ANY implementation (reference gradsens confirms), so only its size is held
            # Otherwise max(6e-3, 3 x gradsens): every route differs from the others by fp32 rounding somewhere in the trunk
            # (Winograd F(6x6) ~1.2e-5 relative per layer, F(4x4) ~4e-6; the two stem formulations and summation orders
            # ~1e-6), and the REFERENCE's own gradients respond to a 4e-6 jitter of the trunk outputs with up to 8.4e-3
            # rel-L2 (bot_fine.weight; gradsens_* in deepwv3plus_train_step_2x592x600.npz, measured on the reference by
            # tools/gen_golden.py) -- cancellation in sums over 2 M pixels, not a kernel property. Measured here: F(4x4)
            # route 3e-3, F(6x6) route 7e-3 on that one tensor, <= 2e-3 on all others.
            loose = pairs == 1 and k.startswith("aspp.img_conv")
            if v > (0.5 if loose else max(6e-3, 3 * sens.get(k, 0.0))):
                bad.append((route, k, v))
    _note(f"three_routes[{tag}]", rep)
    assert not nondet, nondet
    assert not bad, bad


@pytest.mark.parametrize("n,h,w", [(16, 64, 96)])
def test_batch16_train_mode_vs_torch_oracle(model, deeplab_params, n, h, w):
    """Batch-16 BatchNorm statistics (the partial sums left by the conv/Winograd epilogues) against the oracle."""
    from multishiftseg_amd import synth
    from oracle import deepv3_torch
    img = synth.synth_image(13, n, h, w)
    rng = np.random.default_rng(6)
    masks = {"mod6": ((rng.random((n, 1024)) >= 0.3) / 0.7).astype(np.float32),
             "mod7": ((rng.random((n, 2048)) >= 0.5) / 0.5).astype(np.float32)}
    saved = {k: v.detach().clone() for k, v in model.state_dict().items()}
    try:
        model.train()
        model.dropout_masks = {k: torch.from_numpy(v) for k, v in masks.items()}
        with torch.no_grad():
            score, logit = model(torch.from_numpy(img).cuda())
        stats = {}
        rs, rl = deepv3_torch.forward(deeplab_params, img, train=True, stats_out=stats, drop_masks=masks)
        assert np.abs(logit.cpu().numpy() - rl).max() < 1e-3
        assert np.abs(score.cpu().numpy() - rs).max() < 1e-3
        sd = model.state_dict()
        for k, v in stats.items():
            np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=2e-3, atol=2e-4, err_msg=k)
    finally:
        model.dropout_masks = None
        model.load_state_dict(saved)
        model.eval()


# ------------------------------------------------------------------------------------ loss + tail at C2 size
def test_loss_16x19x700x700_vs_oracle():
    """a-5 at the size exps/DeepLab.yaml trains on: value, dscore (all), dlogit (strided slice + checksum), target
    mutation. The easiest-80 % selection runs over 8*700*700 = 3.92 M augmented pixels."""
    from multishiftseg_amd import synth
    from multishiftseg_amd.loss import RelContrastiveLoss
    from oracle import loss as oloss
    B, C, H, W = 16, 19, 700, 700
    rng = np.random.default_rng(31)
    logits = rng.standard_normal((B, C, H, W), dtype=np.float32) * 3
    score = rng.standard_normal((B, H, W), dtype=np.float32) * 4
    target = synth.synth_targets(31, B // 2, H, W)
    idx = [np.flatnonzero(m) for m in ((target[:B // 2] < 99).reshape(-1), (target[B // 2:] < 99).reshape(-1),
                                        ((target > 99) & (target != 255)).reshape(-1))]
    n = min(len(i) for i in idx)
    perms = [rng.permutation(len(i))[:n].astype(np.int64) for i in idx]
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    st = torch.from_numpy(score).cuda().requires_grad_(True)
    tt = torch.from_numpy(target.copy()).cuda()
    loss = RelContrastiveLoss(LOSS_PARAMS)(lt, st, tt, perms=[torch.from_numpy(p) for p in perms])
    loss.backward()
    tgt_ref = target.copy()
    r = oloss.rel_contrastive_loss(logits, score, tgt_ref, LOSS_PARAMS, perms)
    np.testing.assert_allclose(loss.item(), float(r["loss"]), rtol=2e-6)
    np.testing.assert_allclose(st.grad.cpu().numpy(), r["dscore"], rtol=1e-5, atol=1e-10)
    dl = lt.grad
    np.testing.assert_allclose(dl[:, :, ::7, ::5].cpu().numpy(), r["dlogit"][:, :, ::7, ::5], rtol=1e-4, atol=1e-10)
    np.testing.assert_allclose(dl.double().abs().sum().item(), np.abs(r["dlogit"].astype(np.float64)).sum(), rtol=1e-5)
    mism = float((tt.cpu().numpy() != tgt_ref).mean())
    _note("loss_16x19x700x700", {"loss": loss.item(), "target_mismatch": mism})
    assert mism < 1e-6          # only exact ties at the selection threshold may differ


def test_ood_tail_16x700x700_vs_oracle():
    """a-4 tail at C2 size: -logsumexp + x2 bilinear (align_corners=True) + NHWC->NCHW logits + argmax."""
    from multishiftseg_amd import kernels as K
    from oracle import nnops
    rng = np.random.default_rng(32)
    N, H, W = 16, 700, 700
    dec = rng.standard_normal((N, 350, 350, 48), dtype=np.float32) * 3
    act = K.Act(torch.from_numpy(dec).cuda())
    score, logit, label = K.ood_score(act.slice(20, 19), act.slice(0, 19), H, W, want_label=True)
    d1 = np.ascontiguousarray(dec[..., 0:19].transpose(0, 3, 1, 2))
    d2 = np.ascontiguousarray(dec[..., 20:39].transpose(0, 3, 1, 2))
    rs, rl = nnops.ood_score_tail(d2, d1, (H, W))
    np.testing.assert_allclose(score.cpu().numpy(), rs, rtol=0, atol=2e-5)
    np.testing.assert_allclose(logit.cpu().numpy(), rl, rtol=0, atol=2e-5)
    top2 = np.sort(rl, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    np.testing.assert_array_equal(label.cpu().numpy()[clear], rl.argmax(1)[clear])


@pytest.mark.parametrize("tag,pairs,h,w,tiles", [("c3_2x1024x2048", 1, 1024, 2048, [6, 6, 4]), ("c2_4x700x700", 2, 700, 700, [4, 4, 4])])
def test_aspp_single_read_transform_changes_no_bit(deeplab_params, monkeypatch, tag, pairs, h, w, tiles):
    """The three dilated ASPP branches get their Winograd-domain inputs from ONE kernel that reads the 4096-channel map once
    (kernels.aspp_input_transforms, round 4) -- against the three separate transforms (MSS_WINO_ASPP3=0): a stage-2
    forward/backward (X' kept for the weight gradients) and the one-image eval forward (dilations 12 / 24 in one paired GEMM)
    give identical bits, and the fused kernel is really the one taken at these sizes."""
    from multishiftseg_amd import kernels as K, synth
    m = _new_model(deeplab_params)
    m.uncertainty_func_init()
    saved = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(78)
    n = 2 * pairs
    img = torch.randn((n, 3, h, w), device="cuda", generator=gen)
    target = torch.from_numpy(synth.synth_targets(6, pairs, h, w)).cuda()
    rng = np.random.default_rng(10)
    masks = {"mod6": torch.from_numpy(((rng.random((n, 1024)) >= 0.3) / 0.7).astype(np.float32)),
             "mod7": torch.from_numpy(((rng.random((n, 2048)) >= 0.5) / 0.5).astype(np.float32))}
    h8, w8 = -(-h // 8), -(-w // 8)
    assert [K.wino_tile(h8, w8, r) for r in (12, 24, 36)] == tiles
    out, taken = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MSS_WINO_ASPP3", mode)
        prof = K.ConvProfile()
        K.set_conv_profile(prof)
        try:
            m.load_state_dict(saved)
            out[mode] = _stage2_step(m, img, target, masks, seed=99)
            m.eval()
            m.dropout_masks = None
            with torch.no_grad():
                out[mode] += tuple(m(img[:1]))
            torch.cuda.synchronize()
            taken[mode] = sum(1 for r in prof.per_launch() if r[0] == "wino_transform" and r[1][0] == "input_aspp3")
        finally:
            K.set_conv_profile(None)
    assert taken == {"1": 2, "0": 0}, taken
    a, b = out["1"], out["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])
    assert set(a[3]) == set(b[3]) and len(a[3]) == 18
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
