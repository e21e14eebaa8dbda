"""Pins the CPU oracle (oracle/*.py) against vectors produced by the reference itself
(tools/gen_golden.py -> tests/golden/*.npz). CPU only."""
import ast

import numpy as np
import pytest

from conftest import golden
from oracle import deepv3, loss as oloss, m2f, msda, nnops


def test_conv_classes():
    g = golden("ops")
    x = g["conv_x"]
    for tag, (stride, dil) in {"1x1": (1, 1), "3x3_d1": (1, 1), "3x3_d2": (1, 2), "3x3_d4": (1, 4), "3x3_d12": (1, 12),
                               "3x3_d24": (1, 24), "3x3_d36": (1, 36), "3x3_s2": (2, 1), "1x1_s2": (2, 1)}.items():
        w = g[f"conv_{tag}_w"]
        pad = dil if w.shape[2] == 3 else 0
        y = nnops.conv2d(x, w, stride=stride, dilation=dil, padding=pad)
        np.testing.assert_allclose(y, g[f"conv_{tag}_y"], rtol=1e-5, atol=2e-6, err_msg=tag)


def test_batchnorm_pool_upsample_lse():
    g = golden("ops")
    y, _, _ = nnops.batchnorm(g["bn_x"], g["bn_gamma"], g["bn_beta"], g["bn_rm"], g["bn_rv"], train=False)
    np.testing.assert_allclose(y, g["bn_eval_y"], rtol=1e-5, atol=1e-6)
    y, rm, rv = nnops.batchnorm(g["bn_x"], g["bn_gamma"], g["bn_beta"], g["bn_rm"], g["bn_rv"], train=True)
    np.testing.assert_allclose(y, g["bn_train_y"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(rm, g["bn_train_rm"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv, g["bn_train_rv"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(nnops.maxpool3s2(g["pool_x"]), g["pool_y"])
    for tag in "abc":
        y = nnops.upsample_bilinear_ac(g["up_x"], g[f"up_{tag}_y"].shape[2:])
        np.testing.assert_allclose(y, g[f"up_{tag}_y"], rtol=1e-5, atol=1e-6)
        gx = nnops.upsample_bilinear_ac_bwd(g[f"up_{tag}_gy"], g["up_x"].shape[2:])
        np.testing.assert_allclose(gx, g[f"up_{tag}_gx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nnops.upsample_bilinear_ac(g["up1_x"], (5, 6)), g["up1_y"], rtol=0, atol=0)
    np.testing.assert_allclose(nnops.logsumexp(g["lse_x"], 1), g["lse_y"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["eval_1x64x128", "eval_2x96x96"])
def test_deepwv3plus_eval(tag, deeplab_params):
    from multishiftseg_amd import synth
    g = golden("deepwv3plus_" + tag)
    n, h, w = (int(v) for v in g["shape"])
    img = synth.synth_image(int(g["image_seed"]), n, h, w)
    taps = {}
    score, logit = deepv3.forward(deeplab_params, img, taps=taps)
    # north_star tolerance: logits and OOD scores within 1e-3 (fp32), argmax bit-exact
    np.testing.assert_allclose(taps["m2"][:, ::8], g["m2"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(taps["x"][:, ::64], g["x"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(taps["dec1"], g["dec1"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(taps["dec2"], g["dec2"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(logit, g["logit"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(score, g["score"], rtol=0, atol=1e-3)
    label = logit.argmax(1)
    clear = g["margin"] > 1e-3           # pixels whose top-2 margin exceeds the tolerance must agree exactly
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(label[clear], g["label"][clear])


@pytest.mark.parametrize("tag", ["testpy_f64", "testpy_f32", "d30_f64", "d32_f64", "d64_f64", "d71_f64", "m8d32_f32"])
def test_msda(tag):
    g = golden("msda_" + tag)
    f64 = tag.endswith("f64")
    out = msda.forward(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"])
    gv, gl, ga = msda.backward(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"], g["grad_out"])
    # ops/test.py:43 torch.allclose defaults for fp64; :59 rtol 1e-2 atol 1e-3 for fp32 (we hold 1e-5)
    tol = dict(rtol=1e-9, atol=1e-12) if f64 else dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out, g["out"], **tol)
    np.testing.assert_allclose(gv, g["grad_value"], **tol)
    np.testing.assert_allclose(gl, g["grad_loc"], **(tol if f64 else dict(rtol=1e-3, atol=1e-4)))
    np.testing.assert_allclose(ga, g["grad_attn"], **tol)
    # the grid_sample-style composition bench.py times as the MSDA CPU baseline
    np.testing.assert_allclose(msda.forward_sampled(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"]), g["out"], **tol)
    # ... and its autograd, the comparator of the full-size GPU tests
    sv, sl, sa = msda.backward_sampled(g["value"], g["shapes"], g["starts"], g["loc"], g["attn"], g["grad_out"])
    np.testing.assert_allclose(sv, g["grad_value"], **tol)
    np.testing.assert_allclose(sl, g["grad_loc"], **(tol if f64 else dict(rtol=1e-3, atol=1e-4)))
    np.testing.assert_allclose(sa, g["grad_attn"], **tol)


def test_m2f_score():
    g = golden("m2f_score")
    s = m2f.anomaly_score(g["cls"], g["mask"], tuple(int(v) for v in g["size"]))
    np.testing.assert_allclose(s, g["score"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["deeplab_4x32x32", "m2f_4x32x32", "ratio1_4x16x16", "no_ood_4x16x16",
                                 "no_in_aug_4x16x16", "deeplab_8x48x40"])
def test_rel_contrastive_loss(tag):
    g = golden("rcl_" + tag)
    params = ast.literal_eval(str(g["params"]))
    B, C, H, W = (int(v) for v in g["shape"])
    if g["logits"].size:
        logits = g["logits"]
    else:
        logits = np.random.default_rng(int(g["seed"])).standard_normal((B, C, H, W), dtype=np.float32) * 3
    target = g["target"].astype(np.int64)
    perms = [g[f"perm{i}"].astype(np.int64) for i in range(3)]
    r = oloss.rel_contrastive_loss(logits, g["score"], target, params, perms)
    if np.isnan(g["loss"]):
        assert np.isnan(r["loss"])
    else:
        np.testing.assert_allclose(r["loss"], g["loss"], rtol=2e-6)
        np.testing.assert_allclose(r["dscore"], g["dscore"], rtol=1e-5, atol=1e-9)
        if "dlogit" in g:
            np.testing.assert_allclose(r["dlogit"], g["dlogit"], rtol=1e-4, atol=1e-8)
        else:
            np.testing.assert_allclose(r["dlogit"][:, :, ::3, ::3], g["dlogit_sub"], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(np.abs(r["dlogit"].astype(np.float64)).sum(), g["dlogit_abs_sum"], rtol=1e-5)
    np.testing.assert_array_equal(target.astype(np.uint8), g["target_mut"])


def test_ood_metrics_oracle_vs_reference():
    """8f-1: oracle/metric.py against the reference's eval_ood_measure (sklearn inside) on the golden maps."""
    from oracle import metric as ometric
    g = golden("ood_metrics")
    tags = sorted(k[:-len("_measures")] for k in g.files if k.endswith("_measures"))
    assert len(tags) == 6
    for tag in tags:
        got = ometric.eval_ood_measure(g[tag + "_score"], g[tag + "_label"].astype(np.int64))
        np.testing.assert_allclose(got, g[tag + "_measures"], rtol=0, atol=1e-12, err_msg=tag)
    assert ometric.eval_ood_measure(np.zeros((1, 4, 4), np.float32), np.zeros((1, 4, 4), np.int64)) is None
    assert ometric.eval_ood_measure(np.zeros((1, 4, 4), np.float32), np.ones((1, 4, 4), np.int64)) is None


def test_m2f_fused_chain_oracle_vs_reference_ops():
    """8f-2: mask prediction einsum -> F.interpolate(bilinear, align_corners=False) -> anomaly score."""
    g = golden("m2f_fused")
    for tag in ("x4", "ragged"):
        masks = m2f.mask_logits(g[tag + "_embed"], g[tag + "_features"])
        np.testing.assert_allclose(masks[:, ::7], g[tag + "_masks_sub"], rtol=1e-5, atol=1e-5)
        image, crop = tuple(int(v) for v in g[tag + "_image"]), tuple(int(v) for v in g[tag + "_crop"])
        up = m2f.upsample_bilinear(masks, image)
        np.testing.assert_allclose(up[:, ::9, ::3, ::3], g[tag + "_up_sub"], rtol=1e-5, atol=1e-5)
        s = m2f.anomaly_score_from_features(g[tag + "_cls"], g[tag + "_embed"], g[tag + "_features"], image, crop)
        np.testing.assert_allclose(s, g[tag + "_score"], rtol=1e-5, atol=1e-5)


# ---- the second restatement: stock torch CPU ops (oracle/deepv3_torch.py), the checker at BASELINE sizes ----------
@pytest.mark.parametrize("tag", ["eval_1x64x128", "eval_2x96x96"])
def test_torch_oracle_eval(tag, deeplab_params):
    from multishiftseg_amd import synth
    from oracle import deepv3_torch
    g = golden("deepwv3plus_" + tag)
    n, h, w = (int(v) for v in g["shape"])
    img = synth.synth_image(int(g["image_seed"]), n, h, w)
    score, logit = deepv3_torch.forward(deeplab_params, img)
    np.testing.assert_allclose(logit, g["logit"], rtol=0, atol=2e-5)     # same ATen kernels as the reference ran
    np.testing.assert_allclose(score, g["score"], rtol=0, atol=2e-5)
    s2, l2 = deepv3.forward(deeplab_params, img)                          # and the numpy restatement agrees with it
    np.testing.assert_allclose(l2, logit, rtol=0, atol=1e-3)
    np.testing.assert_allclose(s2, score, rtol=0, atol=1e-3)


def test_torch_oracle_eval_592x600(deeplab_params):
    """The big reference fixture (F(4x4) territory of the HIP path) pins the torch oracle at a BASELINE-like size."""
    from multishiftseg_amd import synth
    from oracle import deepv3_torch
    g = golden("deepwv3plus_eval_1x592x600")
    n, h, w = (int(v) for v in g["shape"])
    score, logit = deepv3_torch.forward(deeplab_params, synth.synth_image(int(g["image_seed"]), n, h, w))
    np.testing.assert_allclose(logit[:, :, ::4, ::4], g["logit_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(score[:, ::2, ::2], g["score_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(np.abs(logit.astype(np.float64)).sum(), float(g["logit_abs_sum"]), rtol=1e-6)
    clear = np.unpackbits(g["clear_bits"])[:n * h * w].reshape(n, h, w).astype(bool)
    np.testing.assert_array_equal(logit.argmax(1)[clear], g["label"][clear])


@pytest.mark.parametrize("fixture", ["deepwv3plus_train_step", "deepwv3plus_train_step_8pairs"])
def test_torch_oracle_train_step(deeplab_params, fixture):
    """Gradient oracle: the stage-2 step of the torch restatement (autograd on CPU) against the reference's own step
    (loss, gradients, running statistics) with its Dropout2d masks and permutations -- on the (2+2)-image fixture and on the
    C2 batch layout (8 originals + their 8 augmentations, 64x96)."""
    import torch
    from multishiftseg_amd import synth
    from oracle import deepv3_torch
    g = golden(fixture)
    pairs, h, w = (int(v) for v in g["shape"])
    pre = "stage2_"
    p = deepv3_torch.to_torch(deeplab_params)
    p["ood_head.weight"] = p["final.6.weight"].clone()                  # uncertainty_func_init
    names = [k for k in p if any(s in k for s in ["aspp", "bot_fine", "bot_aspp", "ood_head"]) and
             k.endswith(("weight", "bias"))]
    for k in names:
        p[k].requires_grad_(True)
    masks = {"mod6": torch.from_numpy(g[pre + "drop_mod6"]), "mod7": torch.from_numpy(g[pre + "drop_mod7"])}
    img = torch.from_numpy(synth.synth_image(int(g["image_seed"]), 2 * pairs, h, w))
    score, logit = deepv3_torch.forward_t(p, img, train=True, drop_masks=masks)
    target = g["target"].astype(np.int64)
    perms = [g[pre + f"perm{i}"].astype(np.int64) for i in range(3)]
    params = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
              "inoutaug_contras_margins_tri": [10, 5, 5]}
    r = oloss.rel_contrastive_loss(logit.detach().numpy(), score.detach().numpy(), target, params, perms)
    np.testing.assert_allclose(r["loss"], float(g[pre + "loss"]), rtol=1e-5)
    torch.autograd.backward([logit, score], [torch.from_numpy(r["dlogit"]), torch.from_numpy(r["dscore"])])
    np.testing.assert_allclose(score.detach().numpy(), g[pre + "score"], rtol=0, atol=2e-5)
    for k in [k for k in g.files if k.startswith(pre + "grad_") and not k.startswith((pre + "grad_sub_", pre + "grad_l2_"))]:
        name = k[len(pre) + 5:]
        ref = g[k]
        np.testing.assert_allclose(p[name].grad.numpy(), ref, rtol=0, atol=2e-3 * np.abs(ref).max() + 1e-12, err_msg=name)
    for k in [k for k in g.files if k.startswith(pre + "grad_l2_")]:
        name = k[len(pre) + 8:]
        np.testing.assert_allclose(p[name].grad.double().norm().item(), float(g[k]), rtol=1e-3, err_msg=name)
    for k in [k for k in g.files if k.startswith(pre + "rs_")]:
        np.testing.assert_allclose(p[k[len(pre) + 3:]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_pixel_decoder_ops_oracle():
    """GroupNorm, residual + LayerNorm (+ gradients) and half-pixel bilinear + add against ATen's outputs (ops2.npz)."""
    g = golden("ops2")
    np.testing.assert_allclose(nnops.groupnorm(g["gn_x"], 8, g["gn_gamma"], g["gn_beta"]), g["gn_y"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(nnops.add_layernorm(g["ln_a"], g["ln_b"], g["ln_gamma"], g["ln_beta"]), g["ln_y"], rtol=1e-5, atol=2e-6)
    dz, dg, db = nnops.add_layernorm_bwd(g["ln_a"], g["ln_b"], g["ln_gamma"], g["ln_gy"])
    np.testing.assert_allclose(dz, g["ln_da"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(dg, g["ln_dgamma"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, g["ln_dbeta"], rtol=1e-5, atol=1e-5)
    for tag in ("x2", "odd", "same"):
        lat = g[f"up_{tag}_lat"]
        np.testing.assert_allclose(lat + nnops.upsample_bilinear_hp(g["up_top"], lat.shape[2:]), g[f"up_{tag}_y"], rtol=1e-5, atol=2e-6)
