"""GPU parity of the fused RelContrastiveLoss against the reference's own outputs (golden, with the
reference's recorded permutations injected) and against the CPU oracle at a larger size."""
import ast

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import loss as oloss

pytestmark = pytest.mark.gpu


def run(params, logits, score, target, perms, **kw):
    from multishiftseg_amd.loss import RelContrastiveLoss
    crit = RelContrastiveLoss(params, **kw)
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    st = torch.from_numpy(score).cuda().requires_grad_(True)
    tt = torch.from_numpy(target.astype(np.int64)).cuda()
    loss = crit(lt, st, tt, perms=perms)
    if torch.isfinite(loss):
        loss.backward()
    return loss, lt.grad, st.grad, tt


@pytest.mark.parametrize("tag", ["deeplab_4x32x32", "m2f_4x32x32", "ratio1_4x16x16", "no_ood_4x16x16",
                                 "no_in_aug_4x16x16", "deeplab_8x48x40"])
def test_golden(tag):
    g = golden("rcl_" + tag)
    params = ast.literal_eval(str(g["params"]))
    B, C, H, W = (int(v) for v in g["shape"])
    logits = g["logits"] if g["logits"].size else \
        np.random.default_rng(int(g["seed"])).standard_normal((B, C, H, W), dtype=np.float32) * 3
    perms = [torch.from_numpy(g[f"perm{i}"].astype(np.int64)) for i in range(3)]
    loss, dl, ds, tt = run(params, logits, g["score"], g["target"], perms)
    if np.isnan(g["loss"]):
        assert torch.isnan(loss)      # mean of an empty tensor, reproduced not "fixed"
    else:
        np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
        np.testing.assert_allclose(ds.cpu().numpy(), g["dscore"], rtol=1e-4, atol=1e-8)
        d = dl.cpu().numpy()
        if "dlogit" in g:
            np.testing.assert_allclose(d, g["dlogit"], rtol=1e-3, atol=1e-7)
        else:
            np.testing.assert_allclose(d[:, :, ::3, ::3], g["dlogit_sub"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(np.abs(d.astype(np.float64)).sum(), g["dlogit_abs_sum"], rtol=1e-4)
    np.testing.assert_array_equal(tt.cpu().numpy().astype(np.uint8), g["target_mut"])


def test_seeded_reference_pairing_matches_oracle():
    """pairing='reference' draws torch.randperm on the CPU generator exactly as loss.py:129-131."""
    from multishiftseg_amd import synth
    rng = np.random.default_rng(77)
    B, H, W = 4, 96, 80
    logits = rng.standard_normal((B, 19, H, W), dtype=np.float32) * 3
    score = rng.standard_normal((B, H, W), dtype=np.float32) * 4
    target = synth.synth_targets(77, B // 2, H, W)
    params = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
              "inoutaug_contras_margins_tri": [10, 5, 5]}
    torch.manual_seed(5)
    loss, dl, ds, tt = run(params, logits, score, target, None)
    # replay the same three CPU permutations for the oracle
    t = target.copy()
    n_orig = int((t[:B // 2] < 99).sum()); n_aug = int((t[B // 2:] < 99).sum()); n_ood = int(((t > 99) & (t != 255)).sum())
    torch.manual_seed(5)
    perms = [torch.randperm(k).numpy() for k in (n_orig, n_aug, n_ood)]
    r = oloss.rel_contrastive_loss(logits, score, t, params, perms)
    np.testing.assert_allclose(loss.item(), r["loss"], rtol=1e-5)
    np.testing.assert_allclose(ds.cpu().numpy(), r["dscore"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(dl.cpu().numpy(), r["dlogit"], rtol=1e-3, atol=1e-7)
    np.testing.assert_array_equal(tt.cpu().numpy(), t)


def test_device_pairing_statistics():
    """pairing='device' (Feistel bijections): same terms except the two randomly paired hinges,
    which must agree with the reference pairing in expectation; every non-random term is exact."""
    from multishiftseg_amd import synth
    from multishiftseg_amd.loss import RelContrastiveLoss
    rng = np.random.default_rng(78)
    B, H, W = 4, 128, 128
    logits = torch.from_numpy(rng.standard_normal((B, 19, H, W), dtype=np.float32) * 3).cuda()
    score = torch.from_numpy(rng.standard_normal((B, H, W), dtype=np.float32) * 4).cuda()
    target = torch.from_numpy(synth.synth_targets(78, B // 2, H, W)).cuda()
    params = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
              "inoutaug_contras_margins_tri": [10, 5, 5]}
    a = RelContrastiveLoss(params, pairing="reference")
    b = RelContrastiveLoss(params, pairing="device")
    torch.manual_seed(1)
    a(logits, score, target.clone())
    sr = score.clone().requires_grad_(True)
    lb = b(logits, sr, target.clone())
    lb.backward()
    ta, tb = a.last_terms.cpu().numpy(), b.last_terms.cpu().numpy()
    np.testing.assert_allclose(tb[[1, 2, 5]], ta[[1, 2, 5]], rtol=1e-6)      # ce_orig, ce_aug, c_in: deterministic
    np.testing.assert_allclose(tb[[3, 4]], ta[[3, 4]], rtol=0.05)            # c_orig, c_aug: same expectation
    assert torch.isfinite(sr.grad).all() and abs(float(sr.grad.sum())) < 1e-3  # +coef/-coef pairs cancel


def test_odd_batch_is_refused():
    """[orig...; aug...] pairs: an odd batch has no pairing (and used to overrun the augmented-CE buffer)."""
    from multishiftseg_amd.loss import RelContrastiveLoss
    crit = RelContrastiveLoss({"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
                               "inoutaug_contras_margins_tri": [10, 5, 5]})
    lt = torch.zeros(3, 19, 8, 8, device="cuda")
    with pytest.raises(RuntimeError, match="odd"):
        crit(lt, torch.zeros(3, 8, 8, device="cuda"), torch.zeros(3, 8, 8, dtype=torch.int64, device="cuda"))


@pytest.mark.parametrize("n_sets,cap", [((5000, 4000, 900), 1 << 20), ((70000, 90000, 150000), 60000), ((300, 0, 50), 1 << 20), ((1, 1, 1), 1),
                                        ((180000, 190000, 170000), 1 << 20)])
def test_merged_pair_launch_equals_the_two_separate_ones(n_sets, cap):
    """mss_rcl_pairs_device2_f32 (round 4: both hinge terms in one launch, no atomics: thread i is the only writer of the three
    score-gradient elements pair i touches) against two mss_rcl_pairs_device_f32 calls (float atomics): the same pairs, so the
    same hinge sums (to the rounding of a differently grouped block sum) and bit-identical score gradients."""
    from multishiftseg_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(sum(n_sets))
    npx = 600000
    score = torch.randn(npx, device="cuda", generator=g) * 4
    perm = torch.randperm(npx, device="cuda", generator=g).int()
    n0, n1, n2 = n_sets
    idx = [perm[:n0].contiguous(), perm[n0:n0 + n1].contiguous(), perm[n0 + n1:n0 + n1 + n2].contiguous()]
    idx = [t if t.numel() else torch.zeros(1, dtype=torch.int32, device="cuda") for t in idx]
    n_out = torch.tensor([n0, n1, n2, 0], dtype=torch.int32, device="cuda")
    s0, m0, m1, wc = 0x1234567, 10.0, 5.0, 1.0
    c_a, c_b = torch.zeros(16, dtype=torch.float64, device="cuda"), torch.zeros(16, dtype=torch.float64, device="cuda")
    d_a, d_b = torch.zeros(npx, device="cuda"), torch.zeros(npx, device="cuda")
    for slot, margin in enumerate((m0, m1)):
        call("mss_rcl_pairs_device_f32", ptr(score), ptr(idx[slot]), ptr(idx[2]), ptr(n_out), slot, cap, s0 + 1 + slot, s0 + 7, margin,
             ptr(c_a), slot, wc, ptr(d_a))
    call("mss_rcl_pairs_device2_f32", ptr(score), ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(n_out), cap, s0 + 1, s0 + 2, s0 + 7, m0, m1,
         ptr(c_b), wc, ptr(d_b))
    assert torch.equal(d_a, d_b)
    np.testing.assert_allclose(c_b.cpu().numpy(), c_a.cpu().numpy(), rtol=2e-6, atol=0)   # float partial sums grouped by another grid
    n = min(cap, n0, n1, n2)
    assert (d_a != 0).sum().item() <= 4 * n and (n == 0 or c_a.abs().sum().item() > 0)


@pytest.mark.parametrize("case", ["random", "ties", "inf_nan", "k0", "all", "tiny", "one_exponent"])
def test_merged_selection_equals_the_nine_launch_one(case):
    """mss_rcl_select_merged_f32 (each digit's pick in front of the next byte's histogram pass: 5 launches, what the one-call loss
    runs) against mss_rcl_select_f32 (init + 4 x (histogram, pick)): the same threshold key, count below it, k and number of ties
    to take, on value sets with many exact ties at the threshold, +inf / NaN entries (ignored pixels are +inf in ce_aug), k = 0,
    k = n, a single element, and values that share sign and exponent (the first pass sees one digit)."""
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(hash(case) % 1000)
    n, ratio = 300007, 0.8
    if case == "random":
        v = torch.randn(n, device="cuda", generator=g).abs() * 3
    elif case == "ties":
        v = (torch.randint(0, 7, (n,), device="cuda", generator=g).float() * 0.25)
    elif case == "inf_nan":
        v = torch.randn(n, device="cuda", generator=g).abs()
        v[::5] = float("inf")
        v[7::1001] = float("nan")
    elif case == "k0":
        v, ratio = torch.rand(n, device="cuda", generator=g), 1e-9
    elif case == "all":
        v, ratio = torch.rand(n, device="cuda", generator=g), 1.0
    elif case == "tiny":
        n = 1
        v = torch.tensor([0.37], device="cuda")
    else:
        v = 1.0 + torch.rand(n, device="cuda", generator=g) * 0.999
    n_in = int(torch.isfinite(v).sum()) if case == "inf_nan" else n
    counters = torch.zeros(16, dtype=torch.float64, device="cuda")
    counters[2] = n_in                                    # CNT_N_IN_AUG
    hist = torch.zeros(256, dtype=torch.int32, device="cuda")
    sel_a = torch.full((8,), -1, dtype=torch.int32, device="cuda")
    call("mss_rcl_select_f32", ptr(v), n, ptr(counters), ratio, ptr(hist), ptr(sel_a))
    for zeroed in (0, 1):
        scratch = torch.full((4 * 256 + 16,), 0 if zeroed else 12345, dtype=torch.int32, device="cuda")
        sel_b = torch.full((8,), -1, dtype=torch.int32, device="cuda")
        call("mss_rcl_select_merged_f32", ptr(v), n, ptr(counters), ratio, ptr(scratch), zeroed, ptr(sel_b))
        assert sel_a[:6].tolist() == sel_b[:6].tolist(), (case, sel_a.tolist(), sel_b.tolist())
    k = sel_a[2].item()
    assert k == int(np.float32(ratio) * np.float32(n_in))
    if k > 0 and case != "inf_nan":
        kth = torch.sort(v).values[k - 1].item()
        assert sel_a[1].item() == int((v < kth).sum()) and sel_a[3].item() == k - sel_a[1].item()
