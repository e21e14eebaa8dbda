"""The optimizer inside the timed step: multishiftseg_amd/optim.py -> mss_adam_step_f32 (csrc/glue.hip).

Two pins:
  * the kernel against torch.optim.Adam's own arithmetic on the CPU (single-tensor path, the one the reference's run
    here takes), several steps with non-zero moments, weight decay and zero gradients;
  * the three-step fixture of the REFERENCE loop (tools/gen_golden.py gen_train_steps3: train_deeplab.py:134-149 builds
    torch.optim.Adam, :198-204 zero_grad / backward / step), replayed through trainer.TrainStep: parameter deltas and
    both moment buffers after three steps of each stage.
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lr,wd", [(1e-4, 1e-4), (1e-6, 1e-4), (1e-3, 0.0)])
def test_adam_kernel_matches_torch_adam_over_steps(lr, wd):
    """Same gradients into both optimizers for 7 steps: moments agree to fp32 rounding of one fused multiply-add, the
    parameters to one ulp of the parameter per step taken (the update itself is ~1e3 ulps at lr 1e-6)."""
    from multishiftseg_amd.optim import Adam
    rng = np.random.default_rng(0)
    shapes = [(19, 256, 1, 1), (257, 63), (5,), (64, 48, 3, 3)]
    p0 = [rng.standard_normal(s).astype(np.float32) * 0.05 for s in shapes]
    ref_p = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in p0]
    got_p = [torch.nn.Parameter(torch.from_numpy(a.copy()).cuda()) for a in p0]
    ref = torch.optim.Adam(ref_p, lr=lr, weight_decay=wd, foreach=False)
    got = Adam(got_p, lr=lr, weight_decay=wd)
    steps = 7
    for k in range(steps):
        for i, s in enumerate(shapes):
            g = (rng.standard_normal(s) * 10.0 ** rng.uniform(-6, 0)).astype(np.float32)
            g[rng.random(s) < 0.1] = 0.0                       # exact zeros: update decided by weight decay / moments only
            if k == 3 and i == 2:
                g[:] = 0.0                                     # a whole step on moments + weight decay only
            ref_p[i].grad = torch.from_numpy(g.copy())
            got_p[i].grad = torch.from_numpy(g.copy()).cuda()
        ref.step()
        got.step()
    for i in range(len(shapes)):
        rp, gp = ref_p[i].detach().numpy(), got_p[i].detach().cpu().numpy()
        ulp = np.spacing(np.abs(rp).astype(np.float32))
        assert (np.abs(gp - rp) <= steps * ulp + 1e-3 * lr).all(), (i, np.abs(gp - rp).max(), lr)
        moved = np.abs(rp - p0[i])
        assert moved.max() > 0.5 * lr                          # the test would be vacuous on an optimizer that did nothing
        st_r, st_g = ref.state[ref_p[i]], got.state[id(got_p[i])]
        np.testing.assert_allclose(st_g[0].cpu().numpy(), st_r["exp_avg"].numpy(), rtol=2e-6, atol=1e-30)
        np.testing.assert_allclose(st_g[1].cpu().numpy(), st_r["exp_avg_sq"].numpy(), rtol=2e-6, atol=1e-37)


def test_adam_bias_corrections_are_formed_in_double():
    """Step 1 with a constant gradient: p -= lr * g/(|g| + eps*sqrt(1-b2)) ... exactly lr to fp32 rounding. Bias
    corrections formed in float (`1 - 0.999f`) are off by 1.3e-5 relative; asked for here: 2e-7."""
    from multishiftseg_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(1024, device="cuda"))
    p.grad = torch.full((1024,), 0.37, device="cuda")
    opt = Adam([p], lr=1e-4, weight_decay=0.0)
    opt.step()
    ref = torch.nn.Parameter(torch.zeros(1024))
    ref.grad = torch.full((1024,), 0.37)
    torch.optim.Adam([ref], lr=1e-4, foreach=False).step()
    np.testing.assert_allclose(p.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-7)
    for k in range(2, 40):                                      # and at later steps, where beta**step matters
        opt.step()
    ref_opt = torch.optim.Adam([ref], lr=1e-4, foreach=False)
    ref.data.zero_()
    for k in range(39):
        ref_opt.step()
    np.testing.assert_allclose(p.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6)


@pytest.mark.parametrize("stage,lr", [("stage1", 1e-4), ("stage2", 1e-6)])
def test_three_reference_steps_through_trainstep(deeplab_params, stage, lr, gemm_route):
    """Three optimizer steps of the reference loop replayed through TrainStep (fused loss route + HIP Adam): losses of all
    three steps, and after the third: parameter deltas, exp_avg and exp_avg_sq of every trainable tensor."""
    from multishiftseg_amd import synth
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    from multishiftseg_amd.loss import RelContrastiveLoss
    from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep
    g = golden("deepwv3plus_train_3steps")
    pairs, h, w = (int(v) for v in g["shape"])
    steps = int(g["steps"])
    pre = stage + "_"
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in deeplab_params.items()}, strict=True)
    m = m.cuda()
    m.uncertainty_func_init()
    step = TrainStep(m, RelContrastiveLoss(LOSS_PARAMS), stage=int(stage[-1]))
    named = {n: p for n, p in m.named_parameters() if p.requires_grad}
    before = {n: p.detach().clone() for n, p in named.items()}
    for k in range(steps):
        m.dropout_masks = {"mod6": torch.from_numpy(g[pre + f"drop_mod6_{k}"]), "mod7": torch.from_numpy(g[pre + f"drop_mod7_{k}"])}
        img = torch.from_numpy(synth.synth_image(int(g["image_seeds"][k]), 2 * pairs, h, w)).cuda()
        target = torch.from_numpy(g[f"target{k}"].astype(np.int64)).cuda()
        perms = [torch.from_numpy(g[pre + f"perm{i}_{k}"].astype(np.int64)) for i in range(3)]
        loss = step(img, target, perms=perms)
        np.testing.assert_allclose(loss.item(), float(g[pre + f"loss{k}"]), rtol=1e-4, err_msg=f"step {k}")
    assert step.optimizer.step_count == steps

    def sub(t):
        a = t.detach().cpu().numpy()
        if a.size <= 70000:
            return a
        flat = a.reshape(a.shape[0], -1)
        return flat[:, ::max(1, flat.shape[1] // 64)][:, :64]

    report = {}
    for n, p in named.items():
        m1, m2 = step.optimizer.state[id(p)]
        ref_m1, ref_m2, ref_d = g[pre + "exp_avg_" + n], g[pre + "exp_avg_sq_" + n], g[pre + "delta_" + n]
        got_m1, got_m2, got_d = sub(m1), sub(m2), sub(p.detach() - before[n])

        def rel(a, b, full):
            # against the slice's own norm, or -- where the stored slice is (nearly) zero against the rest of its tensor: the
            # dilated ASPP weights' tap (0, 0) never meets the 12 x 16 map, its moments come from the weight decay alone, and
            # a 2e-6-of-rms rounding residue of the gradient there (split-bf16 route) is 5 % of THAT -- 1e-3 of what a slice of
            # this size typically weighs in this tensor
            typical = float(full.double().norm()) * np.sqrt(b.size / full.numel())
            return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / (max(np.sqrt((b.astype(np.float64) ** 2).sum()), 1e-3 * typical) + 1e-300))
        # moments are running means of the gradients: they carry the gradients' own parity (measured round 4: stage 1 4e-5,
        # stage 2 at most 6.7e-3 / 5.3e-3 on this tiny batch, whose ASPP BatchNorm sums are ill-conditioned --
        # test_train_step_golden); bound 2e-2
        r1, r2 = rel(got_m1, ref_m1, m1), rel(got_m2, ref_m2, m2)
        # deltas: elements whose normalised step |m|/sqrt(v) is well defined, i.e. the last gradient is not rounding noise
        lastg = g[pre + "lastgrad_" + n]
        solid = np.abs(ref_m1) > 0.05 * np.sqrt(np.maximum(ref_m2, 1e-30) / (1 - 0.999 ** steps))
        # ... and not where the reference's own raw gradient is EXACTLY zero (round 5): tap (0, 0) of the dilated ASPP weights never meets
        # the 12 x 16 map, so those elements move by Adam's normalised weight decay alone, sign(1e-4 p) -- a step any rounding residue of
        # the gradient (5e-8 = 2e-6 of the tensor's rms on the split-bf16 route; the native Winograd sum cancels exactly) redirects
        # wherever |p| < 5e-4. Their moments are still compared above.
        solid &= lastg != 0
        ok = np.abs(got_d - ref_d) <= 0.05 * lr * steps
        frac = float(ok[solid].mean()) if solid.any() else 1.0
        report[n] = dict(exp_avg_rel_l2=r1, exp_avg_sq_rel_l2=r2, delta_ok_frac=frac, solid=int(solid.sum()), size=int(solid.size),
                         delta_max_err_over_lr=float(np.abs(got_d - ref_d).max() / lr))
        assert np.isfinite(lastg).all()
    import json, os
    from conftest import ROOT
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"train_3steps_{stage}.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
    except OSError:
        pass
    bad = {n: r for n, r in report.items() if r["exp_avg_rel_l2"] > 2e-2 or r["exp_avg_sq_rel_l2"] > 2e-2 or r["delta_ok_frac"] < 0.99}
    assert not bad, bad
