#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box and is
never copied): the reference modules are imported in place, fed deterministic inputs, and only
inputs + outputs (data) are written. Weights are not stored: they come from
multishiftseg_amd.synth (counter-based generator keyed by state-dict name), so the GPU box
regenerates bit-identical parameters.

    python tools/gen_golden.py            # all fixtures
    python tools/gen_golden.py msda loss  # a subset
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from multishiftseg_amd import synth  # noqa: E402


def import_reference():
    """Import recipe from SURVEY.md 8(c): stub the import-time argparse module and the absent
    compiled extension, nothing else."""
    sys.path.insert(0, REF)
    pa = types.ModuleType("lib.configs.parse_arg")
    pa.opt = types.SimpleNamespace()
    pa.args = types.SimpleNamespace()
    sys.modules["lib.configs.parse_arg"] = pa
    from lib.network.deepv3.deepv3 import DeepWV3Plus
    import lib.loss as ref_loss
    sys.path.insert(0, os.path.join(REF, "lib/network/mask2former/modeling/pixel_decoder"))
    sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    from ops.modules.ms_deform_attn import MSDeformAttn
    return DeepWV3Plus, ref_loss, ms_deform_attn_core_pytorch, MSDeformAttn


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def t2n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------- ops
def gen_ops():
    """Per-op vectors for every distinct conv class on the path + BN/pool/upsample/logsumexp."""
    rng = np.random.default_rng(11)
    out = {}
    x = rng.standard_normal((2, 16, 20, 24), dtype=np.float32)
    out["conv_x"] = x
    for tag, (r, stride, dil) in {"1x1": (1, 1, 1), "3x3_d1": (3, 1, 1), "3x3_d2": (3, 1, 2), "3x3_d4": (3, 1, 4),
                                  "3x3_d12": (3, 1, 12), "3x3_d24": (3, 1, 24), "3x3_d36": (3, 1, 36),
                                  "3x3_s2": (3, 2, 1), "1x1_s2": (1, 2, 1)}.items():
        w = rng.standard_normal((8, 16, r, r), dtype=np.float32) * 0.1
        pad = dil if r == 3 else 0
        y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=stride, dilation=dil, padding=pad)
        out[f"conv_{tag}_w"] = w
        out[f"conv_{tag}_y"] = t2n(y)
    # BatchNorm2d train / eval
    bx = rng.standard_normal((3, 8, 7, 9), dtype=np.float32) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(8)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 8).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.standard_normal(8).astype(np.float32)))
        bn.running_mean.copy_(torch.from_numpy(rng.standard_normal(8).astype(np.float32) * 0.1))
        bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 8).astype(np.float32)))
    out.update(bn_x=bx, bn_gamma=t2n(bn.weight), bn_beta=t2n(bn.bias), bn_rm=t2n(bn.running_mean).copy(),
               bn_rv=t2n(bn.running_var).copy())
    bn.eval()
    out["bn_eval_y"] = t2n(bn(torch.from_numpy(bx)))
    bn.train()
    out["bn_train_y"] = t2n(bn(torch.from_numpy(bx)))
    out["bn_train_rm"] = t2n(bn.running_mean)
    out["bn_train_rv"] = t2n(bn.running_var)
    # MaxPool2d(3,2,1), odd and even sizes
    px = rng.standard_normal((2, 4, 11, 14), dtype=np.float32)
    out["pool_x"] = px
    out["pool_y"] = t2n(F.max_pool2d(torch.from_numpy(px), 3, stride=2, padding=1))
    # bilinear align_corners=True: x2-ish, x4-ish (88->350 style), 1x1 -> hxw
    ux = rng.standard_normal((2, 4, 11, 13), dtype=np.float32)
    out["up_x"] = ux
    for tag, size in {"a": (22, 26), "b": (41, 50), "c": (11, 13)}.items():
        xt = torch.from_numpy(ux).requires_grad_(True)
        y = F.interpolate(xt, size=size, mode="bilinear", align_corners=True)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape), dtype=np.float32))
        y.backward(gy)
        out[f"up_{tag}_y"] = t2n(y)
        out[f"up_{tag}_gy"] = t2n(gy)
        out[f"up_{tag}_gx"] = t2n(xt.grad)
    u1 = rng.standard_normal((2, 4, 1, 1), dtype=np.float32)
    out["up1_x"] = u1
    out["up1_y"] = t2n(F.interpolate(torch.from_numpy(u1), size=(5, 6), mode="bilinear", align_corners=True))
    lx = rng.standard_normal((2, 19, 5, 6), dtype=np.float32) * 4
    out["lse_x"] = lx
    out["lse_y"] = t2n(torch.logsumexp(torch.from_numpy(lx), dim=1))
    save("ops", **out)


def gen_ops2():
    """ATen ops the Mask2Former pixel decoder is made of (msdeformattn.py:116-131,215-219,344): GroupNorm(32), residual +
    LayerNorm (with autograd gradients), bilinear align_corners=False up-sampling + add."""
    rng = np.random.default_rng(12)
    out = {}
    x = (rng.standard_normal((2, 64, 7, 9), dtype=np.float32) * 2 + 0.3)
    gn = torch.nn.GroupNorm(8, 64)
    with torch.no_grad():
        gn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32)))
        gn.bias.copy_(torch.from_numpy(rng.standard_normal(64).astype(np.float32)))
    out.update(gn_x=x, gn_gamma=t2n(gn.weight), gn_beta=t2n(gn.bias), gn_y=t2n(gn(torch.from_numpy(x))))
    a = torch.from_numpy(rng.standard_normal((3, 11, 256), dtype=np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal((3, 11, 256), dtype=np.float32)).requires_grad_(True)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 256).astype(np.float32)))
        ln.bias.copy_(torch.from_numpy(rng.standard_normal(256).astype(np.float32)))
    y = ln(a + b)
    gy = torch.from_numpy(rng.standard_normal((3, 11, 256), dtype=np.float32))
    y.backward(gy)
    out.update(ln_a=t2n(a), ln_b=t2n(b), ln_gamma=t2n(ln.weight), ln_beta=t2n(ln.bias), ln_y=t2n(y), ln_gy=t2n(gy),
               ln_da=t2n(a.grad), ln_dgamma=t2n(ln.weight.grad), ln_dbeta=t2n(ln.bias.grad))
    top = rng.standard_normal((2, 8, 5, 7), dtype=np.float32)
    for tag, size in {"x2": (10, 14), "odd": (11, 13), "same": (5, 7)}.items():
        lat = rng.standard_normal((2, 8) + size, dtype=np.float32)
        out[f"up_{tag}_lat"] = lat
        out[f"up_{tag}_y"] = t2n(torch.from_numpy(lat) + F.interpolate(torch.from_numpy(top), size=size, mode="bilinear", align_corners=False))
    out["up_top"] = top
    save("ops2", **out)


# --------------------------------------------------------------------------------------- DeepLab
def build_ref_model(DeepWV3Plus, seed=0):
    torch.manual_seed(0)
    model = DeepWV3Plus(19)
    sd = model.state_dict()
    shapes = synth.deepwv3plus_param_shapes(19)
    assert list(sd.keys()) == list(shapes.keys()), "state-dict names/order differ from synth"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    params = synth.deepwv3plus_params(seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}, strict=True)
    return model


def gen_deeplab(DeepWV3Plus):
    model = build_ref_model(DeepWV3Plus)
    model.eval()
    for tag, (n, h, w) in {"eval_1x64x128": (1, 64, 128), "eval_2x96x96": (2, 96, 96)}.items():
        img = synth.synth_image(1, n, h, w)
        taps = {}
        hooks = [
            model.mod2.register_forward_hook(lambda m, i, o: taps.__setitem__("m2", t2n(o))),
            model.mod7.register_forward_hook(lambda m, i, o: taps.__setitem__("x", t2n(o))),
            model.aspp.register_forward_hook(lambda m, i, o: taps.__setitem__("aspp", t2n(o))),
            model.final[5].register_forward_hook(lambda m, i, o: taps.__setitem__("feature", t2n(o))),
            model.final[6].register_forward_hook(lambda m, i, o: taps.__setitem__("dec1", t2n(o))),
            model.ood_head.register_forward_hook(lambda m, i, o: taps.__setitem__("dec2", t2n(o))),
        ]
        with torch.no_grad():
            score, logit = model(torch.from_numpy(img))
        for hk in hooks:
            hk.remove()
        logit_n = t2n(logit)
        top2 = np.sort(logit_n, axis=1)[:, -2:]
        save("deepwv3plus_" + tag, image_seed=np.int64(1), shape=np.array([n, h, w]), score=t2n(score),
             logit=logit_n, label=logit_n.argmax(1).astype(np.uint8), margin=(top2[:, 1] - top2[:, 0]),
             m2=taps["m2"][:, ::8], x=taps["x"][:, ::64], aspp=taps["aspp"][:, ::16],
             feature=taps["feature"][:, ::16], dec1=taps["dec1"], dec2=taps["dec2"],
             x_absmax=np.float32(np.abs(taps["x"]).max()), feature_absmax=np.float32(np.abs(taps["feature"]).max()))
        print(f"   {tag}: |x|max {np.abs(taps['x']).max():.3g}  |feature|max {np.abs(taps['feature']).max():.3g} "
              f" logit range [{logit_n.min():.3g},{logit_n.max():.3g}]  score range [{t2n(score).min():.3g},"
              f"{t2n(score).max():.3g}]  min top-2 margin {(top2[:, 1] - top2[:, 0]).min():.3g}")
    return model


def margin_bits(logit_n):
    """Top-2 margin of the reference's logits as three packed bit planes (> 1e-3, > 1e-4, > 1e-5): lets a test count
    argmax flips over ALL pixels and say how close to a tie the reference itself was where they happen."""
    top2 = np.partition(logit_n, -2, axis=1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    return {f"clear_bits_{tag}": np.packbits(margin > thr) for tag, thr in (("1e3", 1e-3), ("1e4", 1e-4), ("1e5", 1e-5))}


def gen_deeplab_big(DeepWV3Plus, n=1, h=592, w=600, score_stride=2, logit_stride=4, all_margins=False):
    """Eval forward at a size where the build's Winograd policy picks F(4x4,3x3) for all three ASPP rates on its own
    (the /8 map is 74x75: 4x4 / 4x4 / 3x3 residue sub-grids at dilation 12 / 24 / 36) and nothing divides evenly
    (296x300 -> 148x150 -> 74x75). Big outputs are stored as strided slices + float64 checksums."""
    model = build_ref_model(DeepWV3Plus)
    model.eval()
    img = synth.synth_image(3, n, h, w)
    with torch.no_grad():
        score, logit = model(torch.from_numpy(img))
    logit_n, score_n = t2n(logit), t2n(score)
    top2 = np.sort(logit_n, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    extra = dict(margin_bits(logit_n), score_stride=np.int64(score_stride), logit_stride=np.int64(logit_stride)) if all_margins else {}
    save(f"deepwv3plus_eval_{n}x{h}x{w}", image_seed=np.int64(3), shape=np.array([n, h, w]), **extra,
         score_sub=score_n[:, ::score_stride, ::score_stride], logit_sub=logit_n[:, :, ::logit_stride, ::logit_stride],
         label=logit_n.argmax(1).astype(np.uint8),
         clear_bits=np.packbits(clear), score_abs_sum=np.float64(np.abs(score_n.astype(np.float64)).sum()),
         logit_abs_sum=np.float64(np.abs(logit_n.astype(np.float64)).sum()),
         logit_row=logit_n[:, :, h // 3], score_row=score_n[:, h // 3])
    print(f"   eval {n}x{h}x{w}: logit range [{logit_n.min():.3g},{logit_n.max():.3g}], clear pixels {clear.mean():.4f}")


def gen_train_step(DeepWV3Plus, ref_loss, pairs=2, h=96, w=128, fixture="deepwv3plus_train_step",
                   stages=("stage1", "stage2"), fp64_replay=True, score_stride=1, logit_stride=4, truncate_perms=False,
                   store_labels=False):
    """a-7: one optimizer step of each training stage on a (pairs+pairs)x3xhxw batch (default (2+2)x3x96x128),
    train-mode BN on the frozen trunk, Dropout2d masks and the loss permutations recorded and stored."""
    img = synth.synth_image(2, 2 * pairs, h, w)
    target = synth.synth_targets(2, pairs, h, w)
    loss_params = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
                   "inoutaug_contras_margins_tri": [10, 5, 5]}        # exps/DeepLab.yaml:30-35
    out = dict(image_seed=np.int64(2), shape=np.array([pairs, h, w]), target=target.astype(np.uint8))
    out["score_stride"], out["logit_stride"] = np.int64(score_stride), np.int64(logit_stride)
    for stage, (names, lr) in {"stage1": (["ood_head"], 1e-4),
                               "stage2": (["aspp", "bot_fine", "bot_aspp", "ood_head"], 1e-6)}.items():
        if stage not in stages:
            continue
        model = build_ref_model(DeepWV3Plus)
        model.uncertainty_func_init()                                   # train_deeplab.py:108-111
        params = []
        for name, p in model.named_parameters():                       # train_deeplab.py:124-130
            p.requires_grad = any(s in name for s in names)
            if p.requires_grad:
                params.append(p)
        opt = torch.optim.Adam(params, lr=lr, weight_decay=1e-4)       # train_deeplab.py:145-149
        model.train()
        # record Dropout2d masks: replace the two dropout modules' forward by an explicit mask
        masks = {}
        torch.manual_seed(123)

        def make_drop(key, p):
            def fwd(x):
                m = (torch.rand(x.shape[0], x.shape[1]) >= p).float() / (1 - p)
                masks[key] = t2n(m)
                return x * m[:, :, None, None]
            return fwd
        model.mod6.block1.convs.dropout.forward = make_drop("mod6", 0.3)
        model.mod7.block1.convs.dropout.forward = make_drop("mod7", 0.5)
        perms = []
        real_randperm = torch.randperm

        def logging_randperm(n, *a, **k):
            p = real_randperm(n, *a, **k)
            perms.append(t2n(p))
            return p
        torch.randperm = logging_randperm
        try:
            crit = ref_loss.RelContrastiveLoss(loss_params)
            tgt = torch.from_numpy(target.copy())
            before = {n_: p.detach().clone() for n_, p in model.named_parameters() if p.requires_grad}
            score, logit = model(torch.from_numpy(img))
            loss = crit(logit, score, tgt).mean()
            opt.zero_grad()
            loss.backward()
            grads = {n_: p.grad.detach().clone() for n_, p in model.named_parameters() if p.requires_grad}
            opt.step()
        finally:
            torch.randperm = real_randperm
        pre = stage + "_"
        out[pre + "loss"] = t2n(loss)
        out[pre + "score"] = t2n(score)[:, ::score_stride, ::score_stride]
        out[pre + "logit_sub"] = t2n(logit)[:, :, ::logit_stride, ::logit_stride]
        out[pre + "score_abs_sum"] = np.float64(np.abs(t2n(score).astype(np.float64)).sum())
        out[pre + "logit_abs_sum"] = np.float64(np.abs(t2n(logit).astype(np.float64)).sum())
        out[pre + "target_mut"] = t2n(tgt).astype(np.uint8)
        if store_labels:
            out[pre + "label"] = t2n(logit).argmax(1).astype(np.uint8)
            out.update({pre + k: v for k, v in margin_bits(t2n(logit)).items()})
        out[pre + "drop_mod6"] = masks["mod6"]
        out[pre + "drop_mod7"] = masks["mod7"]
        n_used = min(len(p) for p in perms)        # loss.py:149-156 keeps only the first n = min(set sizes) entries
        for i, p in enumerate(perms):
            out[pre + f"perm{i}"] = (p[:n_used] if truncate_perms else p).astype(np.int32)
        sd = model.state_dict()
        for k in ("mod2.block1.bn1.0.running_mean", "mod2.block1.bn1.0.running_var",
                  "mod7.block1.convs.bn3.0.running_mean", "mod7.block1.convs.bn3.0.running_var",
                  "aspp.features.3.1.running_mean", "aspp.features.3.1.running_var",
                  "aspp.img_conv.1.running_var", "final.4.running_mean", "final.4.running_var"):
            out[pre + "rs_" + k] = t2n(sd[k])
        for n_, g in grads.items():
            gn = t2n(g)
            if gn.size > 70000:   # keep a slice of the big ones + a checksum
                flat = gn.reshape(gn.shape[0], -1)
                out[pre + "grad_sub_" + n_] = flat[:, ::max(1, flat.shape[1] // 64)][:, :64].copy()
                out[pre + "grad_l2_" + n_] = np.float64(np.sqrt((gn.astype(np.float64) ** 2).sum()))
            else:
                out[pre + "grad_" + n_] = gn
        for n_, b in before.items():
            delta = t2n(dict(model.named_parameters())[n_].detach() - b)
            if delta.size <= 70000:
                out[pre + "delta_" + n_] = delta
        if stage == "stage2" and fp64_replay:
            # conditioning of the small gradients: the same step of the reference in float64 (same masks, same
            # permutations). rel-L2(fp32 reference, fp64 reference) per tensor is the noise floor a parity test
            # can ask of any fp32 implementation; stored as gradnoise_<name>.
            model64 = build_ref_model(DeepWV3Plus).double()
            model64.uncertainty_func_init()
            model64.ood_head.double()
            for name, p in model64.named_parameters():
                p.requires_grad = any(s in name for s in names)
            model64.train()
            model64.mod6.block1.convs.dropout.forward = lambda x: x * torch.from_numpy(masks["mod6"]).double()[:, :, None, None]
            model64.mod7.block1.convs.dropout.forward = lambda x: x * torch.from_numpy(masks["mod7"]).double()[:, :, None, None]
            replay = [torch.from_numpy(p_.astype(np.int64)) for p_ in perms]
            torch.randperm = lambda n, *a, **k: replay.pop(0)
            try:
                score64, logit64 = model64(torch.from_numpy(img).double())
                loss64 = ref_loss.RelContrastiveLoss(loss_params)(logit64, score64, torch.from_numpy(target.copy())).mean()
                loss64.backward()
            finally:
                torch.randperm = real_randperm
            for n_, p in model64.named_parameters():
                if p.requires_grad and n_ in grads and grads[n_].numel() <= 70000:
                    g32, g64 = grads[n_].double().numpy(), p.grad.numpy()
                    out[pre + "gradnoise_" + n_] = np.float64(np.sqrt(((g32 - g64) ** 2).sum()) / (np.sqrt((g64 ** 2).sum()) + 1e-300))
        if stage == "stage2":
            # sensitivity of the same gradients to forward rounding: the fp32 reference again, with the two tensors
            # the decoder consumes (mod7 and mod2 outputs) multiplied by (1 + 4e-6 * N(0,1)) -- the size of one
            # Winograd F(4x4,3x3) layer's fp32 error. rel-L2 change per tensor = gradsens_<name>.
            model_p = build_ref_model(DeepWV3Plus)
            model_p.uncertainty_func_init()
            for name, p in model_p.named_parameters():
                p.requires_grad = any(s in name for s in names)
            model_p.train()
            model_p.mod6.block1.convs.dropout.forward = lambda x: x * torch.from_numpy(masks["mod6"])[:, :, None, None]
            model_p.mod7.block1.convs.dropout.forward = lambda x: x * torch.from_numpy(masks["mod7"])[:, :, None, None]
            gen_p = torch.Generator().manual_seed(99)
            jitter = lambda m, i, o: o * (1 + 4e-6 * torch.randn(o.shape, generator=gen_p))
            hooks = [model_p.mod7.register_forward_hook(jitter), model_p.mod2.register_forward_hook(jitter)]
            replay = [torch.from_numpy(p_.astype(np.int64)) for p_ in perms]
            torch.randperm = lambda n, *a, **k: replay.pop(0)
            try:
                score_p, logit_p = model_p(torch.from_numpy(img))
                ref_loss.RelContrastiveLoss(loss_params)(logit_p, score_p, torch.from_numpy(target.copy())).mean().backward()
            finally:
                torch.randperm = real_randperm
                for hk in hooks:
                    hk.remove()
            for n_, p in model_p.named_parameters():
                if p.requires_grad and n_ in grads:
                    g0, g1 = grads[n_].double().numpy(), p.grad.double().numpy()
                    if g0.size > 70000:      # the same slice the gradient itself is stored as
                        g0, g1 = (g.reshape(g.shape[0], -1)[:, ::max(1, g[0].size // 64)][:, :64] for g in (g0, g1))
                    out[pre + "gradsens_" + n_] = np.float64(np.sqrt(((g0 - g1) ** 2).sum()) / (np.sqrt((g0 ** 2).sum()) + 1e-300))
            sens = sorted(((float(v), k) for k, v in out.items() if k.startswith(pre + "gradsens_")), reverse=True)[:6]
            print(f"   stage2 jittered replay (4e-6 relative noise on mod7/mod2 outputs): largest gradient rel-L2 change: {sens}")
            if fp64_replay:
                noisy = sorted(((float(v), k) for k, v in out.items() if k.startswith(pre + "gradnoise_")), reverse=True)[:5]
                print(f"   stage2 float64 replay: loss {float(loss64):.6f}; largest fp32-vs-fp64 gradient rel-L2: {noisy}")
        print(f"   {stage}: loss {float(loss):.6f}, {len(grads)} trainable tensors, perms {[len(p) for p in perms]}")
    save(fixture, **out)


def gen_train_steps3(DeepWV3Plus, ref_loss, pairs=2, h=96, w=128, steps=3, fixture="deepwv3plus_train_3steps"):
    """a-7, optimizer state: `steps` consecutive optimizer steps of each stage of the reference loop (train_deeplab.py:
    134-149 builds torch.optim.Adam; :198-204 zero_grad / backward / step) on a fresh (pairs+pairs)x3xhxw batch per step,
    Dropout2d masks and loss permutations recorded per step. Stored: per-step loss, final parameter deltas (small tensors
    whole, big ones as the usual 64-column slice) and Adam's exp_avg / exp_avg_sq after the last step -- moments are
    non-zero from step 2 on, which the one-step fixture cannot exercise."""
    loss_params = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
                   "inoutaug_contras_margins_tri": [10, 5, 5]}
    out = dict(shape=np.array([pairs, h, w]), steps=np.int64(steps), image_seeds=np.arange(20, 20 + steps))
    for k in range(steps):
        out[f"target{k}"] = synth.synth_targets(20 + k, pairs, h, w).astype(np.uint8)
    real_randperm = torch.randperm
    for stage, (names, lr) in {"stage1": (["ood_head"], 1e-4),
                               "stage2": (["aspp", "bot_fine", "bot_aspp", "ood_head"], 1e-6)}.items():
        pre = stage + "_"
        model = build_ref_model(DeepWV3Plus)
        model.uncertainty_func_init()
        params = []
        for name, p in model.named_parameters():
            p.requires_grad = any(s_ in name for s_ in names)
            if p.requires_grad:
                params.append(p)
        opt = torch.optim.Adam(params, lr=lr, weight_decay=1e-4)
        model.train()
        masks = {}
        torch.manual_seed(321)

        def make_drop(key, p):
            def fwd(x):
                m = (torch.rand(x.shape[0], x.shape[1]) >= p).float() / (1 - p)
                masks[key] = t2n(m)
                return x * m[:, :, None, None]
            return fwd
        model.mod6.block1.convs.dropout.forward = make_drop("mod6", 0.3)
        model.mod7.block1.convs.dropout.forward = make_drop("mod7", 0.5)
        before = {n_: p.detach().clone() for n_, p in model.named_parameters() if p.requires_grad}
        crit = ref_loss.RelContrastiveLoss(loss_params)
        for k in range(steps):
            perms = []

            def logging_randperm(n, *a, **kw):
                p = real_randperm(n, *a, **kw)
                perms.append(t2n(p))
                return p
            torch.randperm = logging_randperm
            try:
                img = torch.from_numpy(synth.synth_image(20 + k, 2 * pairs, h, w))
                tgt = torch.from_numpy(out[f"target{k}"].astype(np.int64))
                score, logit = model(img)
                loss = crit(logit, score, tgt).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
            finally:
                torch.randperm = real_randperm
            out[pre + f"loss{k}"] = t2n(loss)
            out[pre + f"drop_mod6_{k}"], out[pre + f"drop_mod7_{k}"] = masks["mod6"], masks["mod7"]
            for i, p in enumerate(perms):
                out[pre + f"perm{i}_{k}"] = p.astype(np.int32)
            print(f"   {stage} step {k}: loss {float(loss):.6f}")

        def sub(a):
            if a.size <= 70000:
                return a
            flat = a.reshape(a.shape[0], -1)
            return flat[:, ::max(1, flat.shape[1] // 64)][:, :64].copy()
        named = dict(model.named_parameters())
        for n_, b in before.items():
            out[pre + "delta_" + n_] = sub(t2n(named[n_].detach() - b))
            st = opt.state[named[n_]]
            assert int(st["step"]) == steps
            out[pre + "exp_avg_" + n_] = sub(t2n(st["exp_avg"]))
            out[pre + "exp_avg_sq_" + n_] = sub(t2n(st["exp_avg_sq"]))
            out[pre + "lastgrad_" + n_] = sub(t2n(named[n_].grad))
    save(fixture, **out)


# ------------------------------------------------------------------------------------------ loss
def gen_loss(ref_loss):
    def run(tag, B, H, W, params, seed, tmod=None, full=True):
        rng = np.random.default_rng(seed)
        logits = (rng.standard_normal((B, 19, H, W), dtype=np.float32) * 3)
        score = rng.standard_normal((B, H, W), dtype=np.float32) * 4
        target = synth.synth_targets(seed, B // 2, H, W)
        if tmod is not None:
            target = tmod(target)
        perms = []
        real_randperm = torch.randperm

        def logging_randperm(n, *a, **k):
            p = real_randperm(n, *a, **k)
            perms.append(t2n(p))
            return p
        torch.randperm = logging_randperm
        try:
            torch.manual_seed(seed)
            lt = torch.from_numpy(logits).requires_grad_(True)
            st = torch.from_numpy(score).requires_grad_(True)
            tt = torch.from_numpy(target.copy())
            crit = ref_loss.RelContrastiveLoss(params)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                loss = crit(lt, st, tt)
            if torch.isfinite(loss):
                loss.backward()
                dl, ds = t2n(lt.grad), t2n(st.grad)
            else:
                dl, ds = np.zeros_like(logits), np.zeros_like(score)
        finally:
            torch.randperm = real_randperm
        arrays = dict(logits=logits if full else np.zeros(0, np.float32), score=score, target=target.astype(np.uint8),
                      seed=np.int64(seed), shape=np.array([B, 19, H, W]), loss=t2n(loss), dscore=ds,
                      target_mut=t2n(tt).astype(np.uint8), params=np.array(repr(params)))
        arrays["dlogit" if full else "dlogit_sub"] = dl if full else dl[:, :, ::3, ::3].copy()
        arrays["dlogit_abs_sum"] = np.float64(np.abs(dl.astype(np.float64)).sum())
        for i, p in enumerate(perms):
            arrays[f"perm{i}"] = p.astype(np.int32)
        print(f"   {tag}: loss {float(loss.detach()):.6f}  perms {[len(p) for p in perms]}")
        save("rcl_" + tag, **arrays)

    deeplab = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
               "inoutaug_contras_margins_tri": [10, 5, 5]}                       # exps/DeepLab.yaml
    m2f = {"ce_weights": [0, 0], "conduct_pixel_selection": False, "selection_ratio": 1.0,
           "inoutaug_contras_margins_tri": [0.7, 0.5, 0.2]}                       # exps/M2F.yaml
    run("deeplab_4x32x32", 4, 32, 32, deeplab, 5)
    run("deeplab_8x48x40", 8, 48, 40, deeplab, 6, full=False)
    run("m2f_4x32x32", 4, 32, 32, m2f, 7)
    run("ratio1_4x16x16", 4, 16, 16, dict(deeplab, selection_ratio=1.0), 8)
    run("no_ood_4x16x16", 4, 16, 16, deeplab, 9, tmod=lambda t: np.where(t == 254, 3, t))      # NaN loss
    def void_aug(t):                                                                            # select_num == 0
        t = t.copy(); t[t.shape[0] // 2:] = np.where(t[t.shape[0] // 2:] < 99, 255, t[t.shape[0] // 2:]); return t
    run("no_in_aug_4x16x16", 4, 16, 16, deeplab, 10, tmod=void_aug)


# ------------------------------------------------------------------------------------------ MSDA
def gen_msda(core, MSDeformAttn):
    def case(tag, N, M, D, Lq, P, shapes, seed, dtype, loc_range=(0.0, 1.0), value_scale=0.01):
        torch.manual_seed(seed)
        shp = torch.as_tensor(shapes, dtype=torch.long)
        L = len(shapes)
        starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
        S = int(shp.prod(1).sum())
        value = (torch.rand(N, S, M, D) * value_scale).to(dtype).requires_grad_(True)
        lo, hi = loc_range
        loc = (torch.rand(N, Lq, M, L, P, 2) * (hi - lo) + lo).to(dtype).requires_grad_(True)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype).requires_grad_(True)
        out = core(value, shp, loc, attn)
        gout = torch.randn(out.shape, dtype=dtype)
        out.backward(gout)
        save("msda_" + tag, value=t2n(value), shapes=t2n(shp), starts=t2n(starts), loc=t2n(loc), attn=t2n(attn),
             out=t2n(out), grad_out=t2n(gout), grad_value=t2n(value.grad), grad_loc=t2n(loc.grad),
             grad_attn=t2n(attn.grad))

    # ops/test.py:24-31 recipe (N,M,D=1,2,2; Lq,L,P=2,2,2; shapes (6,4),(3,2); seed 3), fp64 and fp32
    case("testpy_f64", 1, 2, 2, 2, 2, [(6, 4), (3, 2)], 3, torch.float64)
    case("testpy_f32", 1, 2, 2, 2, 2, [(6, 4), (3, 2)], 3, torch.float32)
    # the channel counts of ops/test.py:88-89 that hit each backward variant, kept small (fp64)
    for d in (30, 32, 64, 71):
        case(f"d{d}_f64", 1, 2, d, 3, 2, [(6, 4), (3, 2)], 3 + d, torch.float64, loc_range=(-0.15, 1.15))
    # production geometry (M=8, D=32, L=3, P=4) scaled down, locations beyond the borders
    case("m8d32_f32", 1, 8, 32, 100, 4, [(4, 5), (8, 10), (16, 20)], 21, torch.float32, loc_range=(-0.2, 1.2),
         value_scale=1.0)
    # a-10: the module around the op, generator weights
    torch.manual_seed(31)
    mod = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4)
    sd = {k: synth.gen_tensor(7, "msdeformattn." + k, tuple(v.shape), gain=1.0) if v.dim() == 2
          else t2n(v) for k, v in mod.state_dict().items()}
    sd["sampling_offsets.bias"] = t2n(mod.state_dict()["sampling_offsets.bias"])   # the ring init, ms_deform_attn.py:66-80
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    shapes = [(4, 5), (8, 10), (16, 20)]
    shp = torch.as_tensor(shapes, dtype=torch.long)
    starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    S = int(shp.prod(1).sum())
    rng = np.random.default_rng(32)
    query = rng.standard_normal((1, S, 256), dtype=np.float32)
    src = rng.standard_normal((1, S, 256), dtype=np.float32)
    refp = rng.random((1, S, 3, 2), dtype=np.float32)
    with torch.no_grad():
        y = mod(torch.from_numpy(query), torch.from_numpy(refp), torch.from_numpy(src), shp, starts)
    # 2-D weights are regenerated by synth.gen_tensor(7, "msdeformattn."+name, ...); only the biases are stored
    save("msda_module", query=query, src=src, refp=refp, shapes=t2n(shp), starts=t2n(starts), out=t2n(y),
         **{"b_" + k: np.asarray(v) for k, v in sd.items() if np.asarray(v).ndim == 1})


def import_reference_encoder():
    """msdeformattn.py imports detectron2 / fvcore (absent, un-vendored) at module level only for the
    pixel-decoder shell below the encoder classes. Stub those third-party names (never the reference)
    and load the file through a synthetic package whose __path__ points at the real directories, so
    the reference's own __init__ chains (which need detectron2 everywhere) are not executed."""
    import importlib
    base = os.path.join(REF, "lib/network/mask2former/modeling")
    for name, attrs in {
        "fvcore": {}, "fvcore.nn": {}, "fvcore.nn.weight_init": {},
        "detectron2": {}, "detectron2.config": {"configurable": lambda f=None, **k: f if f is not None else (lambda g: g)},
        "detectron2.layers": {"Conv2d": torch.nn.Conv2d, "ShapeSpec": object, "get_norm": lambda *a, **k: None},
        "detectron2.modeling": {"SEM_SEG_HEADS_REGISTRY": types.SimpleNamespace(register=lambda: (lambda c: c))},
    }.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    for name, sub in {"m2fref": "", "m2fref.pixel_decoder": "pixel_decoder", "m2fref.transformer_decoder": "transformer_decoder"}.items():
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(base, sub)]
        sys.modules[name] = m
    mod = importlib.import_module("m2fref.pixel_decoder.msdeformattn")
    pe = importlib.import_module("m2fref.transformer_decoder.position_encoding")
    return mod.MSDeformAttnTransformerEncoderOnly, pe.PositionEmbeddingSine


def gen_encoder():
    """a-11: the reference's encoder-only transformer (2 layers) on 3 levels, CPU, eval mode."""
    Enc, PE = import_reference_encoder()
    torch.manual_seed(51)
    enc = Enc(d_model=256, nhead=8, num_encoder_layers=2, dim_feedforward=1024, dropout=0.0, activation="relu",
              num_feature_levels=3, enc_n_points=4).eval()
    sd = {}
    for k, v in enc.state_dict().items():
        if k.endswith("sampling_offsets.bias"):
            sd[k] = v.clone()                                   # keep the ring initialisation
        else:
            sd[k] = torch.from_numpy(synth.gen_tensor(9, "m2fenc." + k, tuple(v.shape), gain=1.0))
    enc.load_state_dict(sd)
    rng = np.random.default_rng(52)
    shapes = [(4, 5), (8, 10), (16, 20)]
    srcs = [rng.standard_normal((1, 256, h, w), dtype=np.float32) for h, w in shapes]
    pe = PE(128, normalize=True)
    pos = [pe(torch.from_numpy(s)) for s in srcs]
    with torch.no_grad():
        memory, spatial_shapes, starts = enc([torch.from_numpy(s) for s in srcs], pos)
    save("m2f_encoder", memory=t2n(memory), spatial_shapes=t2n(spatial_shapes), starts=t2n(starts),
         pos0=t2n(pos[0]), pos2_sub=t2n(pos[2])[:, ::16], seed=np.int64(52),
         names=np.array(list(sd.keys())), offsets_bias=t2n(sd["encoder.layers.0.self_attn.sampling_offsets.bias"]))


def import_reference_decoder():
    """The pixel-decoder shell (msdeformattn.py:164-358) needs more of detectron2 / fvcore than the encoder classes: the
    Conv2d wrapper (conv -> norm -> activation), get_norm("GN") = GroupNorm(32, C), ShapeSpec, c2_xavier_fill and the
    @configurable decorator. Both libraries are absent and un-vendored (detectron2 is installed from git HEAD, unpinned:
    README.md:53-54), so these five names are restated here from their published behaviour -- third-party code, never
    the reference -- and the reference file itself is imported unchanged. Parity of the shell is pinned up to that
    restatement."""
    import collections
    import importlib

    class Conv2d(torch.nn.Conv2d):                       # detectron2.layers.wrappers.Conv2d
        def __init__(self, *args, **kwargs):
            norm = kwargs.pop("norm", None)
            activation = kwargs.pop("activation", None)
            super().__init__(*args, **kwargs)
            self.norm = norm
            self.activation = activation

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    def get_norm(norm, out_channels):                    # detectron2.layers.batch_norm.get_norm, the cases the configs use
        if norm is None or norm == "":
            return None
        assert norm == "GN", norm
        return torch.nn.GroupNorm(32, out_channels)

    def c2_xavier_fill(module):                          # fvcore.nn.weight_init.c2_xavier_fill
        torch.nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            torch.nn.init.constant_(module.bias, 0)

    base = os.path.join(REF, "lib/network/mask2former/modeling")
    stubs = {
        "fvcore": {}, "fvcore.nn": {}, "fvcore.nn.weight_init": {"c2_xavier_fill": c2_xavier_fill},
        "detectron2": {}, "detectron2.config": {"configurable": lambda f=None, **k: f if f is not None else (lambda g: g)},
        "detectron2.layers": {"Conv2d": Conv2d, "ShapeSpec": collections.namedtuple("ShapeSpec", ["channels", "stride"]),
                              "get_norm": get_norm},
        "detectron2.modeling": {"SEM_SEG_HEADS_REGISTRY": types.SimpleNamespace(register=lambda: (lambda c: c))},
    }
    for name, attrs in stubs.items():
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
    for name, sub in {"m2fdec": "", "m2fdec.pixel_decoder": "pixel_decoder", "m2fdec.transformer_decoder": "transformer_decoder"}.items():
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(base, sub)]
        sys.modules[name] = m
    mod = importlib.import_module("m2fdec.pixel_decoder.msdeformattn")
    return mod.MSDeformAttnPixelDecoder, stubs["detectron2.layers"]["ShapeSpec"]


def _decoder_grad_sensitivity(dec, feats, cot, grads, feat_slice, trials, tag):
    """gsens_<name> (round 5): what the REFERENCE's own fp32 gradients move by under the deviations of an fp32 re-implementation --
    its four input maps jittered by 1e-5 relative AND every contraction's output by one layer's rounding: 6e-6 relative behind a 3x3
    convolution (Winograd F(6x6): 5.8e-6 of the output per layer, tools/wino_matrices.py), 4e-7 behind a 1x1 convolution / Linear.
    gnoise_* (fp32 vs fp64 of the reference) only sees ~1e-7 deviations; the bilinear sampler's location derivative is piecewise
    constant and the ReLUs behind the 3x3 output convolutions flip, and the stored slices are small (feat_res2: 32 x 16 x 16 values:
    one flipped ReLU inside the slice's footprint moves it by ~1.2e-3), so the comparison is only meaningful down to this floor.
    Largest rel-L2 change over `trials` seeded perturbations; tests bound with max(1e-3, 3 x gnoise, 2 x gsens)."""
    rel = lambda a, b: np.float64(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-300))
    sens = {}
    for trial in range(trials):
        jr = np.random.default_rng(650 + trial)
        tg = torch.Generator().manual_seed(660 + trial)

        def jitter(mod, inp, out, tg=tg):
            eps = 6e-6 if isinstance(mod, torch.nn.Conv2d) and tuple(mod.kernel_size) == (3, 3) else 4e-7
            return out * (1.0 + eps * torch.randn(out.shape, generator=tg, dtype=out.dtype))
        hooks = [m.register_forward_hook(jitter) for m in dec.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
        try:
            for p in dec.parameters():
                p.grad = None
            tj = {k: torch.from_numpy(v * (1.0 + 1e-5 * jr.standard_normal(v.shape).astype(np.float32))).requires_grad_(True) for k, v in feats.items()}
            mj, _, msj = dec.forward_features(tj)
            sum((t * c).sum() for t, c in zip((mj, *msj), cot)).backward()
        finally:
            for h in hooks:
                h.remove()
        for k, prm in dec.named_parameters():
            gk = t2n(prm.grad)
            if "g_" + k in grads:
                v = rel(gk, grads["g_" + k].astype(np.float64))
            else:
                flat = gk.reshape(gk.shape[0], -1)
                v = rel(flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)], grads["gsub_" + k].astype(np.float64))
            sens["gsens_" + k] = max(sens.get("gsens_" + k, 0.0), float(v))
        for k, t in tj.items():
            v = rel(feat_slice(t2n(t.grad)), grads["gsub_feat_" + k].astype(np.float64))
            sens["gsens_feat_" + k] = max(sens.get("gsens_feat_" + k, 0.0), float(v))
        del tj, mj, msj
    for p in dec.parameters():
        p.grad = None
    grads.update({k: np.float64(v) for k, v in sens.items()})
    top = sorted(((v, k) for k, v in sens.items()), reverse=True)[:8]
    print(f"   {tag}: largest change of the REFERENCE's gradients under the perturbations of an fp32 re-implementation: {top}")


def gen_decoder():
    """a-11: MSDeformAttnPixelDecoder.forward_features of the reference (2 encoder layers, the anomaly_ft.yaml geometry:
    res2..res5 = 256/512/1024/2048 channels at strides 4..32, GN norm, common_stride 4) on a 2-image batch, CPU."""
    Dec, ShapeSpec = import_reference_decoder()
    torch.manual_seed(61)
    shape = {"res2": ShapeSpec(256, 4), "res3": ShapeSpec(512, 8), "res4": ShapeSpec(1024, 16), "res5": ShapeSpec(2048, 32)}
    dec = Dec(shape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024, transformer_enc_layers=2,
              conv_dim=256, mask_dim=256, norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    sd = {}
    for k, v in dec.state_dict().items():
        if k.endswith("sampling_offsets.bias"):
            sd[k] = v.clone()                               # keep the ring initialisation
        else:                                               # 1-d weights / biases come out non-trivial (synth.gen_tensor)
            sd[k] = torch.from_numpy(synth.gen_tensor(10, "m2fdec." + k, tuple(v.shape), gain=1.0))
    dec.load_state_dict(sd)
    rng = np.random.default_rng(62)
    H, W = 96, 160
    feats = {k: rng.standard_normal((2, s.channels, H // s.stride, W // s.stride), dtype=np.float32) for k, s in shape.items()}
    with torch.no_grad():
        mask, out0, ms = dec.forward_features({k: torch.from_numpy(v) for k, v in feats.items()})
    # backward of the shell + encoder: L = <mask, G> + sum_i <ms[i], G_i> with seeded cotangents; parameters and the four
    # feature maps all receive gradients (the training loop freezes the backbone, train_m2f.py:409-412, but the input
    # gradients pin the 1x1 data-gradient path too)
    for p in dec.parameters():
        p.requires_grad_(True)
    tf = {k: torch.from_numpy(v).requires_grad_(True) for k, v in feats.items()}
    mask_g, _, ms_g = dec.forward_features(tf)
    crng = np.random.default_rng(63)
    cot = [torch.from_numpy(crng.standard_normal(tuple(t.shape), dtype=np.float32)) for t in (mask_g, *ms_g)]
    loss = sum((t * c).sum() for t, c in zip((mask_g, *ms_g), cot))
    loss.backward()
    grads = {}
    for k, prm in dec.named_parameters():
        gk = t2n(prm.grad)
        grads["gl2_" + k] = np.float64(np.sqrt((gk.astype(np.float64) ** 2).sum()))
        if gk.size <= 16384:
            grads["g_" + k] = gk
        else:
            flat = gk.reshape(gk.shape[0], -1)
            grads["gsub_" + k] = flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)].copy()
    for k, t in tf.items():
        gk = t2n(t.grad)
        grads["gl2_feat_" + k] = np.float64(np.sqrt((gk.astype(np.float64) ** 2).sum()))
        grads["gsub_feat_" + k] = gk[:, ::max(1, gk.shape[1] // 32)].copy()
    # conditioning, as in gen_decoder_fullsize (round 5: the small fixture had no floor, and its 3x5 / 6x10 / 12x20 maps put the res4-level
    # gradients of ANY fp32 implementation ~2e-3 from the fp32 reference): the same computation of the reference in float64
    dec64 = Dec(shape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024, transformer_enc_layers=2,
                conv_dim=256, mask_dim=256, norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    dec64.load_state_dict(sd)
    dec64 = dec64.double()
    for p in dec64.parameters():
        p.requires_grad_(True)
    tf64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in feats.items()}
    real_float = torch.Tensor.float        # the reference casts its inputs with .float() (msdeformattn.py:314-320): keep them double here
    torch.Tensor.float = lambda self, *a, **k: self
    try:
        m64, _, ms64 = dec64.forward_features(tf64)
    finally:
        torch.Tensor.float = real_float
    sum((t * c.double()).sum() for t, c in zip((m64, *ms64), cot)).backward()
    rel = lambda a, b: np.float64(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-300))
    for k, prm in dec64.named_parameters():
        g64 = prm.grad.numpy()
        if "g_" + k in grads:
            grads["gnoise_" + k] = rel(grads["g_" + k], g64)
        else:
            flat = g64.reshape(g64.shape[0], -1)
            grads["gnoise_" + k] = rel(grads["gsub_" + k], flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)])
    for k, t in tf64.items():
        g64 = t.grad.numpy()
        grads["gnoise_feat_" + k] = rel(grads["gsub_feat_" + k], g64[:, ::max(1, g64.shape[1] // 32)])
    top = sorted(((float(v), k) for k, v in grads.items() if k.startswith("gnoise_")), reverse=True)[:6]
    print(f"   decoder: largest fp32-vs-fp64 rel-L2 of the REFERENCE's own gradients: {top}")
    _decoder_grad_sensitivity(dec, feats, cot, grads, lambda gk: gk[:, ::max(1, gk.shape[1] // 32)], 4, "decoder")
    save("m2f_decoder", names=np.array(list(sd.keys())), seed=np.int64(62), hw=np.array([H, W]),
         offsets_bias=t2n(sd["transformer.encoder.layers.0.self_attn.sampling_offsets.bias"]),
         mask_sub=t2n(mask)[:, ::4], mask_abs_sum=np.float64(np.abs(t2n(mask).astype(np.float64)).sum()),
         out0=t2n(out0), ms1=t2n(ms[1])[:, ::2], ms2_sub=t2n(ms[2])[:, ::4],
         ms2_abs_sum=np.float64(np.abs(t2n(ms[2]).astype(np.float64)).sum()), cot_seed=np.int64(63), **grads)
    print(f"   decoder: mask {tuple(mask.shape)} |max| {float(mask.abs().max()):.3g}, out0 {tuple(out0.shape)}, "
          f"levels {[tuple(m.shape) for m in ms]}, {len(grads)} gradient entries")


def gen_decoder_fullsize(tag="m2f_decoder_704", n=1, H=704, W=704, layers=6, with_grads=False):
    """a-11 at BASELINE config 4's size: the reference's MSDeformAttnPixelDecoder.forward_features with the SHIPPED depth
    (6 encoder layers, anomaly_ft.yaml:27-31) on the feature maps of a 704x704 crop (levels 22^2 / 44^2 / 88^2, 10 164
    tokens). Big outputs as strided slices + float64 checksums; with_grads: the gradients of every parameter and feature map
    from the reference's autograd as well. ~1 minute forward, a few more with gradients, on 8 cores."""
    Dec, ShapeSpec = import_reference_decoder()
    torch.manual_seed(61)
    shape = {"res2": ShapeSpec(256, 4), "res3": ShapeSpec(512, 8), "res4": ShapeSpec(1024, 16), "res5": ShapeSpec(2048, 32)}
    dec = Dec(shape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024, transformer_enc_layers=layers,
              conv_dim=256, mask_dim=256, norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    sd = {}
    for k, v in dec.state_dict().items():
        sd[k] = v.clone() if k.endswith("sampling_offsets.bias") else \
            torch.from_numpy(synth.gen_tensor(10, "m2fdec." + k, tuple(v.shape), gain=1.0))
    dec.load_state_dict(sd)
    rng = np.random.default_rng(64)
    feats = {k: rng.standard_normal((n, s.channels, H // s.stride, W // s.stride), dtype=np.float32) for k, s in shape.items()}
    with torch.no_grad():
        mask, out0, ms = dec.forward_features({k: torch.from_numpy(v) for k, v in feats.items()})
    absum = lambda t: np.float64(np.abs(t2n(t).astype(np.float64)).sum())
    grads = {}
    if with_grads:
        # the training path at this size (VERDICT r03 missing #3): the reference class's own autograd of
        # L = <mask, G> + sum_i <ms[i], G_i> with seeded cotangents, every parameter and the four feature maps
        for p in dec.parameters():
            p.requires_grad_(True)
        tf = {k: torch.from_numpy(v).requires_grad_(True) for k, v in feats.items()}
        mask_g, _, ms_g = dec.forward_features(tf)
        crng = np.random.default_rng(65)
        cot = [torch.from_numpy(crng.standard_normal(tuple(t.shape), dtype=np.float32)) for t in (mask_g, *ms_g)]
        sum((t * c).sum() for t, c in zip((mask_g, *ms_g), cot)).backward()
        for k, prm in dec.named_parameters():
            gk = t2n(prm.grad)
            grads["gl2_" + k] = np.float64(np.sqrt((gk.astype(np.float64) ** 2).sum()))
            if gk.size <= 16384:
                grads["g_" + k] = gk
            else:
                flat = gk.reshape(gk.shape[0], -1)
                grads["gsub_" + k] = flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)].copy()
        for k, t in tf.items():
            gk = t2n(t.grad)
            grads["gl2_feat_" + k] = np.float64(np.sqrt((gk.astype(np.float64) ** 2).sum()))
            grads["gsub_feat_" + k] = gk[:, ::max(1, gk.shape[1] // 32), ::max(1, gk.shape[2] // 16), ::max(1, gk.shape[3] // 16)].copy()
        grads["cot_seed"] = np.int64(65)
        # conditioning: the same computation of the reference in float64. rel-L2(fp32 reference, fp64 reference) of each stored
        # gradient is the floor any fp32 implementation can be asked to meet (bilinear sampling is only piecewise smooth in the
        # locations, and the FFN's ReLU flips on ~1e-7 pre-activations): gnoise_<name>
        dec64 = Dec(shape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024, transformer_enc_layers=layers,
                    conv_dim=256, mask_dim=256, norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
        dec64.load_state_dict(sd)
        dec64 = dec64.double()
        for p in dec64.parameters():
            p.requires_grad_(True)
        tf64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in feats.items()}
        real_float = torch.Tensor.float        # the reference casts its inputs with .float() (msdeformattn.py:314-320): keep them double here
        torch.Tensor.float = lambda self, *a, **k: self
        try:
            m64, _, ms64 = dec64.forward_features(tf64)
        finally:
            torch.Tensor.float = real_float
        sum((t * c.double()).sum() for t, c in zip((m64, *ms64), cot)).backward()
        rel = lambda a, b: np.float64(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-300))
        # round 5: where the reference's two precisions disagree by more than 1e-4, its float64 gradient is stored too (rounded to
        # fp32, g64*_<name>): the two runs took different sides of a knife edge (a sample on a bilinear cell boundary, a ReLU at 0),
        # and an fp32 re-implementation may land on either -- tests compare with whichever of the two is nearer
        for k, prm in dec64.named_parameters():
            g64 = prm.grad.numpy()
            if "g_" + k in grads:
                sl, key = g64, "g64_" + k
                grads["gnoise_" + k] = rel(grads["g_" + k], g64)
            else:
                flat = g64.reshape(g64.shape[0], -1)
                sl, key = flat[::max(1, flat.shape[0] // 32), ::max(1, flat.shape[1] // 64)], "g64sub_" + k
                grads["gnoise_" + k] = rel(grads["gsub_" + k], sl)
            if grads["gnoise_" + k] > 1e-4:
                grads[key] = sl.astype(np.float32)
        for k, t in tf64.items():
            g64 = t.grad.numpy()
            sl = g64[:, ::max(1, g64.shape[1] // 32), ::max(1, g64.shape[2] // 16), ::max(1, g64.shape[3] // 16)]
            grads["gnoise_feat_" + k] = rel(grads["gsub_feat_" + k], sl)
            if grads["gnoise_feat_" + k] > 1e-4:
                grads["g64sub_feat_" + k] = sl.astype(np.float32)
        top = sorted(((float(v), k) for k, v in grads.items() if k.startswith("gnoise_")), reverse=True)[:6]
        print(f"   {tag}: {len(grads)} gradient entries; largest fp32-vs-fp64 rel-L2 of the REFERENCE: {top}")
        _decoder_grad_sensitivity(dec, feats, cot, grads,
                                  lambda gk: gk[:, ::max(1, gk.shape[1] // 32), ::max(1, gk.shape[2] // 16), ::max(1, gk.shape[3] // 16)], 4, tag)
    save(tag, names=np.array(list(sd.keys())), seed=np.int64(64), nhw=np.array([n, H, W]), layers=np.int64(layers), **grads,
         mask_sub=t2n(mask)[:, ::8, ::4, ::4], mask_abs_sum=absum(mask), mask_row=t2n(mask)[:, :, mask.shape[2] // 3],
         out0_sub=t2n(out0)[:, ::4], out0_abs_sum=absum(out0), ms1_sub=t2n(ms[1])[:, ::8, ::2, ::2], ms1_abs_sum=absum(ms[1]),
         ms2_sub=t2n(ms[2])[:, ::8, ::4, ::4], ms2_abs_sum=absum(ms[2]),
         absmax=np.array([float(t.abs().max()) for t in (mask, out0, ms[1], ms[2])]))
    print(f"   {tag}: mask {tuple(mask.shape)} |max| {float(mask.abs().max()):.3g}, levels {[tuple(m.shape) for m in ms]}")


def reference_get_anomaly_score():
    """TrainM2FOOD.get_anomaly_score of the reference ITSELF (train_m2f.py:387-407). The module cannot be imported (detectron2 is
    absent), so the method's own source is cut out of the reference file with `ast` and executed here -- nothing is re-typed
    (VERDICT r04 missing #3). Returns fn(other_outputs, size) -> score."""
    import ast
    import typing
    path = os.path.join(REF, "train_m2f.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "get_anomaly_score")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "Dict": typing.Dict, "Tuple": typing.Tuple}
    exec(compile(mod, path, "exec"), ns)
    return lambda other_outputs, size: ns["get_anomaly_score"](None, other_outputs, size)


def gen_m2f():
    """a-6: the reference's own get_anomaly_score (train_m2f.py:387-407, executed from its source: reference_get_anomaly_score)
    on random inputs."""
    rng = np.random.default_rng(41)
    cls = rng.standard_normal((2, 100, 20), dtype=np.float32) * 2
    mask = rng.standard_normal((2, 100, 24, 32), dtype=np.float32) * 3
    size = (22, 30)
    score = reference_get_anomaly_score()({"pred_logits_ood": torch.from_numpy(cls), "pred_masks_ood": torch.from_numpy(mask)}, size)
    save("m2f_score", cls=cls, mask=mask, size=np.array(size), score=t2n(score))
    # 8f-2: the chain in front of it -- mask prediction (mask2former_transformer_decoder.py:544-548), the x4 upsample
    # of maskformer_model.py:264-277 and the score -- evaluated with the same torch ops on random inputs
    out = {}
    for tag, (b, c, hm, wm, image, crop) in {"x4": (2, 64, 24, 32, (96, 128), (90, 120)),
                                             "ragged": (1, 32, 13, 17, (50, 70), (50, 70))}.items():
        emb = (rng.standard_normal((b, 100, c), dtype=np.float32) / np.sqrt(c) * 3).astype(np.float32)
        feat = rng.standard_normal((b, c, hm, wm), dtype=np.float32)
        cl = rng.standard_normal((b, 100, 20), dtype=np.float32) * 2
        masks = torch.einsum("bqc,bchw->bqhw", torch.from_numpy(emb), torch.from_numpy(feat))
        up = torch.nn.functional.interpolate(masks, size=image, mode="bilinear", align_corners=False)
        probs = torch.softmax(torch.from_numpy(cl), dim=-1)[..., :-1]
        u = torch.einsum("bqc,bqhw->bchw", probs, up.sigmoid())[:, :, :crop[0], :crop[1]]
        out.update({tag + "_embed": emb, tag + "_features": feat, tag + "_cls": cl, tag + "_image": np.array(image),
                    tag + "_crop": np.array(crop), tag + "_masks_sub": t2n(masks)[:, ::7], tag + "_up_sub": t2n(up)[:, ::9, ::3, ::3],
                    tag + "_score": t2n(1 - torch.max(u, dim=1)[0])})
    save("m2f_fused", **out)


def gen_datapath():
    """f-4: per-sample data path. lib/utils/img_utils.py is imported with cv2 / torchvision stubbed (third-party, absent;
    mix_func / normalize / extract_bboxes use neither) and its own mix_func does the COCO paste; mixup is the three lines of
    cityscapes.py:161-164 evaluated verbatim; ToTensor / F.crop / Normalize are the torch expressions torchvision
    documents (uint8 -> float32 / 255; slicing; sub_(mean).div_(std) with float32 mean / std)."""
    import importlib.util
    import random
    for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.InterpolationMode = types.SimpleNamespace(BILINEAR=2, NEAREST=0)
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    spec = importlib.util.spec_from_file_location("ref_img_utils", os.path.join(REF, "lib", "utils", "img_utils.py"))
    iu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(iu)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    rng = np.random.default_rng(71)
    B, H, W, h, w = 3, 40, 56, 24, 32
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    gen = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    tgt = rng.integers(0, 19, (B, H, W)).astype(np.uint8)
    gen_tgt = np.where(rng.random((B, H, W)) < 0.1, 254, tgt).astype(np.uint8)
    out = dict(img=img, gen=gen, tgt=tgt, gen_tgt=gen_tgt, crop=np.array([h, w]))
    random.seed(1234)
    o_imgs, o_tgts, a_imgs, a_tgts = [], [], [], []
    for b in range(B):
        p = min(random.random(), 0.3)                                                      # cityscapes.py:162
        mix_array = (p * np.array(img[b]) + (1 - p) * np.array(gen[b])).astype(np.uint8)   # cityscapes.py:163
        top = random.randint(0, H - h)                                                     # img_utils.py:256-257
        left = random.randint(0, W - w)

        def tf(a):                                    # ToTensor -> F.crop -> Normalize (torchvision semantics)
            t = torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).div(255)
            t = t[:, top:top + h, left:left + w].clone()
            return t.sub_(torch.as_tensor(mean, dtype=torch.float32)[:, None, None]).div_(
                torch.as_tensor(std, dtype=torch.float32)[:, None, None])
        x, xg = tf(img[b]), tf(mix_array)
        t = torch.tensor(np.array(tgt[b], dtype=np.uint8), dtype=torch.long)[top:top + h, left:left + w].clone()
        tg = torch.tensor(np.array(gen_tgt[b], dtype=np.uint8), dtype=torch.long)[top:top + h, left:left + w].clone()
        # anomaly mix: an already rescaled object (the cv2.resize of random_scale cannot run here) through the
        # reference's own mix_func; its two randint draws are recorded
        oh, ow = int(rng.integers(6, 14)), int(rng.integers(8, 18))
        o_img = (rng.random((oh, ow, 3)) * 255).astype(np.float32)
        o_mask = np.zeros((oh, ow), dtype=np.uint8)
        o_mask[2:oh - 1, 1:ow - 3] = np.where(rng.random((oh - 3, ow - 4)) < 0.7, 254, 0)
        o_mask[0, 0] = 255                                                                  # a void pixel of the object crop
        drawn = []
        real_randint = random.randint

        def logging_randint(a, c):
            v = real_randint(a, c)
            drawn.append(v)
            return v
        random.randint = logging_randint
        try:
            x2, t2 = iu.mix_func(x, t, o_img.copy(), o_mask.copy())
        finally:
            random.randint = real_randint
        boxes = iu.extract_bboxes(np.expand_dims((o_mask != 0) & (o_mask != 255), axis=2))[0]
        out.update({f"s{b}_p": np.float64(p), f"s{b}_top": np.int64(top), f"s{b}_left": np.int64(left), f"s{b}_obj_img": o_img,
                    f"s{b}_obj_mask": o_mask, f"s{b}_bbox": boxes.astype(np.int64), f"s{b}_corner": np.array(drawn, dtype=np.int64)})
        o_imgs.append(t2n(x2)); o_tgts.append(t2n(t2)); a_imgs.append(t2n(xg)); a_tgts.append(t2n(tg))
    out["images"] = np.stack(o_imgs + a_imgs).astype(np.float32)             # train_deeplab.py:194: cat([img, div_img])
    out["targets"] = np.stack(o_tgts + a_tgts).astype(np.int64)
    save("datapath", **out)


def gen_metric():
    """8f-1: lib/utils/metric.py eval_ood_measure (loaded by file path: the package __init__ needs wget/h5py) on
    small score/label maps -- continuous scores, heavily tied scores, signed zeros, a recall level that falls
    between thresholds, tiny positive sets."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metric", os.path.join(REF, "lib", "utils", "metric.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    rng = np.random.default_rng(77)
    out = {}
    cases = {"cont": (3, 40, 56, None, 0.25), "ties16": (2, 48, 64, 16, 0.2), "ties2": (1, 32, 32, 2, 0.5),
             "fewpos": (1, 64, 64, None, 0.002), "fewneg": (1, 24, 40, 8, 0.97), "zeros": (1, 16, 16, 1, 0.4)}
    for tag, (b, h, w, q, ppos) in cases.items():
        r = rng.random((b, h, w))
        label = np.where(r < ppos, 1, np.where(r < ppos + (1 - ppos) * 0.85, 0, 255)).astype(np.int64)
        score = (rng.standard_normal((b, h, w)) + 0.9 * (label == 1)).astype(np.float32)
        if q:
            score = (np.round(score * q) / q).astype(np.float32)
        if tag == "zeros":
            score[score == 0] = np.where(rng.random((score == 0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
        res = rm.eval_ood_measure(score.copy(), label.copy())
        out[tag + "_score"], out[tag + "_label"] = score, label.astype(np.uint8)
        out[tag + "_measures"] = np.array(res, dtype=np.float64)
        print(f"   {tag}: P={int((label == 1).sum())} N={int((label == 0).sum())} auroc/aupr/fpr {res}")
    lab = np.zeros((1, 8, 8), np.int64)
    assert rm.eval_ood_measure(np.zeros((1, 8, 8), np.float32), lab) is None          # no OOD pixel -> None
    save("ood_metrics", **out)


def main():
    which = set(sys.argv[1:]) or {"ops", "deeplab", "train", "loss", "msda", "m2f", "encoder", "decoder", "metric", "datapath"}
    torch.set_num_threads(8)
    DeepWV3Plus, ref_loss, core, MSDeformAttn = import_reference()
    if "ops" in which:
        print("ops"); gen_ops()
    if "ops2" in which or "ops" in which:
        print("ops2"); gen_ops2()
    if "msda" in which:
        print("msda"); gen_msda(core, MSDeformAttn)
    if "m2f" in which:
        print("m2f"); gen_m2f()
    if "encoder" in which:
        print("encoder"); gen_encoder()
    if "decoder" in which:
        print("decoder"); gen_decoder()
    if "decoder_704" in which:           # minutes: only on request
        print("decoder_704"); gen_decoder_fullsize(with_grads=True)
    if "decoder_c5" in which:
        print("decoder_c5"); gen_decoder_fullsize("m2f_decoder_1024x2048", 1, 1024, 2048, with_grads=True)
    if "datapath" in which:
        print("datapath"); gen_datapath()
    if "loss" in which:
        print("loss"); gen_loss(ref_loss)
    if "metric" in which:
        print("metric"); gen_metric()
    if "deeplab" in which:
        print("deeplab"); gen_deeplab(DeepWV3Plus)
    if "train" in which:
        print("train"); gen_train_step(DeepWV3Plus, ref_loss)
    if "train" in which or "train3" in which:
        print("train3"); gen_train_steps3(DeepWV3Plus, ref_loss)
    if "train" in which or "train8" in which:
        # the C2 batch layout (VERDICT r04 missing #4): 8 (original, augmented) pairs -- [0:8] originals, [8:16] their augmentations,
        # pairing i <-> i + 8 (train_deeplab.py:190-204, lib/loss.py:59-60,141-145) -- at a small crop, both stages
        print("train8"); gen_train_step(DeepWV3Plus, ref_loss, pairs=8, h=64, w=96, fixture="deepwv3plus_train_step_8pairs")
    # the two big fixtures take minutes on 8 cores: only on request (python tools/gen_golden.py deeplab_big train_big)
    if "deeplab_big" in which:
        print("deeplab_big"); gen_deeplab_big(DeepWV3Plus)
    if "train_big" in which:
        print("train_big")
        gen_train_step(DeepWV3Plus, ref_loss, pairs=1, h=592, w=600, fixture="deepwv3plus_train_step_2x592x600",
                       stages=("stage2",), fp64_replay=False, score_stride=4, logit_stride=8, truncate_perms=True)
    # the headline configuration itself (BASELINE config 3, per GPU): about 2 + 10 minutes on 8 cores
    if "deeplab_c3" in which:
        print("deeplab_c3"); gen_deeplab_big(DeepWV3Plus, n=1, h=1024, w=2048, score_stride=4, logit_stride=8, all_margins=True)
    if "train_c3" in which:
        print("train_c3")
        gen_train_step(DeepWV3Plus, ref_loss, pairs=1, h=1024, w=2048, fixture="deepwv3plus_train_step_2x1024x2048",
                       stages=("stage2",), fp64_replay=False, score_stride=8, logit_stride=16, truncate_perms=True,
                       store_labels=True)


if __name__ == "__main__":
    main()
