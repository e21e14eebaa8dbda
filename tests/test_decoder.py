"""Row a-11: MSDeformAttnPixelDecoder.forward_features (input_proj + GroupNorm, encoder, FPN top-down step, mask_features)
against the reference class itself (tests/golden/m2f_decoder.npz, tools/gen_golden.py decoder), and its building blocks
(GroupNorm, residual + LayerNorm with gradients, half-pixel bilinear + add, NHWC->NCHW) against the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import golden

SHAPE = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}


def build():
    from multishiftseg_amd import synth
    from multishiftseg_amd.msdeformattn_decoder import MSDeformAttnPixelDecoder, ShapeSpec
    g = golden("m2f_decoder")
    dec = MSDeformAttnPixelDecoder({k: ShapeSpec(*v) for k, v in SHAPE.items()}, transformer_dropout=0.0, transformer_nheads=8,
                                   transformer_dim_feedforward=1024, transformer_enc_layers=2, conv_dim=256, mask_dim=256, norm="GN",
                                   transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    sd = dec.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]              # parameter names and order = the reference's
    new = {}
    for k, v in sd.items():
        new[k] = v.clone() if k.endswith("sampling_offsets.bias") else \
            torch.from_numpy(synth.gen_tensor(10, "m2fdec." + k, tuple(v.shape), gain=1.0))
    np.testing.assert_allclose(new["transformer.encoder.layers.0.self_attn.sampling_offsets.bias"].numpy(), g["offsets_bias"], atol=1e-6)
    dec.load_state_dict(new)
    return dec, g


def test_decoder_state_dict_contract_cpu():
    dec, g = build()
    assert dec.num_fpn_levels == 1 and dec.transformer_in_features == ["res3", "res4", "res5"]
    feats = {k: torch.zeros(1, c, 64 // s, 64 // s) for k, (c, s) in SHAPE.items()}
    with pytest.raises(NotImplementedError):          # parameter gradients requested: the shell has no backward
        dec.forward_features(feats)
    with torch.no_grad(), pytest.raises(RuntimeError, match="MI355X"):
        dec.forward_features(feats)                   # no CPU path


@pytest.mark.gpu
def test_decoder_forward_features_golden():
    dec, g = build()
    dec = dec.cuda()
    rng = np.random.default_rng(int(g["seed"]))
    H, W = (int(v) for v in g["hw"])
    feats = {k: torch.from_numpy(rng.standard_normal((2, c, H // s, W // s), dtype=np.float32)).cuda() for k, (c, s) in SHAPE.items()}
    with torch.no_grad():
        mask, out0, ms = dec.forward_features(feats)
    assert ms[0] is out0 or torch.equal(ms[0], out0)
    assert [tuple(m.shape) for m in ms] == [(2, 256, 3, 5), (2, 256, 6, 10), (2, 256, 12, 20)] and tuple(mask.shape) == (2, 256, 24, 40)
    np.testing.assert_allclose(out0.cpu().numpy(), g["out0"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ms[1].cpu().numpy()[:, ::2], g["ms1"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(ms[2].cpu().numpy()[:, ::4], g["ms2_sub"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(mask.cpu().numpy()[:, ::4], g["mask_sub"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(mask.double().abs().sum().item(), float(g["mask_abs_sum"]), rtol=1e-4)
    np.testing.assert_allclose(ms[2].double().abs().sum().item(), float(g["ms2_abs_sum"]), rtol=1e-4)


@pytest.mark.gpu
def test_groupnorm_layernorm_upsample_ops_vs_oracle():
    from multishiftseg_amd import kernels as K
    from oracle import nnops
    rng = np.random.default_rng(3)
    # GroupNorm(32, 256) on ragged maps, with and without ReLU, plain and into a token buffer
    for (n, h, w) in [(2, 7, 9), (1, 33, 20), (3, 1, 1)]:
        x = rng.standard_normal((n, 256, h, w), dtype=np.float32) * 2 + 0.5
        gn = torch.nn.GroupNorm(32, 256).cuda()
        with torch.no_grad():
            gn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 256).astype(np.float32)))
            gn.bias.copy_(torch.from_numpy(rng.standard_normal(256).astype(np.float32)))
        ref = nnops.groupnorm(x, 32, gn.weight.detach().cpu().numpy(), gn.bias.detach().cpu().numpy())
        xa = K.Act.from_nchw(torch.from_numpy(x).cuda())
        y = K.groupnorm(xa, gn)
        np.testing.assert_allclose(y.nchw().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        y = K.groupnorm(xa, gn, relu=True)
        np.testing.assert_allclose(y.nchw().cpu().numpy(), np.maximum(ref, 0), rtol=1e-4, atol=1e-5)
        tokens = torch.full((n, h * w + 5, 256), float("nan"), device="cuda")
        K.groupnorm(xa, gn, out=tokens[0, 3:], out_sample_stride=(h * w + 5) * 256, out_ld=256)
        lvl = K.TokenLevel(tokens, 3, h, w)
        np.testing.assert_allclose(K.nhwc_to_nchw(lvl).cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        assert torch.isnan(tokens[:, :3]).all() and torch.isnan(tokens[:, 3 + h * w:]).all()       # nothing outside the level
    # residual + LayerNorm, forward and backward
    for C in (256, 512):
        a = torch.from_numpy(rng.standard_normal((3, 37, C), dtype=np.float32)).cuda().requires_grad_(True)
        b = torch.from_numpy(rng.standard_normal((3, 37, C), dtype=np.float32)).cuda().requires_grad_(True)
        ln = torch.nn.LayerNorm(C).cuda()
        with torch.no_grad():
            ln.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)))
            ln.bias.copy_(torch.from_numpy(rng.standard_normal(C).astype(np.float32)))
        y = K.add_layernorm(a, b, ln)
        gy = torch.from_numpy(rng.standard_normal((3, 37, C), dtype=np.float32)).cuda()
        y.backward(gy)
        an, bn_, gm, bt = (t.detach().cpu().numpy() for t in (a, b, ln.weight, ln.bias))
        np.testing.assert_allclose(y.detach().cpu().numpy(), nnops.add_layernorm(an, bn_, gm, bt), rtol=1e-4, atol=1e-5)
        dz, dg, db = nnops.add_layernorm_bwd(an, bn_, gm, gy.cpu().numpy())
        np.testing.assert_allclose(a.grad.cpu().numpy(), dz, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(b.grad.cpu().numpy(), dz, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(ln.weight.grad.cpu().numpy(), dg, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ln.bias.grad.cpu().numpy(), db, rtol=1e-4, atol=1e-4)
        y2 = K.add_layernorm(a.detach(), None, ln)
        np.testing.assert_allclose(y2.detach().cpu().numpy(), nnops.add_layernorm(an, None, gm, bt), rtol=1e-4, atol=1e-5)
    # half-pixel bilinear + add (FPN top-down), x2 / odd ratio / identity
    top = rng.standard_normal((2, 16, 5, 7), dtype=np.float32)
    ta = K.Act.from_nchw(torch.from_numpy(top).cuda())
    for size in [(10, 14), (11, 13), (5, 7), (24, 40)]:
        lat = rng.standard_normal((2, 16) + size, dtype=np.float32)
        y = K.upsample_bilinear_add(ta, K.Act.from_nchw(torch.from_numpy(lat).cuda()))
        np.testing.assert_allclose(K.nhwc_to_nchw(y).cpu().numpy(), lat + nnops.upsample_bilinear_hp(top, size), rtol=1e-5, atol=1e-5)
