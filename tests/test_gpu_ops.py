"""GPU parity of the per-op HIP kernels against the CPU oracle and the reference's own vectors."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import m2f as om2f, nnops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from multishiftseg_amd import kernels
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_conv(K, x, w, stride=1, dil=1, pad=0, **kw):
    xa = K.Act.from_nchw(dev(x), ld=((x.shape[1] + 15) // 16) * 16)
    pw = K.pack_weight(dev(w))
    y = K.conv2d(xa, pw, stride=stride, dil=dil, pad=pad, **kw)
    return y.nchw().cpu().numpy()


def test_conv_classes_golden(K, gemm_route):
    g = golden("ops")
    x = g["conv_x"]
    for tag, (stride, dil) in {"1x1": (1, 1), "3x3_d1": (1, 1), "3x3_d2": (1, 2), "3x3_d4": (1, 4), "3x3_d12": (1, 12),
                               "3x3_d24": (1, 24), "3x3_d36": (1, 36), "3x3_s2": (2, 1), "1x1_s2": (2, 1)}.items():
        w = g[f"conv_{tag}_w"]
        pad = dil if w.shape[2] == 3 else 0
        y = run_conv(K, x, w, stride, dil, pad)
        np.testing.assert_allclose(y, g[f"conv_{tag}_y"], rtol=1e-5, atol=2e-6, err_msg=tag)


@pytest.mark.parametrize("cin,cout,r,stride,dil,n,h,w", [
    (64, 128, 3, 1, 1, 2, 33, 47),      # mod2-style, ragged M tile
    (48, 48, 1, 1, 1, 1, 17, 19),       # narrow-N tile, C % 32 != 0
    (304, 256, 3, 1, 1, 1, 24, 20),     # final.0 (C = 304 -> BK 16)
    (256, 512, 3, 2, 1, 2, 31, 29),     # mod4.block1.conv1 stride 2, odd size
    (128, 256, 3, 1, 12, 1, 16, 32),    # ASPP rate 12 on a small map: dead taps
    (128, 64, 3, 1, 36, 2, 16, 16),     # rate 36: only the centre tap is live
    (512, 19 + 29, 1, 1, 1, 2, 9, 7),   # heads-like K = 48
    (64, 304, 3, 1, 1, 1, 12, 10),      # K = 2*128 + 48: main + narrow tail launch
])
def test_conv_vs_oracle(K, cin, cout, r, stride, dil, n, h, w, gemm_route):
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((n, cin, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, r, r), dtype=np.float32) / np.sqrt(cin * r * r)).astype(np.float32)
    pad = dil if r == 3 else 0
    ref = nnops.conv2d(x, wt, stride, dil, pad)
    y = run_conv(K, x, wt, stride, dil, pad)
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)


def test_conv_fusions(K, gemm_route):
    """prologue BN+ReLU (shared and per-sample), epilogue affine + residual + ReLU."""
    rng = np.random.default_rng(3)
    n, c, k, h, w = 2, 64, 128, 12, 15
    x = rng.standard_normal((n, c, h, w), dtype=np.float32)
    wt = rng.standard_normal((k, c, 3, 3), dtype=np.float32) / 24
    sc = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sh = rng.standard_normal(c).astype(np.float32)
    res = rng.standard_normal((n, k, h, w), dtype=np.float32)
    osc = rng.uniform(0.5, 1.5, k).astype(np.float32)
    osh = rng.standard_normal(k).astype(np.float32)
    act = np.maximum(x * sc[None, :, None, None] + sh[None, :, None, None], 0)
    ref = nnops.conv2d(act, wt, 1, 1, 1)
    ref = np.maximum(ref * osc[None, :, None, None] + osh[None, :, None, None] + res, 0)
    xa = K.Act.from_nchw(dev(x))
    ra = K.Act.from_nchw(dev(res))
    y = K.conv2d(xa, K.pack_weight(dev(wt)), pad=1, in_affine=(dev(sc), dev(sh)), in_relu=True,
                 out_affine=(dev(osc), dev(osh)), out_relu=True, res=ra)
    np.testing.assert_allclose(y.nchw().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    # per-sample affine (Dropout2d fold); 12x15 tiles straddle images, the 16x16 case below does not
    mask = (rng.random((n, c)) > 0.3).astype(np.float32) / 0.7
    act = np.maximum(x * (sc[None] * mask)[:, :, None, None] + (sh[None] * mask)[:, :, None, None], 0)
    ref = nnops.conv2d(act, wt, 1, 1, 1)
    y = K.conv2d(xa, K.pack_weight(dev(wt)), pad=1, in_affine=(dev(sc[None] * mask), dev(sh[None] * mask)), in_relu=True)
    np.testing.assert_allclose(y.nchw().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    x2 = rng.standard_normal((3, c, 16, 16), dtype=np.float32)
    mask2 = (rng.random((3, c)) > 0.5).astype(np.float32) / 0.5
    act = np.maximum(x2 * (sc[None] * mask2)[:, :, None, None] + (sh[None] * mask2)[:, :, None, None], 0)
    ref = nnops.conv2d(act, wt, 1, 1, 1)
    y = K.conv2d(K.Act.from_nchw(dev(x2)), K.pack_weight(dev(wt)), pad=1,
                 in_affine=(dev(sc[None] * mask2), dev(sh[None] * mask2)), in_relu=True)
    np.testing.assert_allclose(y.nchw().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_conv_channel_slices(K, gemm_route):
    """read from / write into slices of a wider NHWC buffer (in-place concat)."""
    rng = np.random.default_rng(4)
    n, h, w = 1, 10, 11
    big = rng.standard_normal((n, 96, h, w), dtype=np.float32)
    wt = rng.standard_normal((48, 32, 1, 1), dtype=np.float32) / 6
    ba = K.Act.from_nchw(dev(big))
    out = K.Act.zeros(n, h, w, 304, "cuda")
    K.conv2d(ba.slice(32, 32), K.pack_weight(dev(wt)), out=out.slice(48, 48))
    ref = nnops.conv2d(big[:, 32:64], wt)
    got = out.buf.permute(0, 3, 1, 2).cpu().numpy()
    np.testing.assert_allclose(got[:, 48:96], ref, rtol=1e-4, atol=1e-5)
    assert np.all(got[:, :48] == 0) and np.all(got[:, 96:] == 0)


def test_dgrad_and_wgrad(K, gemm_route):
    rng = np.random.default_rng(5)
    for (cin, cout, r, dil, n, h, w) in [(64, 128, 3, 1, 2, 14, 13), (256, 48, 1, 1, 1, 9, 8), (128, 64, 3, 12, 1, 20, 18)]:
        pad = dil if r == 3 else 0
        x = torch.from_numpy(rng.standard_normal((n, cin, h, w), dtype=np.float32)).requires_grad_(True)
        wt = torch.from_numpy((rng.standard_normal((cout, cin, r, r), dtype=np.float32) / np.sqrt(cin * r * r)).astype(np.float32)).requires_grad_(True)
        y = torch.nn.functional.conv2d(x, wt, dilation=dil, padding=pad)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape), dtype=np.float32))
        y.backward(gy)
        ga = K.Act.from_nchw(gy.cuda(), ld=((cout + 15) // 16) * 16)
        dx = K.conv2d(ga, K.pack_weight(wt.detach().cuda(), flip=True), dil=dil, pad=pad)
        np.testing.assert_allclose(dx.nchw().cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-4)
        dw = K.conv2d_wgrad(K.Act.from_nchw(x.detach().cuda()), ga.slice(0, cout), cout, cin, r, r, dil=dil, pad=pad)
        np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_batchnorm_train_eval(K):
    g = golden("ops")
    bn = torch.nn.BatchNorm2d(8).cuda()
    with torch.no_grad():
        bn.weight.copy_(dev(g["bn_gamma"])); bn.bias.copy_(dev(g["bn_beta"]))
        bn.running_mean.copy_(dev(g["bn_rm"])); bn.running_var.copy_(dev(g["bn_rv"]))
    xa = K.Act.from_nchw(dev(g["bn_x"]))
    st = K.bn_fold(bn, train=False)
    y = torch.empty_like(xa.buf)
    from multishiftseg_amd._lib import call, ptr
    call("mss_affine_relu_nhwc_f32", xa.ptr, xa.ld, ptr(y), 8, xa.M, 8, ptr(st.scale), ptr(st.shift), 0)
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), g["bn_eval_y"], rtol=1e-5, atol=1e-5)
    st = K.bn_fold(bn, xa, train=True)
    call("mss_affine_relu_nhwc_f32", xa.ptr, xa.ld, ptr(y), 8, xa.M, 8, ptr(st.scale), ptr(st.shift), 0)
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), g["bn_train_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), g["bn_train_rm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), g["bn_train_rv"], rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_bn_relu_backward(K):
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.standard_normal((3, 16, 9, 7), dtype=np.float32) * 2 + 0.3).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(16)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 16).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.standard_normal(16).astype(np.float32) * 0.3))
    y = torch.relu(bn(x))
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(gy)
    bn_g = torch.nn.BatchNorm2d(16).cuda()
    bn_g.load_state_dict({k: v for k, v in bn.state_dict().items()})
    with torch.no_grad():
        bn_g.running_mean.zero_(); bn_g.running_var.fill_(1)
    xa = K.Act.from_nchw(x.detach().cuda())
    st = K.bn_fold(bn_g, xa, train=True)
    dx, dg, db = K.bn_relu_backward(K.Act.from_nchw(gy.cuda()), xa, st, want_param_grads=True)
    np.testing.assert_allclose(dx.nchw().cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), bn.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bn.bias.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_pool_gap_upsample(K):
    g = golden("ops")
    px = np.repeat(g["pool_x"], 1, axis=1)
    y = K.maxpool3s2(K.Act.from_nchw(dev(px)))
    np.testing.assert_array_equal(y.nchw().cpu().numpy(), g["pool_y"])
    ux = g["up_x"]
    for tag in "abc":
        oh, ow = g[f"up_{tag}_y"].shape[2:]
        y = K.upsample_ac(K.Act.from_nchw(dev(ux)), oh, ow)
        np.testing.assert_allclose(y.nchw().cpu().numpy(), g[f"up_{tag}_y"], rtol=1e-5, atol=5e-6)
        gx = K.upsample_ac_bwd(K.Act.from_nchw(dev(g[f"up_{tag}_gy"])), ux.shape[2], ux.shape[3])
        np.testing.assert_allclose(gx.nchw().cpu().numpy(), g[f"up_{tag}_gx"], rtol=1e-4, atol=1e-5)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 64, 5, 9), dtype=np.float32)
    np.testing.assert_allclose(K.gap(K.Act.from_nchw(dev(x))).cpu().numpy(), x.mean((2, 3)), rtol=1e-5, atol=1e-6)


def test_ood_score_tail(K):
    rng = np.random.default_rng(6)
    n, h, w = 2, 13, 17
    d = rng.standard_normal((n, 48, h, w), dtype=np.float32) * 3
    da = K.Act.from_nchw(dev(d))
    for (oh, ow) in [(26, 34), (25, 33), (13, 17), (300, 270)]:   # tiled x2 path, odd sizes, identity (fallback), big ratio
        score, logit, label = K.ood_score(da.slice(20, 19), da.slice(0, 19), oh, ow, want_label=True)
        rs, rl = nnops.ood_score_tail(d[:, 20:39], d[:, 0:19], (oh, ow))
        np.testing.assert_allclose(score.cpu().numpy(), rs, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(logit.cpu().numpy(), rl, rtol=1e-5, atol=1e-5)
        got = logit.cpu().numpy()
        np.testing.assert_array_equal(label.cpu().numpy(), got.argmax(1))   # bit-exact vs its own logits
    # backward vs torch autograd of the same composition: (26, 34) takes the per-pixel gather kernel (OW % 4 != 0), the other
    # two the LDS-tiled one (several 4x64 tiles, ragged edges); there the 48-channel rows are written whole, so the
    # destination starts as NaN and the padding channels must come back as zeros
    for (hh, ww, oh, ow, tiled) in [(13, 17, 26, 34, False), (13, 18, 26, 36, True), (37, 70, 74, 140, True)]:
        d2 = rng.standard_normal((n, 48, hh, ww), dtype=np.float32) * 3
        da2 = K.Act.from_nchw(dev(d2))
        dt = torch.from_numpy(d2).requires_grad_(True)
        lg = torch.nn.functional.interpolate(dt[:, 0:19], size=(oh, ow), mode="bilinear", align_corners=True)
        sc = torch.nn.functional.interpolate(-torch.logsumexp(dt[:, 20:39], 1, keepdim=True), size=(oh, ow), mode="bilinear",
                                             align_corners=True)[:, 0]
        gl = torch.from_numpy(rng.standard_normal(tuple(lg.shape), dtype=np.float32))
        gs = torch.from_numpy(rng.standard_normal(tuple(sc.shape), dtype=np.float32))
        (lg * gl).sum().backward(retain_graph=True)
        (sc * gs).sum().backward()
        dd = K.Act(torch.full((n, hh, ww, 48), float("nan"), device="cuda")) if tiled else K.Act.zeros(n, hh, ww, 48, "cuda")
        K.ood_score_bwd(da2.slice(20, 19), gs.cuda(), gl.cuda(), dd.slice(20, 19), dd.slice(0, 19), oh, ow)
        np.testing.assert_allclose(dd.nchw().cpu().numpy(), dt.grad.numpy(), rtol=1e-4, atol=1e-5, err_msg=str((hh, ww)))
        if tiled:   # only one of the two gradients present: the other head's channels are zeros, not garbage
            dd = K.Act(torch.full((n, hh, ww, 48), float("nan"), device="cuda"))
            K.ood_score_bwd(da2.slice(20, 19), gs.cuda(), None, dd.slice(20, 19), dd.slice(0, 19), oh, ow)
            got = dd.nchw().cpu().numpy()
            np.testing.assert_allclose(got[:, 20:39], dt.grad.numpy()[:, 20:39], rtol=1e-4, atol=1e-5)
            assert np.all(got[:, 0:20] == 0) and np.all(got[:, 39:] == 0)


def test_m2f_score(K):
    g = golden("m2f_score")
    size = tuple(int(v) for v in g["size"])
    s = K.m2f_score(dev(g["cls"]), dev(g["mask"]), size)
    np.testing.assert_allclose(s.cpu().numpy(), g["score"], rtol=1e-5, atol=1e-5)
    rng = np.random.default_rng(9)
    cls = rng.standard_normal((1, 100, 20), dtype=np.float32)
    mask = rng.standard_normal((1, 100, 64, 96), dtype=np.float32) * 4
    s = K.m2f_score(dev(cls), dev(mask), (64, 96))
    np.testing.assert_allclose(s.cpu().numpy(), om2f.anomaly_score(cls, mask, (64, 96)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tile", [2, 4, 6])
@pytest.mark.parametrize("cin,cout,dil,n,h,w", [(32, 64, 1, 2, 12, 14), (64, 32, 2, 1, 13, 17), (32, 128, 4, 2, 16, 22),
                                                 (48, 64, 1, 1, 7, 9), (64, 64, 4, 1, 5, 6),
                                                 # several channel chunks of the LDS-staged input transform, the last one
                                                 # ragged: 2x4-tile blocks x 64 channels / 1x2-tile blocks x 256 channels
                                                 (160, 32, 1, 1, 11, 18), (336, 16, 12, 1, 30, 41)])
def test_winograd_conv_vs_oracle(K, cin, cout, dil, n, h, w, tile, monkeypatch):
    """Winograd F(2x2,3x3) / F(4x4,3x3) path: dilation handled through residue sub-grids, ragged tiles,
    fused BatchNorm+ReLU prologue and residual epilogue. MSS_WINO_INPUT_LDS=2 sends every F(4x4) input transform
    through the LDS-staged kernel (the policy would keep these small, ragged maps on the one-thread-per-tile kernel,
    which the whole-network tests and the other op tests cover)."""
    monkeypatch.setenv("MSS_WINO_INPUT_LDS", "2")
    rng = np.random.default_rng(cin + cout + dil)
    x = rng.standard_normal((n, cin, h, w), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    sh = rng.standard_normal(cin).astype(np.float32)
    res = rng.standard_normal((n, cout, h, w), dtype=np.float32)
    act = np.maximum(x * sc[None, :, None, None] + sh[None, :, None, None], 0)
    ref = nnops.conv2d(act, wt, 1, dil, dil) + res
    xa = K.Act.from_nchw(dev(x))
    y = K.conv2d_winograd(xa, K.pack_weight_wino(dev(wt), tile=tile), dil=dil, in_affine=(dev(sc), dev(sh)), in_relu=True,
                          res=K.Act.from_nchw(dev(res)))
    np.testing.assert_allclose(y.nchw().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    y2 = K.conv2d_winograd(xa, K.pack_weight_wino(dev(wt), tile=tile), dil=dil)        # no prologue / residual
    np.testing.assert_allclose(y2.nchw().cpu().numpy(), nnops.conv2d(x, wt, 1, dil, dil), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tile", [2, 4, 6])
@pytest.mark.parametrize("cin,cout,dil,n,h,w", [(32, 64, 1, 2, 12, 14), (64, 32, 2, 1, 13, 17), (48, 36, 12, 2, 16, 22)])
def test_winograd_wgrad_vs_autograd(K, cin, cout, dil, n, h, w, tile, gemm_route):
    rng = np.random.default_rng(cin * 3 + cout + dil)
    x = torch.from_numpy(rng.standard_normal((n, cin, h, w), dtype=np.float32))
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)).requires_grad_(True)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(cin).astype(np.float32))
    act = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    y = torch.nn.functional.conv2d(act, wt, dilation=dil, padding=dil)
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(gy)
    dw = K.conv2d_wgrad_winograd(K.Act.from_nchw(x.cuda()), K.Act.from_nchw(gy.cuda()), cout, cin, dil=dil,
                                 in_affine=(sc.cuda(), sh.cuda()), in_relu=True, tile=tile)
    np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=1e-3)
    # data gradient through the flipped Winograd filter
    xg = torch.from_numpy(rng.standard_normal((n, cin, h, w), dtype=np.float32)).requires_grad_(True)
    torch.nn.functional.conv2d(xg, wt.detach(), dilation=dil, padding=dil).backward(gy)
    if cout % 16 == 0:
        dx = K.conv2d_winograd(K.Act.from_nchw(gy.cuda()), K.pack_weight_wino(wt.detach().cuda(), flip=True, tile=tile), dil=dil)
        np.testing.assert_allclose(dx.nchw().cpu().numpy(), xg.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_pack_cache_dies_with_parameter(K):
    """Successive same-shaped parameters (recycled id() and storage address) must never see each other's packs."""
    for tile in (0, 2, 4, 6):
        for seed in range(6):
            w = torch.nn.Parameter(torch.full((32, 16, 3, 3), float(seed + 1), device="cuda"))
            pw = K.packed(w) if tile == 0 else K.packed_wino(w, tile=tile)
            assert pw is (K.packed(w) if tile == 0 else K.packed_wino(w, tile=tile))       # cached
            ref = K.pack_weight(w.detach()) if tile == 0 else K.pack_weight_wino(w.detach(), tile=tile)
            assert torch.equal(pw.t, ref.t)
            with torch.no_grad():
                w.mul_(2.0)                                                                 # in-place update -> re-pack
            pw2 = K.packed(w) if tile == 0 else K.packed_wino(w, tile=tile)
            assert pw2 is not pw and torch.allclose(pw2.t, 2 * ref.t)
            del w, pw, pw2, ref


@pytest.mark.parametrize("tag", ["x4", "ragged"])
def test_m2f_fused_score_golden(K, tag):
    """8f-2: batched-GEMM mask prediction + fused upsample/sigmoid/class-mix/max against the reference's op chain."""
    g = golden("m2f_fused")
    image, crop = tuple(int(v) for v in g[tag + "_image"]), tuple(int(v) for v in g[tag + "_crop"])
    lg = K.m2f_mask_logits(dev(g[tag + "_embed"]), dev(g[tag + "_features"]))                 # [B,h,w,Q]
    np.testing.assert_allclose(lg.permute(0, 3, 1, 2).cpu().numpy()[:, ::7], g[tag + "_masks_sub"], rtol=1e-5, atol=1e-5)
    s = K.m2f_score_fused(dev(g[tag + "_cls"]), lg, image, crop)
    np.testing.assert_allclose(s.cpu().numpy(), g[tag + "_score"], rtol=1e-5, atol=1e-5)
    s_full = K.m2f_score_fused(dev(g[tag + "_cls"]), lg, image)                               # no crop
    want = om2f.anomaly_score_from_features(g[tag + "_cls"], g[tag + "_embed"], g[tag + "_features"], image, image)
    np.testing.assert_allclose(s_full.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


def test_m2f_fused_equals_unfused_at_full_size(K):
    """BASELINE C5 size (1x100x256x512 -> 1024x2048): the fused kernel equals the unfused chain (explicit upsample,
    then the full-resolution score kernel) and never materialises the 839 MB tensor."""
    gen = torch.Generator(device="cuda").manual_seed(3)
    emb = torch.randn(1, 100, 256, device="cuda", generator=gen) * 0.2
    feat = torch.randn(1, 256, 256, 512, device="cuda", generator=gen)
    cls = torch.randn(1, 100, 20, device="cuda", generator=gen) * 2
    lg = K.m2f_mask_logits(emb, feat)
    ref_lg = torch.einsum("bqc,bchw->bqhw", emb, feat)
    assert (lg.permute(0, 3, 1, 2) - ref_lg).abs().max().item() < 2e-4 * ref_lg.abs().max().item()
    fused = K.m2f_score_fused(cls, lg, (1024, 2048))
    up = torch.nn.functional.interpolate(lg.permute(0, 3, 1, 2).contiguous(), size=(1024, 2048), mode="bilinear", align_corners=False)
    unfused = K.m2f_score(cls, up, (1024, 2048))
    assert (fused - unfused).abs().max().item() < 2e-5


@pytest.mark.parametrize("cin,cout,r,stride,dil,n,h,w,wino", [
    (64, 128, 1, 1, 1, 2, 16, 24, False),      # persistent GEMM kernel
    (32, 128, 1, 1, 1, 1, 9, 7, False),        # ragged rows (63 pixels: one partial row group)
    (32, 96, 3, 1, 2, 2, 11, 13, False),       # implicit GEMM, 128x128 tile
    (32, 48, 3, 2, 1, 2, 17, 19, False),       # 256x64 tile (K <= 64), stride 2
    (32, 64, 3, 1, 1, 2, 12, 14, True),        # Winograd F(4x4): statistics from the output transform
    (64, 32, 3, 1, 4, 1, 13, 17, True),
    (32, 64, 3, 1, 1, 2, 12, 14, 6),           # Winograd F(6x6): the LDS output transform's statistics (two tiles per wave at K=64)
    (64, 32, 3, 1, 2, 1, 13, 17, 6),           # ... four tiles per wave
    (16, 272, 3, 1, 1, 1, 9, 8, 6),            # ... one tile per wave, two channel groups (the second ragged)
])
def test_batchnorm_statistics_from_the_producing_kernel(K, cin, cout, r, stride, dil, n, h, w, wino, gemm_route):
    """want_stats: the conv epilogue / Winograd output transform leaves per-channel partial sums; bn_fold(train=True)
    built from them must equal bn_fold on a statistics pass over the stored activation (residual included)."""
    rng = np.random.default_rng(cin + cout + r + dil)
    x = K.Act.from_nchw(dev(rng.standard_normal((n, cin, h, w), dtype=np.float32)))
    wt = dev((rng.standard_normal((cout, cin, r, r), dtype=np.float32) / np.sqrt(cin * r * r)).astype(np.float32))
    pad = dil if r == 3 else 0
    oh, ow = K.conv_out_size(h, r, stride, dil, pad), K.conv_out_size(w, r, stride, dil, pad)
    res = K.Act.from_nchw(dev(rng.standard_normal((n, cout, oh, ow), dtype=np.float32)))
    if wino:
        y = K.conv2d_winograd(x, K.pack_weight_wino(wt, tile=6 if wino == 6 else 4), dil=dil, res=res, want_stats=True)
    else:
        y = K.conv2d(x, K.pack_weight(wt), stride=stride, dil=dil, pad=pad, res=res, want_stats=True)
    assert y.stats is not None
    bn_a, bn_b = torch.nn.BatchNorm2d(cout).cuda(), torch.nn.BatchNorm2d(cout).cuda()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_()
        bn_b.load_state_dict(bn_a.state_dict())
    st_a = K.bn_fold(bn_a, y, train=True)
    assert y.stats is None                       # consumed
    st_b = K.bn_fold(bn_b, y, train=True)
    for a, b in ((st_a.scale, st_b.scale), (st_a.shift, st_b.shift), (st_a.save_mean, st_b.save_mean),
                 (st_a.save_invstd, st_b.save_invstd), (bn_a.running_mean, bn_b.running_mean), (bn_a.running_var, bn_b.running_var)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    ref = torch.nn.functional.batch_norm(y.nchw(), None, None, bn_a.weight, bn_a.bias, training=True, eps=bn_a.eps)
    got = y.nchw() * st_a.scale[None, :, None, None] + st_a.shift[None, :, None, None]
    np.testing.assert_allclose(got.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_conv_paths_random_shapes(K, gemm_route):
    """Seeded sweep over the three forward paths (persistent GEMM, implicit GEMM, Winograd m = 2 / 4) with odd image
    sizes, channel counts around the tile edges, every dilation of the network, prologue / residual / ReLU / statistics
    switches -- against torch's CPU convolution."""
    rng = np.random.default_rng(2024)
    checked = {"gemm_nt": 0, "conv_igemm": 0, "wino2": 0, "wino4": 0, "wino6": 0}
    for case in range(40):
        r = int(rng.choice([1, 3, 3]))
        cin = int(rng.choice([16, 32, 48, 64, 80, 144, 272]))
        cout = int(rng.choice([20, 48, 64, 68, 128, 132, 192, 260]))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(3, 40)), int(rng.integers(3, 40))
        stride = int(rng.choice([1, 1, 1, 2]))
        dil = int(rng.choice([1, 2, 4, 12, 24, 36])) if r == 3 else 1
        pad = dil if r == 3 else 0
        use_aff, use_res, out_relu = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        x = rng.standard_normal((n, cin, h, w), dtype=np.float32)
        wt = (rng.standard_normal((cout, cin, r, r), dtype=np.float32) / np.sqrt(cin * r * r)).astype(np.float32)
        sc, sh = rng.uniform(0.5, 1.5, cin).astype(np.float32), rng.standard_normal(cin).astype(np.float32)
        a = torch.from_numpy(x)
        if use_aff:
            a = torch.relu(a * torch.from_numpy(sc)[None, :, None, None] + torch.from_numpy(sh)[None, :, None, None])
        ref = torch.nn.functional.conv2d(a, torch.from_numpy(wt), stride=stride, dilation=dil, padding=pad)
        res = rng.standard_normal(tuple(ref.shape), dtype=np.float32)
        if use_res:
            ref = ref + torch.from_numpy(res)
        ref = ref.numpy()
        xa = K.Act.from_nchw(dev(x))
        aff = (dev(sc), dev(sh)) if use_aff else None
        resa = K.Act.from_nchw(dev(res), ld=((cout + 3) // 4) * 4) if use_res else None
        tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
        # direct / GEMM path
        y = K.conv2d(xa, K.pack_weight(dev(wt)), stride=stride, dil=dil, pad=pad, in_affine=aff, in_relu=use_aff, res=resa,
                     out_relu=out_relu, want_stats=True)
        want = np.maximum(ref, 0) if out_relu else ref
        np.testing.assert_allclose(y.nchw().cpu().numpy(), want, rtol=0, atol=tol, err_msg=f"case {case} direct")
        if y.stats is not None:                          # statistics of exactly what was stored
            got_sum = y.stats[:, 0].double().sum(0).cpu().numpy()
            np.testing.assert_allclose(got_sum, want.astype(np.float64).sum((0, 2, 3)), rtol=1e-4, atol=1e-2, err_msg=f"case {case} stats")
        checked["gemm_nt" if (r == 1 and stride == 1 and cout > 64 and cin >= 32) else "conv_igemm"] += 1
        # Winograd path
        if r == 3 and stride == 1 and cout % 4 == 0:
            for tile in (2, 4, 6):
                yw = K.conv2d_winograd(xa, K.pack_weight_wino(dev(wt), tile=tile), dil=dil, in_affine=aff, in_relu=use_aff, res=resa)
                np.testing.assert_allclose(yw.nchw().cpu().numpy(), ref, rtol=0, atol=tol, err_msg=f"case {case} wino{tile}")
                checked[f"wino{tile}"] += 1
    assert all(v >= 4 for v in checked.values()), checked


def _bf16_planes_decode(planes, batch, kpad, c):
    """The three bf16 planes of mss_gemm_split_weights_bf16x3 (include/mss_hip.h) back as float64 [3][batch][kpad][c]."""
    nk = c // 16
    raw = planes.view(torch.int16).view(batch, kpad // 128, nk, 3, 128, 2, 8)       # [b][block][K-step][plane][row][half][8 k]
    # (rows as they are, k 0..7 then k 8..15: no half swap since round 6, csrc/mss_bf16x3.h)
    bits = (raw.to(torch.int32) & 0xffff) << 16
    vals = bits.view(torch.float32).double()                                        # bf16 -> fp32 is exact
    return vals.permute(3, 0, 1, 4, 2, 5, 6).reshape(3, batch, kpad, c)


@pytest.mark.parametrize("batch,kpad,c", [(1, 128, 16), (3, 256, 80), (2, 384, 512)])
def test_bf16x3_weight_planes_are_an_exact_three_term_split(K, batch, kpad, c):
    """hi + mid + lo == w exactly, each term a bf16 (16 low bits zero by construction of the format), |mid| <= 2^-8 |hi|-ish and
    |lo| <= 2^-16: the layout of include/mss_hip.h and the exactness the six-product form relies on."""
    torch.manual_seed(kpad + c)
    w = torch.randn(batch, kpad, c, device="cuda") * torch.exp(torch.randn(batch, kpad, 1, device="cuda") * 3)
    w[0, 0, :4] = torch.tensor([0.0, 1.0, -3.0e-30, 65504.0], device="cuda")
    planes = K.split_planes(w, kpad, c)
    assert planes.numel() == batch * kpad * c * 6
    hi, mid, lo = _bf16_planes_decode(planes, batch, kpad, c)
    assert torch.equal(hi + mid + lo, w.double())
    nz = w != 0
    assert (mid.abs()[nz] <= w.double().abs()[nz] * 2.0 ** -8).all() and (lo.abs()[nz] <= w.double().abs()[nz] * 2.0 ** -16).all()


def _run_gemm(K, x, w, k, split, in_affine=None, out_affine=None, out_relu=False, res=None, res_mask=False, want_stats=False, y=None):
    """x [batch][rows][c], w [batch][kpad][c] (rows >= k zero) through mss_conv2d_forward_f32 on the chosen GEMM route.
    `y`: an output VIEW [batch][rows][k] with unit stride in k (row pitch and start address are the view's) instead of a fresh tensor."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    batch, rows, c = x.shape
    kpad = w.shape[1]
    if y is None:
        y = torch.full((batch, rows, k), float("nan"), device="cuda")
    assert y.shape == (batch, rows, k) and y.stride(2) == 1
    a = MssConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), y.data_ptr()
    planes = K.split_planes(w, kpad, c) if split else None
    a.w_split = ptr(planes)
    if in_affine is not None:
        a.in_scale, a.in_shift, a.in_relu = ptr(in_affine[0]), ptr(in_affine[1]), 1
    if out_affine is not None:
        a.out_scale, a.out_shift = ptr(out_affine[0]), ptr(out_affine[1])
    a.out_relu = int(out_relu)
    if res is not None:
        a.res, a.ldres, a.res_mask = ptr(res), k, int(res_mask)
    stats = None
    if want_stats:
        stats = torch.empty((-(-rows // 64), 2, k), device="cuda")
        a.stats = ptr(stats)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, rows, c, c
    a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, rows, k, kpad, y.stride(1)
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    if batch > 1:
        a.batch, a.x_bs, a.w_bs, a.y_bs = batch, rows * c, kpad * c, y.stride(0)
    assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == (3 if split else 1)
    call("mss_conv2d_forward_f32", ctypes.byref(a))
    return y, stats


@pytest.mark.parametrize("rows,c,k,batch", [(300, 64, 128, 1), (1000, 256, 192, 1), (257, 512, 256, 3), (4096, 1024, 128, 2), (70000, 48, 512, 1),
                                            (131, 2048, 1024, 1), (66000, 256, 256, 2),
                                            (9216, 4096, 256, 36)])       # 5.4 GB of X': beyond 32-bit offsets as a whole (16 x 700 x 700 ASPP), 151 MB per entry
def test_bf16x3_gemm_is_fp32_accurate(K, rows, c, k, batch):
    """The split-bf16 route (MssConvArgs.w_split; six bf16 MFMAs per block on operands split into three bf16 terms): against
    float64 it must be as accurate as the native fp32 MFMA kernel (both ~1e-7 of max|y|). Shapes cover both tile widths
    (128 / 256), ragged row counts, channel counts that leave a padded column tile, the shortest reduction (3 K-steps) and
    batched products; the operands span 2^+-10 in scale so the low planes are exercised."""
    from multishiftseg_amd import _lib
    torch.manual_seed(rows + c)
    x = torch.relu(torch.randn(batch, rows, c, device="cuda")) * torch.exp2(torch.randint(-10, 11, (1, 1, c), device="cuda").float())
    kpad = _lib.value("mss_conv2d_kpad", k)
    w = torch.zeros(batch, kpad, c, device="cuda")
    w[:, :k] = torch.randn(batch, k, c, device="cuda") / c ** 0.5 * torch.exp2(-torch.randint(-10, 11, (1, 1, c), device="cuda").float())
    sel = torch.randint(0, rows, (min(rows, 2048),), device="cuda")
    ref = torch.einsum("bmc,bkc->bmk", x[:, sel].double(), w[:, :k].double())
    y32, _ = _run_gemm(K, x, w, k, split=False)
    y3, _ = _run_gemm(K, x, w, k, split=True)
    assert torch.isfinite(y3).all()
    scale = ref.abs().max().item()
    e32 = (y32[:, sel].double() - ref).abs().max().item() / scale
    e3 = (y3[:, sel].double() - ref).abs().max().item() / scale
    assert e32 < 2e-6 and e3 < 2e-6 and e3 < 2 * e32 + 1e-7, (e32, e3)
    assert (y3.double() - y32.double()).abs().max().item() < 4e-6 * scale       # every element, not only the sampled rows
    assert not torch.equal(y32, y3)          # it really is a different evaluation


@pytest.mark.parametrize("rows,c,k,batch,pitch,lead", [
    (1000, 256, 130, 1, 130, 0),       # 130 output channels: rows are 8-byte aligned only -> the 32x32x16 form with scalar stores
    (900, 128, 128, 2, 131, 1),        # a column window of a wider buffer, start and pitch off the 16-byte grid
    (1300, 256, 256, 2, 260, 4),       # 16-byte aligned window with a pitch wider than the product: the 16x16x32 form's float4 stores
    (700, 512, 192, 1, 200, 8)])       # padded column tile (192 of 256) inside an aligned window
def test_bf16x3_gemm_into_an_output_window(K, rows, c, k, batch, pitch, lead):
    """The 16x16x32 split kernels store 16 bytes per lane and are taken only when the output's start, pitch and batch stride sit on the
    16-byte grid (split_mf16_ok, gemm_bf16x3.hip); everything else goes to the 32x32x16 kernels' 4-byte stores. Both ways the product
    lands in a WINDOW of a larger buffer: the window against float64 and against the native route, the rest of the buffer untouched."""
    from multishiftseg_amd import _lib
    torch.manual_seed(rows + k)
    x = torch.randn(batch, rows, c, device="cuda")
    kpad = _lib.value("mss_conv2d_kpad", k)
    w = torch.zeros(batch, kpad, c, device="cuda")
    w[:, :k] = torch.randn(batch, k, c, device="cuda") / c ** 0.5
    ref = torch.einsum("bmc,bkc->bmk", x.double(), w[:, :k].double())
    scale = ref.abs().max().item()
    out = {}
    for split in (False, True):
        buf = torch.full((batch, rows + 2, pitch), 7.0, device="cuda")
        win = buf[:, 1:rows + 1, lead:lead + k] if lead + k <= pitch else None
        assert win is not None and win.data_ptr() % 16 == (4 * (pitch + lead)) % 16
        _run_gemm(K, x, w, k, split=split, y=win)
        if split:                                  # the form really follows the window's alignment
            aligned = k % 4 == 0 and pitch % 4 == 0 and lead % 4 == 0
            assert _lib.value("mss_gemm_split_last_mfma") == (16 if aligned else 32)
        assert (win.double() - ref).abs().max().item() < 2e-6 * scale, split
        outside = buf.clone()
        outside[:, 1:rows + 1, lead:lead + k] = 7.0
        assert (outside == 7.0).all(), ("wrote outside its window", split)
        out[split] = win.clone()
    assert (out[True].double() - out[False].double()).abs().max().item() < 4e-6 * scale


@pytest.mark.parametrize("rows,c,k,per_sample", [(1500, 256, 384, False), (3 * 128 * 5, 128, 256, True), (40000, 512, 1024, False),
                                                 (3 * 1320, 256, 384, True)])     # 1320 rows per image: 128-row tiles straddle images (ROWAFF kernel)
def test_bf16x3_gemm_fused_prologue_and_epilogue(K, rows, c, k, per_sample):
    """Everything gemm_nt_kernel fuses, on the split route: BatchNorm + ReLU prologue (one affine, or one per sample = the Dropout2d
    fold), per-channel output affine + residual + ReLU, the ReLU-gate form of the residual (res_mask), and the per-64-row partial
    sums for the next layer's BatchNorm statistics -- against float64, and against the native route to fp32 rounding."""
    from multishiftseg_amd import _lib
    torch.manual_seed(rows)
    x = torch.randn(1, rows, c, device="cuda")
    kpad = _lib.value("mss_conv2d_kpad", k)
    w = torch.zeros(1, kpad, c, device="cuda")
    w[:, :k] = torch.randn(1, k, c, device="cuda") / c ** 0.5
    n_img = 3 if per_sample else 1
    sc, sh = torch.rand(n_img, c, device="cuda") + 0.5, torch.randn(n_img, c, device="cuda") * 0.3
    osc, osh = torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda")
    res = torch.randn(rows, k, device="cuda")
    img_of_row = torch.arange(rows, device="cuda") // (rows // n_img)
    xa = torch.relu(x[0].double() * sc.double()[img_of_row] + sh.double()[img_of_row])
    lin = (xa @ w[0, :k].double().T) * osc.double() + osh.double()

    def run(split, **kw):
        import ctypes
        from multishiftseg_amd._lib import MssConvArgs, call, ptr
        y = torch.full((rows, k), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(x), ptr(w), ptr(y)
        planes = K.split_planes(w, kpad, c) if split else None
        a.w_split = ptr(planes)
        a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
        a.in_ss_stride = c if per_sample else 0
        a.out_scale, a.out_shift, a.out_relu = ptr(osc), ptr(osh), int(kw.get("relu", False))
        a.res, a.ldres, a.res_mask = ptr(res), k, int(kw.get("mask", False))
        stats = torch.empty((-(-rows // 64), 2, k), device="cuda")
        a.stats = ptr(stats)
        hw = rows // n_img
        a.N, a.H, a.W, a.C, a.ldx = n_img, 1, hw, c, c
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, hw, k, kpad, k
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        straddle = per_sample and (rows // n_img) % 128 != 0       # native: conv_igemm_kernel's per-sample path; split: the ROWAFF kernel
        assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == ((4 if split else 0) if straddle else (3 if split else 1))
        call("mss_conv2d_forward_f32", ctypes.byref(a))
        return y, stats
    for kw, ref in (({"relu": True}, torch.relu(lin + res.double())), ({"mask": True}, torch.where(res.double() > 0, lin, torch.zeros_like(lin)))):
        y32, st32 = run(False, **kw)
        y3, st3 = run(True, **kw)
        scale = ref.abs().max().item()
        e32, e3 = (y32.double() - ref).abs().max().item() / scale, (y3.double() - ref).abs().max().item() / scale
        assert e32 < 2e-6 and e3 < 2e-6 and e3 < 2 * e32 + 1e-7, (kw, e32, e3)
        pad = (-rows) % 64
        yp = torch.cat([y3.double(), torch.zeros(pad, k, device="cuda", dtype=torch.float64)]).view(-1, 64, k)
        np.testing.assert_allclose(st3[:, 0].double().cpu().numpy(), yp.sum(1).cpu().numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(st3[:, 1].double().cpu().numpy(), (yp * yp).sum(1).cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cin,cout,r,stride,dil,n,h,w,affine", [
    (64, 128, 3, 1, 1, 2, 70, 94, True),       # mod2.block1.conv1: 64 input channels stay off Winograd F(4x4); BN + ReLU prologue
    (256, 512, 3, 2, 1, 2, 61, 59, True),      # mod4.block1.conv1: stride 2, odd size, wide tile when large
    (256, 512, 1, 2, 1, 2, 61, 59, True),      # mod4.block1.proj_conv: 1x1 stride 2
    (32, 200, 3, 1, 2, 1, 40, 33, False),      # K = 200 (padded column tile), dilation 2, no prologue
    (128, 256, 3, 1, 12, 1, 16, 32, False),    # rate 12 on a small map: most taps in the padding
])
def test_bf16x3_implicit_gemm_vs_float64(K, cin, cout, r, stride, dil, n, h, w, affine):
    """The implicit-GEMM layers on the split route (gemm_nt_bf16x3_kernel<..., CONV>; mss_conv2d_forward_route answers 4): taps folded
    into one long reduction on the weight side, per-tap pixel addressing with zero padding AFTER the BatchNorm + ReLU prologue on
    the activation side -- against a float64 convolution and against the native implicit-GEMM kernel."""
    import ctypes
    from multishiftseg_amd import _lib
    torch.manual_seed(cin + cout + r)
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = torch.randn(cout, cin, r, r, device="cuda") / (cin * r * r) ** 0.5
    aff = (torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3) if affine else None
    pad = dil if r == 3 else 0
    outs = {}
    for route in ("native", "bf16x3"):
        K.set_gemm_route(route)
        try:
            pw = K.pack_weight(wt)
            a = K._conv_args(x, pw, None, stride, dil, pad, aff, affine, None, False, None)
            a.OH, a.OW = K.conv_out_size(h, r, stride, dil, pad), K.conv_out_size(w, r, stride, dil, pad)
            assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == (4 if route == "bf16x3" else 0)
            outs[route] = K.conv2d(x, pw, stride=stride, dil=dil, pad=pad, in_affine=aff, in_relu=affine, want_stats=True)
        finally:
            K.set_gemm_route(None)
    xa = x.nchw().double()
    if affine:
        xa = torch.relu(xa * aff[0].double().view(1, -1, 1, 1) + aff[1].double().view(1, -1, 1, 1))
    ref = torch.nn.functional.conv2d(xa, wt.double(), stride=stride, dilation=dil, padding=pad)
    scale = ref.abs().max().item()
    e0 = (outs["native"].nchw().double() - ref).abs().max().item() / scale
    e1 = (outs["bf16x3"].nchw().double() - ref).abs().max().item() / scale
    assert e0 < 2e-6 and e1 < 2e-6 and e1 < 2 * e0 + 1e-7, (e0, e1)
    assert not torch.equal(outs["native"].nchw(), outs["bf16x3"].nchw())
    st = outs["bf16x3"].stats                                     # the BatchNorm partial sums of the epilogue, per 64 output rows
    y = outs["bf16x3"].buf.view(-1, cout).double()
    padr = (-y.shape[0]) % 64
    yp = torch.cat([y, torch.zeros(padr, cout, device="cuda", dtype=torch.float64)]).view(-1, 64, cout)
    np.testing.assert_allclose(st[:, 0].double().cpu().numpy(), yp.sum(1).cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("tile", [2, 4, 6])
@pytest.mark.parametrize("k,c", [(256, 128), (200, 48), (384, 1040)])
def test_bf16x3_winograd_planes_straight_from_the_weight(K, tile, k, c):
    """mss_wino_pack_split_bf16x3 (3x3 weight -> the split-bf16 planes of its Winograd-domain form, U never written in fp32) must be
    bit-identical to mss_wino_pack_weights_f32 followed by mss_gemm_split_weights_bf16x3 -- also on the padding rows (k >= K) -- and a
    WinoWeight made lazily under the split route must hand out the same U on demand."""
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import call, ptr
    torch.manual_seed(tile * 1000 + k + c)
    w = (torch.randn(k, c, 3, 3, device="cuda") * torch.logspace(-3, 1, k, device="cuda")[:, None, None, None]).contiguous()
    kpad = _lib.value("mss_conv2d_kpad", k)
    P = (tile + 2) ** 2
    u = torch.empty(P, kpad, c, device="cuda")
    call("mss_wino_pack_weights_f32", ptr(w), ptr(u), k, c, kpad, c, tile)
    want = K.split_planes(u, kpad, c)
    got = torch.full_like(want, 0x5a)
    call("mss_wino_pack_split_bf16x3", ptr(w), ptr(got), k, c, kpad, tile)
    assert torch.equal(got, want)
    K.set_gemm_route("bf16x3")
    try:
        ww = K.pack_weight_wino(w, tile=tile)
        assert ww._t is None                                   # nothing in fp32 yet
        K.w_split_of(ww)
        assert ww._t is None and torch.equal(ww.planes, want)
        assert torch.equal(ww.t, u)                            # ... and U on demand
    finally:
        K.set_gemm_route(None)


def test_bf16x3_ticket_tile_order_on_two_streams_at_once(K):
    """The 128 x 256 split kernels with a prologue walk their tiles off a device ticket counter (gemm_bf16x3.hip, split_dyn_tiles): a
    counter pair may only be shared by launches that cannot overlap, so every stream has its own ring. Two streams launch such
    products at the same time, 40 each -- one long, one short, so launches of one stream overlap several of the other -- and
    every output must be bit-identical to the same product computed alone."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(11)

    def product(rows, c, k):
        x = torch.randn(rows, c, device="cuda")
        kpad = _lib.value("mss_conv2d_kpad", k)
        w = torch.zeros(1, kpad, c, device="cuda")
        w[:, :k] = torch.randn(1, k, c, device="cuda") / c ** 0.5
        sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.3
        planes = K.split_planes(w, kpad, c)

        def run(y):
            a = MssConvArgs()
            a.x, a.w, a.y, a.w_split = ptr(x), ptr(w), ptr(y), ptr(planes)
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
            a.N, a.H, a.W, a.C, a.ldx = 1, 1, rows, c, c
            a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, rows, k, kpad, k
            a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
            assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == 3
            call("mss_conv2d_forward_f32", ctypes.byref(a))
        keep = (x, w, sc, sh, planes)
        return run, keep, (rows, k)
    jobs = [product(40000, 512, 512), product(9000, 256, 256)]
    want = []
    for run, _, (rows, k) in jobs:
        y = torch.empty(rows, k, device="cuda")
        run(y)
        want.append(y)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.full((rows, k), float("nan"), device="cuda") for _ in range(40)] for _, _, (rows, k) in jobs]
    for i in range(40):
        for j, (run, _, _) in enumerate(jobs):
            with torch.cuda.stream(streams[j]):
                run(outs[j][i])
    torch.cuda.synchronize()
    for j in range(2):
        for y in outs[j]:
            assert torch.equal(y, want[j])


def test_bf16x3_ticket_kernels_in_two_graphs_replayed_concurrently(K, monkeypatch):
    """VERDICT r05 weak 9 / ADVICE r05: a captured launch bakes its ticket slot into the graph, and the graph may be replayed on any
    stream, beside other graphs and beside eager launches of the stream it was captured on. Captured launches therefore get PRIVATE
    slots (gemm_bf16x3.hip, mss_sched_slot). Two graphs -- both captured on torch's capture stream, each holding two prologue (ticket
    order) products -- are replayed 30 times at once on two streams while a third stream launches the same kernels eagerly: every output
    bit-identical to the product computed alone. The static-walk fallback the launcher takes when no slot can be had
    (MSS_GEMM_SPLIT_STATIC=1 forces it) computes the same bits."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(12)

    def product(rows, c, k):
        x = torch.randn(rows, c, device="cuda")
        kpad = _lib.value("mss_conv2d_kpad", k)
        w = torch.zeros(1, kpad, c, device="cuda")
        w[:, :k] = torch.randn(1, k, c, device="cuda") / c ** 0.5
        sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.3
        planes = K.split_planes(w, kpad, c)

        def run(y):
            a = MssConvArgs()
            a.x, a.w, a.y, a.w_split = ptr(x), ptr(w), ptr(y), ptr(planes)
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
            a.N, a.H, a.W, a.C, a.ldx = 1, 1, rows, c, c
            a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, rows, k, kpad, k
            a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
            assert _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == 3
            call("mss_conv2d_forward_f32", ctypes.byref(a))
        return run, (x, w, sc, sh, planes), (rows, k)
    jobs = [product(40000, 512, 512), product(9000, 256, 256), product(30000, 256, 1024), product(5000, 1024, 256)]
    want = []
    for run, _, (rows, k) in jobs:
        y = torch.empty(rows, k, device="cuda")
        run(y)
        want.append(y)
    torch.cuda.synchronize()
    monkeypatch.setenv("MSS_GEMM_SPLIT_STATIC", "1")          # the fallback: same tiles, static order
    for (run, _, (rows, k)), ref in zip(jobs, want):
        y = torch.full((rows, k), float("nan"), device="cuda")
        run(y)
        assert torch.equal(y, ref)
    monkeypatch.delenv("MSS_GEMM_SPLIT_STATIC")
    outs = [torch.full((rows, k), float("nan"), device="cuda") for _, _, (rows, k) in jobs]
    graphs = []
    for pair in ((0, 1), (2, 3)):                              # graph A: jobs 0, 1; graph B: jobs 2, 3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for j in pair:
                jobs[j][0](outs[j])
        graphs.append(g)
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    eager = [[torch.full((rows, k), float("nan"), device="cuda") for _ in range(30)] for _, _, (rows, k) in jobs[:2]]
    for it in range(30):
        for o in outs:
            o.fill_(float("nan"))
        torch.cuda.synchronize()
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
        with torch.cuda.stream(streams[2]):
            for j in range(2):
                jobs[j][0](eager[j][it])
        torch.cuda.synchronize()
        for o, ref in zip(outs, want):
            assert torch.equal(o, ref), it
    for j in range(2):
        for y in eager[j]:
            assert torch.equal(y, want[j])



@pytest.mark.parametrize("rows,c,k,batch", [(65536, 512, 1024, 1), (2112, 1024, 512, 16), (777, 256, 256, 5)])
def test_bf16x3_mfma16_lds_dma_has_no_race(K, monkeypatch, rows, c, k, batch):
    """The 16x16x32 form stages the weight planes by LDS-DMA and orders them by hand (counted s_waitcnt vmcnt + raw s_barrier; a read
    placed one wait too early passes whenever the DMA happens to land first -- cdna_hip_programming.md 5, "Read a staged buffer one phase
    AFTER the wait that retires it"). 60 launches of the same product beside a second stream that keeps the memory system busy: every
    output bit-identical to the first, and within fp32 rounding of the 32x32x16 form (another summation order, same six products)."""
    torch.manual_seed(rows)
    x = torch.randn(batch, rows, c, device="cuda")
    from multishiftseg_amd import _lib
    kpad = _lib.value("mss_conv2d_kpad", k)
    w = torch.zeros(batch, kpad, c, device="cuda")
    w[:, :k] = torch.randn(batch, k, c, device="cuda") / c ** 0.5
    monkeypatch.setenv("MSS_GEMM_SPLIT_MFMA", "16")
    first, _ = _run_gemm(K, x, w, k, split=True)
    assert _lib.value("mss_gemm_split_last_mfma") == 16
    noise = torch.empty(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    for it in range(60):
        with torch.cuda.stream(side):
            noise.normal_()                                # HBM / L2 traffic beside the launch: DMA latencies vary
        y, _ = _run_gemm(K, x, w, k, split=True)
        assert torch.equal(y, first), it
    torch.cuda.synchronize()
    monkeypatch.setenv("MSS_GEMM_SPLIT_MFMA", "32")
    y32, _ = _run_gemm(K, x, w, k, split=True)
    assert _lib.value("mss_gemm_split_last_mfma") == 32
    scale = first.abs().max().item()
    assert (first.double() - y32.double()).abs().max().item() < 2e-6 * scale
    assert not torch.equal(first, y32)                     # it really is the other kernel



def test_bf16x3_route_is_taken_by_the_layer_wrappers(K):
    """kernels.set_gemm_route("bf16x3") / MSS_GEMM_SPLIT=1 must reach the kernel through every wrapper that builds MssConvArgs: a 1x1
    layer, a Winograd layer (with the 304 = 256 + 48 output split, whose 48-channel tail stays on the native narrow tile) and a
    Linear -- checked by the outputs differing in the last bits from the native route while agreeing to fp32 rounding."""
    from multishiftseg_amd import linear as L
    torch.manual_seed(3)
    x = K.Act(torch.randn(2, 40, 56, 256, device="cuda"))
    w1 = torch.randn(512, 256, 1, 1, device="cuda") / 16
    w3 = torch.randn(304, 256, 3, 3, device="cuda") / 48
    wl = torch.nn.Parameter(torch.randn(192, 256, device="cuda") / 16)
    outs = {}
    for route in ("native", "bf16x3"):
        K.set_gemm_route(route)
        try:
            outs[route] = (K.conv2d(x, K.pack_weight(w1)).nchw(), K.conv3x3(x, w3).nchw(), L.linear(x.buf.view(-1, 256), wl).detach().clone())
        finally:
            K.set_gemm_route(None)
    for a, b in zip(outs["native"], outs["bf16x3"]):
        assert not torch.equal(a, b)
        assert (a - b).abs().max().item() < 1e-4 * a.abs().max().item()         # (the Winograd layer: 2e-5 of max|y| between two fp32 evaluations)


@pytest.mark.parametrize("P,T,C,Ko,affine,ld_extra", [(36, 1100, 512, 256, False, 0), (1, 162629, 256, 256, False, 32), (1, 70000, 1280, 256, True, 0),
                                                      (64, 1936, 256, 256, False, 0), (2, 40000, 256, 384, False, 0),
                                                      (36, 9216, 4096, 256, False, 0)])     # 5.4 GB of X' in all: 64-bit position bases
def test_bf16x3_wgrad_tn_vs_float64(K, P, T, C, Ko, affine, ld_extra):
    """The TN weight-gradient product on the split-bf16 route (MssConvArgs.route = 1, gemm_tn_bf16x3_kernel: both operands split and
    transposed in the loader) against float64 and against the native kernels: row counts that are not multiples of 16 (the last
    split is shifted back and the rows it shares with its predecessor enter as zeros -- up to 52 of them here), batched positions,
    a dy that is a channel slice of a wider buffer, the BatchNorm + ReLU prologue on x. Deterministic (two runs bit-identical)."""
    import ctypes
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(T + C)
    xt = torch.randn(P, T, C, device="cuda")
    ldy = Ko + ld_extra
    dy_buf = torch.randn(P, T, ldy, device="cuda")
    sc = sh = None
    if affine:
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
    outs = {}
    for route in (0, 1, 1):
        du = torch.full((P, Ko, C), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x = ptr(xt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.route = route
        if affine:
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
        if P > 1:
            a.batch, a.x_bs, a.y_bs = P, T * C, T * ldy
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dy_buf), ldy, ptr(du), C, ptr(ws), wsb)
        outs.setdefault(route, []).append(du)
    assert torch.equal(outs[1][0], outs[1][1])                      # deterministic
    assert not torch.equal(outs[0][0], outs[1][0])                  # it really ran the other kernel
    rows = torch.randint(0, Ko, (32,), device="cuda")
    xa = xt.double()
    if affine:
        xa = torch.relu(xa * sc.double() + sh.double())
    for b in {0, P - 1}:
        ref = dy_buf[b][:, rows].double().T @ xa[b]
        scale = ref.abs().max().item()
        e0 = (outs[0][0][b, rows].double() - ref).abs().max().item() / scale
        e1 = (outs[1][0][b, rows].double() - ref).abs().max().item() / scale
        assert e0 < 5e-6 and e1 < 5e-6 and e1 < 2 * e0 + 2e-7, (e0, e1)
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-5 * outs[0][0].abs().max().item()


@pytest.mark.parametrize("mode", ["0", "1", "2", "4", "5", "6", "7"])
@pytest.mark.parametrize("P,T,C,Ko", [(3, 700, 512, 128), (2, 1000, 256, 72), (4, 37, 768, 256)])
def test_batched_wgrad_routes_vs_float64(K, monkeypatch, mode, P, T, C, Ko):
    """dU[p] = dY'[p]^T X'[p] (the Winograd-domain weight gradient) on its kernels -- the convolution-loader kernel (0),
    the TN kernel (1), the transposing-loader kernel with 128-wide (2) and 256-wide (4) c tiles, the round-4 LDS-free kernel
    with 128 x 128 (7; 5 = the default rule, which keeps these small products on the older kernels) and 64 x 128 tiles per wave
    (6), ragged and odd T, K not a multiple of 128 (falls back), with and without a pixel split -- against a float64 product;
    and twice with identical bits."""
    import ctypes
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    monkeypatch.setenv("MSS_WGRAD_TN", mode)
    torch.manual_seed(P * T + C)
    xt = torch.randn(P, T, C, device="cuda")
    dyt = torch.randn(P, T, Ko, device="cuda")
    kpad = (Ko + 3) // 4 * 4
    a = MssConvArgs()
    a.x = ptr(xt)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
    a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, kpad
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
    outs = []
    for _ in range(2):
        du = torch.full((P, kpad, C), float("nan"), device="cuda")
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb)
        outs.append(du)
    assert torch.equal(outs[0], outs[1])
    want = torch.einsum("ptk,ptc->pkc", dyt.double(), xt.double())
    err = (outs[0][:, :Ko].double() - want).abs().max().item()
    assert err <= 2e-6 * T ** 0.5 * 16, err
    assert not torch.isnan(outs[0]).any()


@pytest.mark.parametrize("n,h,w,cin,k,ld,c0,affine", [(1, 128, 160, 256, 19, 48, 20, True), (2, 96, 100, 128, 48, 48, 0, False),
                                                     (1, 130, 131, 256, 20, 20, 0, True), (1, 128, 129, 384, 34, 36, 0, True),
                                                     (1, 128, 128, 128, 64, 64, 0, False), (1, 150, 120, 256, 19, 48, 0, False)])
def test_narrow_wgrad_direct_kernel_vs_float64(K, monkeypatch, n, h, w, cin, k, ld, c0, affine):
    """r04: weight gradients with <= 64 output channels (the 19-channel heads over a slice of the 48-wide gradient buffer, bot_fine's
    48) on the LDS-free streaming kernel gemm_tn_narrow_kernel, with and without the BatchNorm + ReLU prologue on x, against the
    LDS kernel (MSS_WGRAD_NARROW=0) and a float64 product; odd pixel counts; twice with identical bits."""
    torch.manual_seed(h * w + k)
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    dybuf = torch.randn(n, h, w, ld, device="cuda")
    dy = K.Act(dybuf, k, c0)
    aff = (torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3) if affine else None
    outs = {}
    for mode in ("1", "1", "0"):
        monkeypatch.setenv("MSS_WGRAD_NARROW", mode)
        outs.setdefault(mode, []).append(K.conv2d_wgrad(x, dy, k, cin, 1, 1, in_affine=aff, in_relu=affine))
    assert torch.equal(outs["1"][0], outs["1"][1])
    xa = x.buf.double().view(-1, cin)
    if affine:
        xa = torch.relu(xa * aff[0].double() + aff[1].double())
    want = dybuf.double().view(-1, ld)[:, c0:c0 + k].t() @ xa
    scale = want.abs().max().item()
    e_new = (outs["1"][0].view(k, cin).double() - want).abs().max().item() / scale
    e_old = (outs["0"][0].view(k, cin).double() - want).abs().max().item() / scale
    assert e_new < 2e-6 and e_new < 4 * e_old + 1e-7, (e_new, e_old)
    assert not torch.equal(outs["1"][0], outs["0"][0]) or True      # (different summation orders; equality is not required)


@pytest.mark.parametrize("n,h,w,cin,k,relu", [(2, 128, 256, 1280, 256, True), (1, 200, 333, 256, 128, True), (1, 128, 256, 512, 256, False)])
def test_direct_wgrad_with_bn_relu_prologue_vs_float64(K, monkeypatch, n, h, w, cin, k, relu, gemm_route):
    """r04: a 1x1 layer's weight gradient whose forward reads relu(x * scale + shift) (bot_aspp over the five ASPP branches'
    BatchNorm + ReLU, deepv3.py:235-240) on the LDS-free kernel, the affine applied to the x registers at consume time, against
    the LDS kernel (MSS_WGRAD_TN_AFFINE=0) and a float64 product; twice with identical bits."""
    torch.manual_seed(h + w + cin)
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    dy = K.Act(torch.randn(n, h, w, k, device="cuda"))
    aff = (torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3)
    outs = {}
    for mode in ("1", "1", "0"):
        monkeypatch.setenv("MSS_WGRAD_TN_AFFINE", mode)
        outs.setdefault(mode, []).append(K.conv2d_wgrad(x, dy, k, cin, 1, 1, in_affine=aff, in_relu=relu))
    assert torch.equal(outs["1"][0], outs["1"][1])
    xa = x.buf.double().view(-1, cin) * aff[0].double() + aff[1].double()
    if relu:
        xa = torch.relu(xa)
    want = dy.buf.double().view(-1, k).t() @ xa
    scale = want.abs().max().item()
    e_new = (outs["1"][0].view(k, cin).double() - want).abs().max().item() / scale
    e_old = (outs["0"][0].view(k, cin).double() - want).abs().max().item() / scale
    assert e_new < 2e-6 and e_new < 4 * e_old + 1e-7, (e_new, e_old)


@pytest.mark.parametrize("c0,ld", [(0, 288), (32, 288), (4, 260)])
def test_direct_wgrad_takes_a_channel_slice_of_a_wider_gradient_buffer(K, monkeypatch, c0, ld):
    """r04: dy as 256 columns of a wider buffer (row stride ld) on the LDS-free kernel (forced: MSS_WGRAD_TN=7) against the same
    columns copied out densely -- the same bits -- and a float64 product."""
    monkeypatch.setenv("MSS_WGRAD_TN", "7")
    torch.manual_seed(c0 + ld)
    n, h, w, cin, k = 1, 200, 333, 256, 256
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    buf = torch.randn(n, h, w, ld, device="cuda")
    got = K.conv2d_wgrad(x, K.Act(buf, k, c0), k, cin, 1, 1)
    dense = K.conv2d_wgrad(x, K.Act(buf[..., c0:c0 + k].contiguous()), k, cin, 1, 1)
    assert torch.equal(got, dense)
    want = buf[..., c0:c0 + k].double().reshape(-1, k).t() @ x.buf.double().view(-1, cin)
    assert (got.view(k, cin).double() - want).abs().max().item() / want.abs().max().item() < 2e-6


@pytest.mark.parametrize("P,T,C,Ko,tail", [(36, 1100, 4096, 256, "1"), (36, 1100, 4096, 256, "0"), (9, 777, 2048, 1024, "1"), (5, 300, 4096, 512, "1")])
def test_direct_wgrad_tail_plan_vs_float64(K, monkeypatch, P, T, C, Ko, tail):
    """r04: more output tiles than wave slots and a mostly empty last round (36 x 2 x 32 = 2304 tiles on 1024 SIMDs, the ASPP F(4x4)
    product): gemm_tn_direct_kernel runs the whole rounds as unsplit tiles written straight to the result and cuts only the remaining
    tiles into row ranges (MSS_WGRAD_TN_TAIL=0: every tile split, all slabs reduced). Both against a float64 product, twice with
    identical bits, every element written."""
    import ctypes
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    monkeypatch.setenv("MSS_WGRAD_TN", "7")
    monkeypatch.setenv("MSS_WGRAD_TN_TAIL", tail)
    torch.manual_seed(P * T + C)
    xt = torch.randn(P, T, C, device="cuda")
    dyt = torch.randn(P, T, Ko, device="cuda")
    a = MssConvArgs()
    a.x = ptr(xt)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
    a.OH, a.OW, a.K, a.Kpad = 1, T, Ko, Ko
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    a.batch, a.x_bs, a.y_bs = P, T * C, T * Ko
    outs = []
    for _ in range(2):
        du = torch.full((P, Ko, C), float("nan"), device="cuda")
        ws, wsb = K._wgrad_workspace(a, C, "cuda")
        if ws is not None:
            ws.fill_(float("nan"))
        call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), Ko, ptr(du), C, ptr(ws), wsb)
        outs.append(du)
    assert torch.equal(outs[0], outs[1])
    assert not torch.isnan(outs[0]).any()
    err = 0.0
    for p0 in range(0, P, 4):                      # float64 reference in slices (the whole einsum would take 2 x 36 x 1100 x 4096 x 8 B)
        want = torch.einsum("ptk,ptc->pkc", dyt[p0:p0 + 4].double(), xt[p0:p0 + 4].double())
        err = max(err, (outs[0][p0:p0 + 4].double() - want).abs().max().item())
    assert err <= 2e-6 * T ** 0.5 * 16, err


@pytest.mark.parametrize("n,h,w", [(1, 64, 128), (2, 37, 53), (1, 1, 1), (1, 2, 33), (3, 90, 150), (1, 70, 1000), (1, 33, 2048)])
def test_fused_stem_conv_pool(K, n, h, w):
    """csrc/stem.hip: conv3x3(3 -> 64, padding 1) + MaxPool2d(3, 2, 1) in one kernel (wave per pooled row, MFMA from registers,
    pooling in the accumulator layout) against (a) the two-kernel path of this repository (im2col + K = 32 GEMM, then the pool
    kernel) and (b) torch in float64; odd and tiny sizes, widths beyond one segment, every border case of both paddings."""
    torch.manual_seed(n * h + w)
    img = torch.randn(n, 3, h, w, device="cuda")
    wt = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device="cuda") * 0.2)
    got = K.stem_conv_pool(img, wt)
    two = K.maxpool3s2(K.conv2d(K.stem_im2col(img), K.packed_stem(wt)))
    assert (got.N, got.H, got.W, got.C) == (two.N, two.H, two.W, two.C) == (n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 64)
    want = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(img.double(), wt.detach().double(), padding=1), 3, 2, 1)
    g = got.nchw().double()
    assert torch.isfinite(g).all()
    err, err2 = (g - want).abs().max().item(), (two.nchw().double() - want).abs().max().item()
    assert err <= 2e-6 * max(1.0, want.abs().max().item()), (err, err2)
    # the same products in the same K-order as the GEMM of the two-kernel path: the same numbers
    # (images of <= 8 pixels take the few-rows kernel in the two-kernel path: another summation order)
    assert n * h * w <= 8 or torch.equal(got.nchw(), two.nchw()), (got.nchw() - two.nchw()).abs().max().item()


@pytest.mark.parametrize("m,c,k", [(1, 4096, 256), (2, 4096, 256), (8, 64, 19), (3, 528, 100), (9, 4096, 256), (16, 4096, 256), (32, 512, 48),
                                   (33, 4096, 256)])
def test_few_rows_1x1_route(K, m, c, k):
    """1x1 convolutions over <= 32 pixels in all (ASPP's image-pooling branch: [N, 4096] -> 256, N = 16 at 16 x 768 x 768) take the
    wave-per-output-channel kernel of gemm.hip instead of one MFMA tile walking the whole reduction; 33 rows take the MFMA kernels.
    Against float64."""
    torch.manual_seed(m + c + k)
    x = torch.randn(m, 1, 1, c, device="cuda")
    w = torch.randn(k, c, 1, 1, device="cuda") / c ** 0.5
    y = K.conv2d(K.Act(x), K.pack_weight(w))
    want = x.view(m, c).double() @ w.view(k, c).double().t()
    got = y.buf.view(m, -1)[:, :k].double()
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())    # fp32 accumulation over up to 4096 terms


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 40, 56, 128, 128), (1, 96, 96, 256, 304), (2, 24, 24, 128, 256), (1, 7, 9, 128, 128)])
def test_dgrad_after_bn_fused_equals_separate_steps(K, monkeypatch, n, h, w, cin, cout):
    """kernels.conv3x3_dgrad_after_bn: the train-mode BatchNorm+ReLU backward's apply pass inside the Winograd input transform of the
    data-gradient convolution in front of it (mss_wino_input_transform_bnbwd_f32) against the two separate steps -- the same
    arithmetic per element, so bit for bit -- and against torch's float64 autograd of conv -> BatchNorm(train) -> ReLU."""
    torch.manual_seed(n * h + cin)
    wt = torch.nn.Parameter(torch.randn(cin, cout, 3, 3, device="cuda") * 0.05)      # forward layer: cout -> cin channels
    bn = torch.nn.BatchNorm2d(cin).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
    xin = torch.randn(n, cout, h, w, device="cuda")
    x = K.conv3x3(K.Act.from_nchw(xin), wt, want_stats=True)                          # BN input [n,h,w,cin]
    st = K.bn_fold(bn, x, train=True)
    dy = K.Act.from_nchw(torch.randn(n, cin, h, w, device="cuda"))
    monkeypatch.setenv("MSS_BNBWD_FUSED", "0")
    sep = K.conv3x3_dgrad_after_bn(dy, x, st, wt)
    monkeypatch.setenv("MSS_BNBWD_FUSED", "1")
    fus = K.conv3x3_dgrad_after_bn(dy, x, st, wt)
    assert sep.ld == fus.ld and sep.ld % 32 == 0              # fresh Winograd outputs sit on 128-byte pixel rows (304 -> ld 320)
    assert torch.equal(sep.buf[..., :sep.C], fus.buf[..., :fus.C])
    # float64 reference ON THE SAME BatchNorm input (the fp32 values of x: a ReLU mask decided by a differently rounded x would
    # flip at the elements within 1e-5 of zero and dominate the comparison): BN(train) -> ReLU backward, then the convolution's
    bnd = torch.nn.BatchNorm2d(cin).cuda().double().train()
    with torch.no_grad():
        bnd.weight.copy_(bn.weight.double()); bnd.bias.copy_(bn.bias.double())
    xq = x.nchw().double().requires_grad_(True)
    torch.relu(bnd(xq)).backward(dy.nchw().double())
    xd = xin.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd, wt.detach().double(), padding=1).backward(xq.grad)
    rel = (fus.nchw().double() - xd.grad).norm().item() / xd.grad.norm().item()
    assert rel <= 2e-4, rel


@pytest.mark.parametrize("n,ih,iw,scale_up,ca,cs,k", [(2, 12, 20, 4, 48, 256, 256), (1, 32, 64, 4, 48, 80, 64), (1, 15, 19, 3, 16, 112, 128)])
def test_conv_on_upsampled_concat_fused_equals_separate_steps(K, monkeypatch, n, ih, iw, scale_up, ca, cs, k):
    """kernels.conv3x3_on_upsampled_concat: the decoder's first 3x3 layer on cat(a, Upsample(small)) with the align_corners=True
    bilinear upsample interpolated inside the Winograd input transform, against upsample + concat + conv3x3 of this repository
    (same arithmetic: equal to rounding of the interpolation's fused multiply-adds) and against torch in float64."""
    torch.manual_seed(n + ih + cs)
    h, w = (ih - 1) * scale_up + 1 + 3, (iw - 1) * scale_up + 1 + 1          # not an exact multiple: generic align_corners geometry
    a = K.Act.from_nchw(torch.randn(n, ca, h, w, device="cuda"))
    small = K.Act.from_nchw(torch.randn(n, cs, ih, iw, device="cuda"))
    wt = torch.nn.Parameter(torch.randn(k, ca + cs, 3, 3, device="cuda") * 0.05)
    fus = K.conv3x3_on_upsampled_concat(a, small, wt, want_stats=True)
    assert fus is not None
    cat = K.Act.empty(n, h, w, ca + cs, "cuda")
    cat.slice(0, ca).buf[..., cat.c0:cat.c0 + ca] = a.buf[..., :ca]
    K.upsample_ac(small, h, w, out=cat.slice(ca, cs))
    sep = K.conv3x3(cat, wt, want_stats=True)
    d = (fus.nchw() - sep.nchw()).abs().max().item()
    assert d <= 2e-6 * sep.nchw().abs().max().item(), d
    up = torch.nn.functional.interpolate(small.nchw().double(), size=(h, w), mode="bilinear", align_corners=True)
    want = torch.nn.functional.conv2d(torch.cat((a.nchw().double(), up), 1), wt.detach().double(), padding=1)
    rel = (fus.nchw().double() - want).norm().item() / want.norm().item()
    assert rel <= 5e-5, rel
    monkeypatch.setenv("MSS_UPCAT_FUSED", "0")
    assert K.conv3x3_on_upsampled_concat(a, small, wt) is None


@pytest.mark.parametrize("n,c,h,w,cp", [(2, 256, 37, 53, 256), (1, 70, 9, 130, 80), (3, 16, 1, 1, 16), (1, 3, 20, 30, 16), (1, 2048, 22, 22, 2048)])
def test_nchw_to_nhwc_conversion(K, n, c, h, w, cp):
    """mss_nchw_to_nhwc_pad_f32: the tiled LDS transpose (feature maps, >= 16 channels) and the per-pixel kernel (images) against a
    permute; ragged pixel / channel counts, zero channel padding."""
    torch.manual_seed(c + h)
    t = torch.randn(n, c, h, w, device="cuda")
    a = K.nchw_to_act(t, Cp=cp)
    assert (a.N, a.H, a.W, a.C) == (n, h, w, cp)
    assert torch.equal(a.buf[..., :c], t.permute(0, 2, 3, 1)) and bool((a.buf[..., c:] == 0).all())


def test_winograd_pair_equals_two_separate_layers(K, monkeypatch):
    """kernels.conv3x3_pair: two dilated 3x3 layers on the same input with their 2 x 64 Winograd-domain products in ONE gemm_nt
    launch (the eval forward's ASPP branches) -- every output element is the same sum in the same order as in the separate
    calls: bit-identical, BatchNorm partial sums included; and the rule takes the pair at the one-image ASPP shape only."""
    torch.manual_seed(5)
    x = K.Act(torch.randn(1, 24, 24, 64, device="cuda"))
    w1 = torch.nn.Parameter(torch.randn(128, 64, 3, 3, device="cuda") * 0.05)
    w2 = torch.nn.Parameter(torch.randn(128, 64, 3, 3, device="cuda") * 0.05)
    a1, a2 = K.Act.empty(1, 24, 24, 128, "cuda"), K.Act.empty(1, 24, 24, 128, "cuda")
    K.conv2d_winograd(x, K.pack_weight_wino(w1.detach(), tile=6), dil=1, out=a1, want_stats=True)
    K.conv2d_winograd(x, K.pack_weight_wino(w2.detach(), tile=6), dil=2, out=a2, want_stats=True)
    b1, b2 = K.Act.empty(1, 24, 24, 128, "cuda"), K.Act.empty(1, 24, 24, 128, "cuda")
    K.conv3x3_pair(x, w1, w2, 1, 2, b1, b2, 6, want_stats=True)
    assert torch.equal(a1.buf, b1.buf) and torch.equal(a2.buf, b2.buf)
    assert torch.equal(a1.stats, b1.stats) and torch.equal(a2.stats, b2.stats)
    # the halves of the pair buffer serve the single-layer path too (no second packed copy), and follow a weight update
    assert K.packed_wino(w1, False, 6).t.data_ptr() == K.packed_wino_pair(w1, w2, 6).t.data_ptr()
    with torch.no_grad():
        w2.mul_(2.0)
    K.conv3x3_pair(x, w1, w2, 1, 2, b1, b2, 6)
    assert torch.equal(a1.buf, b1.buf) and torch.allclose(2 * a2.buf, b2.buf, rtol=1e-6, atol=1e-6)
    wa = torch.empty(256, 4096, 3, 3, device="meta")
    class Shape:            # geometry only: the rule never touches data
        def __init__(self, n): self.N, self.H, self.W = n, 128, 256
    assert K.conv3x3_pair_tile(Shape(1), wa, wa, 12, 24) == 6
    assert K.conv3x3_pair_tile(Shape(2), wa, wa, 12, 24) == 0      # 2304 tiles per position: three full rounds already
    assert K.conv3x3_pair_tile(Shape(1), wa, wa, 12, 36) == 0      # different tile sizes / tile counts
    monkeypatch.setenv("MSS_WINO_PAIR", "0")
    assert K.conv3x3_pair_tile(Shape(1), wa, wa, 12, 24) == 0


@pytest.mark.parametrize("T,C,Ko", [(162624, 256, 256), (50001, 256, 192), (40000, 1024, 256)])
def test_linear_wgrad_many_rows_vs_float64(K, T, C, Ko, gemm_route):
    """The decoder's Linear weight gradients: ONE position, a handful of output tiles and very many rows (16 x 10 164 tokens), i.e.
    up to 192 row splits on the TN kernel (the cap was 64: a third of the slots) summed by the ordered reduction; against a
    float64 product on a sample of rows of dW, and twice with identical bits."""
    torch.manual_seed(T + C)
    x = torch.randn(T, C, device="cuda")
    dy = torch.randn(T, Ko, device="cuda")
    outs = [K.conv2d_wgrad(K.Act(x.view(1, 1, T, C)), K.Act(dy.view(1, 1, T, Ko)), Ko, C, 1, 1).view(Ko, C) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    rows = torch.arange(0, Ko, 7, device="cuda")
    want = dy[:, rows].double().t() @ x.double()
    err = (outs[0][rows].double() - want).abs().max().item()
    assert err <= 2e-6 * T ** 0.5 * 16, err


@pytest.mark.parametrize("P,T,C,Ko,affine", [(64, 1892, 512, 512, False), (34, 2112, 320, 1024, False), (1, 70000, 512, 256, True)])
def test_gemm_hybrid_last_round_is_bitwise_the_wide_result(K, monkeypatch, P, T, C, Ko, affine):
    """Wide-tile GEMMs whose last round would be mostly idle finish on narrow tiles in a second launch (gemm.hip): every
    output element is the same sum in the same order, so the result equals the all-wide launch bit for bit; a sample of rows
    is checked against float64."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(T + C)
    x = torch.randn(P, T, C, device="cuda")
    kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.zeros(P, kpad, C, device="cuda")
    w[:, :Ko] = torch.randn(P, Ko, C, device="cuda") / C ** 0.5
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.1
    outs = {}
    for tail in ("0", "1"):
        monkeypatch.setenv("MSS_GEMM_TAIL", tail)
        y = torch.full((P, T, Ko), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(x), ptr(w), ptr(y)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        if P > 1:
            a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, kpad * C, T * Ko
        if affine:
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
        call("mss_conv2d_forward_f32", ctypes.byref(a))
        outs[tail] = y
    assert torch.equal(outs["0"], outs["1"])
    rows = torch.randint(0, T, (64,), device="cuda")
    xin = x[:, rows].double()
    if affine:
        xin = torch.relu(xin * sc.double() + sh.double())
    want = torch.einsum("ptc,pkc->ptk", xin, w[:, :Ko].double())
    assert (outs["1"][:, rows].double() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("P,T,C,Ko,affine,extras", [
    (36, 4000, 128, 128, False, False),      # narrow tile, 8 K-steps per tile, ragged last row tile
    (64, 1892, 512, 512, False, False),      # wide tile, partial last round
    (5, 3000, 48, 256, False, False),        # 3 K-steps per tile: the shortest reduction variant 3 takes
    (1, 70000, 512, 256, True, True),        # BatchNorm + ReLU prologue, residual, ReLU epilogue, statistics
    (1, 5000, 304, 304, True, False),        # C = 304, 304 = 256 + 48 output channels (narrow tail launch)
])
def test_gemm_variant3_is_bitwise_variant2(K, monkeypatch, P, T, C, Ko, affine, extras):
    """gemm.hip variant 3 (round 3: 32-bit operand offsets, branch-free advance, loader instructions interleaved with the MFMAs)
    changes the SCHEDULE only: every output element is the same fmaf chain as in variant 2, so outputs and the BatchNorm partial
    sums are equal bit for bit; a sample of rows is checked against float64."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(T + C)
    x = torch.randn(P, T, C, device="cuda")
    kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.zeros(P, kpad, C, device="cuda")
    w[:, :Ko] = torch.randn(P, Ko, C, device="cuda") / C ** 0.5
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.1
    res = torch.randn(P, T, Ko, device="cuda")
    outs, stats = {}, {}
    for var in ("2", "3"):
        monkeypatch.setenv("MSS_GEMM_VARIANT", var)
        y = torch.full((P, T, Ko), float("nan"), device="cuda")
        st = torch.full((-(-T // 64), 2, Ko), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(x), ptr(w), ptr(y)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        if P > 1:
            a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, kpad * C, T * Ko
        if affine:
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
        if extras:
            a.res, a.ldres, a.out_relu, a.stats = ptr(res), Ko, 1, ptr(st)
        if Ko % 128 and Ko > 128:               # as kernels.conv2d does: 256 + 48 -> a narrow tail launch on the rest
            a.K = Ko - Ko % 128
            call("mss_conv2d_forward_f32", ctypes.byref(a))
            a.K, a.Kpad = Ko % 128, kpad - (Ko - Ko % 128)
            a.w = ctypes.c_void_p(w.data_ptr() + 4 * (Ko - Ko % 128) * C)
            a.y = ctypes.c_void_p(y.data_ptr() + 4 * (Ko - Ko % 128))
            if P > 1:
                pytest.skip("tail split of a batched call needs per-batch weight strides the test does not model")
        call("mss_conv2d_forward_f32", ctypes.byref(a))
        outs[var], stats[var] = y, st
    assert torch.equal(outs["2"], outs["3"])
    assert not torch.isnan(outs["3"]).any()
    if extras:
        assert torch.equal(stats["2"], stats["3"])
    rows = torch.randint(0, T, (64,), device="cuda")
    xin = x[:, rows].double()
    if affine:
        xin = torch.relu(xin * sc.double() + sh.double())
    want = torch.einsum("ptc,pkc->ptk", xin, w[:, :Ko].double())
    if extras:
        want = torch.relu(want + res[:, rows].double())
    assert (outs["3"][:, rows].double() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("n,h,w,c,d,tiles", [
    (2, 128, 256, 128, 12, (6, 6, 4)),      # the ASPP map of 2 x 1024 x 2048: 11 x 22 base sub-grids, CB = 32
    (1, 64, 128, 192, 12, (6, 6, 4)),       # C1 (512 x 1024): 6 x 11 sub-grids, ragged last channel chunk
    (2, 88, 88, 64, 12, (4, 4, 4)),         # the 700 x 700 crop
    (1, 45, 75, 68, 6, (6, 4, 6)), (1, 45, 75, 68, 6, (4, 6, 4)),   # other tile mixes, sizes nothing divides
    (3, 12, 16, 32, 12, (4, 4, 4)),         # 96 x 128 inputs of the fixtures: sub-grids of 1 x 2 pixels
    (1, 37, 41, 36, 6, (6, 6, 6)), (1, 9, 40, 16, 1, (4, 6, 4)),
])
def test_aspp_fused_input_transform_is_bitwise_the_three_separate_ones(n, h, w, c, d, tiles):
    """mss_wino_input_transform_aspp3_f32: X' of one map for dilations d, 2d, 3d from ONE read of it (VERDICT r03 missing #2)
    against three calls of mss_wino_input_transform_f32 -- every element of every X' identical, nothing written outside."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import call, ptr
    torch.manual_seed(h * w + c)
    ld = c + 12                                                     # a channel slice of a wider buffer
    buf = torch.randn(n, h, w, ld, device="cuda")
    x = buf[..., 4:]                                                # 16-byte aligned offset
    xptr = ctypes.c_void_p(buf.data_ptr() + 16)
    want, got = [], []
    for m, ts in enumerate(tiles):
        T = _lib.value("mss_wino_num_tiles", n, h, w, (m + 1) * d, ts)
        P = (ts + 2) ** 2
        a = torch.full((P * T * c + 64,), float("nan"), device="cuda")
        b = torch.full((P * T * c + 64,), float("nan"), device="cuda")
        call("mss_wino_input_transform_f32", xptr, ld, n, h, w, c, (m + 1) * d, ts, None, None, 0, ptr(a))
        want.append(a)
        got.append(b)
    rc = _lib.status("mss_wino_input_transform_aspp3_f32", xptr, ld, n, h, w, c, d, (ctypes.c_int * 3)(*tiles), ptr(got[0]), ptr(got[1]),
                     ptr(got[2]))
    assert rc == 0, rc
    for m in range(3):
        assert torch.isnan(got[m][-64:]).all() and not torch.isnan(got[m][:-64]).any()
        assert torch.equal(got[m][:-64], want[m][:-64]), (m, (got[m][:-64] - want[m][:-64]).abs().max().item())
    del x


def test_aspp_fused_input_transform_refuses_what_it_does_not_take():
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import ptr
    x = torch.zeros(1, 16, 16, 16, device="cuda")
    o = torch.zeros(1 << 20, device="cuda")
    rc = _lib.status("mss_wino_input_transform_aspp3_f32", ptr(x), 16, 1, 16, 16, 16, 4, (ctypes.c_int * 3)(2, 4, 4), ptr(o), ptr(o), ptr(o))
    assert rc == _lib.MSS_ERR_UNSUPPORTED                          # a 2 x 2 tile
    big = torch.zeros(1, 512, 1024, 4, device="cuda")              # 43 x 86 base sub-grids: more than LDS holds
    rc = _lib.status("mss_wino_input_transform_aspp3_f32", ptr(big), 4, 1, 512, 1024, 4, 12, (ctypes.c_int * 3)(6, 6, 6), ptr(o), ptr(o), ptr(o))
    assert rc == _lib.MSS_ERR_UNSUPPORTED
    rc = _lib.status("mss_wino_input_transform_aspp3_f32", ptr(x), 16, 1, 16, 16, 16, 4, None, ptr(o), ptr(o), ptr(o))
    assert rc == _lib.MSS_ERR_BAD_ARG


@pytest.mark.parametrize("b,q,c,hm,wm,image,crop", [(2, 100, 19, 24, 32, (96, 128), (90, 120)), (1, 100, 19, 13, 17, (50, 70), (50, 70)),
                                                   (1, 20, 7, 9, 9, (33, 65), (31, 64)), (3, 8, 19, 6, 40, (7, 161), (7, 161)),
                                                   (1, 100, 19, 64, 128, (256, 512), (256, 512))])
def test_m2f_fused_score_mfma_equals_valu_kernel(K, monkeypatch, b, q, c, hm, wm, image, crop):
    """The class mix of the fused Mask2Former score on the matrix cores (round 4: A = prob^T [32 classes x 2 queries], B = the sigmoids
    of 32 pixels, v_mfma_f32_32x32x2_f32) against the all-VALU kernel it replaces (MSS_M2F_MFMA=0), over ragged tiles, 4 / 8 / 16-row
    tiles, query counts that leave a 2-step tail, fewer classes, and against the numpy oracle."""
    g = torch.Generator(device="cuda").manual_seed(b * q + hm)
    cls = torch.randn(b, q, c + 1, device="cuda", generator=g) * 2
    lg = torch.randn(b, hm, wm, q, device="cuda", generator=g) * 3
    got = K.m2f_score_fused(cls, lg, image, crop)
    monkeypatch.setenv("MSS_M2F_MFMA", "0")
    want = K.m2f_score_fused(cls, lg, image, crop)
    assert got.shape == want.shape == (b,) + tuple(crop)
    assert (got - want).abs().max().item() < 2e-6, (got - want).abs().max().item()
    up = torch.nn.functional.interpolate(lg.permute(0, 3, 1, 2).double(), size=image, mode="bilinear", align_corners=False)
    ref = 1 - torch.einsum("bqc,bqhw->bchw", torch.softmax(cls.double(), -1)[..., :-1], up.sigmoid())[:, :, :crop[0], :crop[1]].max(1)[0]
    assert (got.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("P,T,C,Ko,extras", [
    (1, 3000, 256, 1024, False),       # 24 row tiles x 4 wide column tiles: three full groups of 8
    (1, 1300, 128, 4096, True),        # 11 row tiles (last group has 3) x 16 column tiles, affine prologue + residual + statistics
    (3, 1000, 64, 640, False),         # batched, narrow 128-wide tiles: 8 x 5 per batch entry
    (1, 130, 512, 512, False),         # 2 row tiles: fewer than one group
])
def test_gemm_grouped_tile_order_is_a_permutation_of_the_same_tiles(K, monkeypatch, P, T, C, Ko, extras):
    """gemm.hip tile_mn (round 4): walking the tile grid in groups of 8 row tiles (row tile fastest inside a group) instead of column
    tile fastest changes WHICH workgroup computes a tile and when -- the L2 sharing between the workgroups of an XCD -- never the
    tile itself: outputs and BatchNorm partial sums are equal bit for bit with MSS_GEMM_GROUP_M=0, and nothing is left unwritten."""
    import ctypes
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import MssConvArgs, call, ptr
    torch.manual_seed(T + Ko)
    x = torch.randn(P, T, C, device="cuda")
    kpad = _lib.value("mss_conv2d_kpad", Ko)
    w = torch.zeros(P, kpad, C, device="cuda")
    w[:, :Ko] = torch.randn(P, Ko, C, device="cuda") / C ** 0.5
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    res = torch.randn(P, T, Ko, device="cuda")
    outs, stats = {}, {}
    for g in ("0", "8", "3"):
        monkeypatch.setenv("MSS_GEMM_GROUP_M", g)
        y = torch.full((P, T, Ko), float("nan"), device="cuda")
        st = torch.full((-(-T // 64), 2, Ko), float("nan"), device="cuda")
        a = MssConvArgs()
        a.x, a.w, a.y = ptr(x), ptr(w), ptr(y)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        if P > 1:
            a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, kpad * C, T * Ko
        if extras:
            a.in_scale, a.in_shift, a.in_relu = ptr(sc), ptr(sh), 1
            a.res, a.ldres, a.out_relu, a.stats = ptr(res), Ko, 1, ptr(st)
        call("mss_conv2d_forward_f32", ctypes.byref(a))
        outs[g], stats[g] = y, st
    assert not torch.isnan(outs["8"]).any()
    assert torch.equal(outs["0"], outs["8"]) and torch.equal(outs["0"], outs["3"])
    if extras:
        assert torch.equal(stats["0"], stats["8"]) and torch.equal(stats["0"], stats["3"])
    xin = x.double()
    if extras:
        xin = torch.relu(xin * sc.double() + sh.double())
    want = torch.einsum("ptc,pkc->ptk", xin[:, ::7], w[:, :Ko].double())
    if extras:
        want = torch.relu(want + res[:, ::7].double())
    assert (outs["8"][:, ::7].double() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("n,h,w,c", [(2, 128, 256, 4096), (3, 24, 40, 512), (1, 8, 8, 256), (2, 6, 10, 256)])
def test_gap_from_the_producers_partial_sums(K, monkeypatch, n, h, w, c):
    """r04: AdaptiveAvgPool2d(1) from the per-64-row column sums a producing GEMM's epilogue left (kernels.gap with x.stats) against
    the pass over the map (MSS_GAP_FROM_STATS=0) and float64; image sizes that are not multiples of 64 pixels keep the pass."""
    torch.manual_seed(n * h + c)
    cin = 64
    xin = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = torch.randn(c, cin, 1, 1, device="cuda") / 8
    y = K.conv2d(xin, K.pack_weight(wt), want_stats=True)
    assert y.stats is not None
    got = K.gap(y)
    monkeypatch.setenv("MSS_GAP_FROM_STATS", "0")
    old = K.gap(y)
    want = y.buf[..., :c].double().mean((1, 2))
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() <= 2e-6 * scale + 1e-7
    assert (old.double() - want).abs().max().item() <= 2e-6 * scale + 1e-7
    if (h * w) % 64:
        assert torch.equal(got, old)                # the fallback ran
