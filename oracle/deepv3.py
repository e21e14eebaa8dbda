"""TEST INFRASTRUCTURE ONLY -- numpy restatement of DeepWV3Plus.forward.

Follows lib/network/deepv3/deepv3.py:258-285 (network), :47-92 (ASPP) and
lib/network/deepv3/wider_resnet.py:169-182 (IdentityResidualBlock), :303-350 (module plan).
Parameters come in as a {state_dict name: ndarray} mapping with the reference's own names.
Pinned against tests/golden/deepwv3plus_*.npz (outputs of the reference model itself).
"""
import numpy as np

from . import nnops as ops

STRUCTURE = [3, 3, 6, 3, 1, 1]
CHANNELS = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]


def _bnrelu(p, prefix, x, train, stats_out):
    y, rm, rv = ops.batchnorm(x, p[prefix + ".weight"], p[prefix + ".bias"], p[prefix + ".running_mean"],
                              p[prefix + ".running_var"], train)
    if train and stats_out is not None:
        stats_out[prefix + ".running_mean"] = rm
        stats_out[prefix + ".running_var"] = rv
    return ops.relu(y)


def _block(p, prefix, x, stride, dil, bottleneck, train, stats_out, drop_mask):
    """wider_resnet.py:169-182."""
    bn1 = _bnrelu(p, prefix + ".bn1.0", x, train, stats_out)
    if (prefix + ".proj_conv.weight") in p:
        shortcut = ops.conv2d(bn1, p[prefix + ".proj_conv.weight"], stride=stride)
    else:
        shortcut = x
    if not bottleneck:
        out = ops.conv2d(bn1, p[prefix + ".convs.conv1.weight"], stride=stride, dilation=dil, padding=dil)
        out = _bnrelu(p, prefix + ".convs.bn2.0", out, train, stats_out)
        if drop_mask is not None:
            out = out * drop_mask[:, :, None, None]
        out = ops.conv2d(out, p[prefix + ".convs.conv2.weight"], dilation=dil, padding=dil)
    else:
        out = ops.conv2d(bn1, p[prefix + ".convs.conv1.weight"], stride=stride)
        out = _bnrelu(p, prefix + ".convs.bn2.0", out, train, stats_out)
        out = ops.conv2d(out, p[prefix + ".convs.conv2.weight"], dilation=dil, padding=dil)
        out = _bnrelu(p, prefix + ".convs.bn3.0", out, train, stats_out)
        if drop_mask is not None:  # Dropout2d sits before conv3 (wider_resnet.py:161-162)
            out = out * drop_mask[:, :, None, None]
        out = ops.conv2d(out, p[prefix + ".convs.conv3.weight"])
    return out + shortcut


def trunk(p, inp, train=False, stats_out=None, drop_masks=None, taps=None):
    """mod1 .. mod7 (deepv3.py:261-268). drop_masks: {'mod6': [N,1024], 'mod7': [N,2048]} already
    scaled by 1/(1-p) (Dropout2d, wider_resnet.py:334-337); None in eval mode."""
    x = ops.conv2d(inp, p["mod1.conv1.weight"], padding=1)
    m2 = None
    for mod_id, num in enumerate(STRUCTURE):
        name = f"mod{mod_id + 2}"
        if mod_id < 2:
            x = ops.maxpool3s2(x)
        for b in range(num):
            dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
            stride = 2 if (b == 0 and mod_id == 2) else 1
            dm = drop_masks.get(name) if drop_masks else None
            x = _block(p, f"{name}.block{b + 1}", x, stride, dil, len(CHANNELS[mod_id]) == 3, train, stats_out, dm)
        if mod_id == 0:
            m2 = x
        if taps is not None:
            taps[name] = x
    return x, m2


def aspp(p, x, train=False, stats_out=None):
    """deepv3.py:84-92; concat order [img, 1x1, d12, d24, d36]."""
    n, c, h, w = x.shape
    img = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64).astype(x.dtype)
    img = ops.conv2d(img, p["aspp.img_conv.0.weight"])
    img = _bnrelu(p, "aspp.img_conv.1", img, train, stats_out)
    out = [np.broadcast_to(img, (n, img.shape[1], h, w))]
    for i, rate in enumerate((None, 12, 24, 36)):
        wt = p[f"aspp.features.{i}.0.weight"]
        y = ops.conv2d(x, wt) if rate is None else ops.conv2d(x, wt, dilation=rate, padding=rate)
        out.append(_bnrelu(p, f"aspp.features.{i}.1", y, train, stats_out))
    return np.concatenate(out, axis=1)


def head(p, x, m2, out_size, train=False, stats_out=None, taps=None):
    """deepv3.py:270-285."""
    dec = aspp(p, x, train, stats_out)
    dec0_up = ops.conv2d(dec, p["bot_aspp.weight"])
    dec0_fine = ops.conv2d(m2, p["bot_fine.weight"])
    dec0_up = ops.upsample_bilinear_ac(dec0_up, m2.shape[2:])
    dec0 = np.concatenate([dec0_fine, dec0_up], axis=1)
    f = ops.conv2d(dec0, p["final.0.weight"], padding=1)
    f = _bnrelu(p, "final.1", f, train, stats_out)
    f = ops.conv2d(f, p["final.3.weight"], padding=1)
    feature = _bnrelu(p, "final.4", f, train, stats_out)
    dec1 = ops.conv2d(feature, p["final.6.weight"])
    dec2 = ops.conv2d(feature, p["ood_head.weight"])
    score, logit = ops.ood_score_tail(dec2, dec1, out_size)
    if taps is not None:
        taps.update(aspp=dec, feature=feature, dec1=dec1, dec2=dec2)
    return score, logit


def forward(p, inp, train=False, stats_out=None, drop_masks=None, taps=None):
    """DeepWV3Plus.forward -> (anomaly_score [B,H,W], logit [B,19,H,W])."""
    x, m2 = trunk(p, inp, train, stats_out, drop_masks, taps)
    if taps is not None:
        taps["m2"] = m2
        taps["x"] = x
    return head(p, x, m2, inp.shape[2:], train, stats_out, taps)
