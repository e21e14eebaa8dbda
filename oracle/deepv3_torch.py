"""TEST INFRASTRUCTURE ONLY -- DeepWV3Plus.forward composed from STOCK torch CPU ops.

The second CPU restatement SURVEY 8(d) asks for ("stock torch CPU ops composing the same graph -- not
the reference files"): F.conv2d / F.batch_norm / F.max_pool2d / F.interpolate(align_corners=True) /
torch.logsumexp over a {state_dict name: array} mapping. Follows lib/network/deepv3/deepv3.py:258-285
(network), :84-92 (ASPP), lib/network/deepv3/wider_resnet.py:169-182 (IdentityResidualBlock) and
:303-350 (module plan), lib/network/deepv3/mynn.py:28-33 (Upsample).

Used by (and only by) tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg: it is the
checker at the BASELINE sizes (C1 1x512x1024 in seconds on the GPU box's host cores; train steps with
autograd for the gradient parity tests) and the timed CPU baseline -- never the product path.
Pinned against tests/golden/deepwv3plus_*.npz (outputs of the reference model itself) by
tests/test_oracle_golden.py, and against the numpy restatement oracle/deepv3.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

STRUCTURE = [3, 3, 6, 3, 1, 1]
CHANNELS = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
ASPP_RATES = (12, 24, 36)


def to_torch(params, dtype=torch.float32):
    """{name: ndarray} -> {name: tensor}; integer buffers (num_batches_tracked) are kept as they are."""
    out = {}
    for k, v in params.items():
        t = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v
        out[k] = t.to(dtype).clone() if t.is_floating_point() else t.clone()
    return out


def _bnrelu(p, prefix, x, train):
    """mynn.Norm2d = nn.BatchNorm2d (momentum 0.1, eps 1e-5) + ReLU; train=True updates p[...running_*] in place."""
    y = F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"], p[prefix + ".bias"],
                     training=train, momentum=0.1, eps=1e-5)
    return F.relu(y)


def _block(p, prefix, x, stride, dil, bottleneck, train, drop_mask):
    bn1 = _bnrelu(p, prefix + ".bn1.0", x, train)
    shortcut = F.conv2d(bn1, p[prefix + ".proj_conv.weight"], stride=stride) if (prefix + ".proj_conv.weight") in p else x
    if not bottleneck:
        out = F.conv2d(bn1, p[prefix + ".convs.conv1.weight"], stride=stride, dilation=dil, padding=dil)
        out = _bnrelu(p, prefix + ".convs.bn2.0", out, train)
        if drop_mask is not None:
            out = out * drop_mask[:, :, None, None]
        out = F.conv2d(out, p[prefix + ".convs.conv2.weight"], dilation=dil, padding=dil)
    else:
        out = F.conv2d(bn1, p[prefix + ".convs.conv1.weight"], stride=stride)
        out = _bnrelu(p, prefix + ".convs.bn2.0", out, train)
        out = F.conv2d(out, p[prefix + ".convs.conv2.weight"], dilation=dil, padding=dil)
        out = _bnrelu(p, prefix + ".convs.bn3.0", out, train)
        if drop_mask is not None:
            out = out * drop_mask[:, :, None, None]
        out = F.conv2d(out, p[prefix + ".convs.conv3.weight"])
    return out + shortcut


def trunk(p, inp, train=False, drop_masks=None):
    x = F.conv2d(inp, p["mod1.conv1.weight"], padding=1)
    m2 = None
    for mod_id, num in enumerate(STRUCTURE):
        name = f"mod{mod_id + 2}"
        if mod_id < 2:
            x = F.max_pool2d(x, 3, stride=2, padding=1)
        for b in range(num):
            dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
            stride = 2 if (b == 0 and mod_id == 2) else 1
            dm = drop_masks.get(name) if drop_masks else None
            x = _block(p, f"{name}.block{b + 1}", x, stride, dil, len(CHANNELS[mod_id]) == 3, train, dm)
        if mod_id == 0:
            m2 = x
    return x, m2


def head(p, x, m2, out_size, train=False, taps=None):
    n, c, h, w = x.shape
    img = F.adaptive_avg_pool2d(x, 1)
    img = _bnrelu(p, "aspp.img_conv.1", F.conv2d(img, p["aspp.img_conv.0.weight"]), train)
    out = [F.interpolate(img, size=(h, w), mode="bilinear", align_corners=True)]
    for i, rate in enumerate((None,) + ASPP_RATES):
        wt = p[f"aspp.features.{i}.0.weight"]
        y = F.conv2d(x, wt) if rate is None else F.conv2d(x, wt, dilation=rate, padding=rate)
        out.append(_bnrelu(p, f"aspp.features.{i}.1", y, train))
    dec = torch.cat(out, 1)
    dec0_up = F.conv2d(dec, p["bot_aspp.weight"])
    dec0_fine = F.conv2d(m2, p["bot_fine.weight"])
    dec0_up = F.interpolate(dec0_up, size=m2.shape[2:], mode="bilinear", align_corners=True)
    dec0 = torch.cat([dec0_fine, dec0_up], 1)
    f = _bnrelu(p, "final.1", F.conv2d(dec0, p["final.0.weight"], padding=1), train)
    feature = _bnrelu(p, "final.4", F.conv2d(f, p["final.3.weight"], padding=1), train)
    dec1 = F.conv2d(feature, p["final.6.weight"])
    dec2 = F.conv2d(feature, p["ood_head.weight"])
    logit = F.interpolate(dec1, size=out_size, mode="bilinear", align_corners=True)
    energy = -torch.logsumexp(dec2, dim=1)
    score = F.interpolate(energy[:, None], size=out_size, mode="bilinear", align_corners=True)[:, 0]
    if taps is not None:
        taps.update(aspp=dec, feature=feature, dec1=dec1, dec2=dec2, x=x, m2=m2)
    return score, logit


def forward_t(p, inp, train=False, drop_masks=None, taps=None):
    """Tensor in, tensors out; p as returned by to_torch (running statistics updated in place when train)."""
    with torch.no_grad():
        x, m2 = trunk(p, inp, train, drop_masks)        # frozen in both training stages (exps/DeepLab.yaml:10-11)
    return head(p, x, m2, tuple(inp.shape[2:]), train, taps)


def forward(params, inp, train=False, stats_out=None, drop_masks=None):
    """Same calling convention as oracle.deepv3.forward (numpy in, numpy out)."""
    p = to_torch(params)
    dm = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in drop_masks.items()} if drop_masks else None
    with torch.no_grad():
        score, logit = forward_t(p, torch.from_numpy(np.asarray(inp, dtype=np.float32)), train, dm)
    if train and stats_out is not None:
        for k, v in p.items():
            if k.endswith(("running_mean", "running_var")):
                stats_out[k] = v.numpy()
    return score.numpy(), logit.numpy()
