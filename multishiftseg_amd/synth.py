"""Deterministic synthetic parameters and inputs.

There are no checkpoints or datasets on the GPU box, so weights are regenerated from a
counter-based generator keyed by the state-dict name: the golden fixtures (made in the build
container by importing the reference) and the GPU tests see bit-identical parameters without a
548 MB file ever being shipped. numpy's PCG64 stream is platform independent.
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def deepwv3plus_param_shapes(num_classes=19):
    """State-dict of DeepWV3Plus in the reference's order and naming
    (lib/network/deepv3/deepv3.py:217-249, wider_resnet.py:303-350): name -> shape."""
    shapes = {}

    def conv(name, k, c, r):
        shapes[name + ".weight"] = (k, c, r, r)

    def bn(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)
        shapes[name + ".running_mean"] = (c,)
        shapes[name + ".running_var"] = (c,)
        shapes[name + ".num_batches_tracked"] = ()

    conv("mod1.conv1", 64, 3, 3)
    structure = [3, 3, 6, 3, 1, 1]
    channels = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
    in_c = 64
    for mod_id, num in enumerate(structure):
        for block_id in range(num):
            p = f"mod{mod_id + 2}.block{block_id + 1}"
            ch = channels[mod_id]
            stride = 2 if (block_id == 0 and mod_id == 2) else 1
            bn(p + ".bn1.0", in_c)
            if len(ch) == 2:
                conv(p + ".convs.conv1", ch[0], in_c, 3)
                bn(p + ".convs.bn2.0", ch[0])
                conv(p + ".convs.conv2", ch[1], ch[0], 3)
            else:
                conv(p + ".convs.conv1", ch[0], in_c, 1)
                bn(p + ".convs.bn2.0", ch[0])
                conv(p + ".convs.conv2", ch[1], ch[0], 3)
                bn(p + ".convs.bn3.0", ch[1])
                conv(p + ".convs.conv3", ch[2], ch[1], 1)
            if stride != 1 or in_c != ch[-1]:
                conv(p + ".proj_conv", ch[-1], in_c, 1)
            in_c = ch[-1]
    conv("aspp.features.0.0", 256, 4096, 1)
    bn("aspp.features.0.1", 256)
    for i in (1, 2, 3):
        conv(f"aspp.features.{i}.0", 256, 4096, 3)
        bn(f"aspp.features.{i}.1", 256)
    conv("aspp.img_conv.0", 256, 4096, 1)
    bn("aspp.img_conv.1", 256)
    conv("bot_fine", 48, 128, 1)
    conv("bot_aspp", 256, 1280, 1)
    conv("final.0", 256, 304, 3)
    bn("final.1", 256)
    conv("final.3", 256, 256, 3)
    bn("final.4", 256)
    conv("final.6", num_classes, 256, 1)
    conv("ood_head", num_classes, 256, 1)
    return shapes


def gen_tensor(seed, name, shape, gain=1.0):
    """One parameter / buffer. Conv weights: N(0, gain/fan_in) so activations stay O(1) through
    the 17 un-normalised residual adds; BN affine/statistics are non-trivial on purpose so that
    eval-mode BatchNorm is exercised."""
    r = _rng(seed, name)
    if name.endswith("num_batches_tracked"):
        return np.zeros((), dtype=np.int64)
    if name.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if name.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1 and name.endswith(".weight"):  # BN gamma
        return r.uniform(0.8, 1.2, shape).astype(np.float32)
    if len(shape) == 1 and name.endswith(".bias"):    # BN beta
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = np.sqrt(gain / fan_in)
    return (std * r.standard_normal(shape, dtype=np.float32)).astype(np.float32)


def deepwv3plus_params(seed=0, num_classes=19, names=None):
    """dict name -> np.ndarray for the whole network (or the subset `names`)."""
    out = {}
    for name, shape in deepwv3plus_param_shapes(num_classes).items():
        if names is not None and name not in names:
            continue
        out[name] = gen_tensor(seed, name, shape)
    return out


def synth_image(seed, n, h, w):
    """Post-Normalize image statistics: N(0,1) fp32, NCHW."""
    return _rng(seed, f"image{n}x{h}x{w}").standard_normal((n, 3, h, w), dtype=np.float32)


def synth_targets(seed, pairs, h, w, num_classes=19, ood_in_aug_prob=0.5):
    """Cityscapes-like label maps for `pairs` (original, augmented) pairs -> int64 [2*pairs,h,w]
    laid out [orig...; aug...] as train_deeplab.py:194-195 concatenates them.
    Blocky 19-class layout, ~3 % void (255), one OOD rectangle (254) of 1-10 % area in every
    original image (COCO paste, cityscapes.py:168-169) and in about half of the augmented ones."""
    r = _rng(seed, f"target{pairs}x{h}x{w}")
    bh, bw = max(1, h // 8), max(1, w // 8)
    t = np.empty((2 * pairs, h, w), dtype=np.int64)
    for p in range(pairs):
        coarse = r.integers(0, num_classes, size=((h + bh - 1) // bh, (w + bw - 1) // bw))
        base = np.kron(coarse, np.ones((bh, bw), dtype=np.int64))[:h, :w]
        void = r.random((h, w)) < 0.03
        base = np.where(void, 255, base)
        for k, has_ood in ((p, True), (pairs + p, r.random() < ood_in_aug_prob)):
            m = base.copy()
            if has_ood:
                area = r.uniform(0.01, 0.10) * h * w
                rh = int(np.clip(np.sqrt(area * r.uniform(0.5, 2.0)), 1, h))
                rw = int(np.clip(area / rh, 1, w))
                y0 = int(r.integers(0, h - rh + 1))
                x0 = int(r.integers(0, w - rw + 1))
                m[y0:y0 + rh, x0:x0 + rw] = 254
            t[k] = m
    return t
