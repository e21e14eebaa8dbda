"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the Mask2Former anomaly score
(train_m2f.py:387-407): 1 - max_c sum_q softmax(cls)[b,q,c<C] * sigmoid(mask)[b,q,h,w].
Pinned against tests/golden/m2f_score.npz (reference arithmetic reproduced with torch ops in
tools/gen_golden.py because train_m2f.py itself needs detectron2 to import)."""
import numpy as np


def anomaly_score(class_logits, mask_logits, size):
    """class_logits [B,Q,C+1], mask_logits [B,Q,Hm,Wm], size (H,W) crop -> [B,H,W]."""
    m = class_logits.max(-1, keepdims=True)
    e = np.exp(class_logits - m)
    probs = (e / e.sum(-1, keepdims=True))[..., :-1]
    sig = 1.0 / (1.0 + np.exp(-mask_logits))
    u = np.einsum("bqc,bqhw->bchw", probs, sig, optimize=True)
    u = u[:, :, :size[0], :size[1]]
    return (1 - u.max(axis=1)).astype(class_logits.dtype)


def mask_logits(mask_embed, mask_features):
    """einsum("bqc,bchw->bqhw") (mask2former_transformer_decoder.py:544-548)."""
    return np.einsum("bqc,bchw->bqhw", mask_embed, mask_features, optimize=True).astype(np.float32)


def upsample_bilinear(x, size):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) (maskformer_model.py:264-277):
    src = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbour index clamped at the border."""
    n, c, h, w = x.shape
    H, W = size

    def axis(out, inp):
        s = np.maximum((np.arange(out, dtype=np.float32) + np.float32(0.5)) * np.float32(inp / out) - np.float32(0.5), 0).astype(np.float32)
        i0 = np.minimum(s.astype(np.int64), inp - 1)
        i1 = np.minimum(i0 + 1, inp - 1)
        return i0, i1, (s - i0).astype(np.float32)
    y0, y1, ly = axis(H, h)
    x0, x1, lx = axis(W, w)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return (top * (1 - ly)[:, None] + bot * ly[:, None]).astype(np.float32)


def anomaly_score_from_features(class_logits, mask_embed, mask_features, image_size, size):
    """The whole 8f-2 chain: mask prediction -> upsample to the image -> score cropped to `size`."""
    return anomaly_score(class_logits, upsample_bilinear(mask_logits(mask_embed, mask_features), image_size), size)
