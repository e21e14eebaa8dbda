"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the per-sample data path of the DeepLab trainer.

Follows lib/dataset/cityscapes.py:153-171 (__getitem__: mixup, transform, anomaly mix), lib/utils/img_utils.py:110-123
(ToTensor), :246-259 (RandCrop), :147-152 (Normalize = torchvision sub_/div_ in float32), :355-361 (normalize() of the
pasted object: float32 / 255.0, then float64 through the Python-float mean / std), :398-435 (mix_func) and
train_deeplab.py:190-195 (batch = cat([img, gen_img]), cat([target, gen_target])).
Pinning: tests/golden/datapath.npz holds outputs of the reference's own mix_func / normalize / extract_bboxes (imported
from img_utils.py with cv2 / torchvision stubbed: neither is used by those three functions) and of the torch expressions
of ToTensor / crop / Normalize; the cv2.resize of the COCO object (random_scale, :345-352) cannot run here (no cv2):
PARITY UNPINNED for that step, objects enter already rescaled.
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def mixup(image_u8, gen_u8, p):
    """cityscapes.py:161-164."""
    return (p * image_u8 + (1 - p) * gen_u8).astype(np.uint8)


def to_tensor_crop_normalize(img_u8, top, left, h, w, mean=MEAN, std=STD, flip=False):
    """ToTensor -> F.crop -> (hflip) -> Normalize on one HWC uint8 image -> CHW float32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    x = x[:, top:top + h, left:left + w]
    if flip:
        x = x[:, :, ::-1]
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def paste(image_chw, target_hw, obj_img, obj_mask, geom, mean=MEAN, std=STD):
    """mix_func (img_utils.py:398-435) with the bounding box and the paste corner already chosen:
    geom = (y1, x1, bh, bw, h0, w0)."""
    y1, x1, bh, bw, h0, w0 = geom
    image = image_chw.transpose(1, 2, 0).copy()
    target = target_hw.copy()
    if bh <= 0:
        return image.transpose(2, 0, 1), target
    cm = obj_mask[y1:y1 + bh, x1:x1 + bw]
    ci = obj_img[y1:y1 + bh, x1:x1 + bw, :]
    ci = ci.astype(np.float32) / 255.0                 # normalize(): float32 / python float stays float32 ...
    ci = (ci - mean) / std                             # ... minus / over tuples of python floats: float64
    sel = (cm != 0) & (cm != 255)
    image[h0:h0 + bh, w0:w0 + bw, :][sel] = ci[sel]    # assignment rounds to the image's float32
    target[h0:h0 + bh, w0:w0 + bw][sel] = cm[sel]
    return image.transpose(2, 0, 1), target


def pair_batch(img, gen, tgt, gen_tgt, crop_size, params, flip=None):
    """Batch of samples -> (images [2B,3,h,w] float32, targets [2B,h,w] int64), originals first."""
    B = img.shape[0]
    h, w = crop_size
    oi, ot, ai, at = [], [], [], []
    for b in range(B):
        p = params[b]
        g = mixup(img[b], gen[b], p["p"]) if p["p"] is not None else gen[b]
        f = bool(flip[b]) if flip is not None else False
        x = to_tensor_crop_normalize(img[b], p["top"], p["left"], h, w, flip=f)
        xg = to_tensor_crop_normalize(g, p["top"], p["left"], h, w, flip=f)
        t = tgt[b, p["top"]:p["top"] + h, p["left"]:p["left"] + w].astype(np.int64)
        tg = gen_tgt[b, p["top"]:p["top"] + h, p["left"]:p["left"] + w].astype(np.int64)
        if f:
            t, tg = t[:, ::-1], tg[:, ::-1]
        if "geom" in p:
            x, t = paste(x, t, p["obj_img"], p["obj_mask"], p["geom"])
        oi.append(x); ot.append(t); ai.append(xg); at.append(tg)
    return np.stack(oi + ai).astype(np.float32), np.stack(ot + at).astype(np.int64)
