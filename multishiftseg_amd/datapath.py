"""On-device data path of the DeepLab trainer.

The reference builds every training sample on CPU workers -- PIL decode of four files, the mixup of the generated image,
ToTensor, one random 700x700 crop shared by image / target / generated image / generated target, Normalize, the COCO-object
paste of the anomaly mix (lib/dataset/cityscapes.py:153-171, lib/utils/img_utils.py:110-153,246-259,367-435) -- and the
trainer then concatenates originals and augmented images (train_deeplab.py:190-195). With four workers that loader caps
at a few images per second; eight MI355X need two orders of magnitude more.

Here the decoded uint8 maps of a batch live in HBM and ONE HIP kernel (csrc/data.hip) produces the normalised
``[orig...; aug...]`` float32 NCHW batch and the int64 targets. What stays on the host are the random DECISIONS, drawn from
Python's ``random`` in the order the reference's code draws them for each sample, so that a seeded run selects the same
mixing weights, crop corners, objects, scales and paste corners:

    mixup weight      random.random()                       cityscapes.py:162
    crop corner       random.randint x 2                    img_utils.py:256-257
    COCO object       random.randint                        img_utils.py:369
    object scale      random.choice                         img_utils.py:346
    paste corner      random.randint x 2                    img_utils.py:412-415

File decoding and the cv2 rescale of the COCO object are out of scope (no image codecs / cv2 in the image): objects are
handed over already rescaled. There is no CPU path: a CPU tensor raises.
"""
import random as _random

import numpy as np
import torch

from ._lib import call, ptr

MEAN = (0.485, 0.456, 0.406)          # lib/dataset/cityscapes.py:63-64, lib/configs/config.py
STD = (0.229, 0.224, 0.225)
OOD_SCALES = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]       # cityscapes.py:89


def mask_bbox(mask):
    """extract_bboxes of img_utils.py:378-394 for one object mask (numpy [oh, ow] uint8): (y1, x1, y2, x2), 0s if empty."""
    m = (mask != 0) & (mask != 255)
    cols = np.where(np.any(m, axis=0))[0]
    rows = np.where(np.any(m, axis=1))[0]
    if cols.shape[0] == 0:
        return 0, 0, 0, 0
    return int(rows[0]), int(cols[0]), int(rows[-1]) + 1, int(cols[-1]) + 1


def draw_sample_params(H, W, crop_size, mixup, objects=None, scaled_object=None, rng=_random, scales=OOD_SCALES):
    """Random decisions of ONE sample in the reference's order. objects: list of candidates for the anomaly mix;
    scaled_object(obj, scale) -> (image float32 [oh, ow, 3], mask uint8 [oh, ow]) performs the rescale (the reference uses
    cv2.resize, img_utils.py:349-350). Returns a dict with p, top, left and, with objects, obj_img / obj_mask / geom."""
    out = {"p": min(rng.random(), 0.3) if mixup else None}                   # cityscapes.py:161-162
    out["top"] = rng.randint(0, H - crop_size[0])                            # img_utils.py:256
    out["left"] = rng.randint(0, W - crop_size[1])                           # img_utils.py:257
    if objects:
        obj = objects[rng.randint(0, len(objects) - 1)]                      # img_utils.py:369
        scale = rng.choice(scales)                                           # img_utils.py:346
        o_img, o_mask = scaled_object(obj, scale)
        y1, x1, y2, x2 = mask_bbox(o_mask)
        bh, bw = y2 - y1, x2 - x1
        h0 = rng.randint(0, crop_size[0] - bh)                               # img_utils.py:412
        w0 = rng.randint(0, crop_size[1] - bw)                               # img_utils.py:414
        out.update(obj_img=o_img, obj_mask=o_mask, geom=(y1, x1, bh, bw, h0, w0))
    return out


def make_pair_batch(img, gen, tgt, gen_tgt, crop_size, params, flip=None, mean=MEAN, std=STD):
    """img / gen: uint8 [B,H,W,3] device tensors, tgt / gen_tgt: uint8 [B,H,W]; params: one dict per sample as returned by
    draw_sample_params. Returns (images float32 [2B,3,h,w], targets int64 [2B,h,w]), originals first."""
    for t in (img, gen, tgt, gen_tgt):
        if not t.is_cuda:
            raise RuntimeError("multishiftseg_amd.datapath runs on an MI355X only; there is no CPU path")
        if t.dtype != torch.uint8 or not t.is_contiguous():
            raise TypeError("pre-decoded maps must be contiguous uint8 tensors")
    B, H, W, _ = img.shape
    h, w = crop_size
    dev = img.device
    if len(params) != B:
        raise ValueError(f"{len(params)} parameter sets for a batch of {B}")
    mix = None
    if params[0]["p"] is not None:
        mix = torch.tensor([p["p"] for p in params], dtype=torch.float64).to(dev)
    crop = torch.tensor([[p["top"], p["left"]] for p in params], dtype=torch.int32).to(dev)
    flip_t = torch.tensor([int(f) for f in flip], dtype=torch.int32).to(dev) if flip is not None else None
    obj_img = obj_mask = geom = None
    ohm = owm = 0
    if "geom" in params[0]:
        ohm = max(p["obj_mask"].shape[0] for p in params)
        owm = max(p["obj_mask"].shape[1] for p in params)
        oi = np.zeros((B, ohm, owm, 3), dtype=np.float32)
        om = np.zeros((B, ohm, owm), dtype=np.uint8)
        for b, p in enumerate(params):
            oh, ow = p["obj_mask"].shape
            oi[b, :oh, :ow] = p["obj_img"]
            om[b, :oh, :ow] = p["obj_mask"]
        obj_img, obj_mask = torch.from_numpy(oi).to(dev), torch.from_numpy(om).to(dev)
        geom = torch.tensor([list(p["geom"]) for p in params], dtype=torch.int32).to(dev)
    out_img = torch.empty((2 * B, 3, h, w), device=dev, dtype=torch.float32)
    out_tgt = torch.empty((2 * B, h, w), device=dev, dtype=torch.int64)
    import ctypes
    mean3 = (ctypes.c_double * 3)(*[float(v) for v in mean])
    std3 = (ctypes.c_double * 3)(*[float(v) for v in std])
    call("mss_data_pair_f32", ptr(img), ptr(gen), ptr(tgt), ptr(gen_tgt), B, H, W, h, w, ptr(mix), ptr(crop), ptr(flip_t),
         ctypes.cast(mean3, ctypes.c_void_p), ctypes.cast(std3, ctypes.c_void_p), ptr(obj_img), ptr(obj_mask), ptr(geom),
         ohm, owm, ptr(out_img), ptr(out_tgt))
    return out_img, out_tgt
