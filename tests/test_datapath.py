"""Row f-4: the on-device data path (mixup, ToTensor, paired crop, Normalize, COCO paste, [orig; aug] batch) against
tests/golden/datapath.npz -- produced by the reference's own mix_func / extract_bboxes and the torchvision-documented torch
expressions (tools/gen_golden.py datapath) -- and against the numpy restatement oracle/data.py."""
import random

import numpy as np
import pytest
import torch

from conftest import golden


def _params(g, B):
    ps = []
    for b in range(B):
        y1, x1, y2, x2 = (int(v) for v in g[f"s{b}_bbox"])
        h0, w0 = (int(v) for v in g[f"s{b}_corner"])
        ps.append(dict(p=float(g[f"s{b}_p"]), top=int(g[f"s{b}_top"]), left=int(g[f"s{b}_left"]), obj_img=g[f"s{b}_obj_img"],
                       obj_mask=g[f"s{b}_obj_mask"], geom=(y1, x1, y2 - y1, x2 - x1, h0, w0)))
    return ps


def test_oracle_and_host_logic_vs_reference_golden():
    """CPU: the numpy restatement reproduces the reference outputs BIT-exactly; mask_bbox == the reference's extract_bboxes;
    draw_sample_params consumes Python's random stream in the reference's order."""
    from multishiftseg_amd import datapath
    from oracle import data as odata
    g = golden("datapath")
    B = g["img"].shape[0]
    ps = _params(g, B)
    imgs, tgts = odata.pair_batch(g["img"], g["gen"], g["tgt"], g["gen_tgt"], tuple(int(v) for v in g["crop"]), ps)
    np.testing.assert_array_equal(imgs, g["images"])
    np.testing.assert_array_equal(tgts, g["targets"])
    for b in range(B):
        y1, x1, y2, x2 = datapath.mask_bbox(g[f"s{b}_obj_mask"])
        np.testing.assert_array_equal([y1, x1, y2, x2], g[f"s{b}_bbox"])
    # the generator seeded random with 1234 and drew p, top, left (+ two paste corners inside mix_func) per sample
    random.seed(1234)
    H, W = g["img"].shape[1:3]
    crop = tuple(int(v) for v in g["crop"])
    for b in range(B):
        objs = [(g[f"s{b}_obj_img"], g[f"s{b}_obj_mask"])]

        class _Rng:                      # the fixture's objects were not chosen / rescaled through `random`
            random = staticmethod(random.random)
            randint = staticmethod(lambda a, c: 0 if (a, c) == (0, 0) else random.randint(a, c))
            choice = staticmethod(lambda seq: seq[-1])
        d = datapath.draw_sample_params(H, W, crop, True, objects=objs, scaled_object=lambda o, s: o, rng=_Rng)
        assert d["p"] == float(g[f"s{b}_p"]) and d["top"] == int(g[f"s{b}_top"]) and d["left"] == int(g[f"s{b}_left"])
        assert d["geom"][4:] == tuple(int(v) for v in g[f"s{b}_corner"])
    with pytest.raises(RuntimeError, match="MI355X"):
        datapath.make_pair_batch(*(torch.from_numpy(g[k]) for k in ("img", "gen", "tgt", "gen_tgt")), crop, ps)


@pytest.mark.gpu
def test_device_data_path_bit_exact_vs_reference_golden():
    from multishiftseg_amd import datapath
    g = golden("datapath")
    B = g["img"].shape[0]
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    imgs, tgts = datapath.make_pair_batch(dev("img"), dev("gen"), dev("tgt"), dev("gen_tgt"), tuple(int(v) for v in g["crop"]),
                                          _params(g, B))
    np.testing.assert_array_equal(imgs.cpu().numpy(), g["images"])          # float32 values bit-exact
    np.testing.assert_array_equal(tgts.cpu().numpy(), g["targets"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,h,w,mix,paste,flip", [(2, 1024, 2048, 700, 700, True, True, False), (3, 37, 53, 37, 53, False, False, True),
                                                      (1, 64, 96, 33, 47, True, True, True)])
def test_device_data_path_vs_oracle(B, H, W, h, w, mix, paste, flip):
    """Full Cityscapes frame -> 700x700 crops (exps/DeepLab.yaml), no-mixup / no-paste / flip variants, ragged sizes."""
    from multishiftseg_amd import datapath
    from oracle import data as odata
    rng = np.random.default_rng(B * 1000 + h)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    gen = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    tgt = rng.integers(0, 19, (B, H, W)).astype(np.uint8)
    gen_tgt = np.where(rng.random((B, H, W)) < 0.05, 254, tgt).astype(np.uint8)
    objs = []
    for _ in range(4):
        oh, ow = int(rng.integers(5, min(h, 90))), int(rng.integers(5, min(w, 120)))
        m = np.where(rng.random((oh, ow)) < 0.5, 254, 0).astype(np.uint8)
        m[rng.random((oh, ow)) < 0.05] = 255
        m[oh // 2, ow // 2] = 254
        objs.append(((rng.random((oh, ow, 3)) * 255).astype(np.float32), m))
    random.seed(B + h)
    ps = [datapath.draw_sample_params(H, W, (h, w), mix, objects=objs if paste else None, scaled_object=lambda o, s: o) for _ in range(B)]
    fl = [bool(v) for v in rng.integers(0, 2, B)] if flip else None
    t = lambda a: torch.from_numpy(a).cuda()
    imgs, tgts = datapath.make_pair_batch(t(img), t(gen), t(tgt), t(gen_tgt), (h, w), ps, flip=fl)
    ri, rt = odata.pair_batch(img, gen, tgt, gen_tgt, (h, w), ps, flip=fl)
    np.testing.assert_array_equal(tgts.cpu().numpy(), rt)
    np.testing.assert_array_equal(imgs.cpu().numpy(), ri)
