#!/usr/bin/env python3
"""The Mask2Former legs of SURVEY 8(d) as ONE dictionary (bench.py puts it into its JSON line as `m2f` at N = 1; standalone:
`python tools/m2f_legs.py`). All on synthetic tensors of the BASELINE shapes, inputs resident in HBM, HIP events on the
current stream:

  msda            the op alone at C4 (704^2 crops: levels 22^2 / 44^2 / 88^2, 10 164 tokens) with N = 1 and N = 16, and at C5
                  (1024x2048: 32x64 / 64x128 / 128x256, 43 008 tokens) with N = 1: forward, fused forward (raw offsets + logits),
                  backward; rates against the COMPULSORY-byte roofline of SURVEY 8(d) (8 TB/s) and the L2 row-gather rate
  pixel_decoder   MSDeformAttnPixelDecoder.forward_features (6 encoder layers): forward and forward + backward
  fused_score     mask-logit GEMM output -> x4 bilinear + sigmoid + class mix + max (f-2), one 1024x2048 image
  metric_sweep    AUROC / AUPRC / FPR@95 over 64 score maps of 1024x2048 on the device (f-1)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
L2_GATHER_GBS = 18800.0     # MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served from the XCDs' L2
C4 = [(22, 22), (44, 44), (88, 88)]
C5 = [(32, 64), (64, 128), (128, 256)]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def msda_inputs(N, shapes, device="cuda", seed=0):
    """value / locations / weights as the encoder produces them: reference points at the pixel centres of each query's own
    level, offsets ~ N(0, 3 px), softmax weights."""
    g = torch.Generator(device=device).manual_seed(seed)
    shp = torch.as_tensor(shapes, dtype=torch.long, device=device)
    shp._mss_host = [tuple(s) for s in shapes]
    starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    S = int(shp.prod(1).sum())
    value = torch.randn(N, S, 8, 32, device=device, generator=g)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=device) + 0.5) / h, (torch.arange(w, device=device) + 0.5) / w,
                                                indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes])[None, :, None, :].expand(N, S, 3, 2).contiguous()
    off = torch.randn(N, S, 8, 3, 4, 2, device=device, generator=g) * 3
    lg = torch.randn(N, S, 8, 12, device=device, generator=g)
    loc = (ref[:, :, None, :, None, :] + off / shp.flip(-1)[None, None, None, :, None, :].float()).contiguous()
    attn = torch.softmax(lg, -1).view(N, S, 8, 3, 4).contiguous()
    gout = torch.randn(N, S, 256, device=device, generator=g)
    return dict(shp=shp, starts=starts, S=S, value=value, ref=ref, off=off, lg=lg, loc=loc, attn=attn, gout=gout)


def msda_leg(N, shapes):
    from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
    from multishiftseg_amd.ms_deform_attn import _FusedSampleFn
    t = msda_inputs(N, shapes)
    S = t["S"]
    f = timeit(lambda: MSDA.ms_deform_attn_forward(t["value"], t["shp"], t["starts"], t["loc"], t["attn"], 128), iters=20)
    with torch.no_grad():
        ff = timeit(lambda: _FusedSampleFn.apply(t["value"], t["shp"], t["starts"], t["off"], t["lg"], t["ref"]), iters=20)
    b = timeit(lambda: MSDA.ms_deform_attn_backward(t["value"], t["shp"], t["starts"], t["loc"], t["attn"], t["gout"], 128), iters=10)
    # forward 4(N S M D + 3 N Lq M L P + N Lq M D) bytes; backward = forward inputs + grad_out + grad_value
    # (zero + read-modify-write, counted 2x) + grad_loc + grad_attn  (= 75 MB at N = 1, Lq = S = 10 164, as the survey states)
    fwd_b = 4 * (N * S * 256 + 3 * N * S * 8 * 12 + N * S * 256)
    bwd_b = 4 * (N * S * 256 + 3 * N * S * 8 * 12 + N * S * 256 + 2 * N * S * 256 + 3 * N * S * 8 * 12)
    gather = N * S * 8 * 48 * 128
    # rows the kernel really fetches per (query, head): corners outside the image cost no memory access (out-of-range buffer offset)
    # and samples of one level may share rows -- counted exactly on image 0 ("report unique rows")
    with torch.no_grad():
        ids = []
        for l, (h, w) in enumerate(shapes):
            x, y = t["loc"][0, :, :, l, :, 0] * w - 0.5, t["loc"][0, :, :, l, :, 1] * h - 0.5
            x0, y0 = torch.floor(x).long(), torch.floor(y).long()
            for dy in (0, 1):
                for dx in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
                    ids.append(torch.where(ok, int(t["starts"][l]) + yy * w + xx, torch.full_like(xx, -1)))
        ids = torch.cat(ids, -1).sort(-1).values                               # [S, 8, 48]
        distinct = (ids[..., 1:] != ids[..., :-1]).sum(-1) + 1 - (ids[..., 0] < 0).long()   # the -1 group (if any) is not a row
        rows = float(distinct.float().mean())
    gather_u = N * S * 8 * rows * 128
    return {"N": N, "tokens": S, "forward_ms": round(f, 4), "fused_forward_ms": round(ff, 4), "backward_ms": round(b, 4),
            "forward_compulsory_MB": round(fwd_b / 1e6, 1), "forward_GBs": round(fwd_b / f / 1e6, 1),
            "forward_frac_of_hbm_peak": round(fwd_b / f / 1e6 / HBM_PEAK_GBS, 4),
            "forward_L2_row_gather_GBs": round(gather / f / 1e6, 1),
            "forward_frac_of_L2_gather_ceiling": round(gather / f / 1e6 / L2_GATHER_GBS, 3),
            "forward_unique_rows_per_query_head": round(rows, 2),
            "forward_L2_unique_row_gather_GBs": round(gather_u / f / 1e6, 1),
            "forward_frac_of_L2_gather_ceiling_unique_rows": round(gather_u / f / 1e6 / L2_GATHER_GBS, 3),
            "backward_algorithmic_MB": round(bwd_b / 1e6, 1), "backward_GBs": round(bwd_b / b / 1e6, 1),
            "backward_frac_of_hbm_peak": round(bwd_b / b / 1e6 / HBM_PEAK_GBS, 4)}


def decoder_leg():
    from multishiftseg_amd.msdeformattn_decoder import MSDeformAttnPixelDecoder, ShapeSpec
    shape = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
    torch.manual_seed(0)
    dec = MSDeformAttnPixelDecoder({k: ShapeSpec(*v) for k, v in shape.items()}, transformer_dropout=0.0, transformer_nheads=8,
                                   transformer_dim_feedforward=1024, transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                                   transformer_in_features=["res3", "res4", "res5"], common_stride=4).cuda()
    out = {}
    for tag, N, H, W in (("c4_704x704_n16", 16, 704, 704), ("c4_704x704_n1", 1, 704, 704), ("c5_1024x2048_n1", 1, 1024, 2048)):
        feats = {k: torch.randn(N, c, H // s, W // s, device="cuda") for k, (c, s) in shape.items()}
        with torch.no_grad():
            f = timeit(lambda: dec.forward_features(feats), iters=5, warm=2)

        def fb():
            for p in dec.parameters():
                p.grad = None
            mask, out0, ms = dec.forward_features(feats)
            (mask.sum() + sum(m.sum() for m in ms)).backward()
        b = timeit(fb, iters=5, warm=3)
        out[tag] = {"forward_ms": round(f, 2), "forward_backward_ms": round(b, 2), "images_per_s_fwd_bwd": round(N / b * 1e3, 1)}
        if N > 1:                                         # the same on the split-bf16 GEMM route (kernels.set_gemm_route; co-headline only)
            from multishiftseg_amd import kernels as K
            K.set_gemm_route("bf16x3")
            try:
                with torch.no_grad():
                    f3 = timeit(lambda: dec.forward_features(feats), iters=5, warm=2)
                b3 = timeit(fb, iters=5, warm=3)
                out[tag]["bf16x3_route"] = {"forward_ms": round(f3, 2), "forward_backward_ms": round(b3, 2), "images_per_s_fwd_bwd": round(N / b3 * 1e3, 1)}
            finally:
                K.set_gemm_route(None)
        if N == 1:
            try:                                          # the inference form: the same forward as ONE hipGraph launch
                from multishiftseg_amd.msdeformattn_decoder import GraphedFeatures
                for p in dec.parameters():
                    p.grad = None
                gf = GraphedFeatures(dec, feats)
                out[tag]["forward_hipgraph_ms"] = round(timeit(lambda: gf(feats), iters=10, warm=2), 3)
                del gf
            except Exception as exc:
                out[tag]["forward_hipgraph_error"] = repr(exc)[:200]
        del feats
    return out


def fused_score_leg():
    from multishiftseg_amd import kernels as K
    emb = torch.randn(1, 100, 256, device="cuda") * 0.2
    feat = torch.randn(1, 256, 256, 512, device="cuda")
    cls = torch.randn(1, 100, 20, device="cuda") * 2
    t_gemm = timeit(lambda: K.m2f_mask_logits(emb, feat), iters=10, warm=3)
    lg = K.m2f_mask_logits(emb, feat)
    t = timeit(lambda: K.m2f_score_fused(cls, lg, (1024, 2048)), iters=10, warm=3)
    px = 1024 * 2048
    return {"image": "1x1024x2048, 100 queries, 256x512 mask features", "mask_gemm_ms": round(t_gemm, 3), "fused_score_ms": round(t, 3),
            "gpix_s": round(px / t / 1e6, 2), "algorithmic_GBs": round((256 * 512 * 400 + px * 4) / t / 1e6, 1),
            "frac_of_hbm_peak": round((256 * 512 * 400 + px * 4) / t / 1e6 / 8000.0, 4),
            "bound": "VALU: one bilinear interpolation + sigmoid per (pixel, query) = 100 per pixel (v_exp_f32 / v_rcp_f32 are quarter "
                     "rate); since round 4 the 100 x 19 class mix runs on the matrix cores beside them (v_mfma_f32_32x32x2_f32, "
                     "csrc/m2f.hip); MSS_M2F_MFMA=0 = the all-VALU kernel of rounds 2-3"}


def metric_leg(images=64, h=1024, w=2048):
    from multishiftseg_amd import metric as M
    g = torch.Generator(device="cuda").manual_seed(images)
    batches = []
    for _ in range(images):
        lab = (torch.rand(1, h, w, device="cuda", generator=g) < 0.03).long()
        lab[torch.rand(1, h, w, device="cuda", generator=g) < 0.05] = 255
        batches.append((torch.randn(1, h, w, device="cuda", generator=g) + 1.2 * (lab == 1), lab))
    for _ in range(2):
        meter = M.OODMeter()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, l in batches:
            meter.update(s, l)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = meter.compute()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    px = images * h * w
    # the same sweep fed two maps per update -- the batch the reference's test loader uses (test_deeplab.py:47 valid_batch = 2): an
    # update is ONE ~20 us kernel behind ~25 us of Python, so the per-update host cost is what the single-image figure measures
    pairs = [(torch.cat((batches[i][0], batches[i + 1][0])), torch.cat((batches[i][1], batches[i + 1][1]))) for i in range(0, images - 1, 2)]
    for _ in range(2):
        meter2 = M.OODMeter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for s, l in pairs:
            meter2.update(s, l)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
    res2 = meter2.compute()
    # ... and handed over sixteen maps per launch (OODMeter.update_many, mss_oodm_compact_lanes_batch_f32): what a sweep that holds its
    # maps does (the reference appends every batch and evaluates at the end, test_deeplab.py:84-102)
    for _ in range(2):
        meter3 = M.OODMeter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        meter3.update_many(batches)
        torch.cuda.synchronize()
        t6 = time.perf_counter()
    res3 = meter3.compute()
    return {"images": images, "update_many_ms": round(1e3 * (t6 - t5), 2), "update_many_GBs_of_12B_per_pixel": round(px * 12 / (t6 - t5) / 1e9, 1),
            "auroc_update_many": round(float(res3[0]), 6) if res3 is not None else None, "pixels": px, "update_ms": round(1e3 * (t1 - t0), 2), "compute_ms": round(1e3 * (t2 - t1), 2),
            "gpix_s": round(px / (t2 - t0) / 1e9, 2), "update_GBs_of_12B_per_pixel": round(px * 12 / (t1 - t0) / 1e9, 1),
            "update_batch2_ms": round(1e3 * (t4 - t3), 2),
            "update_batch2_GBs_of_12B_per_pixel": round(2 * len(pairs) * h * w * 12 / (t4 - t3) / 1e9, 1),
            "auroc": round(float(res[0]), 6) if res is not None else None,
            "auroc_batch2": round(float(res2[0]), 6) if res2 is not None else None}


def measure():
    out = {"msda": {"c4_n1": msda_leg(1, C4), "c4_n16": msda_leg(16, C4), "c5_n1": msda_leg(1, C5),
                    "roofline_note": "HBM bound on the compulsory bytes of SURVEY 8(d) (8 TB/s); the forward is a row gather served "
                                     "by L2 (48 x 128 B per (query, head)), its rate is forward_L2_row_gather_GBs (forward_L2_unique_row_gather_GBs counts only the distinct in-image rows of a (query, head), measured on image 0); the chip's measured ceiling "
                                     "for L2-resident row gathers is 16.8-18.8 TB/s (MI355X_MICROARCH.md, Indexed rows): forward_frac_of_L2_gather_ceiling"},
           "pixel_decoder_forward_features": decoder_leg(), "fused_score": fused_score_leg(), "metric_sweep": metric_leg()}
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
