#!/usr/bin/env python3
"""HBM-bound kernels of the path at their BASELINE sizes: achieved GB/s of algorithmic bytes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K, synth
from multishiftseg_amd.loss import RelContrastiveLoss
from multishiftseg_amd.trainer import LOSS_PARAMS
from tools.microbench import timeit


def row(name, ms, nbytes, note=""):
    print(json.dumps(dict(kernel=name, ms=round(ms, 4), alg_MB=round(nbytes / 1e6, 1), GBs=round(nbytes / ms / 1e6, 1), note=note)), flush=True)


def main():
    dev = "cuda"
    # OOD-score tail at 2x1024x2048 (train) : dec 48-wide half-res -> score + logits
    for n, H, W in ((1, 1024, 2048), (2, 1024, 2048)):
        dec = K.Act(torch.randn(n, H // 2, W // 2, 48, device=dev))
        ms = timeit(lambda: K.ood_score(dec.slice(20, 19), dec.slice(0, 19), H, W))
        row(f"ood_score score+logit {n}x{H}x{W}", ms, n * H * W * (2 * 19 * 4 / 4 + 4 + 19 * 4))
        ms = timeit(lambda: K.ood_score(dec.slice(20, 19), None, H, W, want_logit=False))
        row(f"ood_score score only {n}x{H}x{W}", ms, n * H * W * 23.0, "23 B/px")
    # M2F score, C5: cls [1,100,20], mask [1,100,1024,2048]
    cls = torch.randn(1, 100, 20, device=dev)
    mask = torch.randn(1, 100, 1024, 2048, device=dev)
    ms = timeit(lambda: K.m2f_score(cls, mask, (1024, 2048)), iters=5)
    row("m2f_score 1x100x1024x2048", ms, 1024 * 2048 * 404.0, "404 B/px")
    del mask
    # fused loss fwd+bwd at C3 (2x1024x2048) and C2 (16x768x768)
    for B, H, W in ((2, 1024, 2048), (16, 768, 768)):
        logits = (torch.randn(B, 19, H, W, device=dev) * 3).requires_grad_(True)
        score = (torch.randn(B, H, W, device=dev) * 4).requires_grad_(True)
        tgt = torch.from_numpy(synth.synth_targets(3, B // 2, H, W)).to(dev)
        crit = RelContrastiveLoss(LOSS_PARAMS, pairing="device")
        # the loss mutates its targets in place (loss.py:110-115): a fresh copy per call, made OUTSIDE the timed region (the 33 MB
        # clone used to be ~27 us of the figure)
        pool = [tgt.clone() for _ in range(16)]
        def f():
            crit.value_and_grads(logits.detach(), score.detach(), pool.pop())
        ms = timeit(f, iters=5)
        row(f"rel_contrastive_loss value+grads {B}x19x{H}x{W}", ms, B * H * W * 168.0, "168 B/px minimum")
    # BN statistics + maxpool + upsample at train sizes
    x = K.Act(torch.randn(2, 512, 1024, 128, device=dev))
    acc = K._col_accum(x.M, 128, dev)
    from multishiftseg_amd._lib import call, ptr
    ms = timeit(lambda: call("mss_bn_stats_nhwc_f32", x.ptr, x.M, 128, x.ld, ptr(acc)))
    row("bn_stats 2x512x1024x128", ms, x.M * 128 * 4.0)
    x1 = K.Act(torch.randn(2, 1024, 2048, 64, device=dev))
    ms = timeit(lambda: K.maxpool3s2(x1))
    row("maxpool3s2 2x1024x2048x64", ms, x1.M * 64 * 4.0 * 1.25)
    u = K.Act(torch.randn(2, 128, 256, 256, device=dev))
    ms = timeit(lambda: K.upsample_ac(u, 512, 1024))
    row("upsample_ac 2x128x256x256 -> 512x1024", ms, 2 * 512 * 1024 * 256 * 4.0 * (1 + 1 / 16))
    g = K.Act(torch.randn(2, 512, 1024, 256, device=dev))
    ms = timeit(lambda: K.upsample_ac_bwd(g, 128, 256))
    row("upsample_ac_bwd 2x512x1024x256 -> 128x256", ms, 2 * 512 * 1024 * 256 * 4.0 * (1 + 1 / 16))
    xx = K.Act(torch.randn(2, 128, 256, 4096, device=dev))
    ms = timeit(lambda: K.gap(xx))
    row("gap 2x128x256x4096", ms, xx.M * 4096 * 4.0)
    # backward of the tail and BatchNorm+ReLU backward at the 2x1024x2048 step's sizes
    dec = K.Act(torch.randn(2, 512, 1024, 48, device=dev))
    ds = torch.randn(2, 1024, 2048, device=dev)
    dl = torch.randn(2, 19, 1024, 2048, device=dev)
    dd = K.Act.zeros(2, 512, 1024, 48, dev)
    ms = timeit(lambda: K.ood_score_bwd(dec.slice(20, 19), ds, dl, dd.slice(20, 19), dd.slice(0, 19), 1024, 2048))
    row("ood_score_bwd 2x1024x2048", ms, 2 * 1024 * 2048 * (19 * 4 + 4) + dec.M * (19 * 4 + 48 * 4.0))
    del dec, ds, dl, dd
    import torch.nn as nn
    xb = K.Act(torch.randn(2, 512, 1024, 256, device=dev))
    dyb = K.Act(torch.randn(2, 512, 1024, 256, device=dev))
    bn = nn.BatchNorm2d(256).to(dev)
    st = K.bn_fold(bn, xb, train=True)
    ms = timeit(lambda: K.bn_relu_backward(dyb, xb, st, want_param_grads=True), iters=5)
    row("bn_relu_backward (reduce + apply) 2x512x1024x256", ms, xb.M * 256 * 4.0 * 5, "reduce: dy,x; apply: dy,x -> dx")
    ms = timeit(lambda: K.bn_fold(bn, xb, train=True), iters=5)
    row("bn_fold train (stats over x) 2x512x1024x256", ms, xb.M * 256 * 4.0)
    del xb, dyb
    if os.environ.get("MSS_BENCH_SKIP_WINO") != "1":
        wino_transforms(dev)


def wino_transforms(dev):
    """The three streaming transforms of the Winograd layers at the shapes of the 2x1024x2048 step."""
    from multishiftseg_amd import _lib
    from multishiftseg_amd._lib import call, ptr
    for (N, H, W, C, dil) in ((2, 512, 1024, 128, 1), (2, 512, 1024, 256, 1), (2, 256, 512, 256, 1), (2, 128, 256, 512, 1),
                              (2, 128, 256, 1024, 4), (2, 128, 256, 4096, 12), (2, 128, 256, 4096, 24)):
        ts = K.wino_tile(H, W, dil)
        P = (ts + 2) ** 2
        T = _lib.value("mss_wino_num_tiles", N, H, W, dil, ts)
        x = K.Act(torch.randn(N, H, W, C, device=dev))
        xt = torch.empty((P, T, C), device=dev)
        sc = torch.rand(C, device=dev) + 0.5
        sh = torch.randn(C, device=dev)
        ms = timeit(lambda: call("mss_wino_input_transform_f32", x.ptr, x.ld, N, H, W, C, dil, ts, ptr(sc), ptr(sh), 1, ptr(xt)), iters=10)
        row(f"wino_input_transform F{ts} {N}x{H}x{W}x{C} d{dil}", ms, 4.0 * (x.M * C + P * T * C), "x once + X' once")
        y = K.Act.empty(N, H, W, C, dev)
        ms = timeit(lambda: call("mss_wino_output_transform_f32", ptr(xt), N, H, W, C, dil, ts, x.ptr, x.ld, y.ptr, y.ld, None), iters=10)
        row(f"wino_output_transform(+res) F{ts} {N}x{H}x{W}x{C} d{dil}", ms, 4.0 * (2 * x.M * C + P * T * C), "Y' once + res + y")
        ms = timeit(lambda: call("mss_wino_grad_output_transform_f32", x.ptr, x.ld, N, H, W, C, dil, ts, ptr(xt)), iters=10)
        row(f"wino_grad_output_transform F{ts} {N}x{H}x{W}x{C} d{dil}", ms, 4.0 * (x.M * C + P * T * C))
        del x, xt, y


def layout_experiment(dev):
    """Does the [P][T][C] layout (36 write streams tens of MB apart) cost bandwidth against a tile-block-major one?"""
    from multishiftseg_amd._lib import call, ptr
    n = 1 << 25                                  # 128 MB read, 36x = 4.8 GB written
    src = torch.randn(n, device=dev)
    for ns in (16, 36):
        dst = torch.empty(ns * n, device=dev)
        for blocked, blk in ((0, 4), (1, 128 * 128), (1, 128 * 256), (1, 128 * 1024), (1, 16 * 128)):
            ms = timeit(lambda: call("mss_peak_scatter_f32", ptr(src), ptr(dst), n, ns, blocked, blk), iters=5)
            row(f"scatter ns={ns} blocked={blocked} blk_floats={blk}", ms, 4.0 * n * (1 + ns))
        del dst


if __name__ == "__main__":
    main()
