#!/usr/bin/env python3
"""Achievable fp32-MFMA and HBM-stream rates of the device (calibration for the rooflines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd._lib import call, ptr


def measure():
    out = torch.empty(256 * 8 * 256, device="cuda")
    res = {}
    for waves_per_simd in (1, 2):
        blocks, iters = 256 * waves_per_simd, 20000
        call("mss_peak_mfma_f32", ptr(out), blocks, 200)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        call("mss_peak_mfma_f32", ptr(out), blocks, iters)
        e.record()
        torch.cuda.synchronize()
        res[f"mfma_f32_tflops_{waves_per_simd}w"] = blocks * 4 * iters * 16 * 4096 / (s.elapsed_time(e) * 1e-3) / 1e12
    # bf16 matrix cores under load, random operands, two waves per SIMD: the delivered ceiling of the split-bf16 GEMM route
    for shape, name in ((0, "32x32x16"), (1, "16x16x32")):
        blocks, iters = 512, 4000
        call("mss_peak_mfma_bf16", ptr(out), blocks, 200, shape)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        call("mss_peak_mfma_bf16", ptr(out), blocks, iters, shape)
        e.record()
        torch.cuda.synchronize()
        res[f"mfma_bf16_{name}_tflops_2w"] = blocks * 4 * iters * 1572864.0 / (s.elapsed_time(e) * 1e-3) / 1e12
    n = 1 << 28   # 1 GiB source, 1 GiB destination: far beyond the 256 MiB Infinity Cache
    a = torch.empty(n, device="cuda").normal_()
    b = torch.empty(n, device="cuda")
    best = 0.0
    for variant in range(4):
        call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        e.record()
        torch.cuda.synchronize()
        res[f"stream_copy_v{variant}_GBs"] = 5 * 8.0 * n / (s.elapsed_time(e) * 1e-3) / 1e9
        best = max(best, res[f"stream_copy_v{variant}_GBs"])
    res["stream_copy_GBs"] = best
    # r06: the two directions apart and the 1 : 2 read : write mix of the Winograd input transforms (bytes: 4 n, 4 n, 6 n)
    for variant, name, byt in ((4, "stream_write_only_GBs", 4.0), (5, "stream_read_only_GBs", 4.0), (6, "stream_read1_write2_GBs", 6.0)):
        call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            call("mss_peak_stream_f32", ptr(a), ptr(b), n, variant)
        e.record()
        torch.cuda.synchronize()
        res[name] = 5 * byt * n / (s.elapsed_time(e) * 1e-3) / 1e9
    return res


if __name__ == "__main__":
    print(measure())
