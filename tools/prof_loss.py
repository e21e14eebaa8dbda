#!/usr/bin/env python3
"""Fused RelContrastiveLoss (value + both gradients) at the C3 / C2 sizes, for `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import synth
from multishiftseg_amd.loss import RelContrastiveLoss
from multishiftseg_amd.trainer import LOSS_PARAMS

for B, H, W in ((2, 1024, 2048), (16, 700, 700)):
    logits = (torch.randn(B, 19, H, W, device="cuda") * 3).requires_grad_(True)
    score = (torch.randn(B, H, W, device="cuda") * 4).requires_grad_(True)
    tgt = torch.from_numpy(synth.synth_targets(3, B // 2, H, W)).cuda()
    crit = RelContrastiveLoss(LOSS_PARAMS, pairing="device")
    for _ in range(10):
        crit(logits, score, tgt.clone())
    torch.cuda.synchronize()
