"""PMC target: the MSDeformAttn forward at C4 N = 16 (op form and fused form) on the per-sample-record kernel, three launches each.   rocprofv3 --pmc ... -- python3 tools/pmc_msda_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib, MultiScaleDeformableAttention as MSDA
from multishiftseg_amd.ms_deform_attn import _FusedSampleFn
from tools.m2f_legs import msda_inputs

t = msda_inputs(16, [(88, 88), (44, 44), (22, 22)])
if True:
    for _ in range(3):
        MSDA.ms_deform_attn_forward(t["value"], t["shp"], t["starts"], t["loc"], t["attn"], 128)
        with torch.no_grad():
            _FusedSampleFn.apply(t["value"], t["shp"], t["starts"], t["off"], t["lg"], t["ref"])
torch.cuda.synchronize()
