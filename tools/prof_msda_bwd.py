"""MSDA backward at C4 N = 16 in isolation (a few calls) -- target for `rocprofv3 --pmc ...` passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(32, 64), (64, 128), (128, 256)] if (len(sys.argv) > 2 and sys.argv[2] == "c5") else [(22, 22), (44, 44), (88, 88)]
shp = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
S = int(shp.prod(1).sum())
torch.manual_seed(0)
value = torch.randn(N, S, 8, 32, device="cuda")
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                            indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes])[None, :, None, :].expand(N, S, 3, 2)
off = torch.randn(N, S, 8, 3, 4, 2, device="cuda") * 3
loc = (ref[:, :, None, :, None, :] + off / shp.flip(-1)[None, None, None, :, None, :].float()).contiguous()
attn = torch.softmax(torch.randn(N, S, 8, 12, device="cuda"), -1).view(N, S, 8, 3, 4).contiguous()
g = torch.randn(N, S, 256, device="cuda")
shp._mss_host = shapes
for _ in range(8):
    MSDA.ms_deform_attn_backward(value, shp, starts, loc, attn, g, 128)
torch.cuda.synchronize()
