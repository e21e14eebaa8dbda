"""MSDeformAttnPixelDecoder.forward_features at the BASELINE shapes: C4 (704x704 crops, N = 1 and 16) and
C5 (1x1024x2048): forward (no_grad) and forward + backward (parameter gradients, frozen backbone features) time."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd.msdeformattn_decoder import MSDeformAttnPixelDecoder, ShapeSpec
from tools.microbench import timeit

SHAPE = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
torch.manual_seed(0)
dec = MSDeformAttnPixelDecoder({k: ShapeSpec(*v) for k, v in SHAPE.items()}, transformer_dropout=0.0, transformer_nheads=8,
                               transformer_dim_feedforward=1024, transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                               transformer_in_features=["res3", "res4", "res5"], common_stride=4).cuda()
for tag, N, H, W in (("c4_n1", 1, 704, 704), ("c4_n16", 16, 704, 704), ("c5_n1", 1, 1024, 2048)):
    feats = {k: torch.randn(N, c, H // s, W // s, device="cuda") for k, (c, s) in SHAPE.items()}
    with torch.no_grad():
        f = timeit(lambda: dec.forward_features(feats), iters=5, warm=2)

    def fb():
        for p in dec.parameters():
            p.grad = None
        mask, out0, ms = dec.forward_features(feats)
        (mask.sum() + sum(m.sum() for m in ms)).backward()
    b = timeit(fb, iters=5, warm=3)            # the first iterations grow the allocator's pool: 92.6 vs 86.2 ms with warm=1
    print(json.dumps(dict(workload=tag, N=N, H=H, W=W, forward_ms=round(f, 2), fwd_bwd_ms=round(b, 2))), flush=True)
    del feats
