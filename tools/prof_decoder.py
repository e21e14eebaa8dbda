"""MSDeformAttnPixelDecoder.forward_features forward + backward in isolation (C4: 16 x 704^2 crops by default) -- target for
`rocprofv3 --kernel-trace --stats`.   python tools/prof_decoder.py [N] [H] [W] [iters] [layers]
With a fifth argument: the per-launch table (HIP events) of the MFMA kernels of one forward + backward, aggregated by shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd.msdeformattn_decoder import MSDeformAttnPixelDecoder, ShapeSpec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 704
W = int(sys.argv[3]) if len(sys.argv) > 3 else 704
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
shape = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
torch.manual_seed(0)
dec = MSDeformAttnPixelDecoder({k: ShapeSpec(*v) for k, v in shape.items()}, transformer_dropout=0.0, transformer_nheads=8,
                               transformer_dim_feedforward=1024, transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                               transformer_in_features=["res3", "res4", "res5"], common_stride=4).cuda()
feats = {k: torch.randn(N, c, H // s, W // s, device="cuda") for k, (c, s) in shape.items()}
fwd_only = os.environ.get("MSS_PROF_FORWARD_ONLY") == "1"        # the inference form (C5: one 1024x2048 image)
for _ in range(iters):
    for p in dec.parameters():
        p.grad = None
    if fwd_only:
        with torch.no_grad():
            dec.forward_features(feats)
        continue
    mask, out0, ms = dec.forward_features(feats)
    (mask.sum() + sum(m.sum() for m in ms)).backward()
torch.cuda.synchronize()
if len(sys.argv) > 5:
    from multishiftseg_amd import kernels as K
    prof = K.ConvProfile(); K.set_conv_profile(prof)
    for p in dec.parameters():
        p.grad = None
    mask, out0, ms = dec.forward_features(feats)
    (mask.sum() + sum(m.sum() for m in ms)).backward()
    K.set_conv_profile(None)
    agg = {}
    for kind, tag, t, tf in prof.per_launch():
        d = agg.setdefault((kind, tag), [0, 0.0, 0.0])
        d[0] += 1; d[1] += t; d[2] += tf * t
    tot = sum(d[1] for d in agg.values())
    print(f"MFMA launches of one forward + backward: {tot:.2f} ms")
    for (kind, tag), (n, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{kind:14s} {str(tag):40s} x{n:<3d} {t:8.3f} ms  {w / t if t else 0:6.1f} TF/s")
