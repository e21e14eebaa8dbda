"""MSDeformAttn encoder (6 layers, C4 shapes 22^2/44^2/88^2, N=16): forward and forward+backward time."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd.msdeformattn_encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine
from tools.microbench import timeit
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
enc = MSDeformAttnTransformerEncoderOnly(num_feature_levels=3, dropout=0.0).cuda()
shapes = [(22, 22), (44, 44), (88, 88)]
srcs = [torch.randn(N, 256, h, w, device="cuda") for h, w in shapes]
pe = PositionEmbeddingSine(128, normalize=True)
pos = [pe(s) for s in srcs]
with torch.no_grad():
    f = timeit(lambda: enc(srcs, pos), iters=5, warm=2)
def fb():
    for p in enc.parameters():
        p.grad = None
    out, _, _ = enc(srcs, pos)
    out.sum().backward()
b = timeit(fb, iters=3, warm=1)
print(json.dumps(dict(N=N, forward_ms=round(f, 2), fwd_bwd_ms=round(b, 2))))
