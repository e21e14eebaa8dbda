"""Where does grad_loc of the MSDA backward differ from the oracle at C4 N = 16? The only mismatches are samples whose pixel
coordinate is an integer to within fp32 rounding (a kink of the piecewise-linear interpolation): tests/test_gpu_msda.py masks them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
from oracle import msda as omsda
tag, N, shapes = "c4_n16", 16, [(22, 22), (44, 44), (88, 88)]
rng = np.random.default_rng(len(tag) + N)
shp = np.array(shapes, dtype=np.int64)
starts = np.concatenate([[0], np.cumsum(shp.prod(1))[:-1]]).astype(np.int64)
S, L = int(shp.prod(1).sum()), len(shapes)
ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2) for h, w in shapes])
off = rng.standard_normal((N, S, 8, L, 4, 2)).astype(np.float32) * 3
off[:, ::17] *= 12
loc = (ref[None, :, None, None, None, :] + off / shp[None, None, None, :, None, ::-1]).astype(np.float32)
attn = rng.random((N, S, 8, L, 4), dtype=np.float32)
attn /= attn.sum((-1, -2), keepdims=True)
value = rng.standard_normal((N, S, 8, 32), dtype=np.float32)
gout = rng.standard_normal((N, S, 256), dtype=np.float32)
t = {k: torch.from_numpy(v).cuda() for k, v in dict(value=value, loc=loc, attn=attn, gout=gout).items()}
ts, tst = torch.from_numpy(shp).cuda(), torch.from_numpy(starts).cuda()
for mode in ("binned", "old"):
    os.environ["MSS_MSDA_BWD_BINNED"] = "1" if mode == "binned" else "0"
    gv, gl, ga = MSDA.ms_deform_attn_backward(t["value"], ts, tst, t["loc"], t["attn"], t["gout"], 128)
    gl = gl.cpu().numpy()
    for i in range(0, N, 2):
        wv, wl, wa = omsda.backward_sampled(value[i:i + 2], shp, starts, loc[i:i + 2], attn[i:i + 2], gout[i:i + 2])
        err = np.abs(gl[i:i + 2] - wl)
        bad = np.argwhere(err > 2e-5 * np.abs(wl).max() + 1e-3)
        print(mode, "images", i, i + 1, "bad", len(bad), "max err", err.max())
        for b in bad[:6]:
            n, q, m, l, p, xy = b
            H, W = shapes[l]
            lx, ly = loc[i + n, q, m, l, p]
            print("   ", b, "got", gl[i + n, q, m, l, p], "want", wl[n, q, m, l, p], "pix x,y", lx * W - 0.5, ly * H - 0.5,
                  "x frac bits", float(np.float32(lx * W - 0.5)) % 1.0, float(np.float32(ly * H - 0.5)) % 1.0)
        if i >= 2 and mode == "old":
            break
