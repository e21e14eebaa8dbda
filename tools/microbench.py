#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (hipEvent via torch.cuda.Event on the current stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import kernels as K


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


def conv_case(name, n, h, w, cin, cout, r, stride=1, dil=1):
    pad = dil if r == 3 else 0
    x = K.Act(torch.randn(n, h, w, cin, device="cuda"))
    wt = torch.randn(cout, cin, r, r, device="cuda") / (cin * r * r) ** 0.5
    pw = K.pack_weight(wt)
    oh = K.conv_out_size(h, r, stride, dil, pad); ow = K.conv_out_size(w, r, stride, dil, pad)
    out = K.Act.empty(n, oh, ow, cout, "cuda")
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda")
    ms = timeit(lambda: K.conv2d(x, pw, stride=stride, dil=dil, pad=pad, in_affine=(sc, sh), in_relu=True, out=out))
    flop = 2.0 * n * oh * ow * cout * cin * r * r
    print(json.dumps(dict(kernel="conv", name=name, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2),
                          shape=[n, h, w, cin, cout, r, stride, dil])), flush=True)


def main():
    print(torch.cuda.get_device_name(0))
    # C1-like (1x512x1024) and C3-like (2x1024x2048) layer shapes
    for n, H, W in ((1, 512, 1024), (2, 1024, 2048)):
        h8, w8, h2, w2 = H // 8, W // 8, H // 2, W // 2
        conv_case("mod2 3x3 128", n, h2, w2, 128, 128, 3)
        conv_case("mod4 3x3 512", n, h8, w8, 512, 512, 3)
        conv_case("mod5 3x3 d2 512->1024", n, h8, w8, 512, 1024, 3, dil=2)
        conv_case("mod7 3x3 d4 1024->2048", n, h8, w8, 1024, 2048, 3, dil=4)
        conv_case("mod7 1x1 2048->4096", n, h8, w8, 2048, 4096, 1)
        conv_case("aspp 3x3 d12 4096->256", n, h8, w8, 4096, 256, 3, dil=12)
        conv_case("aspp 3x3 d36 4096->256", n, h8, w8, 4096, 256, 3, dil=36)
        conv_case("final.0 3x3 304->256", n, h2, w2, 304, 256, 3)
        conv_case("heads 1x1 256->48", n, h2, w2, 256, 48, 1)
    # MSDA C4 / C5
    from multishiftseg_amd import MultiScaleDeformableAttention as MSDA
    for N, shapes in ((1, [(22, 22), (44, 44), (88, 88)]), (16, [(22, 22), (44, 44), (88, 88)]), (1, [(32, 64), (64, 128), (128, 256)])):
        shp = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
        starts = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
        S = int(shp.prod(1).sum())
        value = torch.randn(N, S, 8, 32, device="cuda")
        # reference points = pixel centres + small offsets, like the encoder produces
        loc = torch.rand(N, S, 8, 3, 4, 2, device="cuda")
        attn = torch.softmax(torch.randn(N, S, 8, 12, device="cuda"), -1).view(N, S, 8, 3, 4)
        g = torch.randn(N, S, 256, device="cuda")
        ms_f = timeit(lambda: MSDA.ms_deform_attn_forward(value, shp, starts, loc, attn, 128))
        ms_b = timeit(lambda: MSDA.ms_deform_attn_backward(value, shp, starts, loc, attn, g, 128))
        byt = 4 * (N * S * 256 + 3 * N * S * 8 * 12 + N * S * 256)
        print(json.dumps(dict(kernel="msda", N=N, S=S, fwd_ms=round(ms_f, 4), bwd_ms=round(ms_b, 4),
                              fwd_GBs=round(byt / ms_f / 1e6, 1))), flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def encoder_bench():
    """C4: the 6-layer MSDeformAttn encoder of the pixel decoder (msdeformattn.py:21-153), 704^2 crops."""
    from multishiftseg_amd.msdeformattn_encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine
    enc = MSDeformAttnTransformerEncoderOnly(256, 8, 6, 1024, 0.0, "relu", 3, 4).cuda()
    pe = PositionEmbeddingSine(128, normalize=True)
    for N in (1, 16):
        srcs = [torch.randn(N, 256, h, w, device="cuda", requires_grad=True) for h, w in ((22, 22), (44, 44), (88, 88))]
        pos = [pe(s) for s in srcs]
        with torch.no_grad():
            ms_f = timeit(lambda: enc(srcs, pos), iters=5, warm=2)
        def fb():
            m, _, _ = enc(srcs, pos)
            m.sum().backward()
        ms_fb = timeit(fb, iters=5, warm=2)
        print(json.dumps(dict(kernel="msda_encoder_6layers", N=N, fwd_ms=round(ms_f, 3), fwd_bwd_ms=round(ms_fb, 3))), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "encoder":
    encoder_bench()
