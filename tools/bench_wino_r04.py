"""Round-4 transform micro-benchmarks under an A/B of one environment switch (alternating protocol of tools/bench_gemm_variant.py):
the fused three-dilation ASPP input transform, the upsample-in-transform of the decoder's first layer, the F(6x6) output transform at
304 / 256 channels, and the plain input transforms of the step. TB/s of algorithmic bytes; bit-equality of A and B.
    python tools/bench_wino_r04.py MSS_WINO_ASPP3 0 1 [aspp|upcat|out|in ...]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multishiftseg_amd import _lib
from multishiftseg_amd._lib import call, ptr
from tools.microbench import timeit

VAR, VA, VB = sys.argv[1], sys.argv[2], sys.argv[3]
WHICH = set(sys.argv[4:]) or {"aspp", "upcat", "out", "in"}


def ab(name, make, nbytes):
    """make(v) -> (fn, outputs)"""
    fns, outs = {}, {}
    for v in (VA, VB):
        f, o = make()

        def g(v=v, f=f):
            os.environ[VAR] = v
            _lib.reset_env_cache()
            f()
        fns[v], outs[v] = g, o
    timeit(fns[VA], iters=30, warm=10)
    best = {VA: 1e9, VB: 1e9}
    for _ in range(4):
        for v in (VA, VB):
            best[v] = min(best[v], timeit(fns[v], iters=10, warm=2))
    eq = all(torch.equal(a, b) for a, b in zip(outs[VA], outs[VB]))
    print(json.dumps({"case": name, VAR + "=" + VA: [round(best[VA], 4), round(nbytes / best[VA] / 1e9, 2)],
                      VAR + "=" + VB: [round(best[VB], 4), round(nbytes / best[VB] / 1e9, 2)], "gain": round(best[VA] / best[VB], 3), "equal": eq}), flush=True)


if "aspp" in WHICH:
    for (n, h, w, c, d, tiles) in [(2, 128, 256, 4096, 12, (6, 6, 4)), (1, 128, 256, 4096, 12, (6, 6, 4)), (16, 88, 88, 4096, 12, (4, 4, 4))]:
        x = torch.randn(n, h, w, c, device="cuda")
        Ts = [_lib.value("mss_wino_num_tiles", n, h, w, (m + 1) * d, t) for m, t in enumerate(tiles)]
        Ps = [(t + 2) ** 2 for t in tiles]

        def make():
            xts = [torch.empty(p, t, c, device="cuda") for p, t in zip(Ps, Ts)]
            return (lambda: call("mss_wino_input_transform_aspp3_f32", ptr(x), c, n, h, w, c, d, (ctypes.c_int * 3)(*tiles), ptr(xts[0]), ptr(xts[1]), ptr(xts[2]))), xts
        ab(f"aspp3 {n}x{h}x{w}x{c} d{d} {tiles}", make, 4.0 * (n * h * w * c + sum(p * t for p, t in zip(Ps, Ts)) * c))

        def make3():
            xts = [torch.empty(p, t, c, device="cuda") for p, t in zip(Ps, Ts)]

            def f():
                for m in range(3):
                    call("mss_wino_input_transform_f32", ptr(x), c, n, h, w, c, (m + 1) * d, tiles[m], None, None, 0, ptr(xts[m]))
            return f, xts
        ab(f"three separate {n}x{h}x{w}x{c}", make3, 4.0 * (3 * n * h * w * c + sum(p * t for p, t in zip(Ps, Ts)) * c))
        del x
if "upcat" in WHICH:
    n, h, w, ih, iw, ca, cs, ts = 2, 512, 1024, 128, 256, 48, 256, 6
    a = torch.randn(n, h, w, ca, device="cuda")
    small = torch.randn(n, ih, iw, cs, device="cuda")
    T = _lib.value("mss_wino_num_tiles", n, h, w, 1, ts)

    def make():
        xt = torch.empty(64, T, ca + cs, device="cuda")
        return (lambda: call("mss_wino_input_transform_upcat_f32", ptr(a), ca, ca, ptr(small), cs, ih, iw, n, h, w, ca + cs, ts, ptr(xt))), [xt]
    ab("upcat 2x512x1024x(48+256)", make, 4.0 * (n * h * w * ca + n * ih * iw * cs + 64 * T * (ca + cs)))
if "out" in WHICH:
    for (n, h, w, k, ts, res) in [(2, 512, 1024, 304, 6, False), (2, 512, 1024, 256, 6, False), (2, 512, 1024, 128, 4, True), (2, 256, 512, 256, 4, True),
                                  (2, 128, 256, 1024, 6, False)]:
        T = _lib.value("mss_wino_num_tiles", n, h, w, 1, ts)
        P = (ts + 2) ** 2
        yt = torch.randn(P, T, k, device="cuda")
        r = torch.randn(n, h, w, k, device="cuda") if res else None

        ldy = (k + 31) // 32 * 32 if os.environ.get("MSS_BENCH_LDY_ALIGN") == "1" else k       # pixel rows on 128-byte lines

        def make():
            y = torch.zeros(n, h, w, ldy, device="cuda")
            st = torch.empty(65536 * 2 * k // 64 + 2 * k * 4096, device="cuda")
            return (lambda: call("mss_wino_output_transform_f32", ptr(yt), n, h, w, k, 1, ts, ptr(r), k if res else 0, ptr(y), ldy, ptr(st))), [y[..., :k]]
        ab(f"output F{ts} {n}x{h}x{w}x{k}" + (" +res" if res else ""), make, 4.0 * (P * T * k + n * h * w * k * (2 if res else 1)))
if "in" in WHICH:
    for (n, h, w, c, dil, ts) in [(2, 512, 1024, 128, 1, 4), (2, 512, 1024, 256, 1, 6), (2, 256, 512, 256, 1, 4), (2, 128, 256, 512, 1, 4), (2, 128, 256, 512, 2, 6),
                                  (2, 128, 256, 1024, 2, 6), (2, 128, 256, 1024, 4, 6)]:
        x = torch.randn(n, h, w, c, device="cuda")
        sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
        P, T = (ts + 2) ** 2, _lib.value("mss_wino_num_tiles", n, h, w, dil, ts)

        def make():
            xt = torch.empty(P, T, c, device="cuda")
            return (lambda: call("mss_wino_input_transform_f32", ptr(x), c, n, h, w, c, dil, ts, ptr(sc), ptr(sh), 1, ptr(xt))), [xt]
        ab(f"input F{ts} {n}x{h}x{w}x{c} d{dil}", make, 4.0 * (n * h * w * c + P * T * c))
