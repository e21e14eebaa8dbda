"""Thin tensor-level wrappers over the C ABI (include/mss_hip.h).

torch is used for device memory and streams only; every computation below is a HIP kernel in
libmss_hip.so. An activation is an `Act`: a channel slice [c0, c0+C) of a contiguous NHWC buffer,
so concats are written in place and never copied.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import MssConvArgs, call, ptr


class Act:
    """Channel slice of an NHWC fp32 buffer [N,H,W,ld]. `stats`: per-channel partial sums [parts,2,C] left by the
    kernel that produced this slice (conv epilogue / Winograd output transform), consumed by bn_fold(train=True)."""
    __slots__ = ("buf", "N", "H", "W", "C", "ld", "c0", "stats")

    def __init__(self, buf, C=None, c0=0):
        assert buf.dim() == 4 and buf.is_contiguous() and buf.dtype == torch.float32
        self.buf = buf
        self.stats = None
        self.N, self.H, self.W, self.ld = buf.shape
        self.C = self.ld - c0 if C is None else C
        self.c0 = c0
        assert self.c0 % 4 == 0 and self.ld % 4 == 0 and self.c0 + self.C <= self.ld

    @staticmethod
    def empty(N, H, W, C, device, ld=None):
        return Act(torch.empty((N, H, W, ld or C), device=device, dtype=torch.float32), C)

    @staticmethod
    def zeros(N, H, W, C, device, ld=None):
        return Act(torch.zeros((N, H, W, ld or C), device=device, dtype=torch.float32), C)

    def slice(self, c0, C):
        return Act(self.buf, C, self.c0 + c0)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + 4 * self.c0)

    @property
    def M(self):
        return self.N * self.H * self.W

    def nchw(self):
        """Materialise as an NCHW torch tensor (tests / debugging only)."""
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def from_nchw(t, ld=None):
        n, c, h, w = t.shape
        a = Act.zeros(n, h, w, c, t.device, ld)
        a.buf[..., :c] = t.permute(0, 2, 3, 1)
        return a


class PackedWeight:
    """[R*S][Kpad][Cp] layout consumed by the MFMA kernels. When the output-channel count leaves a
    narrow last 128-wide tile (e.g. 304 = 2*128 + 48, the data gradient of final.0), `tail` holds
    those channels as a separate pack that runs on the 64-wide tile instead of wasting 2/3 of a tile."""
    __slots__ = ("t", "K", "C", "R", "S", "Kpad", "Cp", "tail", "planes")

    def __init__(self, t, K, C, R, S, Kpad, Cp, tail=None):
        self.t, self.K, self.C, self.R, self.S, self.Kpad, self.Cp, self.tail = t, K, C, R, S, Kpad, Cp, tail
        self.planes = None         # split-bf16 form, made on first use under that route (w_split_of)


def _round_up(v, m):
    return (v + m - 1) // m * m


# ---- the split-bf16 GEMM route (csrc/gemm_bf16x3.hip) ---------------------------------------------------------------------------
# MSS_GEMM_SPLIT=1 (or set_gemm_route("bf16x3")): every product the persistent GEMM kernel takes runs as six bf16 MFMAs on operands
# split into three bf16 terms (fp32 accumulate, fp32 accuracy); the weights' planes are made once per packed weight.
_route_override = None


def set_gemm_route(route):
    """"native" (fp32 MFMA), "bf16x3" (split-bf16) or None (back to the MSS_GEMM_SPLIT environment switch)."""
    global _route_override
    assert route in (None, "native", "bf16x3")
    _route_override = route


def gemm_route():
    if _route_override is not None:
        return _route_override
    return "bf16x3" if os.environ.get("MSS_GEMM_SPLIT", "0") == "1" else "native"


def split_planes(t, Kpad, C):
    """The three-bf16-plane form (MssConvArgs.w_split) of packed fp32 weights t [batch][Kpad][C]; None where the route does not
    apply (Kpad not a multiple of 128: the <= 64-channel packs run on the native narrow tile)."""
    batch = t.numel() // (Kpad * C)
    nbytes = _lib.value("mss_gemm_split_weights_bytes", batch, Kpad, C)
    if nbytes <= 0:
        return None
    planes = torch.empty(nbytes, device=t.device, dtype=torch.uint8)
    call("mss_gemm_split_weights_bf16x3", ptr(t), ptr(planes), batch, Kpad, C, Kpad * C)
    return planes


def w_split_of(pw):
    """Pointer for MssConvArgs.w_split of a PackedWeight / WinoWeight under the current route (None: native). The planes are cached
    on the pack object, which is itself re-made whenever the parameter changes (_cached_pack)."""
    if gemm_route() != "bf16x3":
        return None
    if pw.planes is None:
        taps = getattr(pw, "R", 1) * getattr(pw, "S", 1)
        if taps != 1 and isinstance(pw, PackedWeight):
            # an implicit-GEMM layer: the taps folded into one long reduction (mss_conv_split_weights_bf16x3)
            nbytes = _lib.value("mss_gemm_split_weights_bytes", taps, pw.Kpad, pw.Cp)
            if nbytes > 0:
                pw.planes = torch.empty(nbytes, device=pw.t.device, dtype=torch.uint8)
                call("mss_conv_split_weights_bf16x3", ptr(pw.t), ptr(pw.planes), taps, pw.Kpad, pw.Cp)
        elif isinstance(pw, WinoWeight) and pw._t is None:
            pw.planes = pw.fused_planes()
            if pw.planes is None:
                pw.planes = split_planes(pw.t, pw.Kpad, pw.Cp)
        else:
            pw.planes = split_planes(pw.t, pw.Kpad, pw.Cp)
        if pw.planes is None:
            pw.planes = False
    return ptr(pw.planes) if pw.planes is not False else None


def pack_weight(w, flip=False, min_c=16):
    """nn.Conv2d.weight [K,C,R,S] -> PackedWeight. flip=True packs the data-gradient filter."""
    K, C, R, S = w.shape
    if w.dtype != torch.float32 or not w.is_cuda:
        raise TypeError(f"pack_weight needs a float32 CUDA tensor, got {w.dtype} on {w.device}")
    w = w.detach().contiguous()
    if not flip:
        k_out, c_in = K, C
    else:
        k_out, c_in = C, K
    rem = k_out % 128
    if k_out > 128 and 0 < rem <= 64:
        split = k_out - rem
        if not flip:
            main, tail = pack_weight(w[:split], flip, min_c), pack_weight(w[split:], flip, min_c)
        else:
            main, tail = pack_weight(w[:, :split], flip, min_c), pack_weight(w[:, split:], flip, min_c)
        main.tail = tail
        return main
    Kpad = _lib.value("mss_conv2d_kpad", k_out)
    Cp = _round_up(max(c_in, min_c), 16)
    t = torch.empty((R * S, Kpad, Cp), device=w.device, dtype=torch.float32)
    call("mss_conv2d_pack_weights_f32", ptr(w), ptr(t), K, C, R, S, Kpad, Cp, 1 if flip else 0)
    return PackedWeight(t, k_out, c_in, R, S, Kpad, Cp)


def _cached_pack(param, key, make):
    """Packed forms live on the parameter object itself and die with it. (A module-level dict keyed by
    id(param) can hand a new parameter the packed weights of a dead one when Python recycles the id and the
    allocator recycles the storage address -- same-shaped layers of two successive models.) Re-packed when the
    parameter is modified in place (optimizer step, load_state_dict: tensor._version) or moved (data_ptr)."""
    cache = param.__dict__.setdefault("_mss_packed", {})
    ent = cache.get(key)
    if ent is not None and ent[0] == param._version and ent[1] == param.data_ptr():
        return ent[2]
    val = make()
    cache[key] = (param._version, param.data_ptr(), val)
    return val


_ONES = {}


def ones(n, device):
    """A shared all-ones fp32 vector (the unit scale of a bias-only epilogue): one tensor per (length, device) for the life of the
    process instead of a fill kernel per Linear call."""
    key = (int(n), str(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(int(n), device=device, dtype=torch.float32)
    return t


def packed(param, flip=False):
    """Cached pack_weight of an nn.Parameter."""
    return _cached_pack(param, ("direct", flip), lambda: pack_weight(param, flip))


class ConvProfile:
    """Optional live timing of every MFMA conv launch (bench.py's roofline leg): a pair of HIP events
    on the launch stream around each launch plus the launch's algorithmic FLOPs (dense 2*MAC)."""

    def __init__(self):
        self.records = []   # (kind, flops, start_event, stop_event)
        self.tags = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, flops, s, e in self.records:
            ms = s.elapsed_time(e)
            d = out.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += ms
        return out

    def per_launch(self):
        """[(kind, tag, ms, TFLOP/s)] in launch order (debugging aid: tools/layer_table.py)."""
        torch.cuda.synchronize()
        rows = []
        for (kind, flops, s, e), tag in zip(self.records, self.tags):
            ms = s.elapsed_time(e)
            rows.append((kind, tag, ms, flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))
        return rows


_profile = None


def set_conv_profile(prof):
    global _profile
    _profile = prof


class _Timed:
    def __init__(self, kind, flops, tag=None):
        self.kind, self.flops, self.tag = kind, flops, tag

    def __enter__(self):
        if _profile is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if _profile is not None:
            self.e.record()
            _profile.records.append((self.kind, self.flops, self.s, self.e))
            _profile.tags.append(self.tag)
        return False


def _fwd_kind(a):
    """Profiling label of a forward launch: which of the two MFMA kernels the C side picks."""
    if _profile is None:
        return "conv_igemm"
    return ("conv_igemm", "gemm_nt", "gemm_few_rows", "gemm_nt_bf16x3", "conv_igemm_bf16x3")[_lib.value("mss_conv2d_forward_route", ctypes.byref(a))]


def _wgrad_kind(a, lddy):
    """Profiling label of a weight-gradient launch: the split-bf16 TN kernel or one of the native fp32 MFMA kernels."""
    if _profile is None or not a.route:
        return "conv_wgrad"
    return "conv_wgrad_bf16x3" if _lib.value("mss_conv2d_wgrad_route", ctypes.byref(a), lddy) else "conv_wgrad"


def conv_out_size(h, r, stride, dil, pad):
    return (h + 2 * pad - dil * (r - 1) - 1) // stride + 1


def _conv_args(x, pw, y, stride, dil, pad, in_affine, in_relu, out_affine, out_relu, res, res_mask=False):
    a = MssConvArgs()
    a.x, a.w, a.y = x.ptr, ptr(pw.t), (y.ptr if y is not None else None)
    a.w_split = w_split_of(pw)
    if in_affine is not None:
        sc, sh = in_affine
        a.in_scale, a.in_shift = ptr(sc), ptr(sh)
        a.in_ss_stride = sc.shape[1] if sc.dim() == 2 else 0
        assert sc.shape[-1] >= pw.Cp and sc.is_contiguous() and sh.is_contiguous()
    if out_affine is not None:
        a.out_scale, a.out_shift = ptr(out_affine[0]), ptr(out_affine[1])
    if res is not None:
        a.res, a.ldres = res.ptr, res.ld
        a.res_mask = int(res_mask)
    a.N, a.H, a.W, a.C, a.ldx = x.N, x.H, x.W, pw.Cp, x.ld
    a.R, a.S, a.stride, a.dil, a.pad = pw.R, pw.S, stride, dil, pad
    a.K, a.Kpad = pw.K, pw.Kpad
    a.in_relu, a.out_relu = int(in_relu), int(out_relu)
    return a


def conv2d(x, pw, stride=1, dil=1, pad=0, in_affine=None, in_relu=False, out_affine=None, out_relu=False, res=None,
           out=None, want_stats=False, res_mask=False):
    """y = epilogue(conv(prologue(x))). x: Act with C == pw.Cp channels visible. want_stats: the epilogue also leaves
    the per-channel partial sums of y on the returned Act (for the next layer's train-mode BatchNorm). res_mask: `res`
    gates the output (y = res > 0 ? y : 0, a ReLU backward) instead of being added."""
    assert x.C == pw.Cp or (x.C >= pw.C and x.C <= pw.Cp and x.c0 + pw.Cp <= x.ld), (x.C, pw.C, pw.Cp)
    OH = conv_out_size(x.H, pw.R, stride, dil, pad)
    OW = conv_out_size(x.W, pw.S, stride, dil, pad)
    k_total = pw.K + (pw.tail.K if pw.tail is not None else 0)
    if out is None:
        out = Act.empty(x.N, OH, OW, k_total, x.buf.device, ld=_round_up(k_total, 4))
    assert (out.N, out.H, out.W) == (x.N, OH, OW) and out.C >= k_total
    if pw.tail is not None:
        def sub(t, lo, n):
            return None if t is None else t[lo:lo + n]
        conv2d(x, pw.tail, stride, dil, pad, in_affine, in_relu,
               None if out_affine is None else (sub(out_affine[0], pw.K, pw.tail.K), sub(out_affine[1], pw.K, pw.tail.K)),
               out_relu, None if res is None else res.slice(pw.K, pw.tail.K), out=out.slice(pw.K, pw.tail.K), res_mask=res_mask)
        out_affine = None if out_affine is None else (sub(out_affine[0], 0, pw.K), sub(out_affine[1], 0, pw.K))
        res = None if res is None else res.slice(0, pw.K)
        out_main = out.slice(0, pw.K)
        _conv_launch(x, pw, out_main, OH, OW, stride, dil, pad, in_affine, in_relu, out_affine, out_relu, res, res_mask=res_mask)
        return out
    _conv_launch(x, pw, out, OH, OW, stride, dil, pad, in_affine, in_relu, out_affine, out_relu, res, want_stats, res_mask)
    return out


def _conv_launch(x, pw, out, OH, OW, stride, dil, pad, in_affine, in_relu, out_affine, out_relu, res, want_stats=False, res_mask=False):
    a = _conv_args(x, pw, out, stride, dil, pad, in_affine, in_relu, out_affine, out_relu, res, res_mask)
    a.OH, a.OW, a.ldy = OH, OW, out.ld
    out.stats = None
    if want_stats and out.C == pw.K:
        out.stats = torch.empty((-(-(x.N * OH * OW) // 64), 2, pw.K), device=x.buf.device, dtype=torch.float32)
        a.stats = ptr(out.stats)
    with _Timed(_fwd_kind(a), 2.0 * x.N * OH * OW * pw.K * pw.C * pw.R * pw.S,
                (x.N, x.H, x.W, pw.C, pw.K, pw.R, stride, dil)):
        call("mss_conv2d_forward_f32", ctypes.byref(a))


def _wgrad_workspace(a, Cp, device):
    """Scratch for the partial slabs of a pixel-split weight gradient (deterministic two-stage sum)."""
    nbytes = _lib.value("mss_conv2d_wgrad_workspace_bytes", ctypes.byref(a), Cp)
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32), nbytes


def conv2d_wgrad(x, dy, K, C, R, S, stride=1, dil=1, pad=0, in_affine=None, in_relu=False):
    """Weight gradient [K,C,R,S] of y = conv(prologue(x)); dy: Act with K channels."""
    Kpad = _round_up(K, 4)
    Cp = _round_up(C, 4)
    dwp = torch.empty((R * S, Kpad, Cp), device=x.buf.device, dtype=torch.float32)   # fully overwritten by the kernel
    a = MssConvArgs()
    a.x = x.ptr
    if in_affine is not None:
        sc, sh = in_affine
        a.in_scale, a.in_shift = ptr(sc), ptr(sh)
        a.in_ss_stride = sc.shape[1] if sc.dim() == 2 else 0
    a.N, a.H, a.W, a.C, a.ldx = x.N, x.H, x.W, C, x.ld
    a.OH, a.OW, a.K, a.Kpad = dy.H, dy.W, K, Kpad
    a.R, a.S, a.stride, a.dil, a.pad = R, S, stride, dil, pad
    a.in_relu = int(in_relu)
    a.route = 1 if gemm_route() == "bf16x3" else 0
    ws, ws_bytes = _wgrad_workspace(a, Cp, x.buf.device)
    with _Timed(_wgrad_kind(a, dy.ld), 2.0 * x.N * dy.H * dy.W * K * C * R * S, (x.N, x.H, x.W, C, K, R, stride, dil)):
        call("mss_conv2d_wgrad_f32", ctypes.byref(a), dy.ptr, dy.ld, ptr(dwp), Cp, ptr(ws), ws_bytes)
    grad = torch.empty((K, C, R, S), device=x.buf.device, dtype=torch.float32)
    call("mss_conv2d_unpack_wgrad_f32", ptr(dwp), ptr(grad), K, C, R, S, Kpad, Cp, 0)
    return grad


class WinoWeight:
    """Winograd-domain filter U [(tile+2)^2][Kpad][Cp] of a 3x3 weight. `src` (optional): the contiguous [K][C][3][3] weight it comes
    from; with it `t` may start as None and is made on first use -- the split-bf16 route fills its planes straight from `src`
    (w_split_of -> mss_wino_pack_split_bf16x3) and never reads U in fp32 unless the layer has a narrow native tail."""
    __slots__ = ("_t", "src", "K", "C", "Kpad", "Cp", "tile", "planes", "_src_version")

    def __init__(self, t, K, C, Kpad, Cp, tile, src=None):
        assert t is not None or src is not None
        self._t, self.src, self.K, self.C, self.Kpad, self.Cp, self.tile = t, src, K, C, Kpad, Cp, tile
        self.planes = None
        # `src` may alias the live parameter (detach().contiguous() copies nothing): a lazily made U / planes must come from the SAME
        # weight version as whatever was made first, or one convolution would mix two versions. detach() shares the
        # version counter, so an in-place update of the parameter (optimizer.step) shows here.
        self._src_version = None if src is None else src._version

    def _check_fresh(self):
        if self.src is not None and self.src._version != self._src_version:
            raise RuntimeError("stale WinoWeight: the weight it was packed from has been modified in place since (version "
                               f"{self._src_version} -> {self.src._version}); re-pack it (kernels.packed_wino does, through tensor._version)")

    @property
    def t(self):
        if self._t is None:
            self._check_fresh()
            t = torch.empty(((self.tile + 2) ** 2, self.Kpad, self.Cp), device=self.src.device, dtype=torch.float32)
            call("mss_wino_pack_weights_f32", ptr(self.src), ptr(t), self.K, self.C, self.Kpad, self.Cp, self.tile)
            self._t = t
        return self._t

    def fused_planes(self):
        """The split-bf16 planes of U straight from `src` (bit-identical to split_planes(self.t, ...)), or None where that form does
        not apply (no source weight, a padded Cp, Kpad not a multiple of 128)."""
        if self.src is None or self.Cp != self.C or self.Kpad % 128 or self.C % 16 or self.src.data_ptr() % 16:
            return None
        self._check_fresh()
        nbytes = _lib.value("mss_gemm_split_weights_bytes", (self.tile + 2) ** 2, self.Kpad, self.C)
        if nbytes <= 0:
            return None
        planes = torch.empty(nbytes, device=self.src.device, dtype=torch.uint8)
        call("mss_wino_pack_split_bf16x3", ptr(self.src), ptr(planes), self.K, self.C, self.Kpad, self.tile)
        return planes


def pack_weight_wino(w, flip=False, tile=2):
    """flip=True: the data-gradient filter (K<->C swapped, taps rotated 180 degrees), as pack_weight."""
    if flip:
        w = w.detach().flip(2, 3).transpose(0, 1)
    K, C, R, S = w.shape
    assert R == 3 and S == 3 and C % 16 == 0 and K % 4 == 0 and tile in (2, 4, 6)
    if w.dtype != torch.float32 or not w.is_cuda:
        raise TypeError("pack_weight_wino needs a float32 CUDA tensor")
    w = w.detach().contiguous()
    Kpad = _lib.value("mss_conv2d_kpad", K)
    if gemm_route() == "bf16x3" and Kpad % 128 == 0:
        return WinoWeight(None, K, C, Kpad, C, tile, src=w)          # U in fp32 only if somebody asks for it (WinoWeight.t)
    t = torch.empty(((tile + 2) ** 2, Kpad, C), device=w.device, dtype=torch.float32)
    call("mss_wino_pack_weights_f32", ptr(w), ptr(t), K, C, Kpad, C, tile)
    return WinoWeight(t, K, C, Kpad, C, tile)


def packed_wino(param, flip=False, tile=2):
    return _cached_pack(param, ("wino", flip, tile), lambda: pack_weight_wino(param, flip, tile))


# measured on MI355X (tools/bench_wino.py): F(2x2) 0.87x at 128 channels, 1.24-1.28x at 256, 1.7-2.1x at >= 512;
# F(4x4) 1.29x at 128 channels, 2.0x at 256, 2.4-3.4x at >= 512; F(6x6): 64 -> 128 at 512x1024 1.50 -> 0.8 ms
WINOGRAD_MIN_CHANNELS = {2: 256, 4: 128, 6: 64}
WINO_TILE_OF_P = {16: 2, 36: 4, 64: 6}


def use_winograd(c_in, k_out, stride, in_affine=None, tile=2):
    """Policy: 3x3 stride-1 layers whose channel counts make the two HBM-bound transforms cheaper than the
    MFMA work they remove. MSS_WINOGRAD=0 forces the direct implicit GEMM everywhere."""
    if os.environ.get("MSS_WINOGRAD", "1") == "0" or stride != 1:
        return False
    if in_affine is not None and in_affine[0].dim() != 1:
        return False
    lo = WINOGRAD_MIN_CHANNELS[tile]
    return c_in >= lo and k_out >= lo and c_in % 16 == 0 and k_out % 4 == 0


def wino_tile(H, W, dil, max_tile=6):
    """Output-tile edge m of F(m x m, 3x3) for a layer: the one with fewer Winograd-domain elements
    (tiles x (m+2)^2), which decides both the MFMA work and the transform traffic: 1.78 per output pixel for m = 6,
    2.25 for m = 4, 4 for m = 2, unless the dilation sub-grid is so small that the larger tiles are mostly padding
    (e.g. a 12x16 map at dilation 12; the 4x8 sub-grids of dilation 36 at 128x256 stay on 4x4 tiles, the 6x11 ones of
    dilation 24 go from six 4x4 tiles to two 6x6 tiles). m = 6 carries 3x the fp32 rounding error of m = 4 (5.8e-6 vs
    1.9e-6 of the output per layer, tools/wino_matrices.py), so it must save at least 5 % to be chosen, and the caller
    caps it (`max_tile`) on the layers whose error the rest of the network amplifies most (deepv3.wino_cap).
    MSS_WINO_TILE=2|4|6 forces one, MSS_WINO_MAX_TILE=4 keeps the policy off 6x6 tiles everywhere."""
    forced = os.environ.get("MSS_WINO_TILE")
    if forced:
        return int(forced)
    hs, ws = -(-H // dil), -(-W // dil)
    cost = {m: (-(-hs // m)) * (-(-ws // m)) * (m + 2) ** 2 for m in (2, 4, 6)}
    best = 4 if cost[4] < cost[2] else 2
    if min(max_tile, int(os.environ.get("MSS_WINO_MAX_TILE", "6"))) >= 6 and cost[6] <= 0.95 * cost[best]:
        best = 6
    return best


def wino_xt_bytes(N, H, W, C, dil):
    """Size of the Winograd-domain input X' of one layer under the tile policy."""
    ts = wino_tile(H, W, dil)
    return (ts + 2) ** 2 * _lib.value("mss_wino_num_tiles", N, H, W, dil, ts) * C * 4


_tile_hook = None


def set_tile_hook(fn):
    """Attribution / test hook (tools/attribute_wino_error.py): fn(info) is asked once per forward 3x3 layer with
    info = dict(H, W, dil, c_in, k_out, stride, policy_tile) and answers None (keep the policy), 0 (direct implicit
    GEMM) or a tile edge 2 / 4 / 6. Not consulted for data or weight gradients."""
    global _tile_hook
    _tile_hook = fn


def conv3x3(x, weight, dil=1, stride=1, in_affine=None, in_relu=False, res=None, out=None, flip=False, keep_xt=None,
            want_stats=False, max_tile=6, xt=None):
    """3x3 convolution with padding = dilation on nn.Conv2d-layout `weight` (flip=True: its data gradient),
    through Winograd when the policy says so, else through the direct implicit GEMM. max_tile: accuracy cap on the
    Winograd tile edge for this layer (0: no Winograd at all). xt: the layer's Winograd-domain input X' when the caller has
    it already (aspp_input_transforms); its tile edge decides the route."""
    k_out, c_in = (weight.shape[1], weight.shape[0]) if flip else (weight.shape[0], weight.shape[1])
    if xt is not None:
        assert in_affine is None and not in_relu and not flip and stride == 1
        return conv2d_winograd(x, packed_wino(weight, False, WINO_TILE_OF_P[xt.shape[0]]), dil=dil, res=res, out=out,
                               keep_xt=keep_xt, want_stats=want_stats, xt=xt)
    tile = wino_tile(x.H, x.W, dil, max_tile) if max_tile else 0
    if _tile_hook is not None and not flip:
        forced = _tile_hook(dict(H=x.H, W=x.W, dil=dil, c_in=c_in, k_out=k_out, stride=stride, policy_tile=tile))
        if forced == 0:
            return conv2d(x, packed(weight, flip), stride=stride, dil=dil, pad=dil, in_affine=in_affine, in_relu=in_relu,
                          res=res, out=out, want_stats=want_stats)
        if forced:
            tile = forced
    if tile and use_winograd(c_in, k_out, stride, in_affine, tile):
        return conv2d_winograd(x, packed_wino(weight, flip, tile), dil=dil, in_affine=in_affine,
                               in_relu=in_relu, res=res, out=out, keep_xt=keep_xt, want_stats=want_stats)
    return conv2d(x, packed(weight, flip), stride=stride, dil=dil, pad=dil, in_affine=in_affine, in_relu=in_relu, res=res,
                  out=out, want_stats=want_stats)


def conv2d_winograd(x, ww, dil=1, in_affine=None, in_relu=False, res=None, out=None, keep_xt=None, want_stats=False, xt=None):
    """3x3 / stride 1 / padding = dilation convolution through Winograd F(m x m,3x3), m = ww.tile: input
    transform (with the fused BatchNorm+ReLU prologue) -> (m+2)^2 batched MFMA GEMMs -> output transform
    (+ residual). keep_xt: a dict; the transformed input X' is stored under keep_xt["xt"] so that the weight
    gradient of the same layer (conv2d_wgrad_winograd(..., xt=...)) does not have to transform x again."""
    assert x.C == ww.C
    N, H, W, C, K, ts = x.N, x.H, x.W, ww.C, ww.K, ww.tile
    P = (ts + 2) ** 2
    dev = x.buf.device
    T = _lib.value("mss_wino_num_tiles", N, H, W, dil, ts)
    with _Timed("conv_winograd", 2.0 * N * H * W * K * C * 9, (N, H, W, C, K, 3, 1, dil)):
        if xt is None:
            xt = torch.empty((P, T, C), device=dev, dtype=torch.float32)
            sc, sh = in_affine if in_affine is not None else (None, None)
            assert sc is None or sc.dim() == 1
            with _Timed("wino_transform", 4.0 * (N * H * W * C + P * T * C), ("input", N, H, W, C, dil, ts)):   # "flops" = algorithmic BYTES
                call("mss_wino_input_transform_f32", x.ptr, x.ld, N, H, W, C, dil, ts, ptr(sc), ptr(sh), int(in_relu), ptr(xt))
        else:
            assert tuple(xt.shape) == (P, T, C) and in_affine is None and not in_relu
        return _wino_gemm_and_output(xt, ww, N, H, W, dil, res, out, keep_xt, want_stats)


def _set_wino_w(a, ww):
    """MssConvArgs.w of a Winograd-domain product. On the split-bf16 route the kernel reads only the planes (a.w_split): when U has not
    been made in fp32 (WinoWeight._t is None) and the library confirms that exactly these arguments take the split GEMM, U stays
    unmade and `w` carries the planes' address as a non-null stand-in; in every other case it is U (made on demand)."""
    if ww._t is None and a.w_split is not None:
        a.w = a.w_split
        if _lib.value("mss_conv2d_forward_route", ctypes.byref(a)) == 3:
            return
    a.w = ptr(ww.t)


def _wino_gemm_and_output(xt, ww, N, H, W, dil, res, out, keep_xt, want_stats):
    """Steps 2 and 3 of the Winograd convolution on a transformed input X' [P][T][C]: the batched MFMA products and the output
    transform (+ residual, + BatchNorm partial sums)."""
    C, K, ts = ww.C, ww.K, ww.tile
    P, T = xt.shape[0], xt.shape[1]
    dev = xt.device
    if out is None:
        # pixel rows on 128-byte lines: a 304-channel map (the decoder's first data gradient) with ld = 304 shares every other
        # cache line between two pixels that different waves write (0.99 -> 0.94 ms at 2 x 512 x 1024, tools/bench_wino_r04.py)
        out = Act.empty(N, H, W, K, dev, ld=_round_up(K, 32) if K % 32 else None)
    yt = torch.empty((P, T, K), device=dev, dtype=torch.float32)
    a = MssConvArgs()
    a.x, a.y = ptr(xt), ptr(yt)
    a.w_split = w_split_of(ww)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
    a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, K, ww.Kpad, K
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    a.batch, a.x_bs, a.w_bs, a.y_bs = P, T * C, ww.Kpad * ww.Cp, T * K
    rem = K % 128
    split = K - rem if (K > 128 and 0 < rem <= 64) else 0     # e.g. 304 = 256 + 48: narrow tail on the 64-wide tile
    if split:
        a.K = split
    _set_wino_w(a, ww)
    with _Timed(_fwd_kind(a), 2.0 * P * T * C * K, (P, 1, T, C, K, 1, 1, 1)):   # the MFMA work actually executed
        call("mss_conv2d_forward_f32", ctypes.byref(a))
        if split:
            a.K, a.Kpad = rem, ww.Kpad - split
            a.w_split = None                                  # the <= 64-channel tail runs on the native narrow tile
            a.w = ctypes.c_void_p(ww.t.data_ptr() + 4 * split * ww.Cp)
            a.y = ctypes.c_void_p(yt.data_ptr() + 4 * split)
            call("mss_conv2d_forward_f32", ctypes.byref(a))
    if keep_xt is not None:
        keep_xt["xt"] = xt
    del xt
    out.stats = None
    if want_stats and out.C == K:
        out.stats = torch.empty((_lib.value("mss_wino_output_stats_parts", N, H, W, K, dil, ts), 2, K), device=dev,
                                dtype=torch.float32)
    with _Timed("wino_transform", 4.0 * (P * T * K + N * H * W * K * (2 if res is not None else 1)), ("output", N, H, W, K, dil, ts)):
        call("mss_wino_output_transform_f32", ptr(yt), N, H, W, K, dil, ts, res.ptr if res is not None else None,
             res.ld if res is not None else 0, out.ptr, out.ld, ptr(out.stats))
    return out


def packed_wino_pair(p1, p2, tile):
    """The Winograd-domain forms of two same-shaped 3x3 weights in ONE buffer [2P][Kpad][C] (P = (tile+2)^2), for
    conv3x3_pair. Cached on p1 until either parameter changes; the halves are also installed as the two parameters' own
    packed forms (views of the buffer), so the single-layer path does not keep a second copy."""
    key = (p1._version, p1.data_ptr(), p2._version, p2.data_ptr())
    cache = p1.__dict__.setdefault("_mss_packed", {})
    ent = cache.get(("wino_pair", tile))
    if ent is not None and ent[0] == key:
        return ent[1]
    assert p1.shape == p2.shape and p1.shape[2:] == (3, 3)
    K, C = p1.shape[0], p1.shape[1]
    P = (tile + 2) ** 2
    Kpad = _lib.value("mss_conv2d_kpad", K)
    t = torch.empty((2 * P, Kpad, C), device=p1.device, dtype=torch.float32)
    for j, p in enumerate((p1, p2)):
        w = p.detach().contiguous()
        call("mss_wino_pack_weights_f32", ptr(w), ptr(t[j * P:(j + 1) * P]), K, C, Kpad, C, tile)
        p.__dict__.setdefault("_mss_packed", {})[("wino", False, tile)] = (p._version, p.data_ptr(), WinoWeight(t[j * P:(j + 1) * P], K, C, Kpad, C, tile))
    ww = WinoWeight(t, K, C, Kpad, C, tile)
    cache[("wino_pair", tile)] = (key, ww)
    return ww


def conv3x3_pair_tile(x, w1, w2, dil1, dil2):
    """Tile edge when two 3x3 / stride-1 layers on the SAME input (ASPP's dilated branches) can share one batched GEMM launch
    and it pays, else 0: same Winograd tile and tile count, K a multiple of 128, and the joint launch fills its rounds of
    workgroup slots better than each alone -- the one-image eval forward: 64 x 1152 x 4096 -> 256 is 1152 narrow tiles = 1.5
    rounds of 768 slots (106 TFLOP/s), the pair exactly 3 (MSS_WINO_PAIR=0 switches it off)."""
    if os.environ.get("MSS_WINO_PAIR", "1") == "0" or _tile_hook is not None or w1.shape != w2.shape:
        return 0
    k, c = w1.shape[0], w1.shape[1]
    t1, t2 = wino_tile(x.H, x.W, dil1), wino_tile(x.H, x.W, dil2)
    if t1 != t2 or not t1 or not use_winograd(c, k, 1, None, t1) or k % 128:
        return 0
    T1 = _lib.value("mss_wino_num_tiles", x.N, x.H, x.W, dil1, t1)
    if T1 != _lib.value("mss_wino_num_tiles", x.N, x.H, x.W, dil2, t1):
        return 0
    P = (t1 + 2) ** 2

    def eff(positions):                 # the better of the wide (k / 256 column tiles, 512 slots) and narrow (k / 128, 768) launches
        best = 0.0
        for bn, slots in ((256, 512), (128, 768)):
            if k % bn == 0:
                tiles = positions * -(-T1 // 128) * (k // bn)
                best = max(best, tiles / (-(-tiles // slots) * slots))
        return best
    return t1 if eff(2 * P) > eff(P) + 0.1 else 0


def conv3x3_pair(x, w1, w2, dil1, dil2, out1, out2, tile, want_stats=False, xt=None):
    """conv3x3(x, w1, dil1) -> out1 and conv3x3(x, w2, dil2) -> out2 with the 2 x (tile+2)^2 Winograd-domain products in ONE
    gemm_nt launch (see conv3x3_pair_tile). Same transforms, same products, same results as the two separate calls."""
    ww = packed_wino_pair(w1, w2, tile)
    N, H, W, C, Ko = x.N, x.H, x.W, ww.C, ww.K
    P = (tile + 2) ** 2
    dev = x.buf.device
    T = _lib.value("mss_wino_num_tiles", N, H, W, dil1, tile)
    with _Timed("conv_winograd", 2 * 2.0 * N * H * W * Ko * C * 9, (N, H, W, C, 2 * Ko, 3, 1, (dil1, dil2))):
        if xt is None:
            xt = torch.empty((2 * P, T, C), device=dev, dtype=torch.float32)
            for j, d in enumerate((dil1, dil2)):
                with _Timed("wino_transform", 4.0 * (N * H * W * C + P * T * C), ("input", N, H, W, C, d, tile)):
                    call("mss_wino_input_transform_f32", x.ptr, x.ld, N, H, W, C, d, tile, None, None, 0, ptr(xt[j * P:(j + 1) * P]))
        else:
            assert tuple(xt.shape) == (2 * P, T, C)
        yt = torch.empty((2 * P, T, Ko), device=dev, dtype=torch.float32)
        a = MssConvArgs()
        a.x, a.y = ptr(xt), ptr(yt)
        a.w_split = w_split_of(ww)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, T, Ko, ww.Kpad, Ko
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.w_bs, a.y_bs = 2 * P, T * C, ww.Kpad * ww.Cp, T * Ko
        _set_wino_w(a, ww)
        with _Timed(_fwd_kind(a), 2.0 * 2 * P * T * C * Ko, (2 * P, 1, T, C, Ko, 1, 1, 1)):
            call("mss_conv2d_forward_f32", ctypes.byref(a))
        del xt
        for j, (d, out) in enumerate(((dil1, out1), (dil2, out2))):
            out.stats = None
            if want_stats and out.C == Ko:
                out.stats = torch.empty((_lib.value("mss_wino_output_stats_parts", N, H, W, Ko, d, tile), 2, Ko), device=dev,
                                        dtype=torch.float32)
            with _Timed("wino_transform", 4.0 * (P * T * Ko + N * H * W * Ko), ("output", N, H, W, Ko, d, tile)):
                call("mss_wino_output_transform_f32", ptr(yt[j * P:(j + 1) * P]), N, H, W, Ko, d, tile, None, 0, out.ptr, out.ld,
                     ptr(out.stats))
    return out1, out2


def aspp_input_transforms(x, rates, weights, pair_tile=0, max_bytes=None):
    """X' of the ONE map `x` for the three dilated 3x3 branches of ASPP (deepv3.py:84-92; rates d, 2d, 3d) from a single read of x
    (mss_wino_input_transform_aspp3_f32): bit-identical to the three separate input transforms, 1 x instead of 3 x 1.07 GB read at
    2 x 128 x 256 x 4096. Returns [xt_d, xt_2d, xt_3d] -- with pair_tile (conv3x3_pair_tile) the first two are the halves of ONE
    [2P, T, C] buffer, returned as its first element with None second -- or None when the policy / shapes do not take the fused
    kernel (a tile edge of 2, Winograd off, a tile hook, sub-grids too large for LDS; MSS_WINO_ASPP3=0 switches it off)."""
    d = rates[0]
    if os.environ.get("MSS_WINO_ASPP3", "1") == "0" or _tile_hook is not None or tuple(rates) != (d, 2 * d, 3 * d):
        return None
    tiles = [wino_tile(x.H, x.W, r) for r in rates]
    if pair_tile and (tiles[0] != pair_tile or tiles[1] != pair_tile):
        return None
    for w, t in zip(weights, tiles):
        if t not in (4, 6) or w.shape[1] != x.C or not use_winograd(w.shape[1], w.shape[0], 1, None, t):
            return None
    N, H, W, C = x.N, x.H, x.W, x.C
    dev = x.buf.device
    Ts = [_lib.value("mss_wino_num_tiles", N, H, W, r, t) for r, t in zip(rates, tiles)]
    Ps = [(t + 2) ** 2 for t in tiles]
    # all three X' are live at once here (2.25-4x the map each: 10.6 GB at 2 x 1024 x 2048); a caller that does not keep them for the
    # backward passes its budget, beyond which each branch transforms for itself again with one X' live at a time
    if max_bytes is not None and 4.0 * sum(p * t for p, t in zip(Ps, Ts)) * C > max_bytes:
        return None
    if pair_tile:
        assert Ts[0] == Ts[1]
        both = torch.empty((2 * Ps[0], Ts[0], C), device=dev, dtype=torch.float32)
        xts = [both[:Ps[0]], both[Ps[0]:], torch.empty((Ps[2], Ts[2], C), device=dev, dtype=torch.float32)]
    else:
        both = None
        xts = [torch.empty((p, t, C), device=dev, dtype=torch.float32) for p, t in zip(Ps, Ts)]
    nbytes = 4.0 * (N * H * W * C + sum(p * t for p, t in zip(Ps, Ts)) * C)
    with _Timed("wino_transform", nbytes, ("input_aspp3", N, H, W, C, d, tuple(tiles))):
        rc = _lib.status("mss_wino_input_transform_aspp3_f32", x.ptr, x.ld, N, H, W, C, d, (ctypes.c_int * 3)(*tiles), ptr(xts[0]),
                         ptr(xts[1]), ptr(xts[2]))
    if rc == _lib.MSS_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        raise _lib.MssError(f"mss_wino_input_transform_aspp3_f32 failed with code {rc}")
    return [both, None, xts[2]] if pair_tile else xts


def conv2d_wgrad_winograd(x, dy, K, C, dil=1, in_affine=None, in_relu=False, xt=None, tile=None):
    """[K,C,3,3] weight gradient of a 3x3 / stride-1 / padding = dilation conv in the Winograd domain:
    dU[p] = dY'[p]^T X'[p] ((m+2)^2 batched MFMA products over tiles, 2.25x / 4x fewer FLOPs than the 9-tap
    form). xt: the X' kept by the forward (its leading dimension tells the tile size)."""
    N, H, W = x.N, x.H, x.W
    dev = x.buf.device
    if xt is not None:
        ts = WINO_TILE_OF_P[xt.shape[0]]
    else:
        ts = tile or wino_tile(H, W, dil)
    P = (ts + 2) ** 2
    T = _lib.value("mss_wino_num_tiles", N, H, W, dil, ts)
    Kpad, Cp = _round_up(K, 4), _round_up(C, 4)
    with _Timed("wgrad_winograd", 2.0 * N * H * W * K * C * 9, (N, H, W, C, K, 3, 1, dil)):
        if xt is None:
            xt = torch.empty((P, T, C), device=dev, dtype=torch.float32)
            sc, sh = in_affine if in_affine is not None else (None, None)
            with _Timed("wino_transform", 4.0 * (N * H * W * C + P * T * C), ("input", N, H, W, C, dil, ts)):
                call("mss_wino_input_transform_f32", x.ptr, x.ld, N, H, W, C, dil, ts, ptr(sc), ptr(sh), int(in_relu), ptr(xt))
        assert tuple(xt.shape) == (P, T, C)
        dyt = torch.empty((P, T, K), device=dev, dtype=torch.float32)
        with _Timed("wino_transform", 4.0 * (N * H * W * K + P * T * K), ("grad_output", N, H, W, K, dil, ts)):
            call("mss_wino_grad_output_transform_f32", dy.ptr, dy.ld, N, H, W, K, dil, ts, ptr(dyt))
        du = torch.empty((P, Kpad, Cp), device=dev, dtype=torch.float32)     # fully overwritten by the kernel
        a = MssConvArgs()
        a.x = ptr(xt)
        a.N, a.H, a.W, a.C, a.ldx = 1, 1, T, C, C
        a.OH, a.OW, a.K, a.Kpad = 1, T, K, Kpad
        a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
        a.batch, a.x_bs, a.y_bs = P, T * C, T * K
        a.route = 1 if gemm_route() == "bf16x3" else 0
        ws, ws_bytes = _wgrad_workspace(a, Cp, dev)
        with _Timed(_wgrad_kind(a, K), 2.0 * P * T * K * C, (P, 1, T, C, K, 1, 1, 1)):
            call("mss_conv2d_wgrad_f32", ctypes.byref(a), ptr(dyt), K, ptr(du), Cp, ptr(ws), ws_bytes)
        grad = torch.empty((K, C, 3, 3), device=dev, dtype=torch.float32)
        call("mss_wino_weight_grad_transform_f32", ptr(du), ptr(grad), K, C, Kpad, Cp, ts)
    return grad


def conv3x3_wgrad(x, dy, K, C, dil=1, in_affine=None, in_relu=False, xt=None, max_tile=6):
    """Weight gradient of a 3x3 / stride-1 layer. max_tile caps the Winograd tile edge (as conv3x3's): the F(6x6) weight-gradient
    transform G^T dU G carries ~20x the rounding error of F(4x4)'s (measured on the pixel decoder's 3x3 layer, round 4)."""
    tile = WINO_TILE_OF_P[xt.shape[0]] if xt is not None else wino_tile(x.H, x.W, dil, max_tile)
    if use_winograd(C, K, 1, in_affine, tile):
        return conv2d_wgrad_winograd(x, dy, K, C, dil, in_affine, in_relu, xt=xt, tile=tile)
    return conv2d_wgrad(x, dy, K, C, 3, 3, dil=dil, pad=dil, in_affine=in_affine, in_relu=in_relu)


def image_to_nhwc(img, Cp=16):
    n, c, h, w = img.shape
    img = img.contiguous()
    out = Act.empty(n, h, w, Cp, img.device)
    call("mss_nchw_to_nhwc_pad_f32", ptr(img), out.ptr, n, c, h, w, Cp)
    return out


def stem_im2col(img):
    """[N,3,H,W] NCHW image -> Act [N,H,W,32]: the 27 taps of a 3x3 / padding-1 window per pixel (+5 zero channels)."""
    n, c, h, w = img.shape
    if c != 3:
        raise ValueError(f"the stem expects a 3-channel image, got {c}")
    img = img.contiguous()
    out = Act.empty(n, h, w, 32, img.device)
    call("mss_im2col3x3_c3_f32", ptr(img), out.ptr, n, h, w)
    return out


def stem_conv_pool(img, weight):
    """[N,3,H,W] NCHW image, mod1.conv1.weight [64,3,3,3] -> Act [N,(H-1)//2+1,(W-1)//2+1,64] = MaxPool2d(3,2,1)(conv3x3(img)) in
    ONE kernel (csrc/stem.hip): the full-resolution 64-channel map never exists."""
    n, c, h, w = img.shape
    if c != 3 or tuple(weight.shape) != (64, 3, 3, 3):
        raise ValueError(f"the fused stem expects a 3-channel image and a [64,3,3,3] weight, got {tuple(img.shape)} / {tuple(weight.shape)}")
    img = img.contiguous()
    wt = weight.detach().contiguous()
    out = Act.empty(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 64, img.device)
    call("mss_stem_conv_pool_f32", ptr(img), ptr(wt), out.ptr, out.ld, n, h, w)
    return out


def packed_stem(param):
    """mod1.conv1.weight [64,3,3,3] as the 1x1 weight [64,32,1,1] that matches stem_im2col's channel order."""
    def make():
        k = param.shape[0]
        w = torch.zeros((k, 32, 1, 1), device=param.device, dtype=torch.float32)
        w[:, :27, 0, 0] = param.detach().reshape(k, 27)
        return pack_weight(w)
    return _cached_pack(param, ("stem",), make)


class BNState:
    """Folded BatchNorm: y = x*scale + shift (+ what the backward needs in train mode)."""
    __slots__ = ("scale", "shift", "save_mean", "save_invstd", "M", "train")


def _col_accum(M, C, device):
    """Scratch of the deterministic two-stage per-channel reductions: [2C] sums + the first stage's partial rows."""
    return torch.empty(_lib.value("mss_col_reduce_accum_doubles", M, C), device=device, dtype=torch.float64)


def bn_fold(bn, x=None, train=False, M=None, x_rows=None, out=None):
    """Fold nn.BatchNorm2d `bn` into (scale, shift). train=True computes batch statistics of the
    NHWC activation x (Act) and updates the running buffers exactly like F.batch_norm
    (momentum 0.1, unbiased running variance; mynn.py:8-12). out: (scale, shift) to write into -- e.g. a 256-channel
    slice of the 1280-channel vectors of the ASPP concat, instead of a copy afterwards."""
    C = bn.num_features
    dev = bn.weight.device
    st = BNState()
    st.train = train
    if out is not None:
        st.scale, st.shift = out
        assert st.scale.numel() == C and st.shift.numel() == C and st.scale.is_contiguous() and st.shift.is_contiguous()
    else:
        st.scale = torch.empty(C, device=dev, dtype=torch.float32)
        st.shift = torch.empty(C, device=dev, dtype=torch.float32)
    st.save_mean = st.save_invstd = None
    if not train:
        call("mss_bn_fold_eval_f32", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             float(bn.eps), C, ptr(st.scale), ptr(st.shift))
        st.M = None
        return st
    if x_rows is not None:          # plain [M, C] matrix (image-pooling branch)
        M = x_rows.shape[0]
        accum = _col_accum(M, C, dev)
        call("mss_bn_stats_nhwc_f32", ptr(x_rows), M, C, x_rows.shape[1], ptr(accum))
    elif x.stats is not None and x.stats.shape[2] == C and x.C == C:   # left by the producing kernel: no re-read of x
        M = x.M
        accum = _col_accum(x.stats.shape[0], C, dev)
        st.M = M
        st.save_mean = torch.empty(C, device=dev, dtype=torch.float32)
        st.save_invstd = torch.empty(C, device=dev, dtype=torch.float32)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        # partial sums -> column sums -> folded affine + running statistics: two launches (the second stage and the fold are one)
        call("mss_bn_fold_train_from_partials_f32", ptr(x.stats), x.stats.shape[0], C, ptr(accum), M, ptr(bn.weight), ptr(bn.bias),
             float(bn.eps), mom, ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(st.scale),
             ptr(st.shift), ptr(st.save_mean), ptr(st.save_invstd))
        x.stats = None
        if track and bn.num_batches_tracked is not None:
            if _nbt_pending is not None:
                _nbt_pending.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
        return st
    else:
        M = x.M
        accum = _col_accum(M, C, dev)
        call("mss_bn_stats_nhwc_f32", x.ptr, M, C, x.ld, ptr(accum))
    st.M = M
    st.save_mean = torch.empty(C, device=dev, dtype=torch.float32)
    st.save_invstd = torch.empty(C, device=dev, dtype=torch.float32)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    track = bn.track_running_stats and bn.running_mean is not None
    call("mss_bn_finalize_train_f32", ptr(accum), M, C, ptr(bn.weight), ptr(bn.bias), float(bn.eps), mom,
         ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(st.scale),
         ptr(st.shift), ptr(st.save_mean), ptr(st.save_invstd))
    if track and bn.num_batches_tracked is not None:
        if _nbt_pending is not None:
            _nbt_pending.append(bn.num_batches_tracked)      # one launch for all layers of a forward (43 tiny kernels otherwise)
        else:
            bn.num_batches_tracked += 1
    return st


_nbt_pending = None


class batched_counters:
    """with batched_counters(): every train-mode bn_fold inside defers its `num_batches_tracked += 1`; they are applied by ONE
    multi-tensor add on exit (nn.BatchNorm2d's counter, F.batch_norm's bookkeeping: same values, 1 launch instead of one per layer)."""

    def __enter__(self):
        global _nbt_pending
        self.prev, _nbt_pending = _nbt_pending, []
        return self

    def __exit__(self, *exc):
        global _nbt_pending
        pending, _nbt_pending = _nbt_pending, self.prev
        if pending:
            if self.prev is not None:
                self.prev.extend(pending)
            else:
                torch._foreach_add_(pending, 1)
        return False


def bn_relu_backward(dy, x, st, relu=True, want_param_grads=False, x_rows=None, dy_rows=None):
    """Backward of y = relu(x*scale+shift) with (scale, shift) = folded BN `st`.
    Returns (dx Act or [M,C] tensor, dgamma, dbeta)."""
    if x_rows is not None:
        M, C = x_rows.shape
        xp, ldx, dyp, lddy = ptr(x_rows), x_rows.shape[1], ptr(dy_rows), dy_rows.shape[1]
        dx = torch.empty_like(x_rows)
        dxp, lddx = ptr(dx), C
        dev = x_rows.device
    else:
        M, C = x.M, x.C
        xp, ldx, dyp, lddy = x.ptr, x.ld, dy.ptr, dy.ld
        dx = Act.empty(x.N, x.H, x.W, C, x.buf.device)
        dxp, lddx = dx.ptr, dx.ld
        dev = x.buf.device
    dgamma = dbeta = None
    accum = None
    if st.train:
        accum = _col_accum(M, C, dev)
        call("mss_bn_relu_bwd_reduce_f32", dyp, lddy, xp, ldx, M, C, ptr(st.scale), ptr(st.shift), ptr(st.save_mean),
             ptr(st.save_invstd), int(relu), ptr(accum))
        if want_param_grads:
            dgamma = torch.zeros(C, device=dev, dtype=torch.float32)
            dbeta = torch.zeros(C, device=dev, dtype=torch.float32)
    elif want_param_grads:
        raise NotImplementedError("BatchNorm affine gradients in eval mode")
    call("mss_bn_relu_bwd_apply_f32", dyp, lddy, xp, ldx, dxp, lddx, M, C, None, ptr(st.scale), ptr(st.shift),
         ptr(st.save_mean), ptr(st.save_invstd), int(relu), ptr(accum), ptr(dgamma), ptr(dbeta))
    return dx, dgamma, dbeta


def conv3x3_on_upsampled_concat(a, small, weight, want_stats=False):
    """conv3x3(concat([a, upsample_ac(small -> a.H x a.W)], channels), weight) -- the decoder's first 3x3 layer on dec0 =
    cat(bot_fine(m2), Upsample(bot_aspp(...))) (deepv3.py:269-275) -- with the bilinear upsample interpolated inside the Winograd
    input transform (mss_wino_input_transform_upcat_f32): the full-resolution upsampled map is never written or read.
    Returns None where that transform does not apply (non-Winograd or small shapes, MSS_UPCAT_FUSED=0): the caller builds the concat."""
    k_out, c_in = weight.shape[0], weight.shape[1]
    N, H, W = a.N, a.H, a.W
    tile = wino_tile(H, W, 1)
    if not (os.environ.get("MSS_UPCAT_FUSED", "1") != "0" and _tile_hook is None and tile >= 4 and a.C + small.C == c_in and a.C % 4 == 0
            and small.N == N and use_winograd(c_in, k_out, 1, None, tile)):
        return None
    dev = a.buf.device
    ww = packed_wino(weight, False, tile)
    P = (tile + 2) ** 2
    T = _lib.value("mss_wino_num_tiles", N, H, W, 1, tile)
    with _Timed("conv_winograd", 2.0 * N * H * W * k_out * c_in * 9, (N, H, W, c_in, k_out, 3, 1, 1)):
        xt = torch.empty((P, T, c_in), device=dev, dtype=torch.float32)
        with _Timed("wino_transform", 4.0 * (N * H * W * a.C + small.M * small.C + P * T * c_in), ("input+upsample", N, H, W, c_in, 1, tile)):
            rc = _lib.status("mss_wino_input_transform_upcat_f32", a.ptr, a.ld, a.C, small.ptr, small.ld, small.H, small.W, N, H, W, c_in,
                             tile, ptr(xt))
        if rc == _lib.MSS_ERR_UNSUPPORTED:
            return None
        if rc != 0:
            raise _lib.MssError(f"mss_wino_input_transform_upcat_f32 failed with code {rc}")
        return _wino_gemm_and_output(xt, ww, N, H, W, 1, None, None, None, want_stats)


def conv3x3_dgrad_after_bn(dy, x, st, weight, relu=True):
    """conv3x3(bn_relu_backward(dy, x, st)[0], weight, flip=True) -- the data gradient of the 3x3 layer in FRONT of a train-mode
    BatchNorm+ReLU -- with the BatchNorm backward's apply pass folded into the Winograd input transform of that convolution
    (mss_wino_input_transform_bnbwd_f32): the gradient w.r.t. the BatchNorm input is never written or read back. Falls back to the
    two separate steps where the fused transform does not apply (eval-mode statistics, non-Winograd or small shapes,
    MSS_BNBWD_FUSED=0). Same arithmetic per element, so the two routes agree bit for bit."""
    k_out, c_in = weight.shape[1], weight.shape[0]
    tile = wino_tile(x.H, x.W, 1)
    fused = (st.train and os.environ.get("MSS_BNBWD_FUSED", "1") != "0" and _tile_hook is None and tile >= 4
             and use_winograd(c_in, k_out, 1, None, tile) and dy.C == c_in and x.C == c_in)
    if fused:
        N, H, W, C = x.N, x.H, x.W, c_in
        dev = x.buf.device
        accum = _col_accum(x.M, C, dev)
        call("mss_bn_relu_bwd_reduce_f32", dy.ptr, dy.ld, x.ptr, x.ld, x.M, C, ptr(st.scale), ptr(st.shift), ptr(st.save_mean),
             ptr(st.save_invstd), int(relu), ptr(accum))
        ww = packed_wino(weight, True, tile)
        P = (tile + 2) ** 2
        T = _lib.value("mss_wino_num_tiles", N, H, W, 1, tile)
        with _Timed("conv_winograd", 2.0 * N * H * W * k_out * C * 9, (N, H, W, C, k_out, 3, 1, 1)):
            xt = torch.empty((P, T, C), device=dev, dtype=torch.float32)
            with _Timed("wino_transform", 4.0 * (2 * N * H * W * C + P * T * C), ("input+bn_bwd", N, H, W, C, 1, tile)):
                rc = _lib.status("mss_wino_input_transform_bnbwd_f32", dy.ptr, dy.ld, x.ptr, x.ld, N, H, W, C, 1, tile, ptr(st.scale),
                                 ptr(st.shift), ptr(st.save_mean), ptr(st.save_invstd), ptr(accum), int(relu), ptr(xt))
            if rc == 0:
                return _wino_gemm_and_output(xt, ww, N, H, W, 1, None, None, None, False)
            if rc != _lib.MSS_ERR_UNSUPPORTED:
                raise _lib.MssError(f"mss_wino_input_transform_bnbwd_f32 failed with code {rc}")
            del xt
    dx, _, _ = bn_relu_backward(dy, x, st, relu=relu)
    return conv3x3(dx, weight, flip=True)


def maxpool3s2(x):
    OH, OW = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
    y = Act.empty(x.N, OH, OW, x.C, x.buf.device)
    call("mss_maxpool3s2_nhwc_f32", x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, OH, OW)
    return y


def gap(x):
    y = torch.empty((x.N, x.C), device=x.buf.device, dtype=torch.float32)
    # the producing kernel's epilogue left per-64-row column sums (x.stats, wanted for BatchNorm statistics or asked for by the
    # caller): the mean comes from those few MB instead of another pass over the map (r04; MSS_GAP_FROM_STATS=0: always the pass)
    if x.stats is not None and (x.H * x.W) % 64 == 0 and x.stats.shape[0] * 64 == x.M and x.stats.shape[2] == x.C \
            and os.environ.get("MSS_GAP_FROM_STATS", "1") != "0":
        call("mss_gap_from_partials_f32", ptr(x.stats), x.N, x.H * x.W, x.C, ptr(y))
        return y
    ws = torch.empty(_lib.value("mss_colsum_workspace_floats", x.N, x.H * x.W, x.C), device=x.buf.device, dtype=torch.float32)
    call("mss_gap_nhwc_f32", x.ptr, x.ld, ptr(y), x.N, x.H * x.W, x.C, ptr(ws))
    return y


def colsum(dy):
    y = torch.empty((dy.N, dy.C), device=dy.buf.device, dtype=torch.float32)
    ws = torch.empty(_lib.value("mss_colsum_workspace_floats", dy.N, dy.H * dy.W, dy.C), device=dy.buf.device, dtype=torch.float32)
    call("mss_colsum_nhwc_f32", dy.ptr, dy.ld, ptr(y), dy.N, dy.H * dy.W, dy.C, ptr(ws))
    return y


def broadcast_rows(v, out, scale=None, shift=None, relu=False):
    call("mss_broadcast_rows_nhwc_f32", ptr(v), out.ptr, out.ld, out.N, out.H * out.W, out.C, ptr(scale), ptr(shift),
         int(relu))
    return out


def upsample_ac(x, OH, OW, out=None):
    if out is None:
        out = Act.empty(x.N, OH, OW, x.C, x.buf.device)
    call("mss_upsample_ac_nhwc_f32", x.ptr, x.ld, out.ptr, out.ld, x.N, x.H, x.W, OH, OW, x.C)
    return out


def upsample_ac_bwd(dy, IH, IW):
    dx = Act.empty(dy.N, IH, IW, dy.C, dy.buf.device)
    call("mss_upsample_ac_nhwc_bwd_f32", dy.ptr, dy.ld, dx.ptr, dx.ld, dy.N, IH, IW, dy.H, dy.W, dy.C)
    return dx


def ood_score(dec2, dec1, OH, OW, want_score=True, want_logit=True, want_label=False):
    """The OOD-score tail (deepv3.py:279-283). dec2/dec1: Acts with 19 channels at half res."""
    ref = dec2 if dec2 is not None else dec1
    dev = ref.buf.device
    N = ref.N
    score = torch.empty((N, OH, OW), device=dev, dtype=torch.float32) if want_score else None
    logit = torch.empty((N, 19, OH, OW), device=dev, dtype=torch.float32) if want_logit else None
    label = torch.empty((N, OH, OW), device=dev, dtype=torch.uint8) if want_label else None
    call("mss_ood_score_f32", dec2.ptr if dec2 is not None else None, dec2.ld if dec2 is not None else 0,
         dec1.ptr if dec1 is not None else None, dec1.ld if dec1 is not None else 0, N, ref.H, ref.W, 19, OH, OW,
         ptr(score), ptr(logit), ptr(label))
    return score, logit, label


def ood_score_bwd(dec2, dscore, dlogit, ddec2, ddec1, OH, OW):
    call("mss_ood_score_bwd_f32", dec2.ptr, dec2.ld, ptr(dscore), ptr(dlogit), dec2.N, dec2.H, dec2.W, 19, OH, OW,
         ddec2.ptr if ddec2 is not None else None, ddec2.ld if ddec2 is not None else 0,
         ddec1.ptr if ddec1 is not None else None, ddec1.ld if ddec1 is not None else 0)


def m2f_score(class_logits, mask_logits, size):
    """train_m2f.py:387-407 -> [B,H,W]."""
    B, Q, C1 = class_logits.shape
    _, _, Hm, Wm = mask_logits.shape
    H, W = size
    cls = class_logits.contiguous().float()
    mask = mask_logits.contiguous().float()
    out = torch.empty((B, H, W), device=cls.device, dtype=torch.float32)
    call("mss_m2f_score_f32", ptr(cls), ptr(mask), B, Q, C1 - 1, H, W, Hm, Wm, ptr(out))
    return out


def m2f_mask_logits(mask_embed, mask_features):
    """einsum("bqc,bchw->bqhw", mask_embed, mask_features) (mask2former_transformer_decoder.py:544-548) as ONE batched
    MFMA GEMM, written pixel-major: returns [B, h, w, Q] (queries contiguous), the layout m2f_score_fused reads."""
    B, Q, C = mask_embed.shape
    Bf, Cf, h, w = mask_features.shape
    if (B, C) != (Bf, Cf):
        raise ValueError(f"mask_embed {tuple(mask_embed.shape)} does not match mask_features {tuple(mask_features.shape)}")
    x = Act.from_nchw(mask_features.float())
    Kpad = _lib.value("mss_conv2d_kpad", Q)
    wp = torch.zeros((B, Kpad, C), device=x.buf.device, dtype=torch.float32)
    wp[:, :Q] = mask_embed.detach().float()
    out = torch.empty((B, h, w, Q), device=x.buf.device, dtype=torch.float32)
    a = MssConvArgs()
    a.x, a.w, a.y = x.ptr, ptr(wp), ptr(out)
    a.N, a.H, a.W, a.C, a.ldx = 1, 1, h * w, C, x.ld
    a.OH, a.OW, a.K, a.Kpad, a.ldy = 1, h * w, Q, Kpad, Q
    a.R, a.S, a.stride, a.dil, a.pad = 1, 1, 1, 1, 0
    a.batch, a.x_bs, a.w_bs, a.y_bs = B, h * w * x.ld, Kpad * C, h * w * Q
    with _Timed(_fwd_kind(a), 2.0 * B * h * w * C * Q, (B, 1, h * w, C, Q, 1, 1, 1)):
        call("mss_conv2d_forward_f32", ctypes.byref(a))
    return out


def m2f_score_fused(class_logits, mask_logits_nhwc, image_size, size=None):
    """Anomaly score from LOW-resolution pixel-major mask logits [B,hm,wm,Q]: bilinear upsample to `image_size`
    (align_corners=False, maskformer_model.py:264-277) + train_m2f.py:387-407, cropped to `size` -> [B,H,W]."""
    B, Q, C1 = class_logits.shape
    Bm, hm, wm, ldq = mask_logits_nhwc.shape
    Hi, Wi = image_size
    H, W = size if size is not None else image_size
    cls = class_logits.contiguous().float()
    lg = mask_logits_nhwc.contiguous()
    out = torch.empty((B, H, W), device=cls.device, dtype=torch.float32)
    prob = torch.empty((B, Q, 32), device=cls.device, dtype=torch.float32)      # class-probability table of the MFMA kernel
    call("mss_m2f_fused_score_ws_f32", ptr(cls), ptr(lg), B, Q, C1 - 1, hm, wm, ldq, Hi, Wi, H, W, ptr(out), ptr(prob))
    return out


# ---- Mask2Former pixel-decoder glue (csrc/norm.hip) --------------------------------------------------------------------
def groupnorm(x, gn, relu=False, out=None, out_sample_stride=None, out_ld=None, want_stat=False):
    """nn.GroupNorm `gn` on an Act (NHWC). `out`: optional float tensor to write into (e.g. the encoder's token buffer
    [N, sum(HW), C] at a level's offset) with pixel stride `out_ld` and sample stride `out_sample_stride` floats.
    want_stat: also return the [N*groups, 2] (mean, rstd) view the backward needs."""
    N, HW, C = x.N, x.H * x.W, x.C
    dev = x.buf.device
    ws = torch.empty(_lib.value("mss_groupnorm_workspace_floats", N, HW, C, gn.num_groups), device=dev, dtype=torch.float32)
    if out is None:
        y = Act.empty(N, x.H, x.W, C, dev)
        optr, old, oss = y.ptr, y.ld, HW * y.ld
    else:
        y, optr, old, oss = out, ptr(out), out_ld, out_sample_stride
    call("mss_groupnorm_nhwc_f32", x.ptr, x.ld, HW * x.ld, N, HW, C, gn.num_groups, ptr(gn.weight), ptr(gn.bias), float(gn.eps),
         int(relu), optr, old, oss, ptr(ws))
    if want_stat:
        off = _lib.value("mss_groupnorm_stat_offset", N, HW, C)
        return y, ws[off:off + 2 * N * gn.num_groups]
    return y


def groupnorm_backward(gy_ptr, gy_ld, gy_ss, x, gn, stat, relu=False):
    """Backward of groupnorm(x, gn, relu): gy given as (pointer, pixel stride, sample stride) so that it may live inside a
    token buffer. Returns (dx Act, dgamma, dbeta)."""
    N, HW, C = x.N, x.H * x.W, x.C
    dev = x.buf.device
    dx = Act.empty(N, x.H, x.W, C, dev)
    dg = torch.empty(C, device=dev, dtype=torch.float32)
    db = torch.empty(C, device=dev, dtype=torch.float32)
    ws = torch.empty(_lib.value("mss_groupnorm_bwd_workspace_floats", N, HW, C, gn.num_groups), device=dev, dtype=torch.float32)
    call("mss_groupnorm_nhwc_bwd_f32", gy_ptr, gy_ld, gy_ss, x.ptr, x.ld, HW * x.ld, N, HW, C, gn.num_groups, ptr(stat), ptr(gn.weight),
         ptr(gn.bias), int(relu), dx.ptr, dx.ld, ptr(dg), ptr(db), ptr(ws))
    return dx, dg, db


def upsample_bilinear_bwd(dy, dtop_ptr, dtop_ld, dtop_ss, IH, IW, accumulate):
    """Transpose of the bilinear part of upsample_bilinear_add: dtop (+)= B^T dy, dtop given as (pointer, strides)."""
    call("mss_upsample_bilinear_bwd_nhwc_f32", dy.ptr, dy.ld, dy.N, dy.H, dy.W, dtop_ptr, dtop_ld, dtop_ss, IH, IW, dy.C, int(accumulate))


def nchw_into_rows(g, dst_ptr, dst_ld, dst_ss, accumulate=False):
    """NCHW gradient tensor -> rows of an NHWC / token buffer given as (pointer, pixel stride, sample stride)."""
    n, c, h, w = g.shape
    g = g.contiguous().float()
    call("mss_nchw_to_nhwc_strided_f32", ptr(g), n, c, h * w, dst_ptr, dst_ld, dst_ss, int(accumulate))


class TokenLevel:
    """One level of a token buffer [N, S, C] seen as an NHWC map: rows [start, start + H*W) of every sample."""
    __slots__ = ("buf", "start", "N", "H", "W", "C", "ld", "sample_stride")

    def __init__(self, buf, start, H, W):
        assert buf.dim() == 3 and buf.is_contiguous() and buf.dtype == torch.float32
        self.buf, self.start, self.H, self.W = buf, start, H, W
        self.N, S, self.C = buf.shape
        self.ld, self.sample_stride = self.C, S * self.C

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + 4 * self.start * self.C)


def _strided(x):
    """(pointer, pixel stride, sample stride) of an Act or a TokenLevel."""
    return (x.ptr, x.ld, x.sample_stride if isinstance(x, TokenLevel) else x.H * x.W * x.ld)


def upsample_bilinear_add(top, lat):
    """lat + F.interpolate(top, size=lat.shape[-2:], mode="bilinear", align_corners=False) (msdeformattn.py:344).
    top: Act or TokenLevel, lat: Act."""
    assert top.C == lat.C and top.N == lat.N
    y = Act.empty(lat.N, lat.H, lat.W, lat.C, lat.buf.device)
    tp, tld, tss = _strided(top)
    call("mss_upsample_bilinear_add_nhwc_f32", tp, tld, tss, top.N, top.H, top.W, lat.ptr, lat.ld, y.ptr, y.ld, lat.H, lat.W, lat.C)
    return y


def nhwc_to_nchw(x):
    """Act or TokenLevel -> contiguous NCHW tensor (module boundary of the pixel decoder)."""
    y = torch.empty((x.N, x.C, x.H, x.W), device=x.buf.device, dtype=torch.float32)
    xp, xld, xss = _strided(x)
    call("mss_nhwc_to_nchw_f32", xp, xld, xss, x.N, x.H * x.W, x.C, ptr(y))
    return y


def nchw_to_act(t, Cp=None):
    """NCHW tensor -> Act with the channel count padded up to a multiple of 16 (the MFMA kernels' K granularity)."""
    n, c, h, w = t.shape
    Cp = Cp or _round_up(c, 16)
    t = t.contiguous().float()
    out = Act.empty(n, h, w, Cp, t.device)
    call("mss_nchw_to_nhwc_pad_f32", ptr(t), out.ptr, n, c, h, w, Cp)
    return out


class _AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) in one HIP pass each way (msdeformattn.py:116-131 norm1 / norm2 sites)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        C = x.shape[-1]
        x2, r2 = x.contiguous(), (res.contiguous() if res is not None else None)
        rows = x2.numel() // C
        y = torch.empty_like(x2)
        stat = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        call("mss_add_layernorm_f32", ptr(x2), ptr(r2), rows, C, ptr(weight), ptr(bias), float(eps), ptr(y), ptr(stat))
        ctx.save_for_backward(x2, r2, weight, stat)
        ctx.has_res = res is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, res, weight, stat = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        gy = gy.contiguous()
        dz = torch.empty_like(x)
        dg = torch.empty_like(weight)
        db = torch.empty_like(weight)
        ws = torch.empty(_lib.value("mss_add_layernorm_bwd_workspace_floats", rows, C), device=x.device, dtype=torch.float32)
        call("mss_add_layernorm_bwd_f32", ptr(gy), ptr(x), ptr(res), ptr(stat), rows, C, ptr(weight), ptr(dz), ptr(dg), ptr(db), ptr(ws))
        return dz, (dz if ctx.has_res else None), dg, db, None


def add_layernorm(x, res, ln):
    """nn.LayerNorm `ln` applied to x + res (res may be None). Falls back to torch for shapes the kernel does not take."""
    C = x.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and C % 256 == 0 and C <= 1024 and ln.elementwise_affine and ln.bias is not None \
            and tuple(ln.normalized_shape) == (C,):
        return _AddLayerNormFn.apply(x, res, ln.weight, ln.bias, ln.eps)
    return ln(x if res is None else x + res)
