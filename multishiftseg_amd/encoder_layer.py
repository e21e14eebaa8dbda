"""One autograd node per deformable-attention encoder layer (msdeformattn.py:92-131 with Dropout p = 0):

    q    = src + pos
    attn = output_proj( MSDeformAttn-sample( value_proj(src), sampling_offsets(q), attention_weights(q) ) )
    s1   = LayerNorm1(src + attn)
    out  = LayerNorm2(s1 + linear2(relu(linear1(s1))))

Same kernels and the same arithmetic as the layer composed from `linear`, `_FusedSampleFn` and `add_layernorm` (the tests
compare the two bit for bit in the forward and to rounding in the backward); what the single node buys is the BACKWARD:
composed from separate nodes, autograd sums the gradients of every tensor with several consumers (src: query, value
projection, residual; q: two projections; s1: FFN, residual) with one elementwise pass each over [N, S, 256] -- four to five
166 MB passes per layer at 16 x 10 164 tokens. Here those sums ride in the residual input of the data-gradient GEMMs'
epilogues (`MssConvArgs.res`), and one explicit add per layer is left (a GEMM epilogue takes one residual).
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import MultiScaleDeformableAttention as MSDA
from . import _lib
from . import kernels as K
from ._lib import call, ptr
from .linear import _packed, _rows


def _gemm(x, weight, bias=None, relu=False, res=None, res_mask=False, flip=False):
    """x [..., c] -> [..., k] with weight [k, c] (flip: x [..., k] -> [..., c], the data gradient), epilogue bias / ReLU /
    residual add (or ReLU-backward gate) on the fp32 MFMA GEMM."""
    k, c = weight.shape
    cin, cout = (k, c) if flip else (c, k)
    out = torch.empty(x.shape[:-1] + (cout,), device=x.device, dtype=torch.float32)
    aff = (K.ones(bias.numel(), bias.device), bias.detach()) if bias is not None else None
    K.conv2d(_rows(x, cin), _packed(weight, flip), out_affine=aff, out_relu=relu, out=_rows(out, cout),
             res=_rows(res, cout) if res is not None else None, res_mask=res_mask)
    return out


def _packed_pair(w1, w2, flip):
    """The forward (flip=False) or data-gradient (flip=True) pack of the row-concatenation [w1; w2] (two Linears on the same
    input): cached on w1 until either parameter changes (optimizer step: tensor._version; moved: data_ptr)."""
    key = (w1._version, w1.data_ptr(), w2._version, w2.data_ptr())
    cache = w1.__dict__.setdefault("_mss_packed", {})
    name = "pair_flip" if flip else "pair"
    ent = cache.get(name)
    if ent is None or ent[0] != key:
        w = torch.cat((w1.detach(), w2.detach()), 0)
        ent = cache[name] = (key, K.pack_weight(w.view(w.shape[0], w.shape[1], 1, 1), flip=flip))
    return ent[1]


def _packed_pair_flip(w1, w2):
    return _packed_pair(w1, w2, True)


def _bias_pair(b1, b2):
    """[b1; b2] as one epilogue bias vector, cached like the packed weights."""
    key = (b1._version, b1.data_ptr(), b2._version, b2.data_ptr())
    cache = b1.__dict__.setdefault("_mss_packed", {})
    ent = cache.get("bias_pair")
    if ent is None or ent[0] != key:
        ent = cache["bias_pair"] = (key, torch.cat((b1.detach(), b2.detach()), 0))
    return ent[1]


def _wgrad(x, gy, weight):
    k, c = weight.shape
    return K.conv2d_wgrad(_rows(x, c), _rows(gy, k), k, c, 1, 1).view(k, c)


def _bgrad(gy, k):
    return K.colsum(_rows(gy, k)).view(k)


def _layernorm(x, res, weight, bias, eps):
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    stat = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    call("mss_add_layernorm_f32", ptr(x), ptr(res), rows, C, ptr(weight), ptr(bias), float(eps), ptr(y), ptr(stat))
    return y, stat


def _layernorm_q(x, res, weight, bias, eps, pos):
    """_layernorm that also leaves q = y + pos (the next layer's query) with the same pass; pos [1 | N, S, C]."""
    C = x.shape[-1]
    rows = x.numel() // C
    y, q = torch.empty_like(x), torch.empty_like(x)
    stat = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    call("mss_add_layernorm_q_f32", ptr(x), ptr(res), rows, C, ptr(weight), ptr(bias), float(eps), ptr(y), ptr(stat), ptr(pos),
         pos.shape[0] * pos.shape[1], ptr(q))
    return y, stat, q


def _layernorm_bwd(gy, x, res, stat, weight, gy2=None):
    """-> dz, dgamma, dbeta, column sums of dz (= the bias gradient of the Linear that produced `res`: no colsum pass over dz).
    gy2: the gradient from a second consumer of the LayerNorm's output (the next layer's query), added while loading."""
    C = x.shape[-1]
    rows = x.numel() // C
    dz = torch.empty_like(x)
    dg, db, dzsum = torch.empty_like(weight), torch.empty_like(weight), torch.empty_like(weight)
    ws = torch.empty(_lib.value("mss_add_layernorm_bwd_workspace_floats", rows, C) // 2 * 3, device=x.device, dtype=torch.float32)
    call("mss_add_layernorm_bwd_sum2_f32", ptr(gy), ptr(gy2), ptr(x), ptr(res), ptr(stat), rows, C, ptr(weight), ptr(dz), ptr(dg),
         ptr(db), ptr(dzsum), ptr(ws))
    return dz, dg, db, dzsum


def _gemm_with_colsum(x, weight, res, res_mask, flip):
    """_gemm whose epilogue also leaves the column sums of its OUTPUT (MssConvArgs.stats: per-64-row partial sums written by the
    producing kernel, added in float64 by the BatchNorm-statistics reduction) -- the bias gradient of the layer below, without
    re-reading the [rows, d_ffn] tensor."""
    k, c = weight.shape
    cin, cout = (k, c) if flip else (c, k)
    out = torch.empty(x.shape[:-1] + (cout,), device=x.device, dtype=torch.float32)
    o = K.conv2d(_rows(x, cin), _packed(weight, flip), out=_rows(out, cout), res=_rows(res, cout) if res is not None else None,
                 res_mask=res_mask, want_stats=True)
    if o.stats is None:
        return out, _bgrad(out, cout)
    accum = K._col_accum(o.stats.shape[0], cout, x.device)
    call("mss_bn_stats_partials_f32", ptr(o.stats), o.stats.shape[0], cout, ptr(accum))
    return out, accum[:cout].float()


# positions of the parameters in _EncoderLayerFn.apply's argument list (after src, q, pos, ref, shapes, starts, geometry)
_PARAMS = ("off_w", "off_b", "att_w", "att_b", "val_w", "val_b", "out_w", "out_b", "n1_w", "n1_b",
           "l1_w", "l1_b", "l2_w", "l2_b", "n2_w", "n2_b")


class _EncoderLayerFn(Function):
    @staticmethod
    def forward(ctx, src, q_in, pos, ref, shapes, starts, geom, *params):
        """-> (out, q_next). q_in: src + pos when the previous layer's LayerNorm already produced it (None: formed here);
        q_next = out + pos, written by this layer's second LayerNorm when geom asks for it (else None)."""
        p = dict(zip(_PARAMS, params))
        M, L, P, eps1, eps2, want_q, save_loc = geom
        N, S, C = src.shape
        D = C // M
        src = src.contiguous()
        q = src + pos if q_in is None else q_in.contiguous()
        value = _gemm(src, p["val_w"], p["val_b"])
        # `sampling_offsets(q)` and `attention_weights(q)` (ops/modules/ms_deform_attn.py:98-101) as ONE product into a
        # [N, S, 192 + 96] buffer whose two column ranges the sampler reads with a row stride (r04; the backward already did)
        ko, ka = M * L * P * 2, M * L * P
        ol = torch.empty((N, S, ko + ka), device=src.device, dtype=torch.float32)
        bias = _bias_pair(p["off_b"], p["att_b"])
        K.conv2d(_rows(q, C), _packed_pair(p["off_w"], p["att_w"], False), out_affine=(K.ones(ko + ka, bias.device), bias),
                 out=_rows(ol, ko + ka))
        ref = ref.contiguous().float()
        samp = torch.empty((N, S, C), device=src.device, dtype=torch.float32)
        # training: the sampler also writes the locations / weights it formed, so that the backward reads what the forward used
        # instead of re-deriving them from `ol` (r04: one prepare kernel less per layer; `ol` is then not kept)
        loc = aw = None
        plog = ctypes.c_void_p(ol.data_ptr() + 4 * ko)
        if save_loc and any(ctx.needs_input_grad):
            loc = torch.empty((N, S, M, L, P, 2), device=src.device, dtype=torch.float32)
            aw = torch.empty((N, S, M, L, P), device=src.device, dtype=torch.float32)
            rc = _lib.status("mss_msda_forward_fused_save_f32", ptr(value), ptr(shapes), ptr(starts), ptr(ol), ko + ka, plog, ko + ka,
                             ptr(ref), N, S, M, D, L, S, P, ptr(samp), ptr(loc), ptr(aw))
            if rc == _lib.MSS_ERR_UNSUPPORTED:
                loc = aw = None
            elif rc != 0:
                raise _lib.MssError(f"mss_msda_forward_fused_save_f32 failed with code {rc}")
        if loc is None:
            call("mss_msda_forward_fused_ld_f32", ptr(value), ptr(shapes), ptr(starts), ptr(ol), ko + ka, plog, ko + ka, ptr(ref),
                 N, S, M, D, L, S, P, ptr(samp))
        attn = _gemm(samp, p["out_w"], p["out_b"])
        s1, stat1 = _layernorm(src, attn, p["n1_w"], p["n1_b"], eps1)
        h = _gemm(s1, p["l1_w"], p["l1_b"], relu=True)
        f = _gemm(h, p["l2_w"], p["l2_b"])
        qn = None
        if want_q:
            out, stat2, qn = _layernorm_q(s1, f, p["n2_w"], p["n2_b"], eps2, pos.contiguous())
        else:
            out, stat2 = _layernorm(s1, f, p["n2_w"], p["n2_b"], eps2)
        ctx.geom = geom
        ctx.pos_batch = pos.shape[0]
        ctx.q_is_input = q_in is not None
        ctx.shapes_host = getattr(shapes, "_mss_host", None)
        ctx.saved_loc = loc is not None
        keep = (loc, aw) if loc is not None else (ol, None)      # what the backward derives the sampling operands from
        ctx.save_for_backward(src, q, ref, shapes, starts, value, keep[0], keep[1], samp, attn, stat1, s1, h, f, stat2, *params)
        if qn is None:
            return out, None
        return out, qn

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, gqn):
        src, q, ref, shapes, starts, value, k0, k1, samp, attn, stat1, s1, h, f, stat2 = ctx.saved_tensors[:15]
        p = dict(zip(_PARAMS, ctx.saved_tensors[15:]))
        need = dict(zip(("src", "q", "pos") + (None,) * 4 + _PARAMS, ctx.needs_input_grad))
        M, L, P = ctx.geom[:3]
        N, S, C = src.shape
        D = C // M
        if ctx.shapes_host is not None:
            shapes._mss_host = ctx.shapes_host          # the host copy of the level sizes travels with the tensor object only
        g = {}
        # ---- LayerNorm2 and the FFN: d(s1) = g2 + dh W1 rides in the last data-gradient GEMM's residual input
        # (three of the six bias gradients come out of kernels that run anyway: linear2's and output_proj's are the column sums of
        # the LayerNorm backward's dz, linear1's the column sums of the ReLU-gated data-gradient GEMM's output)
        if gout is None:                      # only the query of the next layer was used
            gout, gqn = gqn, None
        g2, g["n2_w"], g["n2_b"], g["l2_b"] = _layernorm_bwd(gout.contiguous(), s1, f, stat2, p["n2_w"],
                                                             gy2=gqn.contiguous() if gqn is not None else None)
        if need["l2_w"]:
            g["l2_w"] = _wgrad(h, g2, p["l2_w"])
        if need["l1_b"]:
            dh, g["l1_b"] = _gemm_with_colsum(g2, p["l2_w"], res=h, res_mask=True, flip=True)   # ReLU backward in the epilogue
        else:
            dh = _gemm(g2, p["l2_w"], res=h, res_mask=True, flip=True)
        if need["l1_w"]:
            g["l1_w"] = _wgrad(s1, dh, p["l1_w"])
        ds1 = _gemm(dh, p["l1_w"], res=g2, flip=True)
        del dh, g2
        # ---- LayerNorm1 and the attention output projection
        g1, g["n1_w"], g["n1_b"], g["out_b"] = _layernorm_bwd(ds1, src, attn, stat1, p["n1_w"])
        del ds1
        if need["out_w"]:
            g["out_w"] = _wgrad(samp, g1, p["out_w"])
        dsamp = _gemm(g1, p["out_w"], flip=True)
        # ---- the sampling op: locations / weights rebuilt by the one-pass prepare kernel (bit-identical to the forward's)
        ko, ka = M * L * P * 2, M * L * P
        if ctx.saved_loc:
            loc, aw = k0, k1                   # what the forward's sampler used
        else:
            ol = k0
            loc = torch.empty((N, S, M, L, P, 2), device=src.device, dtype=torch.float32)
            aw = torch.empty((N, S, M, L, P), device=src.device, dtype=torch.float32)
            call("mss_msda_prepare_ld_f32", ptr(ol), ko + ka, ctypes.c_void_p(ol.data_ptr() + 4 * ko), ko + ka, ptr(ref), ptr(shapes),
                 N, S, M, L, P, ptr(loc), ptr(aw))
        # `sampling_offsets` and `attention_weights` are two Linears on the SAME q: their output gradients go side by side into one
        # [N, S, 192 + 96] buffer, so the weight gradient, the bias gradient and the data gradient are one product each (r04:
        # the narrow 192- / 96-wide products ran at 71 / 59 TFLOP/s for the weight and 119 / 73 for the data gradient); the
        # op's gather pass writes them there itself (softmax / location backward folded in: no grad_loc / grad_attn tensors)
        gol = torch.empty((N, S, ko + ka), device=src.device, dtype=torch.float32)
        gvalue = MSDA.ms_deform_attn_backward_proj(value.view(N, S, M, D), shapes, starts, loc, aw, dsamp, gol, ko)
        if gvalue is None:
            gvalue, gloc, gaw = MSDA.ms_deform_attn_backward(value.view(N, S, M, D), shapes, starts, loc, aw, dsamp, 128)
            call("mss_msda_prepare_backward_ld_f32", ptr(aw), ptr(gaw), ptr(gloc), ptr(shapes), N, S, M, L, P, ptr(gol), ko + ka,
                 ctypes.c_void_p(gol.data_ptr() + 4 * ko), ko + ka)
            del gloc, gaw
        del dsamp, loc, aw
        gvalue = gvalue.view(N, S, C)
        # ---- the three input projections: d(q) = [goff | glog] [Woff ; Watt], d(src) = g1 + gvalue Wv + d(q)
        if need["off_w"] or need["att_w"]:
            # (splitting the 288 columns into 256 on the LDS-free kernel + 32 on the narrow streaming one was measured: +-0 at 16 crops,
            # +0.15-0.3 ms at one image -- two launches for one; the kernels keep the row-stride support, the layer does not use it)
            gw = K.conv2d_wgrad(_rows(q, C), _rows(gol, ko + ka), ko + ka, C, 1, 1).view(ko + ka, C)
            g["off_w"], g["att_w"] = gw[:ko], gw[ko:]
        if need["off_b"] or need["att_b"]:
            gb = _bgrad(gol, ko + ka)
            g["off_b"], g["att_b"] = gb[:ko], gb[ko:]
        if need["val_w"]:
            g["val_w"] = _wgrad(src, gvalue, p["val_w"])
        if need["val_b"]:
            g["val_b"] = _bgrad(gvalue, C)
        dq = dsrc = None
        q_own = not ctx.q_is_input                        # q = src + pos was formed inside this node
        if need["q"] or (q_own and (need["src"] or need["pos"])):
            dq = torch.empty((N, S, C), device=src.device, dtype=torch.float32)
            K.conv2d(_rows(gol, ko + ka), _packed_pair_flip(p["off_w"], p["att_w"]), out=_rows(dq, C))
        if need["src"]:
            dsrc = _gemm(gvalue, p["val_w"], res=g1, flip=True)
            if q_own:
                dsrc += dq
        dpos = None
        if need["pos"]:
            # a position input shared by the whole batch ([1, S, C], msdeformattn_encoder.forward_tokens): its gradient is the batch
            # sum -- of d(q) when q was formed here, and of the gradient of q_next = out + pos when this node produced it
            batch = (lambda t: t.sum(0, keepdim=True)) if ctx.pos_batch == 1 and N > 1 else (lambda t: t)
            if q_own:
                dpos = batch(dq)
            if gqn is not None:
                dpos = batch(gqn) if dpos is None else dpos + batch(gqn)
        return (dsrc, dq if need["q"] else None, dpos, None, None, None, None) + tuple(g.get(n) if need[n] else None for n in _PARAMS)


def eligible(layer, src, pos, reference_points, spatial_shapes, padding_mask):
    """The shapes the fused node takes: the pixel decoder's (fp32, 2-d reference points, 8 heads x 32 channels, <= 20 samples per
    head, ReLU FFN, no dropout, no padding mask). Everything else runs as the composition of the separate nodes."""
    import torch.nn.functional as F
    a = layer.self_attn
    C = src.shape[-1]
    return (src.is_cuda and src.dtype == torch.float32 and pos is not None and pos.dim() == 3 and pos.shape[1:] == src.shape[1:]
            and pos.shape[0] in (1, src.shape[0]) and pos.dtype == torch.float32 and padding_mask is None
            and reference_points.shape[-1] == 2 and not reference_points.requires_grad and spatial_shapes.dtype == torch.int64
            and C == a.d_model and C % 256 == 0 and C <= 1024 and C // a.n_heads == 32 and a.n_levels * a.n_points <= 20
            and a.n_heads * a.n_levels * a.n_points > 64 and (a.n_heads * a.n_levels * a.n_points) % 16 == 0
            and layer.activation is F.relu and (not layer.training or (layer.dropout1.p == 0.0 and layer.dropout2.p == 0.0 and layer.dropout3.p == 0.0))
            and layer.linear1.bias is not None and layer.linear2.bias is not None and layer.linear1.out_features % 16 == 0
            and layer.norm1.elementwise_affine and layer.norm1.bias is not None and layer.norm2.elementwise_affine and layer.norm2.bias is not None)


def encoder_layer(layer, src, pos, reference_points, spatial_shapes, level_start_index, q=None, want_q=False):
    """-> (out, q_next). q: src + pos if the caller already has it (the previous layer's q_next); want_q: also return out + pos,
    written by the layer's last LayerNorm kernel (r04: neither `src + pos` nor the sum of the two gradients of `out` is an
    elementwise pass of its own any more)."""
    a = layer.self_attn
    # grad mode is always off INSIDE Function.forward: what the caller runs under is read here
    save_loc = torch.is_grad_enabled()
    geom = (a.n_heads, a.n_levels, a.n_points, layer.norm1.eps, layer.norm2.eps, bool(want_q), save_loc)
    return _EncoderLayerFn.apply(
        src, q, pos, reference_points, spatial_shapes.contiguous(), level_start_index.contiguous(), geom,
        a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
        a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias,
        layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
        layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias)
