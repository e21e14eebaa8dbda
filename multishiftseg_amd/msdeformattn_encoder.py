"""The callers of the MSDeformAttn op inside Mask2Former's pixel decoder.

Host-side mirror of ``MSDeformAttnTransformerEncoderOnly`` / ``...EncoderLayer`` / ``...Encoder``
(lib/network/mask2former/modeling/pixel_decoder/msdeformattn.py:21-153) and of
``PositionEmbeddingSine`` (modeling/transformer_decoder/position_encoding.py:13-52): same class and
parameter names (``encoder.layers.N.self_attn.sampling_offsets.weight``, ``level_embed`` ...), so the
encoder slice of a Mask2Former checkpoint loads unchanged, and the same op inputs: level order as
given (res5 -> res3 in the decoder, msdeformattn.py:319), ``spatial_shapes`` int64 [L,2],
``level_start_index``, reference points at pixel centres. The detectron2-dependent shell around it
(``MSDeformAttnPixelDecoder``: input_proj convs + GroupNorm, FPN) is out of scope.

The attention itself is the HIP op (multishiftseg_amd.ms_deform_attn.MSDeformAttn); the Linears run on the repository's
fp32 MFMA GEMM kernels (multishiftseg_amd/linear.py); residual add + LayerNorm are one HIP kernel (csrc/norm.hip). Masks are all-False in the reference (msdeformattn.py:62), so valid
ratios are 1 and are folded away here.
"""
import copy
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import encoder_layer
from . import kernels as K
from .linear import ffn_relu, linear
from .ms_deform_attn import MSDeformAttn


class PositionEmbeddingSine(nn.Module):
    """Sine/cosine position code over a [N,C,H,W] map (position_encoding.py:13-52), mask-free."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        n, _, h, w = x.shape
        ys = torch.arange(1, h + 1, dtype=torch.float32, device=x.device)
        xs = torch.arange(1, w + 1, dtype=torch.float32, device=x.device)
        if self.normalize:
            eps = 1e-6
            ys = ys / (ys[-1] + eps) * self.scale
            xs = xs / (xs[-1] + eps) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)

        def code(v):                              # [len] -> [len, num_pos_feats], sin on even / cos on odd slots
            p = v[:, None] / dim_t
            return torch.stack((p[:, 0::2].sin(), p[:, 1::2].cos()), dim=2).flatten(1)
        py = code(ys)[:, None, :].expand(h, w, -1)
        px = code(xs)[None, :, :].expand(h, w, -1)
        return torch.cat((py, px), dim=2).permute(2, 0, 1)[None].expand(n, -1, -1, -1).contiguous()


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    """self-attention (deformable) + FFN, post-norm (msdeformattn.py:92-131)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        if os.environ.get("MSS_ENCODER_FUSED", "1") != "0" and \
                encoder_layer.eligible(self, src, pos, reference_points, spatial_shapes, padding_mask):
            # the whole layer as ONE autograd node: same kernels, the backward's gradient sums in GEMM epilogues (encoder_layer.py)
            return encoder_layer.encoder_layer(self, src, pos, reference_points, spatial_shapes, level_start_index)[0]
        q = src if pos is None else src + pos
        # residual add + LayerNorm in one HIP pass (csrc/norm.hip); Dropout is the identity at the reference's p = 0.0
        src = K.add_layernorm(src, self.dropout1(
            self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask)), self.norm1)
        if self.activation is F.relu and (self.dropout2.p == 0.0 or not self.training):
            # ReLU rides in the GEMM epilogue, its backward in the epilogue of linear2's data gradient (linear.py)
            ffn = ffn_relu(src, self.linear1, self.linear2)
        else:
            if self.activation is F.relu:
                hidden = linear(src, self.linear1.weight, self.linear1.bias, relu=True)
            else:
                hidden = self.activation(linear(src, self.linear1.weight, self.linear1.bias))
            ffn = linear(self.dropout2(hidden), self.linear2.weight, self.linear2.bias)
        return K.add_layernorm(src, self.dropout3(ffn), self.norm2)


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres ((x+0.5)/W, (y+0.5)/H) of every query's own level, shared by all target
        levels (msdeformattn.py:140-153) -> [N, sum(HW), L, 2]."""
        pts = []
        host = getattr(spatial_shapes, "_mss_host", None)          # known on the host: no device read (hipGraph capture)
        for lvl, (h, w) in enumerate(host if host is not None else spatial_shapes.tolist()):
            ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5)
            xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            ry = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            rx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            pts.append(torch.stack((rx, ry), -1))
        ref = torch.cat(pts, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        ref = self.get_reference_points(spatial_shapes, valid_ratios, src.device)
        out, q = src, None
        fused = os.environ.get("MSS_ENCODER_FUSED", "1") != "0"
        carry = os.environ.get("MSS_ENCODER_CARRY_Q", "1") != "0"
        ok = [fused and encoder_layer.eligible(layer, src, pos, ref, spatial_shapes, padding_mask) for layer in self.layers]
        for i, layer in enumerate(self.layers):
            if ok[i]:
                # the layer's last LayerNorm also writes the NEXT layer's query out + pos, and its backward adds the two gradients of
                # `out` while loading them (r04): no elementwise pass for either
                want_q = carry and i + 1 < len(self.layers) and ok[i + 1]
                out, q = encoder_layer.encoder_layer(layer, out, pos, ref, spatial_shapes, level_start_index, q=q, want_q=want_q)
            else:
                out, q = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask), None
        return out


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    """Flattens the multi-scale maps, builds the op's index tensors and runs the encoder
    (msdeformattn.py:21-89)."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        layer = MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                    num_feature_levels, nhead, enc_n_points)
        self.encoder = MSDeformAttnTransformerEncoder(layer, num_encoder_layers)
        self.level_embed = nn.Parameter(torch.empty(num_feature_levels, d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)

    def forward(self, srcs, pos_embeds):
        shapes = [(s.shape[2], s.shape[3]) for s in srcs]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        return self.forward_tokens(src, pos_embeds, shapes)

    def forward_tokens(self, src, pos_embeds, shapes, pe_layer=None):
        """Same as forward for a source that already is the token buffer [N, sum(HW), C] in level order (the pixel decoder
        writes its GroupNorm outputs there directly).

        pe_layer (with pos_embeds None): the mask-free sine code depends on the level sizes only, not on the sample, so it is
        built ONCE per (sizes, device) as [1, sum(HW), C] tokens and the position input of every layer is the broadcastable
        [1, S, C] tensor `sine + level_embed[level]` -- 10 MB instead of N x 10 MB (round 4: at 16 x 704^2 the per-forward
        sin / cos / stack / cat kernels, the three [N, HW, C] adds, and in the backward five 166 MB gradient accumulations plus
        a 166 MB reduction for level_embed were 2 ms of ATen kernels per iteration). Same values as the reference's
        `pos_embed + level_embed` (msdeformattn.py:70-76); level_embed's gradient = the batch-and-level sums of d(pos)."""
        if pos_embeds is None:
            if pe_layer is None:
                raise ValueError("forward_tokens needs pos_embeds or pe_layer")
            key = ("sine", tuple(tuple(int(v) for v in hw) for hw in shapes), str(src.device), pe_layer.num_pos_feats,
                   pe_layer.temperature, pe_layer.normalize, pe_layer.scale)
            # the sine tokens are MEGABYTES per shape (10 MB at 704^2, 44 MB at 1024x2048), unlike the index tensors below: their own,
            # bounded cache (least recently used of 4 shapes is dropped), so a sweep over many image sizes does not grow device
            # memory without bound. A captured hipGraph is unaffected: `pos` below is a fresh tensor made from them,
            # and GraphedFeatures keeps its own references.
            cache = self.__dict__.setdefault("_sine_cache", {})
            sine = cache.pop(key, None)
            if sine is None:
                with torch.no_grad():
                    sine = [pe_layer(torch.empty((1, 1, int(h), int(w)), device=src.device)).flatten(2).transpose(1, 2).contiguous()
                            for h, w in shapes]
                while len(cache) >= 4:
                    cache.pop(next(iter(cache)))
            cache[key] = sine                                        # (re)inserted last = most recently used
            pos = torch.cat([p + self.level_embed[lvl].view(1, 1, -1) for lvl, p in enumerate(sine)], 1)
        else:
            pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1)
                             for lvl, p in enumerate(pos_embeds)], 1)
        # the op's index tensors depend on the level sizes only: built once per (sizes, batch, device) -- also what makes the
        # forward capturable into a hipGraph (no host-to-device copy of the shape list inside the capture)
        key = (tuple(tuple(int(v) for v in hw) for hw in shapes), int(src.shape[0]), str(src.device))
        cache = self.__dict__.setdefault("_index_cache", {})
        ent = cache.get(key)
        if ent is None:
            spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=src.device)
            spatial_shapes._mss_host = [tuple(int(v) for v in hw) for hw in shapes]  # host copy: launch grids, tile geometry
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            valid_ratios = torch.ones((src.shape[0], len(shapes), 2), dtype=torch.float32, device=src.device)
            # Entries are never freed behind a reader's back: a captured hipGraph (msdeformattn_decoder.GraphedFeatures)
            # reads these three tensors by raw pointer. They are a few hundred bytes per (sizes, batch) combination, so the
            # cache simply keeps them all; GraphedFeatures additionally holds its own references (`_keep`).
            ent = cache[key] = (spatial_shapes, level_start_index, valid_ratios)
        spatial_shapes, level_start_index, valid_ratios = ent
        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, None)
        return memory, spatial_shapes, level_start_index
