"""Mask2Former's pixel decoder shell around the MSDeformAttn encoder.

Host-side mirror of ``MSDeformAttnPixelDecoder`` (lib/network/mask2former/modeling/pixel_decoder/msdeformattn.py:164-358):
same constructor keywords, same parameter names (``input_proj.{i}.0/1``, ``transformer.*``, ``adapter_1`` / ``layer_1``
with their ``.norm`` children as detectron2's Conv2d wrapper names them, ``mask_features``), same
``forward_features(features) -> (mask_features, out[0], multi_scale_features)`` with NCHW maps at the boundary.
detectron2 is not a dependency: ``input_shape`` values only need ``.channels`` and ``.stride`` (``ShapeSpec`` below).

What runs (all in libmss_hip.so, NHWC inside):
  * input_proj: 1x1 conv (+bias in the GEMM epilogue) -> GroupNorm(32) whose output is written straight into the
    encoder's token buffer [N, sum(HW), 256] at the level's offset -- the reference's flatten/transpose/cat copies
    (msdeformattn.py:66-79) do not exist;
  * the encoder (msdeformattn_encoder.py) on that buffer;
  * the FPN top-down step for the stride-4 level: lateral 1x1 conv -> GroupNorm, bilinear(align_corners=False) + add in
    one kernel reading the finest encoder level in place, 3x3 conv (Winograd / implicit GEMM) -> GroupNorm+ReLU;
  * mask_features 1x1 conv; NHWC -> NCHW only for the five returned maps.
Forward only: the backward of the shell is not built yet (asking for parameter gradients raises).
"""
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from . import kernels as K
from .msdeformattn_encoder import MSDeformAttnTransformerEncoderOnly, PositionEmbeddingSine

ShapeSpec = namedtuple("ShapeSpec", ["channels", "stride"])


class _NormConv2d(nn.Conv2d):
    """Parameter container with the layout of detectron2.layers.Conv2d: an optional ``norm`` child and an activation
    applied after it (the arithmetic is done by the HIP kernels, never by this module's forward)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation


def _get_norm(norm, channels):
    if norm in (None, ""):
        return None
    if norm == "GN":
        return nn.GroupNorm(32, channels)                 # detectron2 get_norm("GN", c)
    raise NotImplementedError(f"norm={norm!r}: the HIP path implements GroupNorm (the Mask2Former configs' 'GN') or none")


def _xavier_fill(conv):
    """fvcore c2_xavier_fill: kaiming_uniform_(a=1) weight, zero bias."""
    nn.init.kaiming_uniform_(conv.weight, a=1)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, input_shape, *, transformer_dropout, transformer_nheads, transformer_dim_feedforward,
                 transformer_enc_layers, conv_dim, mask_dim, norm=None, transformer_in_features, common_stride):
        super().__init__()
        tshape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        items = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, _ in items]                       # "res2" .. "res5"
        self.feature_strides = [v.stride for _, v in items]
        self.feature_channels = [v.channels for _, v in items]
        titems = sorted(tshape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, _ in titems]
        t_channels = [v.channels for _, v in titems]
        self.transformer_feature_strides = [v.stride for _, v in titems]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        if self.transformer_num_feature_levels > 1:
            projs = [nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in t_channels[::-1]]
        else:
            projs = [nn.Sequential(nn.Conv2d(t_channels[-1], conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim))]
        self.input_proj = nn.ModuleList(projs)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = _NormConv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        _xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral = _NormConv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=_get_norm(norm, conv_dim))
            output = _NormConv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=_get_norm(norm, conv_dim), activation=torch.relu)
            _xavier_fill(lateral)
            _xavier_fill(output)
            self.add_module(f"adapter_{idx + 1}", lateral)
            self.add_module(f"layer_{idx + 1}", output)
            lateral_convs.append(lateral)
            output_convs.append(output)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    # ---- pieces ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _bias_affine(conv):
        if conv.bias is None:
            return None
        return (torch.ones_like(conv.bias), conv.bias.detach())

    def _conv1x1(self, x_act, conv):
        return K.conv2d(x_act, K.packed(conv.weight), out_affine=self._bias_affine(conv))

    def _norm_act(self, y, conv):
        """detectron2 Conv2d.forward after the convolution: norm, then activation."""
        if conv.norm is not None:
            return K.groupnorm(y, conv.norm, relu=conv.activation is not None)
        if conv.activation is not None:
            raise NotImplementedError("activation without a norm does not occur in the pixel decoder")
        return y

    # ---- reference API ---------------------------------------------------------------------------------------------------
    def forward_features(self, features):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("MSDeformAttnPixelDecoder (multishiftseg_amd): the shell's backward (GroupNorm / FPN) is "
                                      "not built; call under torch.no_grad() or freeze the parameters")
        with torch.no_grad():
            return self._forward_features(features)

    def _forward_features(self, features):
        names = self.transformer_in_features[::-1]                      # res5 -> res3 (msdeformattn.py:319)
        xs = [features[f].float() for f in names]
        if not xs[0].is_cuda:
            raise RuntimeError("MSDeformAttnPixelDecoder (multishiftseg_amd) runs on an MI355X only; there is no CPU path")
        N = xs[0].shape[0]
        shapes = [(x.shape[2], x.shape[3]) for x in xs]
        starts = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])]).tolist()
        S, C = starts[-1], self.input_proj[0][1].num_channels
        dev = xs[0].device
        tokens = torch.empty((N, S, C), device=dev, dtype=torch.float32)
        pos = []
        for idx, x in enumerate(xs):
            conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
            y = self._conv1x1(K.nchw_to_act(x), conv)
            # GroupNorm output goes straight to rows [start, start + H*W) of every sample of the token buffer
            K.groupnorm(y, gn, out=tokens[0, starts[idx]:], out_sample_stride=S * C, out_ld=C)
            pos.append(self.pe_layer(x))
        memory, spatial_shapes, level_start_index = self.transformer.forward_tokens(tokens, pos, shapes)
        levels = [K.TokenLevel(memory, starts[i], *shapes[i]) for i in range(len(shapes))]
        out = list(levels)
        for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
            x = features[f].float()
            lateral, output = self.lateral_convs[idx], self.output_convs[idx]
            cur = self._norm_act(self._conv1x1(K.nchw_to_act(x), lateral), lateral)
            y = K.upsample_bilinear_add(out[-1], cur)
            y = K.conv3x3(y, output.weight) if output.bias is None else \
                K.conv2d(y, K.packed(output.weight), pad=1, out_affine=self._bias_affine(output))
            out.append(self._norm_act(y, output))
        multi_scale = [K.nhwc_to_nchw(o) for o in out[:self.maskformer_num_feature_levels]]
        last = out[-1]
        if isinstance(last, K.TokenLevel):      # no FPN level: mask_features reads the finest encoder level
            last = K.Act(K.nhwc_to_nchw(last).permute(0, 2, 3, 1).contiguous())
        mask = K.nhwc_to_nchw(self._conv1x1(last, self.mask_features))
        return mask, multi_scale[0], multi_scale

    def forward(self, features, targets=None):
        raise NotImplementedError("the reference only calls forward_features (msdeformattn.py:314)")
