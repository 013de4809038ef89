"""The frozen image stream of DeMFVoteNet.

``DeMFVoteNet.extract_img_feat`` (demf/modeling/detectors/demfnet.py:124-132) runs, under
``torch.no_grad`` and in eval mode, ResNet-50 -> ChannelMapper -> DeformableDetrEncoder
(configs/deformdetr/imvotenet_image.py:3-20, configs/demf/demf_votenet.py:28-47) and hands the
4-level, 256-channel pyramid to the head.  Here:

  * the convolutional part (ResNet-50, ChannelMapper) is PyTorch-ROCm library code (MIOpen /
    hipBLASLt): plain dense convolutions, frozen, outside the north-star kernels;
  * the encoder (demf/modeling/layers/deform_detr_encoder.py:68-154, 6 x [multi-scale deformable
    self-attention over all 18 609 tokens, LayerNorm, FFN, LayerNorm]) runs on this package's MSDA
    kernel and ``ops.linear``, batch-first so that no token tensor is ever transposed, and
  * it **emits channels-last tokens** (B, S, 256) next to the reference's NCHW views, so the
    head's ``prepare_decoder_inputs`` flatten+concat copy (class_agnostic_vote_head.py:570-591,
    19 MB per scene) disappears when the two are used together.

Module / parameter names follow mmdet's, so released ``img_backbone.* / img_neck.* /
img_encoder.*`` checkpoints load.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..geometry import level_masks
from .transformer import FFN, MultiScaleDeformableAttention


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * 4)) if downsample else None

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = F.relu(self.bn2(self.conv2(out)), inplace=True)
        out = self.bn3(self.conv3(out))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)), inplace=True)


class ResNet50(nn.Module):
    """mmdet ``ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), style='pytorch',
    norm_eval=True)`` - imvotenet_image.py:3-12."""

    def __init__(self, out_indices=(1, 2, 3), base=64, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.out_indices = out_indices
        self.conv1 = nn.Conv2d(3, base, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(base)
        inplanes = base
        for i, n in enumerate(blocks):
            planes, stride = base * 2 ** i, 1 if i == 0 else 2
            layers = [Bottleneck(inplanes, planes, stride, downsample=True)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes) for _ in range(n - 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layers))
        self.out_channels = [base * 4 * 2 ** i for i in out_indices]

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, stride=2, padding=1)
        outs = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _ConvGN(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, groups=32):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.gn = nn.GroupNorm(groups, cout)

    def forward(self, x):
        return self.gn(self.conv(x))


class ChannelMapper(nn.Module):
    """mmdet ``ChannelMapper(in_channels, kernel_size=1, out_channels=256, act_cfg=None,
    norm_cfg=GN32, num_outs=4)`` - imvotenet_image.py:13-20: one 1x1 conv + GN per input level,
    plus 3x3 stride-2 conv + GN levels on top of the last INPUT."""

    def __init__(self, in_channels, out_channels=256, num_outs=4, groups=32):
        super().__init__()
        self.convs = nn.ModuleList([_ConvGN(c, out_channels, 1, groups=groups) for c in in_channels])
        self.extra_convs = nn.ModuleList()
        for i in range(len(in_channels), num_outs):
            cin = in_channels[-1] if i == len(in_channels) else out_channels
            self.extra_convs.append(_ConvGN(cin, out_channels, 3, stride=2, padding=1, groups=groups))

    def forward(self, inputs):
        outs = [c(x) for c, x in zip(self.convs, inputs)]
        for i, c in enumerate(self.extra_convs):
            outs.append(c(inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)


def sine_positional_encoding(mask, num_feats, temperature=10000, normalize=True,
                             scale=2 * math.pi, eps=1e-6, offset=0.0):
    """mmdet SinePositionalEncoding.forward(mask (B,H,W) bool) -> (B,H,W,2*num_feats), i.e. the
    upstream (B,2F,H,W) result in channels-last order [pos_y | pos_x]."""
    not_mask = 1 - mask.to(torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32, device=mask.device)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = mask.shape
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3)


class EncoderLayer(nn.Module):
    """mmcv BaseTransformerLayer with operation_order ('self_attn','norm','ffn','norm')
    (demf_votenet.py:33-40), batch-first."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4,
                 feedforward_channels=1024, ffn_dropout=0.1, attn_dropout=0.1):
        super().__init__()
        self.attentions = nn.ModuleList([MultiScaleDeformableAttention(
            embed_dims, num_heads, num_levels, num_points, dropout=attn_dropout, batch_first=True)])
        self.ffns = nn.ModuleList([FFN(embed_dims, feedforward_channels, ffn_dropout)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims), nn.LayerNorm(embed_dims)])

    def forward(self, query, query_pos, key_padding_mask, **kw):
        query = self.attentions[0](query, None, None, None, query_pos=query_pos,
                                   key_padding_mask=key_padding_mask, **kw)
        query = self.norms[0](query)
        query = self.ffns[0](query, None)
        return self.norms[1](query)


class _Layers(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class DeformableDetrEncoder(nn.Module):
    """demf/modeling/layers/deform_detr_encoder.py:12-154.  ``forward(mlvl_feats, img_metas)``
    returns the reference's list of (B,C,H_l,W_l) maps (views of the token buffer);
    ``forward_tokens`` returns the tokens themselves."""

    def __init__(self, num_layers=6, embed_dims=256, num_heads=8, num_feature_levels=4,
                 num_points=4, feedforward_channels=1024, ffn_dropout=0.1, num_feats=128,
                 normalize=True, offset=-0.5):
        super().__init__()
        self.encoder = _Layers([EncoderLayer(embed_dims, num_heads, num_feature_levels, num_points,
                                             feedforward_channels, ffn_dropout)
                                for _ in range(num_layers)])
        self.level_embeds = nn.Parameter(torch.zeros(num_feature_levels, embed_dims))
        self.pe = dict(num_feats=num_feats, normalize=normalize, offset=offset)
        self.embed_dims = embed_dims

    def init_weights(self):
        for m in self.modules():                                  # :31-36
            if hasattr(m, "weight") and isinstance(m.weight, torch.Tensor) and m.weight.dim() > 1:
                nn.init.xavier_uniform_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0.0)
        nn.init.normal_(self.level_embeds)

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):   # :48-66
        refs = []
        for lvl, (H, W) in enumerate(spatial_shapes):
            ref_y, ref_x = torch.meshgrid(
                torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            refs.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(refs, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def _static(self, img_metas, spatial, device):
        """Everything that depends on the metas and the level shapes only: padding masks,
        positional encodings (+ level embeds are added per call), valid ratios, reference points."""
        key = (id(img_metas), tuple(spatial), str(device))
        cache = self.__dict__.setdefault("_meta_cache", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            masks = [torch.as_tensor(m, device=device) for m in level_masks(img_metas, spatial)]
            pos = [sine_positional_encoding(m, **self.pe).flatten(1, 2) for m in masks]   # (B,hw,C)
            ratios = []
            for m in masks:                                        # :38-46
                _, H, W = m.shape
                vh = torch.sum(~m[:, :, 0], 1).float() / H
                vw = torch.sum(~m[:, 0, :], 1).float() / W
                ratios.append(torch.stack([vw, vh], -1))
            valid_ratios = torch.stack(ratios, 1)
            sizes = [h * w for h, w in spatial]
            cache[key] = dict(
                mask_flatten=torch.cat([m.flatten(1) for m in masks], 1),
                pos=pos, valid_ratios=valid_ratios,
                reference_points=self.get_reference_points(spatial, valid_ratios, device),
                spatial_shapes=torch.as_tensor(list(spatial), dtype=torch.long, device=device),
                level_start_index=torch.as_tensor([0] + list(np.cumsum(sizes)[:-1]),
                                                  dtype=torch.long, device=device),
                keep=img_metas)
        return cache[key]

    @torch.no_grad()
    def forward_tokens(self, mlvl_feats, img_metas):
        """-> dict(tokens (B,S,C), spatial [(h,w)...], mask_flatten (B,S), valid_ratios (B,L,2))."""
        spatial = [tuple(f.shape[-2:]) for f in mlvl_feats]
        st = self._static(img_metas, spatial, mlvl_feats[0].device)
        tokens = torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1)       # (B,S,C)
        pos = torch.cat([p + self.level_embeds[l].view(1, 1, -1) for l, p in enumerate(st["pos"])], 1)
        for layer in self.encoder.layers:
            tokens = layer(tokens, pos, st["mask_flatten"], reference_points=st["reference_points"],
                           spatial_shapes=st["spatial_shapes"],
                           level_start_index=st["level_start_index"])
        return dict(tokens=tokens, spatial=spatial, mask_flatten=st["mask_flatten"],
                    valid_ratios=st["valid_ratios"])

    def forward(self, mlvl_feats, img_metas):
        out = self.forward_tokens(mlvl_feats, img_metas)
        memory = out["tokens"].permute(0, 2, 1)                     # (B,C,S), as :141
        B, C = memory.shape[:2]
        feats, start = [], 0
        for h, w in out["spatial"]:
            feats.append(memory[:, :, start:start + h * w].reshape(B, C, h, w))
            start += h * w
        return feats


class ImageStream(nn.Module):
    """img_backbone + img_neck + img_encoder, frozen.  ``forward(img (B,3,H,W), img_metas)`` ->
    the 4-level pyramid as the reference's ``extract_img_feat``; ``tokens(img, img_metas)`` -> the
    channels-last form the head of this package consumes without the flatten copy."""

    def __init__(self, base=64, blocks=(3, 4, 6, 3), embed_dims=256, num_layers=6, num_heads=8,
                 feedforward_channels=1024, gn_groups=32, num_feats=None):
        super().__init__()
        self.img_backbone = ResNet50((1, 2, 3), base, blocks)
        self.img_neck = ChannelMapper(self.img_backbone.out_channels, embed_dims, 4, gn_groups)
        self.img_encoder = DeformableDetrEncoder(
            num_layers, embed_dims, num_heads, 4, 4, feedforward_channels, 0.1,
            num_feats if num_feats is not None else embed_dims // 2)
        self.img_encoder.init_weights()
        for p in self.parameters():                                 # demfnet.py:103-122: frozen
            p.requires_grad_(False)
        self.eval()

    def train(self, mode=True):
        return super().train(False)                                 # always eval (norm_eval, no_grad)

    @torch.no_grad()
    def tokens(self, img, img_metas):
        return self.img_encoder.forward_tokens(self.img_neck(self.img_backbone(img)), img_metas)

    @torch.no_grad()
    def forward(self, img, img_metas):
        return self.img_encoder(self.img_neck(self.img_backbone(img)), img_metas)
