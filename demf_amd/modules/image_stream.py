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


# the encoder layers on csrc/rows_gemm.hip + the raw-input MSDA kernel (embed_dims 256, 8 heads, 4 levels);
# False / DEMF_ENC_FUSED=0: the module path (ops.linear + torch elementwise), which the golden vectors of the
# REAL reference encoder pin - the fused path is tested against it
FUSED_LAYERS = bool(int(__import__("os").environ.get("DEMF_ENC_FUSED", "1")))
CHANNELS_LAST = bool(int(__import__("os").environ.get("DEMF_IMG_NHWC", "1")))        # A/B switch
# ResNet-50 + ChannelMapper on csrc/conv.hip (implicit-GEMM convolutions on NHWC rows, frozen BatchNorm folded into
# the pre-split weights, GroupNorm written straight into the token buffer); False / DEMF_IMG_CONV=0: the library
# convolutions (MIOpen)
CONV_KERNELS = bool(int(__import__("os").environ.get("DEMF_IMG_CONV", "1")))
MIOPEN_SEARCH = bool(int(__import__("os").environ.get("DEMF_MIOPEN_SEARCH", "0")))   # library path only
PRE_ADD = bool(int(__import__("os").environ.get("DEMF_ENC_PRE_ADD", "0")))           # A/B switch (measured neutral: 13.4 vs 13.2 ms)
SPLIT_FFN_LN = bool(int(__import__("os").environ.get("DEMF_ENC_SPLIT_LN", "1")))     # A/B switch


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
                level_sizes=tuple(int(v) for v in sizes),          # host copy: the MSDA kernel's LDS-resident levels
                keep=img_metas)
        return cache[key]

    # ---- the six layers on csrc/rows_gemm.hip: 5 launches per layer, no library GEMM, no elementwise pass ----
    def _fused_ok(self, tokens):
        """The kernel path hard-codes the reference's encoder layer (demf_votenet.py:33-40): post-norm order
        (self_attn, norm, ffn, norm), a 2-fc ReLU FFN with identity, LayerNorm(256), eval mode (no dropout).
        Anything else takes the module path."""
        if not (FUSED_LAYERS and tokens.is_cuda and tokens.dtype == torch.float32 and self.embed_dims == 256
                and not self.training):
            return False
        for layer in self.encoder.layers:
            if not (type(layer) is EncoderLayer and len(layer.attentions) == 1 and len(layer.ffns) == 1
                    and len(layer.norms) == 2 and not layer.training):
                return False
            a, f = layer.attentions[0], layer.ffns[0]
            if not (type(f) is FFN and len(f.layers) == 3 and isinstance(f.layers[0][0], nn.Linear)
                    and isinstance(f.layers[0][1], nn.ReLU) and isinstance(f.layers[1], nn.Linear)
                    and isinstance(f.layers[2], nn.Dropout)):
                return False
            if not all(isinstance(n, nn.LayerNorm) and tuple(n.normalized_shape) == (256,) for n in layer.norms):
                return False
            if not (a.num_heads == 8 and a.num_levels == 4 and a.num_points in (2, 4) and a.embed_dims == 256
                    and f.layers[0][0].out_features % 128 == 0 and f.layers[1].out_features == 256):
                return False
        return tokens.shape[1] * 1024 < 2 ** 31

    def _layer_pack(self, layer, planes):
        """The layer's frozen weights as the kernels take them: [sampling_offsets; attention_weights; value_proj]
        stacked into one (640, 256) projection, every weight pre-split into ``planes`` bf16 planes
        (ops.split_planes).  Cached per (weight versions, planes): a load_state_dict rebuilds it."""
        from .. import ops
        a, f, n = layer.attentions[0], layer.ffns[0], layer.norms
        fc0, fc1 = f.layers[0][0], f.layers[1]
        prm = [a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight,
               a.attention_weights.bias, a.value_proj.weight, a.value_proj.bias, a.output_proj.weight,
               a.output_proj.bias, fc0.weight, fc0.bias, fc1.weight, fc1.bias, n[0].weight, n[0].bias,
               n[1].weight, n[1].bias]
        key = (planes,) + tuple((q.data_ptr(), q._version) for q in prm)
        pk = layer.__dict__.get("_pack")
        if pk is None or pk["key"] != key:
            c = lambda t: t.detach().float().contiguous()
            # the value columns start at a column-tile boundary (128): zero rows pad [offsets | logits] up to it
            n_q = prm[0].shape[0] + prm[2].shape[0]
            v0 = (n_q + 127) // 128 * 128
            zw = prm[0].new_zeros(v0 - n_q, prm[0].shape[1])
            pk = dict(key=key, lgt0=prm[0].shape[0], v0=v0,
                      w_in=ops.split_planes(torch.cat([c(prm[0]), c(prm[2]), zw, c(prm[4])], 0), planes),
                      b_in=torch.cat([c(prm[1]), c(prm[3]), zw[:, 0], c(prm[5])], 0),
                      w_out=ops.split_planes(c(prm[6]), planes), b_out=c(prm[7]),
                      w0=ops.split_planes(c(prm[8]), planes), b0=c(prm[9]),
                      w1=ops.split_planes(c(prm[10]), planes), b1=c(prm[11]),
                      g1=c(prm[12]), be1=c(prm[13]), g2=c(prm[14]), be2=c(prm[15]),
                      eps1=float(n[0].eps), eps2=float(n[1].eps))
            layer.__dict__["_pack"] = pk
        return pk

    def _fused_layers(self, tokens, pos, st):
        from .. import ops, _ffi
        B, S, C = tokens.shape
        R = B * S
        planes = 1 if ops.get_compute_dtype() == "bf16" else 3
        x = tokens.reshape(R, C).contiguous()
        pos = pos.expand(B, S, C).reshape(R, C).contiguous()
        mask = st["mask_flatten"].reshape(R).contiguous()
        ref = st["reference_points"].contiguous()                    # (B,S,L,2)
        F = self.encoder.layers[0].ffns[0].layers[0][0].out_features
        new = lambda n: torch.empty((R, n), dtype=torch.float32, device=x.device)
        samp, x1, hid, xn, raw = new(C), new(C), new(F), new(C), None
        st_ = torch.cuda.current_stream().cuda_stream
        # q = x + pos as its own operand (PRE_ADD): written by the pass that produces x (the previous layer's
        # residual + LayerNorm pass; one add for layer 0) instead of re-added inside every column tile of the
        # input projection - the projection then needs no addend registers and runs on the 4-wave kernel
        q = new(C) if PRE_ADD and SPLIT_FFN_LN else None
        if q is not None:
            _ffi.call("demf_rows_ln_pos_f32", R, C, x.data_ptr(), None, None, None, 0.0, pos.data_ptr(), None,
                      q.data_ptr(), st_)
        nl = len(self.encoder.layers)
        for li, layer in enumerate(self.encoder.layers):
            a = layer.attentions[0]
            pk = self._layer_pack(layer, planes)
            v0 = pk["v0"]
            if raw is None or raw.shape[1] != v0 + C:
                raw = new(v0 + C)
            # [offsets | logits] from x + pos, value from x (padding rows zeroed): one launch
            if q is not None:
                ops.rows_gemm(x, pk["w_in"], pk["b_in"], raw, a2=q, a2_cols=v0, a2_replace=True, row_mask=mask,
                              mask_col0=v0)
            else:
                ops.rows_gemm(x, pk["w_in"], pk["b_in"], raw, a2=pos, a2_cols=v0, row_mask=mask, mask_col0=v0)
            ops.msda_fwd_raw(raw, v0, 0, pk["lgt0"], ref, st["spatial_shapes"], st["level_start_index"], B, S,
                             a.num_heads, C // a.num_heads, a.num_points, samp, level_sizes=st.get("level_sizes"))
            ops.rows_gemm(samp, pk["w_out"], pk["b_out"], x1, ln=(x, pk["g1"], pk["be1"], pk["eps1"]))
            ops.rows_gemm(x1, pk["w0"], pk["b0"], hid, relu=True)
            # K = 1024 with the LayerNorm epilogue runs one 128 x 256-tile workgroup per CU (~1.0 ms); as a
            # 128-column-tile launch (several workgroups per CU) + one residual / LayerNorm pass: 0.43 + 0.08 ms
            if SPLIT_FFN_LN:
                ops.rows_gemm(hid, pk["w1"], pk["b1"], samp)
                _ffi.call("demf_rows_ln_pos_f32", R, C, samp.data_ptr(), x1.data_ptr(), pk["g2"].data_ptr(),
                          pk["be2"].data_ptr(), pk["eps2"], pos.data_ptr() if q is not None else None, xn.data_ptr(),
                          q.data_ptr() if (q is not None and li + 1 < nl) else None, st_)
            else:
                ops.rows_gemm(hid, pk["w1"], pk["b1"], xn, ln=(x1, pk["g2"], pk["be2"], pk["eps2"]))
            x, xn = xn, x
        return x.view(B, S, C)

    @torch.no_grad()
    def forward_tokens(self, mlvl_feats, img_metas):
        """-> dict(tokens (B,S,C), spatial [(h,w)...], mask_flatten (B,S), valid_ratios (B,L,2))."""
        if isinstance(mlvl_feats, dict):
            # the neck's pyramid already as channels-last tokens (ImageStream on csrc/conv.hip): no concat copy
            spatial, tokens = [tuple(sp) for sp in mlvl_feats["spatial"]], mlvl_feats["tokens"]
            st = self._static(img_metas, spatial, tokens.device)
        else:
            spatial = [tuple(f.shape[-2:]) for f in mlvl_feats]
            st = self._static(img_metas, spatial, mlvl_feats[0].device)
            tokens = torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1)       # (B,S,C)
        pos = torch.cat([p + self.level_embeds[l].view(1, 1, -1) for l, p in enumerate(st["pos"])], 1)
        if self._fused_ok(tokens):
            tokens = self._fused_layers(tokens, pos, st)
        else:
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

    def _pyramid(self, img):
        """ResNet-50 + ChannelMapper (library convolutions).  On the GPU in channels-last memory format: MIOpen's
        NHWC kernels are faster here (8 x 3 x 800 x 1120 fp32: 18.8 -> 16.5 ms) and the neck's (B,256,h,w) outputs
        then ARE (B,h,w,256) in memory, so the encoder's token concat reads contiguous rows."""
        if img.is_cuda:
            from .. import ops
            ops.library_fallback("image backbone / neck at widths csrc/conv.hip does not take (F.conv2d -> MIOpen)")
        if CHANNELS_LAST and img.is_cuda:
            if self.__dict__.get("_nhwc_for") != img.device:
                self.img_backbone.to(memory_format=torch.channels_last)
                self.img_neck.to(memory_format=torch.channels_last)
                self.__dict__["_nhwc_for"] = img.device
            img = img.contiguous(memory_format=torch.channels_last)
            # MIOpen's immediate mode picks slow NHWC kernels (22.7 ms); with its search - first call per shape -
            # the same convolutions run in 16.5 ms.  The search flag is process-global (it races with any other
            # thread that runs convolutions), so it is opt-in: DEMF_MIOPEN_SEARCH=1.  (This library path is the
            # fall-back for widths the kernels of csrc/conv.hip do not take, and the A/B reference of bench.py.)
            if not MIOPEN_SEARCH:
                return self.img_neck(self.img_backbone(img))
            prev = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                return self.img_neck(self.img_backbone(img))
            finally:
                torch.backends.cudnn.benchmark = prev
        return self.img_neck(self.img_backbone(img))

    # ---- ResNet-50 + ChannelMapper on csrc/conv.hip -------------------------------------------------------------
    def _conv_ok(self, img):
        """The kernel path needs channels-last friendly widths (every convolution input a multiple of 32 channels -
        the 3-channel stem has its own form -, every output a multiple of 64), the reference's structure
        (imvotenet_image.py:3-20: bottlenecks, 7x7 s2 stem, 256-channel neck with GroupNorm whose groups are 4 .. 64
        consecutive channels) and frozen statistics (eval mode)."""
        if not (CONV_KERNELS and img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3):
            return False
        bb, nk = self.img_backbone, self.img_neck
        if bb.training or nk.training or tuple(bb.conv1.kernel_size) != (7, 7) or bb.conv1.out_channels % 64:
            return False
        for m in bb.modules():
            if isinstance(m, nn.Conv2d) and m is not bb.conv1 and (m.in_channels % 32 or m.out_channels % 64):
                return False
        for c in list(nk.convs) + list(nk.extra_convs):
            cg = c.conv.out_channels // c.gn.num_groups
            if c.conv.in_channels % 32 or c.conv.out_channels != 256 or cg not in (4, 8, 16, 32, 64):
                return False
        return True

    def _conv_pack(self, planes):
        """Every convolution's weight as demf_conv_nhwc_f32 takes it - (Cout, KH, KW, Cin) reduction order, the frozen
        BatchNorm scale folded in (bias = beta - mean * scale), pre-split into ``planes`` bf16 planes.  Cached per
        (parameter / statistic versions, planes): load_state_dict rebuilds it."""
        from .. import ops
        bb, nk = self.img_backbone, self.img_neck
        tensors = [t for m in (bb, nk) for t in list(m.parameters()) + list(m.buffers())]
        key = (planes,) + tuple((t.data_ptr(), t._version) for t in tensors)
        pk = self.__dict__.get("_conv_pack_cache")
        if pk is not None and pk["key"] == key:
            return pk

        def fold(conv, bn, stem=False):
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
            bias = (bn.bias - bn.running_mean * scale).detach().float().contiguous()
            w = (ops.stem_weight_planes if stem else ops.conv_weight_planes)(conv.weight, planes, scale)
            return dict(w=w, b=bias, k=conv.kernel_size[0], s=conv.stride[0], p=conv.padding[0])
        pk = dict(key=key, stem=fold(bb.conv1, bb.bn1, stem=True), blocks=[], neck=[])
        for i in range(4):
            for blk in getattr(bb, f"layer{i + 1}"):
                pk["blocks"].append(dict(
                    c1=fold(blk.conv1, blk.bn1), c2=fold(blk.conv2, blk.bn2), c3=fold(blk.conv3, blk.bn3),
                    ds=None if blk.downsample is None else fold(blk.downsample[0], blk.downsample[1]), stage=i,
                    last=blk is getattr(bb, f"layer{i + 1}")[-1]))
        for c in list(nk.convs) + list(nk.extra_convs):
            assert c.conv.bias is None
            pk["neck"].append(dict(w=ops.conv_weight_planes(c.conv.weight, planes), k=c.conv.kernel_size[0],
                                   s=c.conv.stride[0], p=c.conv.padding[0], g=c.gn.num_groups, eps=float(c.gn.eps),
                                   gamma=c.gn.weight.detach().float().contiguous(),
                                   beta=c.gn.bias.detach().float().contiguous()))
        self.__dict__["_conv_pack_cache"] = pk
        return pk

    def _pyramid_tokens(self, img):
        """ResNet-50 + ChannelMapper on this package's convolution kernels -> dict(tokens (B,S,256), spatial): the
        neck's four levels normalised straight into the encoder's channels-last token buffer (no NCHW tensor, no
        flatten / concat copy, no MIOpen search on the first call)."""
        from .. import ops
        planes = 1 if ops.get_compute_dtype() == "bf16" else 3
        pk = self._conv_pack(planes)
        cv = lambda x, c, resid=None, relu=False: ops.conv_nhwc(x, c["w"], c.get("b"), c["k"], c["k"], c["s"], c["p"],
                                                                resid=resid, relu=relu)
        x = ops.maxpool3x3s2_nhwc(ops.conv_stem7(img.contiguous(), pk["stem"]["w"], pk["stem"]["b"], relu=True))
        outs = []
        for blk in pk["blocks"]:
            idn = x if blk["ds"] is None else cv(x, blk["ds"])
            y = cv(cv(x, blk["c1"], relu=True), blk["c2"], relu=True)
            x = cv(y, blk["c3"], resid=idn, relu=True)
            if blk["last"] and blk["stage"] in self.img_backbone.out_indices:
                outs.append(x)
        nk = self.img_neck
        raws = [cv(o, c) for o, c in zip(outs, pk["neck"][:len(nk.convs)])]
        for j, c in enumerate(pk["neck"][len(nk.convs):]):
            raws.append(cv(outs[-1] if j == 0 else raws[-1], c))
        # (an extra level beyond the first reads the NORMALISED previous level upstream: ChannelMapper.forward;
        # the reference has exactly one extra level, fed by the last backbone map)
        assert len(nk.extra_convs) <= 1, "more than one extra level: normalised intermediate needed"
        spatial = [tuple(r.shape[1:3]) for r in raws]
        S = sum(h * w for h, w in spatial)
        tokens = torch.empty((img.shape[0], S, raws[0].shape[3]), dtype=torch.float32, device=img.device)
        row0 = 0
        for r, c in zip(raws, pk["neck"]):
            ops.groupnorm_nhwc_into(r, c["g"], c["gamma"], c["beta"], c["eps"], tokens, row0)
            row0 += r.shape[1] * r.shape[2]
        return dict(tokens=tokens, spatial=spatial)

    def pyramid(self, img):
        """The neck's output in the form the encoder takes: token dict (kernel path) or the list of NCHW maps."""
        return self._pyramid_tokens(img) if self._conv_ok(img) else self._pyramid(img)

    @torch.no_grad()
    def tokens(self, img, img_metas):
        return self.img_encoder.forward_tokens(self.pyramid(img), img_metas)

    @torch.no_grad()
    def forward(self, img, img_metas):
        return self.img_encoder(self.pyramid(img), img_metas)
