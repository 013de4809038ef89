"""CPU ORACLE (test infrastructure, NOT a product path) - the reference's THIRD-PARTY
dependencies, restated in their own (channel-major) layout.

The reference (haoy945/DeMF) imports these from mmdet3d==0.18.1 / mmcv-full==1.3.18 /
mmdet==2.14.0 (requirements.txt:2-4), none of which is vendored or installed here, so every
class below restates the published upstream algorithm ("dep-recall": parity with upstream
binaries is unpinned; structure is corroborated by the trainable-parameter count 2 189 975
matching the reference config, and the operators by tests/test_oracle_kernels.py).
Sub-module names follow upstream so state dicts are interchangeable with demf_amd.modules.
Each class cites the reference line that instantiates or calls it.
"""
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import torch_ops as O


# ----------------------------------------------------------------------------- mmcv ConvModule
class ConvModule(nn.Module):
    """mmcv ConvModule(conv -> bn -> relu) for kernel size 1."""

    def __init__(self, cin, cout, dim, bias):
        super().__init__()
        if dim == 2:
            self.conv, self.bn = nn.Conv2d(cin, cout, 1, bias=bias), nn.BatchNorm2d(cout)
        else:
            self.conv, self.bn = nn.Conv1d(cin, cout, 1, bias=bias), nn.BatchNorm1d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


def _mlp(channels, dim, bias):
    seq = nn.Sequential()
    for i in range(len(channels) - 1):
        seq.add_module(f"layer{i}", ConvModule(channels[i], channels[i + 1], dim, bias))
    return seq


# ----------------------------------------------------------------------------- mmdet3d.ops
class PointSAModule(nn.Module):
    """mmdet3d PointSAModule / QueryAndGroup as built by build_sa_module
    (class_agnostic_vote_head.py:383) and PointNet2SASSG (demf_votenet.py:48-62)."""

    def __init__(self, num_point, radius, num_sample, mlp_channels, use_xyz=True,
                 normalize_xyz=False, **unused):
        super().__init__()
        self.num_point, self.radius, self.num_sample = num_point, radius, num_sample
        self.use_xyz, self.normalize_xyz = use_xyz, normalize_xyz
        ch = list(mlp_channels)
        if use_xyz:
            ch[0] += 3
        self.mlps = nn.ModuleList([_mlp(ch, 2, False)])

    def forward(self, points_xyz, features=None, indices=None, target_xyz=None):
        xyz_flipped = points_xyz.transpose(1, 2).contiguous()
        if indices is not None:
            new_xyz = O.gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
        elif target_xyz is not None:
            new_xyz = target_xyz.contiguous()
        else:
            indices = O.furthest_point_sample(points_xyz, self.num_point)
            new_xyz = O.gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
        idx = O.ball_query(0.0, self.radius, self.num_sample, points_xyz.contiguous(), new_xyz)
        grouped_xyz = O.grouping_operation(xyz_flipped, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = O.grouping_operation(features.contiguous(), idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features
        else:
            new_features = grouped_xyz
        new_features = self.mlps[0](new_features)
        new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)]).squeeze(-1)
        return new_xyz, new_features, indices


def build_sa_module(cfg):
    cfg = dict(cfg)
    cfg.pop("type", None)
    return PointSAModule(**cfg)


class PointFPModule(nn.Module):
    """mmdet3d PointFPModule (three_nn + three_interpolate + shared MLP)."""

    def __init__(self, mlp_channels):
        super().__init__()
        self.mlps = _mlp(list(mlp_channels), 2, False)

    def forward(self, target, source, target_feats, source_feats):
        dist, idx = O.three_nn(target.contiguous(), source.contiguous())
        dist_reciprocal = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_reciprocal, dim=2, keepdim=True)
        weight = dist_reciprocal / norm
        interpolated = O.three_interpolate(source_feats.contiguous(), idx, weight)
        new = torch.cat([interpolated, target_feats], dim=1) if target_feats is not None \
            else interpolated
        return self.mlps(new.unsqueeze(-1)).squeeze(-1)


class PointNet2SASSG(nn.Module):
    """mmdet3d PointNet2SASSG (demf_votenet.py:48-62; used at demfnet.py:151-152)."""

    def __init__(self, in_channels=4, num_points=(2048, 1024, 512, 256),
                 radius=(0.2, 0.4, 0.8, 1.2), num_samples=(64, 32, 16, 16),
                 sa_channels=((64, 64, 128), (128, 128, 256), (128, 128, 256), (128, 128, 256)),
                 fp_channels=((256, 256), (256, 256)), use_xyz=True, normalize_xyz=True, **unused):
        super().__init__()
        self.num_sa, self.num_fp = len(sa_channels), len(fp_channels)
        self.SA_modules, self.FP_modules = nn.ModuleList(), nn.ModuleList()
        sa_in = in_channels - 3
        skip = [sa_in]
        for i in range(self.num_sa):
            ch = [sa_in] + list(sa_channels[i])
            self.SA_modules.append(PointSAModule(num_points[i], radius[i], num_samples[i], ch,
                                                 use_xyz, normalize_xyz))
            sa_in = ch[-1]
            skip.append(sa_in)
        src, tgt = skip.pop(), skip.pop()
        for i in range(self.num_fp):
            ch = [src + tgt] + list(fp_channels[i])
            self.FP_modules.append(PointFPModule(ch))
            if i != self.num_fp - 1:
                src, tgt = ch[-1], skip.pop()

    def forward(self, points):
        xyz = points[..., 0:3].contiguous()
        features = points[..., 3:].transpose(1, 2).contiguous() if points.size(-1) > 3 else None
        B, N = xyz.shape[:2]
        indices = xyz.new_tensor(range(N)).unsqueeze(0).repeat(B, 1).long()
        sa_xyz, sa_features, sa_indices = [xyz], [features], [indices]
        for i in range(self.num_sa):
            cx, cf, ci = self.SA_modules[i](sa_xyz[i], sa_features[i])
            sa_xyz.append(cx)
            sa_features.append(cf)
            sa_indices.append(torch.gather(sa_indices[-1], 1, ci.long()))
        fp_xyz, fp_features, fp_indices = [sa_xyz[-1]], [sa_features[-1]], [sa_indices[-1]]
        for i in range(self.num_fp):
            fp_features.append(self.FP_modules[i](sa_xyz[self.num_sa - i - 1], sa_xyz[self.num_sa - i],
                                                  sa_features[self.num_sa - i - 1], fp_features[-1]))
            fp_xyz.append(sa_xyz[self.num_sa - i - 1])
            fp_indices.append(sa_indices[self.num_sa - i - 1])
        return dict(fp_xyz=fp_xyz, fp_features=fp_features, fp_indices=fp_indices,
                    sa_xyz=sa_xyz, sa_features=sa_features, sa_indices=sa_indices)


# ----------------------------------------------------------------------------- mmdet3d model utils
def chamfer_distance(src, dst, src_weight=1.0, dst_weight=1.0, criterion_mode="l2",
                     reduction="mean"):
    """mmdet3d.models.losses.chamfer_distance (class_agnostic_vote_head.py:882)."""
    crit = dict(smooth_l1=F.smooth_l1_loss, l1=F.l1_loss, l2=F.mse_loss)[criterion_mode]
    src_e = src.unsqueeze(2).repeat(1, 1, dst.shape[1], 1)
    dst_e = dst.unsqueeze(1).repeat(1, src.shape[1], 1, 1)
    distance = crit(src_e, dst_e, reduction="none").sum(-1)
    s2d, i1 = torch.min(distance, dim=2)
    d2s, i2 = torch.min(distance, dim=1)
    loss_src, loss_dst = s2d * src_weight, d2s * dst_weight
    if reduction == "sum":
        loss_src, loss_dst = loss_src.sum(), loss_dst.sum()
    elif reduction == "mean":
        loss_src, loss_dst = loss_src.mean(), loss_dst.mean()
    return loss_src, loss_dst, i1, i2


class ChamferDistance(nn.Module):
    def __init__(self, mode="l2", reduction="mean", loss_src_weight=1.0, loss_dst_weight=1.0, **kw):
        super().__init__()
        self.mode, self.reduction = mode, reduction
        self.loss_src_weight, self.loss_dst_weight = loss_src_weight, loss_dst_weight

    def forward(self, source, target, src_weight=1.0, dst_weight=1.0, **kw):
        ls, ld, _, _ = chamfer_distance(source, target, src_weight, dst_weight, self.mode,
                                        self.reduction)
        return ls * self.loss_src_weight, ld * self.loss_dst_weight


class VoteModule(nn.Module):
    """mmdet3d VoteModule (class_agnostic_vote_head.py:382,413-414,641-644)."""

    def __init__(self, in_channels, vote_per_seed=1, gt_per_seed=3, conv_channels=(16, 16),
                 norm_feats=True, vote_loss=None, **unused):
        super().__init__()
        self.in_channels, self.vote_per_seed, self.gt_per_seed = in_channels, vote_per_seed, gt_per_seed
        self.norm_feats = norm_feats
        vl = dict(vote_loss or {})
        vl.pop("type", None)
        self.vote_loss = ChamferDistance(**vl)
        ch = [in_channels] + list(conv_channels)
        self.vote_conv = nn.Sequential(*[ConvModule(ch[i], ch[i + 1], 1, True)
                                         for i in range(len(ch) - 1)])
        self.conv_out = nn.Conv1d(ch[-1], (3 + in_channels) * vote_per_seed, 1)

    def forward(self, seed_points, seed_feats):
        B, C, N = seed_feats.shape
        num_vote = N * self.vote_per_seed
        votes = self.conv_out(self.vote_conv(seed_feats))
        votes = votes.transpose(2, 1).view(B, N, self.vote_per_seed, -1)
        offset = votes[:, :, :, 0:3]
        vote_points = (seed_points.unsqueeze(2) + offset).contiguous().view(B, num_vote, 3)
        offset = offset.reshape(B, num_vote, 3).transpose(2, 1)
        res_feats = votes[:, :, :, 3:]
        vote_feats = (seed_feats.transpose(2, 1).unsqueeze(2) + res_feats).contiguous()
        vote_feats = vote_feats.view(B, num_vote, C).transpose(2, 1).contiguous()
        if self.norm_feats:
            vote_feats = vote_feats.div(torch.norm(vote_feats, p=2, dim=1).unsqueeze(1))
        return vote_points, vote_feats, offset

    def get_loss(self, seed_points, vote_points, seed_indices, vote_targets_mask, vote_targets):
        B, N = seed_points.shape[:2]
        mask = torch.gather(vote_targets_mask, 1, seed_indices).float()
        idx = seed_indices.unsqueeze(-1).repeat(1, 1, 3 * self.gt_per_seed)
        gt = torch.gather(vote_targets, 1, idx)
        gt = gt + seed_points.repeat(1, 1, self.gt_per_seed)
        weight = mask / (torch.sum(mask) + 1e-6)
        distance = self.vote_loss(vote_points.view(B * N, -1, 3), gt.view(B * N, -1, 3),
                                  dst_weight=weight.view(B * N, 1))[1]
        return torch.sum(torch.min(distance, dim=1)[0])


class BaseConvBboxHead(nn.Module):
    """mmdet3d BaseConvBboxHead (class_agnostic_vote_head.py:398-403)."""

    def __init__(self, in_channels=0, shared_conv_channels=(), cls_conv_channels=(),
                 num_cls_out_channels=0, reg_conv_channels=(), num_reg_out_channels=0,
                 bias="auto", **unused):
        super().__init__()
        assert not cls_conv_channels and not reg_conv_channels
        ch = [in_channels] + list(shared_conv_channels)
        self.shared_convs = _mlp(ch, 1, bool(bias))
        self.conv_cls = nn.Conv1d(ch[-1], num_cls_out_channels, 1)
        self.conv_reg = nn.Conv1d(ch[-1], num_reg_out_channels, 1)

    def forward(self, feats):
        x = self.shared_convs(feats)
        return self.conv_cls(x), self.conv_reg(x)


# ----------------------------------------------------------------------------- mmcv transformer
class MultiheadAttention(nn.Module):
    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout=None, **kw):
        super().__init__()
        out_drop = 0.0
        if dropout is not None:  # deprecated kwarg: sets attn_drop AND the output dropout layer
            attn_drop, out_drop = dropout, dropout
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(out_drop) if out_drop > 0 else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        return identity + self.dropout_layer(self.proj_drop(out))


class MultiScaleDeformableAttention(nn.Module):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention (transformer.py:8-15)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, **kw):
        super().__init__()
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.im2col_step, self.batch_first = im2col_step, batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid.view(-1)
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kw):
        value = query if value is None else value
        identity = query if identity is None else identity
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        bs, nv, _ = value.shape
        assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == nv
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        off = self.sampling_offsets(query).view(bs, nq, self.num_heads, self.num_levels,
                                                self.num_points, 2)
        aw = self.attention_weights(query).view(bs, nq, self.num_heads,
                                                self.num_levels * self.num_points).softmax(-1)
        aw = aw.view(bs, nq, self.num_heads, self.num_levels, self.num_points)
        assert reference_points.shape[-1] == 2
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = O.MultiScaleDeformableAttnFunction.apply(value.contiguous(), spatial_shapes,
                                                       level_start_index, loc.contiguous(),
                                                       aw.contiguous(), self.im2col_step)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


class FFN(nn.Module):
    def __init__(self, embed_dims=256, feedforward_channels=1024, ffn_drop=0.0, **kw):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True),
                          nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        return (x if identity is None else identity) + self.layers(x)


class DetrTransformerDecoderLayer(nn.Module):
    """mmcv BaseTransformerLayer, generic over ``operation_order`` (post-norm)."""

    def __init__(self, attn_cfgs=None, feedforward_channels=1024, ffn_dropout=0.0,
                 operation_order=None, **kw):
        super().__init__()
        self.operation_order = tuple(operation_order)
        self.attentions = nn.ModuleList()
        for c in attn_cfgs:
            c = dict(c)
            t = c.pop("type")
            self.attentions.append(dict(MultiheadAttention=MultiheadAttention,
                                        MultiScaleDeformableAttention=MultiScaleDeformableAttention)[t](**c))
        dims = attn_cfgs[0]["embed_dims"]
        self.ffns = nn.ModuleList([FFN(dims, feedforward_channels, ffn_dropout)
                                   for _ in range(self.operation_order.count("ffn"))])
        self.norms = nn.ModuleList([nn.LayerNorm(dims)
                                    for _ in range(self.operation_order.count("norm"))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        ni = ai = fi = 0
        for op in self.operation_order:
            if op == "self_attn":
                query = self.attentions[ai](query, query, query, None, query_pos=query_pos,
                                            key_pos=query_pos, attn_mask=None,
                                            key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
            elif op == "cross_attn":
                query = self.attentions[ai](query, key, value, None, query_pos=query_pos,
                                            key_pos=key_pos, attn_mask=None,
                                            key_padding_mask=key_padding_mask, **kwargs)
                ai += 1
            elif op == "norm":
                query = self.norms[ni](query)
                ni += 1
            elif op == "ffn":
                query = self.ffns[fi](query, None)
                fi += 1
        return query


# ----------------------------------------------------------------------------- losses (mmdet / mmdet3d)
class CrossEntropyLoss(nn.Module):
    def __init__(self, class_weight=None, reduction="mean", loss_weight=1.0, **kw):
        super().__init__()
        self.class_weight, self.reduction, self.loss_weight = class_weight, reduction, loss_weight

    def forward(self, cls_score, label, weight=None, **kw):
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        loss = F.cross_entropy(cls_score, label, weight=cw, reduction="none")
        if weight is not None:
            loss = loss * weight.float()
        loss = loss.sum() if self.reduction == "sum" else loss.mean()
        return self.loss_weight * loss


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0, **kw):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, **kw):
        diff = torch.abs(pred - target)
        loss = torch.where(diff < self.beta, 0.5 * diff * diff / self.beta, diff - 0.5 * self.beta)
        if weight is not None:
            loss = loss * weight
        loss = loss.sum() if self.reduction == "sum" else loss.mean()
        return self.loss_weight * loss


class AxisAlignedIoULoss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0, **kw):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, **kw):
        a1 = (pred[..., 3] - pred[..., 0]) * (pred[..., 4] - pred[..., 1]) * (pred[..., 5] - pred[..., 2])
        a2 = (target[..., 3] - target[..., 0]) * (target[..., 4] - target[..., 1]) * \
            (target[..., 5] - target[..., 2])
        lt = torch.max(pred[..., :3], target[..., :3])
        rb = torch.min(pred[..., 3:], target[..., 3:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
        union = torch.max(a1 + a2 - overlap, overlap.new_tensor([1e-6]))
        loss = 1 - overlap / union
        if weight is not None:
            loss = loss * weight
        loss = loss.sum() if self.reduction == "sum" else loss.mean()
        return self.loss_weight * loss


def build_loss(cfg):
    cfg = dict(cfg)
    return dict(CrossEntropyLoss=CrossEntropyLoss, SmoothL1Loss=SmoothL1Loss,
                AxisAlignedIoULoss=AxisAlignedIoULoss, ChamferDistance=ChamferDistance)[cfg.pop("type")](**cfg)


# ----------------------------------------------------------------------------- geometry (mmdet3d.core)
def rotation_3d_in_axis(points, angles, axis=0):
    """mmdet3d 0.18.1 rotation_3d_in_axis, axis=2 form: einsum('aij,jka->aik')."""
    assert axis in (2, -1)
    s, c = torch.sin(angles), torch.cos(angles)
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    rot_mat_T = torch.stack([torch.stack([c, -s, zero]), torch.stack([s, c, zero]),
                             torch.stack([zero, zero, one])])
    return torch.einsum("aij,jka->aik", (points, rot_mat_T))


class DepthInstance3DBoxes:
    """The slice of mmdet3d DepthInstance3DBoxes that class_agnostic_vote_head.py:714-754,
    :825-911 and coder.py:142-166 touch.  tensor (n,7) = x, y, z_bottom, dx, dy, dz, yaw.
    ``origin`` as upstream: the relative position of (x,y,z) inside the box the caller's tensor
    uses; it is converted to the bottom centre (0.5, 0.5, 0)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        t = torch.as_tensor(tensor)
        t = (t if t.is_floating_point() else t.float()).reshape(-1, 7)
        if tuple(origin) != (0.5, 0.5, 0):
            t = t.clone()
            dst = t.new_tensor((0.5, 0.5, 0))
            src = t.new_tensor(origin)
            t[:, :3] += t[:, 3:6] * (dst - src)
        self.tensor = t

    def to(self, device):
        return DepthInstance3DBoxes(self.tensor.to(device))

    def new_box(self, data):
        return DepthInstance3DBoxes(data)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        return DepthInstance3DBoxes(self.tensor[item].reshape(-1, 7))

    @property
    def corners(self):
        """(n,8,3): dims * {0,1}^3 corners relative to the bottom centre, rotated about z by yaw."""
        dims = self.dims
        import numpy as _np
        cn = torch.from_numpy(_np.stack(_np.unravel_index(_np.arange(8), [2] * 3), axis=1)).to(dims)
        cn = cn[[0, 1, 3, 2, 4, 5, 7, 6]] - dims.new_tensor([0.5, 0.5, 0])
        corners = dims.view(-1, 1, 3) * cn.reshape(1, 8, 3)
        corners = rotation_3d_in_axis(corners, self.tensor[:, 6], axis=2)
        return corners + self.tensor[:, :3].view(-1, 1, 3)

    @property
    def gravity_center(self):
        bc = self.tensor[:, :3]
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    @property
    def dims(self):
        return self.tensor[:, 3:6]

    @property
    def yaw(self):
        return self.tensor[:, 6]

    def points_in_boxes(self, points):
        """points_in_boxes_batch membership: |z - cz| <= dz/2, box-frame |x| < dx/2, |y| < dy/2,
        box frame = rotation_3d_in_axis(p - centre, -yaw) (the frame the head itself uses at
        class_agnostic_vote_head.py:909-911)."""
        rel = points[:, None, :3] - self.gravity_center[None]
        n = self.tensor.shape[0]
        local = torch.stack([rotation_3d_in_axis(rel[:, i:i + 1], -self.yaw[i].expand(rel.shape[0]),
                                                 2).squeeze(1) for i in range(n)], 1)
        half = self.dims[None] / 2
        return (rel[..., 2].abs() <= half[..., 2]) & (local[..., 0].abs() < half[..., 0]) & \
            (local[..., 1].abs() < half[..., 1])


def apply_3d_transformation(pcd, coord_type, img_meta, reverse=False):
    """mmdet3d.models.fusion_layers.apply_3d_transformation (class_agnostic_vote_head.py:530)."""
    assert coord_type == "DEPTH" and reverse
    dtype = pcd.dtype
    rot = torch.tensor(img_meta["pcd_rotation"], dtype=dtype) if "pcd_rotation" in img_meta \
        else torch.eye(3, dtype=dtype)
    scale = img_meta.get("pcd_scale_factor", 1.0)
    trans = torch.tensor(img_meta["pcd_trans"], dtype=dtype) if "pcd_trans" in img_meta \
        else torch.zeros(3, dtype=dtype)
    hflip, vflip = img_meta.get("pcd_horizontal_flip", False), img_meta.get("pcd_vertical_flip", False)
    flow = list(img_meta.get("transformation_3d_flow", []))[::-1]
    pcd = pcd.clone()
    for op in flow:
        if op == "T":
            pcd[:, :3] += -trans
        elif op == "S":
            pcd[:, :3] *= 1.0 / scale
        elif op == "R":
            pcd[:, :3] = pcd[:, :3] @ rot.inverse()
        elif op == "HF":
            if hflip:
                pcd[:, 0] = -pcd[:, 0]
        elif op == "VF":
            if vflip:
                pcd[:, 1] = -pcd[:, 1]
    return pcd


def points_cam2img(points_3d, proj_mat, with_depth=False):
    """mmdet3d.core.bbox.points_cam2img (class_agnostic_vote_head.py:535)."""
    shape = list(points_3d.shape)
    shape[-1] = 1
    d1, d2 = proj_mat.shape[:2]
    if d1 == 3:
        ext = torch.eye(4, device=proj_mat.device, dtype=proj_mat.dtype)
        ext[:d1, :d2] = proj_mat
        proj_mat = ext
    p4 = torch.cat([points_3d, points_3d.new_ones(shape)], dim=-1)
    p2 = p4 @ proj_mat.T
    return p2[..., :2] / p2[..., 2:3]


def coord_2d_transform(img_meta, coord_2d, is_orig2new):
    """mmdet3d.models.fusion_layers.coord_2d_transform (class_agnostic_vote_head.py:540)."""
    assert is_orig2new
    img_h, img_w = img_meta["img_shape"][:2]
    sf = coord_2d.new_tensor(img_meta["scale_factor"][:2]) if "scale_factor" in img_meta \
        else coord_2d.new_tensor([1.0, 1.0])
    off = coord_2d.new_tensor(img_meta["img_crop_offset"]) if "img_crop_offset" in img_meta \
        else coord_2d.new_tensor([0.0, 0.0])
    out = torch.zeros_like(coord_2d)
    out[:, 0] = coord_2d[:, 0] * sf[0] + off[0]
    out[:, 1] = coord_2d[:, 1] * sf[1] + off[1]
    if img_meta.get("flip", False):
        out[:, 0] = img_w - out[:, 0]
    return out


class PartialBinBasedBBoxCoder:
    """mmdet3d PartialBinBasedBBoxCoder angle helpers (base of coder.py:8-14)."""

    def __init__(self, num_dir_bins, num_sizes, mean_sizes, with_rot=True):
        self.num_dir_bins, self.num_sizes, self.mean_sizes, self.with_rot = \
            num_dir_bins, num_sizes, mean_sizes, with_rot

    def angle2class(self, angle):
        angle = angle % (2 * np.pi)
        per = 2 * np.pi / float(self.num_dir_bins)
        shifted = (angle + per / 2) % (2 * np.pi)
        cls = shifted // per
        res = shifted - (cls * per + per / 2)
        return cls.long(), res

    def class2angle(self, angle_cls, angle_res, limit_period=True):
        per = 2 * np.pi / float(self.num_dir_bins)
        angle = angle_cls.float() * per + angle_res
        if limit_period:
            angle[angle > np.pi] -= 2 * np.pi
        return angle


def multi_apply(func, *args, **kwargs):
    """mmdet.core.multi_apply."""
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def aligned_3d_nms(boxes, scores, classes, thresh):
    """mmdet3d.core.post_processing.aligned_3d_nms: greedy NMS on axis-aligned (x1,y1,z1,x2,y2,z2)
    boxes; a box suppresses lower-scored boxes of the SAME class whose IoU exceeds ``thresh``."""
    x1, y1, z1, x2, y2, z2 = (boxes[:, i] for i in range(6))
    area = (x2 - x1) * (y2 - y1) * (z2 - z1)
    zero = boxes.new_zeros(1)
    order = torch.argsort(scores)
    pick = []
    while order.shape[0] != 0:
        last = order.shape[0]
        i = order[-1]
        pick.append(int(i))
        rest = order[:last - 1]
        xx1, yy1, zz1 = torch.max(x1[i], x1[rest]), torch.max(y1[i], y1[rest]), torch.max(z1[i], z1[rest])
        xx2, yy2, zz2 = torch.min(x2[i], x2[rest]), torch.min(y2[i], y2[rest]), torch.min(z2[i], z2[rest])
        l, w, h = torch.max(zero, xx2 - xx1), torch.max(zero, yy2 - yy1), torch.max(zero, zz2 - zz1)
        inter = l * w * h
        iou = inter / (area[i] + area[rest] - inter)
        iou = iou * (classes[i] == classes[rest]).float()
        order = rest[torch.nonzero(iou <= thresh, as_tuple=False).flatten()]
    return torch.as_tensor(pick, dtype=torch.long)


def multiclass_nms_single(head, obj_scores, sem_scores, bbox, points, input_meta):
    """mmdet3d VoteHead.multiclass_nms_single (0.18.1), which DeMFVoteHead inherits and calls at
    class_agnostic_vote_head.py:741-744: boxes with more than 5 points inside -> aligned 3-D NMS
    on their corner extents -> score threshold -> (per-class) outputs."""
    tc = head.test_cfg
    get = (lambda k: tc[k]) if isinstance(tc, dict) else (lambda k: getattr(tc, k))
    bbox = input_meta["box_type_3d"](bbox, box_dim=bbox.shape[-1], with_yaw=head.bbox_coder.with_rot,
                                     origin=(0.5, 0.5, 0.5))
    box_indices = bbox.points_in_boxes(points)
    corner3d = bbox.corners
    minmax = corner3d.new_zeros((corner3d.shape[0], 6))
    minmax[:, :3] = torch.min(corner3d, dim=1)[0]
    minmax[:, 3:] = torch.max(corner3d, dim=1)[0]
    nonempty = box_indices.T.sum(1) > 5
    bbox_classes = torch.argmax(sem_scores, -1)
    nms_selected = aligned_3d_nms(minmax[nonempty], obj_scores[nonempty], bbox_classes[nonempty],
                                  get("nms_thr"))
    scores_mask = obj_scores > get("score_thr")
    nonempty_inds = torch.nonzero(nonempty, as_tuple=False).flatten()
    nonempty_mask = torch.zeros_like(bbox_classes).scatter(0, nonempty_inds[nms_selected], 1)
    selected = nonempty_mask.bool() & scores_mask.bool()
    if get("per_class_proposal"):
        bs, ss, ls = [], [], []
        for k in range(sem_scores.shape[-1]):
            bs.append(bbox[selected].tensor)
            ss.append(obj_scores[selected] * sem_scores[selected][:, k])
            ls.append(torch.zeros_like(bbox_classes[selected]).fill_(k))
        return torch.cat(bs, 0), torch.cat(ss, 0), torch.cat(ls, 0)
    return bbox[selected].tensor, obj_scores[selected], bbox_classes[selected]


# ---------------------------------------------------------------------------------------------
# Frozen image stream dependencies: mmdet 2.14 ResNet (pytorch style),
# ChannelMapper, SinePositionalEncoding and mmcv's BaseTransformerLayer / DetrTransformerEncoder
# for operation_order ('self_attn','norm','ffn','norm') - configs/deformdetr/imvotenet_image.py:3-20,
# configs/demf/demf_votenet.py:28-47.  dep-recall, parity with upstream binaries unpinned.
# ---------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # style='pytorch'
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                        nn.BatchNorm2d(planes * 4)) if downsample else None

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """mmdet ResNet(depth=50, num_stages=4, out_indices=(1,2,3), style='pytorch', norm_eval)."""

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

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
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
    """mmdet ChannelMapper(kernel_size=1, norm GN32, act None, num_outs > len(in): extra 3x3 s2)."""

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


class SinePositionalEncoding(nn.Module):
    """mmdet SinePositionalEncoding(num_feats, temperature=10000, normalize, scale=2pi, eps, offset)."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0.0, **kw):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class EncoderLayer(nn.Module):
    """mmcv BaseTransformerLayer, operation_order ('self_attn','norm','ffn','norm')."""

    def __init__(self, attn_cfgs=None, feedforward_channels=1024, ffn_dropout=0.0,
                 operation_order=("self_attn", "norm", "ffn", "norm"), **kw):
        super().__init__()
        assert tuple(operation_order) == ("self_attn", "norm", "ffn", "norm")
        a = dict(attn_cfgs)
        a.pop("type", None)
        self.attentions = nn.ModuleList([MultiScaleDeformableAttention(**a)])
        e = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList([FFN(e, feedforward_channels, ffn_dropout)])
        self.norms = nn.ModuleList([nn.LayerNorm(e), nn.LayerNorm(e)])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                query_key_padding_mask=None, key_padding_mask=None, **kw):
        query = self.attentions[0](query, query, query, None, query_pos=query_pos, key_pos=query_pos,
                                   key_padding_mask=query_key_padding_mask, **kw)
        query = self.norms[0](query)
        query = self.ffns[0](query, None)
        return self.norms[1](query)


class DetrTransformerEncoder(nn.Module):
    """mmdet DetrTransformerEncoder(num_layers, transformerlayers): post_norm only when pre-norm."""

    def __init__(self, num_layers, transformerlayers, **kw):
        super().__init__()
        t = dict(transformerlayers)
        t.pop("type", None)
        self.layers = nn.ModuleList([EncoderLayer(**t) for _ in range(num_layers)])

    def forward(self, query, *args, **kw):
        for layer in self.layers:
            query = layer(query, *args, **kw)
        return query


def build_transformer_layer_sequence(cfg):
    cfg = dict(cfg)
    assert cfg.pop("type") == "DetrTransformerEncoder"
    return DetrTransformerEncoder(**cfg)


def build_positional_encoding(cfg):
    cfg = dict(cfg)
    assert cfg.pop("type") == "SinePositionalEncoding"
    return SinePositionalEncoding(**cfg)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)
