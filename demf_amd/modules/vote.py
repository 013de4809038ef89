"""VoteModule and the per-layer conv prediction head, on point-major rows.

Mirror mmdet3d's VoteModule / BaseConvBboxHead as the reference builds them
(demf/modeling/heads/class_agnostic_vote_head.py:382,394-403; config
configs/demf/demf_votenet.py:65-67,142-154).  Same parameters/state-dict names;
Conv1d(k=1)+BN1d stacks run as GEMMs over rows = B*N.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .layers import ConvBNReLU, RowsMLP, fused_rows


class VoteModule(nn.Module):
    """forward(seed_points (B,N,3), seed_feats (B,C,N)) ->
    (vote_points (B,N,3), vote_feats (B,C,N), offset (B,3,N))   [vote_per_seed == 1]"""

    def __init__(self, in_channels, vote_per_seed=1, gt_per_seed=3, conv_channels=(16, 16),
                 norm_feats=True, with_res_feat=True, vote_loss_dst_weight=10.0, **unused):
        super().__init__()
        assert vote_per_seed == 1 and with_res_feat
        self.in_channels, self.gt_per_seed, self.norm_feats = in_channels, gt_per_seed, norm_feats
        self.vote_loss_dst_weight = vote_loss_dst_weight
        chans = [in_channels] + list(conv_channels)
        # upstream: nn.Sequential(*[ConvModule(Conv1d, BN1d, bias=True)]) -> keys vote_conv.{i}
        self.vote_conv = nn.Sequential(*[ConvBNReLU(chans[i], chans[i + 1], dim=1, bias=True)
                                         for i in range(len(chans) - 1)])
        self.conv_out = nn.Conv1d(chans[-1], 3 + in_channels, 1)

    def forward(self, seed_points, seed_feats):
        B, N, _ = seed_points.shape
        rows = seed_feats.transpose(1, 2).contiguous().view(B * N, -1)
        x = fused_rows(list(self.vote_conv), rows)
        votes = ops.linear(x, self.conv_out.weight.view(self.conv_out.out_channels, -1),
                         self.conv_out.bias)
        offset = votes[:, 0:3].view(B, N, 3)
        C = rows.shape[1]
        if self.norm_feats and rows.is_cuda and C % 64 == 0 and C <= 1024 and \
                (C // 64) & (C // 64 - 1) == 0:
            # seed + offset, residual add and row normalisation in one kernel each way
            # (csrc/dense.hip: vote_combine_*); the lines below are its specification
            vote_points, vote_feats = ops.vote_combine(rows, votes, seed_points)
            vote_feats = vote_feats.view(B, N, -1).transpose(1, 2)
            return vote_points, vote_feats, offset.transpose(2, 1)
        vote_points = (seed_points + offset).contiguous()
        vote_feats = rows + votes[:, 3:]
        if self.norm_feats:
            C = vote_feats.shape[1]
            if vote_feats.is_cuda and C % 64 == 0 and C <= 1024 and (C // 64) & (C // 64 - 1) == 0:
                vote_feats = ops.l2norm_rows(vote_feats.contiguous())
            else:
                vote_feats = vote_feats / torch.norm(vote_feats, p=2, dim=1, keepdim=True)
        vote_feats = vote_feats.view(B, N, -1).transpose(1, 2)  # (B,C,N) view of point-major
        return vote_points, vote_feats, offset.transpose(2, 1)

    def get_loss(self, seed_points, vote_points, seed_indices, vote_targets_mask, vote_targets):
        """Chamfer-L1 vote loss of upstream VoteModule.get_loss (called at
        class_agnostic_vote_head.py:641-644), vote_per_seed == 1."""
        B, N = seed_points.shape[:2]
        mask = torch.gather(vote_targets_mask, 1, seed_indices).float()
        gt = torch.gather(vote_targets, 1,
                          seed_indices.unsqueeze(-1).expand(-1, -1, 3 * self.gt_per_seed))
        gt = gt + seed_points.repeat(1, 1, self.gt_per_seed)
        weight = mask / (torch.sum(mask) + 1e-6)
        # ChamferDistance(mode='l1', reduction='none'): dst->src distance, src has one point
        d = (vote_points.view(B * N, 1, 3) - gt.view(B * N, self.gt_per_seed, 3)).abs().sum(-1)
        d = d * weight.view(B * N, 1) * self.vote_loss_dst_weight
        return torch.sum(torch.min(d, dim=1)[0])


class BaseConvBboxHead(nn.Module):
    """forward(feats (B,C,N)) -> (cls_score (B,ncls,N), bbox_pred (B,nreg,N))."""

    def __init__(self, in_channels, shared_conv_channels=(), num_cls_out_channels=0,
                 num_reg_out_channels=0, bias=True, **unused):
        super().__init__()
        chans = [in_channels] + list(shared_conv_channels)
        self.shared_convs = RowsMLP(chans, dim=1, bias=bias)
        self.conv_cls = nn.Conv1d(chans[-1], num_cls_out_channels, 1)
        self.conv_reg = nn.Conv1d(chans[-1], num_reg_out_channels, 1)

    def forward(self, feats):
        B, C, N = feats.shape
        x = self.shared_convs.forward_rows(feats.transpose(1, 2).contiguous().view(B * N, C))
        cls = ops.linear(x, self.conv_cls.weight.view(self.conv_cls.out_channels, -1), self.conv_cls.bias)
        reg = ops.linear(x, self.conv_reg.weight.view(self.conv_reg.out_channels, -1), self.conv_reg.bias)
        return cls.view(B, N, -1).transpose(1, 2), reg.view(B, N, -1).transpose(1, 2)
