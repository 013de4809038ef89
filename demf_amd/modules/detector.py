"""The trainable hot path of DeMFVoteNet: point backbone -> DeMF head -> loss.

Mirrors demf/modeling/detectors/demfnet.py:134-170 (forward_train) from line 150 on;
the frozen, no_grad image stream (demfnet.py:124-132) is outside the hot path and its
output pyramid ``img_features`` (list of 4 (B,256,H_l,W_l)) is an input here.
Attribute names (``pts_backbone``, ``pts_bbox_head``) match the reference detector so
checkpoints load with the same keys.
"""
import torch
import torch.nn as nn

from ..config import DeMFCfg, head_kwargs
from .head import DeMFVoteHead
from .pointnet2 import PointNet2SASSG


class DeMFHotPath(nn.Module):
    def __init__(self, cfg: DeMFCfg = None):
        super().__init__()
        self.cfg = cfg or DeMFCfg()
        b = self.cfg.backbone
        self.pts_backbone = PointNet2SASSG(
            in_channels=b.in_channels, num_points=b.num_points, radius=b.radius,
            num_samples=b.num_samples, sa_channels=b.sa_channels, fp_channels=b.fp_channels,
            use_xyz=b.use_xyz, normalize_xyz=b.normalize_xyz)
        self.pts_bbox_head = DeMFVoteHead(**head_kwargs(self.cfg))
        for layer in self.pts_bbox_head.decoder:
            layer.init_weights()

    def extract_pts_feat(self, points):
        """ImVoteNet.extract_pts_feat as reached from demfnet.py:151-152."""
        x = self.pts_backbone(points)
        return x["fp_xyz"][-1], x["fp_features"][-1], x["fp_indices"][-1]

    def forward_head(self, points, img_features, img_metas):
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)                                    # demfnet.py:150
        seeds_3d, seed_3d_features, seed_indices = self.extract_pts_feat(points)
        feat_dict = dict(seed_points=seeds_3d, seed_features=seed_3d_features,
                         seed_indices=seed_indices)
        img_dict = dict(img_features=img_features, img_metas=img_metas)
        return self.pts_bbox_head(feat_dict, self.cfg.head.sample_mod, img_dict)  # :165

    def forward_train(self, points, img_features, img_metas, gt_bboxes_3d, gt_labels_3d):
        """-> dict of losses (demfnet.py:134-170 with the image pyramid precomputed)."""
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        bbox_preds = self.forward_head(points, img_features, img_metas)
        return self.pts_bbox_head.loss(bbox_preds, points, gt_bboxes_3d, gt_labels_3d,
                                       None, None, img_metas)                # :167

    def param_groups(self, lr=0.008, weight_decay=0.01):
        """AdamW groups of demf_votenet.py:16-24: 'decoder' params at lr*0.05."""
        dec, rest = [], []
        for n, p in self.named_parameters():
            if p.requires_grad:
                (dec if "decoder" in n else rest).append(p)
        return [dict(params=rest, lr=lr, weight_decay=weight_decay),
                dict(params=dec, lr=lr * 0.05, weight_decay=weight_decay)]
