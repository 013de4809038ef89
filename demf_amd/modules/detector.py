"""The trainable hot path of DeMFVoteNet: point backbone -> DeMF head -> loss.

Mirrors demf/modeling/detectors/demfnet.py:134-170 (forward_train) from line 150 on;
the frozen, no_grad image stream (demfnet.py:124-132) is outside the hot path and its
output pyramid ``img_features`` (list of 4 (B,256,H_l,W_l)) is an input here.
Attribute names (``pts_backbone``, ``pts_bbox_head``) match the reference detector so
checkpoints load with the same keys.
"""
import torch
import torch.nn as nn

from .. import ops
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

    def extract_pts_feat(self, points, geometry=None):
        """ImVoteNet.extract_pts_feat as reached from demfnet.py:151-152."""
        x = self.pts_backbone(points, geometry)
        return x["fp_xyz"][-1], x["fp_features"][-1], x["fp_indices"][-1]

    @torch.no_grad()
    def index_geometry(self, points):
        """Coordinate-only pre-pass of the step: the backbone's FPS / ball-query / 3-NN indices and
        the head's seed FPS (class_agnostic_vote_head.py:429-430; seeds are input points)."""
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        geo = self.pts_backbone.index_geometry(points)
        bb = self.pts_backbone
        seed_xyz = ([points[..., :3]] + [t[1] for t in geo["sa"]])[bb.num_sa - bb.num_fp]
        if self.cfg.head.sample_mod == "seed":
            geo["sample_indices"] = ops.furthest_point_sample(seed_xyz.contiguous(),
                                                              self.pts_bbox_head.num_proposal)
        return geo

    def _side_streams(self, device):
        ss = self.__dict__.setdefault("_streams", {})
        if str(device) not in ss:
            ss[str(device)] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        return ss[str(device)]

    def forward_head(self, points, img_features, img_metas, image_inputs=None, geometry=None):
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)                                    # demfnet.py:150
        seeds_3d, seed_3d_features, seed_indices = self.extract_pts_feat(points, geometry)
        feat_dict = dict(seed_points=seeds_3d, seed_features=seed_3d_features,
                         seed_indices=seed_indices,
                         sample_indices=None if geometry is None else geometry.get("sample_indices"))
        img_dict = dict(img_features=img_features, img_metas=img_metas, image_inputs=image_inputs)
        return self.pts_bbox_head(feat_dict, self.cfg.head.sample_mod, img_dict)  # :165

    def forward_train(self, points, img_features, img_metas, gt_bboxes_3d, gt_labels_3d,
                      overlap=False, geometry=None):
        """-> dict of losses (demfnet.py:134-170 with the image pyramid precomputed).

        ``overlap=True``: the point stream starts with furthest-point sampling, which keeps 8 of
        the 256 CUs busy for milliseconds; everything that does not depend on it - flattening /
        masking / value-projecting the image tokens, and the per-point vote targets - is issued
        on two side HIP streams.  Measured (profiles/, round 1): inside a captured hipGraph the
        ROCm 7.2 runtime replays the forked branches serially (no kernel starts inside the FPS
        launch) and the joins cost ~3 ms of idle, so the default is off until the step is
        replayed as separate per-stream graphs."""
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        head = self.pts_bbox_head
        if not (overlap and points.is_cuda):
            bbox_preds = self.forward_head(points, img_features, img_metas, geometry=geometry)
            return head.loss(bbox_preds, points, gt_bboxes_3d, gt_labels_3d, None, None, img_metas)
        main = torch.cuda.current_stream()
        s_img, s_tgt = self._side_streams(points.device)
        s_img.wait_stream(main)
        s_tgt.wait_stream(main)
        with torch.cuda.stream(s_img):
            image_inputs = head.prepare_image_inputs(img_features, img_metas)
        with torch.cuda.stream(s_tgt):
            vote_pack = head.vote_targets(points, gt_bboxes_3d, gt_labels_3d)
        seeds_3d, seed_3d_features, seed_indices = self.extract_pts_feat(points)
        main.wait_stream(s_img)
        for t in [image_inputs["feat_flatten"], image_inputs["mask_flatten"],
                  image_inputs["valid_ratios"], *(image_inputs["value_projected"] or []),
                  *(image_inputs["value_tokens"] or [])]:
            t.record_stream(main)
        feat_dict = dict(seed_points=seeds_3d, seed_features=seed_3d_features,
                         seed_indices=seed_indices)
        img_dict = dict(img_features=img_features, img_metas=img_metas, image_inputs=image_inputs)
        bbox_preds = head(feat_dict, self.cfg.head.sample_mod, img_dict)     # :165
        main.wait_stream(s_tgt)
        for t in vote_pack.values():
            t.record_stream(main)
        return head.loss(bbox_preds, points, gt_bboxes_3d, gt_labels_3d, None, None, img_metas,
                         vote_pack=vote_pack)                                # :167

    def param_groups(self, lr=0.008, weight_decay=0.01):
        """AdamW groups of demf_votenet.py:16-24: 'decoder' params at lr*0.05."""
        dec, rest = [], []
        for n, p in self.named_parameters():
            if p.requires_grad:
                (dec if "decoder" in n else rest).append(p)
        return [dict(params=rest, lr=lr, weight_decay=weight_decay),
                dict(params=dec, lr=lr * 0.05, weight_decay=weight_decay)]
