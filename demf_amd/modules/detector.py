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
                      geometry=None):
        """-> dict of losses (demfnet.py:134-170 with the image pyramid precomputed).
        ``geometry``: the coordinate-only pre-pass (``index_geometry``) when it was computed ahead
        of the step (engine.Trainer pipelines it under the previous step's backward)."""
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        bbox_preds = self.forward_head(points, img_features, img_metas, geometry=geometry)
        return self.pts_bbox_head.loss(bbox_preds, points, gt_bboxes_3d, gt_labels_3d, None, None,
                                       img_metas)                            # :167

    def param_groups(self, lr=0.008, weight_decay=0.01):
        """AdamW groups of demf_votenet.py:16-24: 'decoder' params at lr*0.05."""
        dec, rest = [], []
        for n, p in self.named_parameters():
            if p.requires_grad:
                (dec if "decoder" in n else rest).append(p)
        return [dict(params=rest, lr=lr, weight_decay=weight_decay),
                dict(params=dec, lr=lr * 0.05, weight_decay=weight_decay)]


class DeMFVoteNet(DeMFHotPath):
    """The whole detector with the reference's entry points
    (demf/modeling/detectors/demfnet.py:12-283): the trainable hot path above plus the frozen
    image stream under the reference's attribute names (``img_backbone``, ``img_neck``,
    ``img_encoder``), so a released checkpoint - or a stage-1 image checkpoint through the key
    remap of :85-101 - loads with ``load_state_dict``.

      forward_train(points, img, img_metas, gt_bboxes_3d, gt_labels_3d) -> dict of losses  (:134-170)
      simple_test(points, img_metas, img) -> list of dict(boxes_3d, scores_3d, labels_3d)  (:254-283)
      forward_test(points, img_metas, img)  (single-augmentation form of :172-238)
    """

    def __init__(self, cfg: DeMFCfg = None, image_stream=None, **image_stream_kwargs):
        super().__init__(cfg)
        from .image_stream import ImageStream
        stream = image_stream if image_stream is not None else ImageStream(**image_stream_kwargs)
        self.img_backbone, self.img_neck = stream.img_backbone, stream.img_neck
        self.img_encoder = stream.img_encoder
        self.freeze_img_branch_params()

    def freeze_img_branch_params(self):                                     # :103-112
        for m in (self.img_backbone, self.img_neck, self.img_encoder):
            for p in m.parameters():
                p.requires_grad_(False)

    def train(self, mode=True):                                              # :114-122
        super().train(mode)
        for m in (self.img_backbone, self.img_neck, self.img_encoder):
            m.eval()
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        """+ the stage-1 key remap of ``_load_from_state_dict`` (:85-101)."""
        from ..data import remap_checkpoint
        return super().load_state_dict(remap_checkpoint(state_dict), strict=strict, **kw)

    @torch.no_grad()
    def extract_img_feat(self, img, img_metas):                             # :124-132
        """-> the channels-last token form of the 4-level pyramid (dict(tokens (B,S,C), spatial,
        ...)): what this package's head consumes without the flatten copy (:570-591).
        ``self.img_encoder(feats, img_metas)`` gives the reference's list of NCHW maps."""
        return self.img_encoder.forward_tokens(self.img_neck(self.img_backbone(img)), img_metas)

    def forward_train(self, points=None, img=None, img_metas=None, gt_bboxes_3d=None,
                      gt_labels_3d=None, **unused):
        img_features = self.extract_img_feat(img, img_metas)               # :148
        return super().forward_train(points, img_features, img_metas, gt_bboxes_3d, gt_labels_3d)

    @torch.no_grad()
    def simple_test(self, points=None, img_metas=None, img=None, bboxes_2d=None, rescale=False,
                    **unused):
        img_features = self.extract_img_feat(img, img_metas)               # :260
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)                                    # :262
        bbox_preds = self.forward_head(points, img_features, img_metas)    # :263-275
        bbox_list = self.pts_bbox_head.get_bboxes(points, bbox_preds, img_metas, rescale=rescale)
        return [bbox3d2result(b, s, l) for b, s, l in bbox_list]           # :278-282

    def forward_test(self, points=None, img_metas=None, img=None, bboxes_2d=None, **kwargs):
        """:172-238 for one augmentation (the only form the reference implements for points +
        image): ``points`` / ``img_metas`` / ``img`` are lists over augmentations."""
        for var, name in ((points, "points"), (img_metas, "img_metas")):
            if not isinstance(var, list):
                raise TypeError("{} must be a list, but got {}".format(name, type(var)))
        if len(points) != len(img_metas):
            raise ValueError("num of augmentations ({}) != num of image meta ({})".format(
                len(points), len(img_metas)))
        if len(points) != 1:
            raise NotImplementedError("aug_test is not implemented by the reference either (:240-252)")
        for _img, metas in zip(img, img_metas):                             # :180-183
            for m in metas:
                m["batch_input_shape"] = tuple(_img.shape[-2:])
        return self.simple_test(points[0], img_metas[0], img[0],
                                bboxes_2d=bboxes_2d[0] if bboxes_2d is not None else None, **kwargs)


def bbox3d2result(bboxes, scores, labels):
    """mmdet3d.core.bbox3d2result: results to host memory."""
    return dict(boxes_3d=bboxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu())
