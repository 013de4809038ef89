"""CPU ORACLE (test infrastructure, NOT a product path) - the whole trainable hot path.

``OracleDeMF`` = restated PointNet2SASSG backbone (oracle/deps.py) + a restatement of the
reference's IN-TREE head, decoder layer and bbox coder.  Every method cites the reference
lines it follows.  The in-tree restatement is PINNED: tests/test_oracle_model.py checks it
against golden vectors produced by running the real reference files
(oracle/pin_reference.py -> tests/golden/ref_head_*.npz, ref_glue.npz).

Runs anywhere (no /root/reference needed); used by tests, smoke() and bench.py's
cpu_baseline leg only.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import deps
from . import torch_ops as O


# ---------------------------------------------------------------- bbox coder
class Coder:
    """DeMFClassAgnosticBBoxCoder - demf/core/bbox/coders/class_agnostic_bbox_coder.py:130-251."""

    def __init__(self, num_dir_bins):
        self.nb = num_dir_bins
        self.base = deps.PartialBinBasedBBoxCoder(num_dir_bins, 0, [], True)

    def encode(self, boxes):                                   # coder.py:142-166
        cls, res = self.base.angle2class(boxes.yaw)
        return boxes.gravity_center, boxes.dims, cls, res, boxes.yaw

    def split_pred(self, cls_preds, reg_preds, base_xyz):      # coder.py:196-240
        c, r = cls_preds.transpose(2, 1), reg_preds.transpose(2, 1)
        nb = self.nb
        res = dict(center=base_xyz + r[..., 0:3], size=r[..., 3:6].contiguous(),
                   dir_class=r[..., 6:6 + nb].contiguous(),
                   dir_res_norm=r[..., 6 + nb:6 + 2 * nb].contiguous(),
                   obj_scores=c[..., 0:2].contiguous())
        res["dir_res"] = res["dir_res_norm"] * (np.pi / nb)
        if c.shape[-1] > 2:
            res["sem_scores"] = c[..., 2:].contiguous()
        return res

    def decode(self, out):                                     # coder.py:168-194
        B, N, _ = out["center"].shape
        dc = torch.argmax(out["dir_class"], -1)
        dr = torch.gather(out["dir_res"], -1, dc.unsqueeze(-1)).squeeze(-1)
        ang = self.base.class2angle(dc, dr.clone()).reshape(B, N, 1) % (2 * np.pi)
        return torch.cat([out["center"], out["size"], ang], dim=-1)

    @staticmethod
    def decode_corners(center, size):                          # coder.py:242-251
        return torch.cat([center - size / 2.0, center + size / 2.0], dim=-1)


# ---------------------------------------------------------------- fusion decoder layer
class PositionEmbeddingLearned(nn.Module):
    """demf/modeling/layers/transformer.py:18-36."""

    def __init__(self, cin, cpos):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(cin, cpos, 1), nn.BatchNorm1d(cpos), nn.ReLU(inplace=True),
            nn.Conv1d(cpos, cpos, 1))

    def forward(self, xyz):
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class FusionLayer(nn.Module):
    """DeMFTransformerDecoderLayer - demf/modeling/layers/transformer.py:39-80."""

    def __init__(self, transformerlayers, posembed):
        super().__init__()
        t = dict(transformerlayers)
        t.pop("type", None)
        self.layer = deps.DetrTransformerDecoderLayer(**t)
        self.posembed = PositionEmbeddingLearned(posembed["input_channel"], posembed["num_pos_feats"])

    def forward(self, query, query_pos, reference_points, valid_ratios, **kw):
        ref = reference_points[:, :, None] * valid_ratios[:, None]          # :62-68 (2-d case)
        pos = self.posembed(query_pos).permute(2, 0, 1)                      # :70-71
        return self.layer(query, query_pos=pos, reference_points=ref, **kw)  # :73-78


# ---------------------------------------------------------------- the head
class OracleHead(nn.Module):
    """DeMFVoteHead - demf/modeling/heads/class_agnostic_vote_head.py:335-941."""

    def __init__(self, kw):
        super().__init__()
        self.kw = kw
        self.num_classes = kw["num_classes"]
        self.gt_per_seed = kw["vote_module_cfg"]["gt_per_seed"]                 # :361
        self.num_proposal = kw["vote_aggregation_cfg"]["num_point"]             # :362
        self.coder = Coder(kw["bbox_coder"]["num_dir_bins"])
        self.nb = self.coder.nb
        self.losses = {k: deps.build_loss(kw[k]) for k in
                       ("objectness_loss", "center_loss", "dir_res_loss", "dir_class_loss",
                        "size_res_loss", "semantic_loss", "iou_loss")}          # :364-376
        self.vote_module = deps.VoteModule(**kw["vote_module_cfg"])             # :382
        self.vote_aggregation = deps.build_sa_module(kw["vote_aggregation_cfg"])  # :383
        dec = kw["decoder"]
        self.num_layers = dec["num_layers"]                                     # :386
        self.decoder = nn.ModuleList([FusionLayer(dec["transformerlayers"], dec["posembed"])
                                      for _ in range(self.num_layers)])         # :388-391
        pl = dict(kw["pred_layer_cfg"])
        assert pl.pop("conv_pred_layers") == self.num_layers + 1                # :394-395
        for i in range(self.num_layers + 1):                                    # :397-403
            self.add_module(f"conv_pred{i}", deps.BaseConvBboxHead(
                **pl, num_cls_out_channels=self.num_classes + 2,
                num_reg_out_channels=6 + 2 * self.nb))
        self.train_cfg = kw["train_cfg"]

    def conv_pred(self, i):
        return getattr(self, f"conv_pred{i}")

    # :714-754 (+ the inherited VoteHead.multiclass_nms_single, oracle/deps.py)
    def get_bboxes(self, points, decode_res_all, use_nms=True):
        """-> per scene (boxes (n,7) bottom-centre form, scores (n,), labels (n,))."""
        tc = self.kw["test_cfg"]
        obj, sem, box = [], [], []
        for i in tc["ensemble_layers"]:                                        # :724-731
            d = decode_res_all[i]
            obj.append(F.softmax(d["obj_scores"], dim=-1)[..., -1])
            sem.append(F.softmax(d["sem_scores"], dim=-1))
            box.append(self.coder.decode(d))
        obj, sem, box = torch.cat(obj, 1), torch.cat(sem, 1), torch.cat(box, 1)  # :733-735
        if not use_nms:
            return box
        shell = type("H", (), dict(test_cfg=tc, bbox_coder=type("C", (), dict(with_rot=True))()))()
        meta = dict(box_type_3d=deps.DepthInstance3DBoxes)
        return [deps.multiclass_nms_single(shell, obj[b], sem[b], box[b], points[b, ..., :3], meta)
                for b in range(box.shape[0])]                                  # :737-751

    # :405-466 (sample_mod == 'seed', the mode configs/demf/demf_votenet.py:171 selects)
    def forward(self, seed_points, seed_features, seed_indices, img_features, img_metas):
        vote_points, vote_features, vote_offset = self.vote_module(seed_points, seed_features)
        sample = O.furthest_point_sample(seed_points, self.num_proposal)        # :429-430
        agg_pts, feats, agg_idx = self.vote_aggregation(points_xyz=vote_points,
                                                        features=vote_features, indices=sample)
        return dict(seed_points=seed_points, seed_indices=seed_indices, vote_points=vote_points,
                    vote_features=vote_features, vote_offset=vote_offset,
                    aggregated_points=agg_pts, aggregated_indices=agg_idx,
                    decode_res_all=self.fuse(feats, agg_pts, img_features, img_metas))

    # :468-512
    def fuse(self, features, agg_pts, img_features, img_metas):
        res_all = []
        res = self.coder.split_pred(*self.conv_pred(0)(features), agg_pts)
        res_all.append(res)
        inp = self.decoder_inputs(agg_pts, img_features, img_metas)
        query = features.permute(2, 0, 1)
        for i in range(self.num_layers):
            qpos = torch.cat([res["center"], res["size"]], dim=-1).detach().clone()
            query = self.decoder[i](query, qpos, inp["reference_points"], inp["valid_ratios"],
                                    key=None, value=inp["feat_flatten"],
                                    key_padding_mask=inp["mask_flatten"],
                                    spatial_shapes=inp["spatial_shapes"],
                                    level_start_index=inp["level_start_index"])
            res = self.coder.split_pred(*self.conv_pred(i + 1)(query.permute(1, 2, 0)), agg_pts)
            res_all.append(res)
        return res_all

    # :524-547
    @staticmethod
    def reference_points(seeds, img_metas):
        uv_all = []
        for pts, meta in zip(seeds, img_metas):
            h, w = meta["img_shape"][:2]
            depth = deps.apply_3d_transformation(pts, "DEPTH", meta, reverse=True)
            uv = deps.points_cam2img(depth, depth.new_tensor(meta["depth2img"]))
            uv = deps.coord_2d_transform(meta, uv, True)
            uv = torch.stack([uv[:, 0] / (w - 1), uv[:, 1] / (h - 1)], -1)
            uv_all.append(torch.clamp(uv, 0, 1))
        return torch.stack(uv_all, 0)

    # :549-594 (+ get_valid_ratio :514-522)
    def decoder_inputs(self, seeds, mlvl_feats, img_metas):
        B = mlvl_feats[0].size(0)
        H, W = img_metas[0]["batch_input_shape"]
        pad = mlvl_feats[0].new_ones((B, H, W))
        for b, meta in enumerate(img_metas):
            pad[b, :meta["img_shape"][0], :meta["img_shape"][1]] = 0
        masks = [F.interpolate(pad[None], size=f.shape[-2:]).to(torch.bool).squeeze(0)
                 for f in mlvl_feats]
        shapes = torch.as_tensor([f.shape[-2:] for f in mlvl_feats], dtype=torch.long)
        ratios = []
        for m in masks:
            _, h, w = m.shape
            ratios.append(torch.stack([(~m[:, 0, :]).sum(1).float() / w,
                                       (~m[:, :, 0]).sum(1).float() / h], -1))
        return dict(
            reference_points=self.reference_points(seeds, img_metas),
            feat_flatten=torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1).permute(1, 0, 2),
            mask_flatten=torch.cat([m.flatten(1) for m in masks], 1),
            spatial_shapes=shapes,
            level_start_index=torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])),
            valid_ratios=torch.stack(ratios, 1))

    # :818-941, one scene
    def targets_single(self, points, boxes, labels, agg_pts):
        N = points.shape[0]
        vt = points.new_zeros((N, 3 * self.gt_per_seed))
        vmask = points.new_zeros((N,), dtype=torch.long)
        slot = points.new_zeros((N,), dtype=torch.long)
        inside = boxes.points_in_boxes(points)                                   # :834
        centers = boxes.gravity_center
        for i in range(labels.shape[0]):                                         # :835-858
            sel = torch.nonzero(inside[:, i], as_tuple=False).squeeze(-1)
            vote = centers[i].unsqueeze(0) - points[sel, :3]
            vmask[sel] = 1
            cur = vt[sel]
            for j in range(self.gt_per_seed):
                rows = torch.nonzero(slot[sel] == j, as_tuple=False).squeeze(-1)
                cur[rows, 3 * j:3 * j + 3] = vote[rows]
                if j == 0:
                    cur[rows] = vote[rows].repeat(1, self.gt_per_seed)
            vt[sel] = cur
            slot[sel] = torch.clamp(slot[sel] + 1, max=2)
        ctr, size, dcls, dres, yaw = self.coder.encode(boxes)                    # :877-879
        d1, _, assign, _ = deps.chamfer_distance(agg_pts.unsqueeze(0), ctr.unsqueeze(0),
                                                 reduction="none")               # :882-885
        assign = assign.squeeze(0)
        euc = torch.sqrt(d1.squeeze(0) + 1e-6)
        pos, neg = self.train_cfg["pos_distance_thr"], self.train_cfg["neg_distance_thr"]
        omask = points.new_zeros(agg_pts.shape[0])
        omask[euc < pos] = 1.0
        omask[euc > neg] = 1.0
        ctr_t, size_t = ctr[assign], size[assign]
        dres_t = dres[assign] / (np.pi / self.nb)                                # :898
        canon = deps.rotation_3d_in_axis((agg_pts - ctr_t).unsqueeze(0).transpose(0, 1),
                                         -yaw[assign], 2).squeeze(1)             # :905-911
        half = size_t / 2.0
        dist_t = torch.cat([half - canon, half + canon], dim=-1)                 # :913-929
        otgt = ((euc < pos) & (dist_t >= 0.0).all(dim=-1)).long()                # :930-934
        return (vt, vmask, size_t, ctr_t, dcls[assign], dres_t, labels[assign].long(), otgt,
                omask, dist_t, yaw[assign])

    # :756-816
    def targets(self, points, gt_boxes, gt_labels, agg_pts):
        gt_boxes, gt_labels = list(gt_boxes), list(gt_labels)
        for i in range(len(gt_labels)):                                          # :766-773
            if len(gt_labels[i]) == 0:
                gt_boxes[i] = deps.DepthInstance3DBoxes(torch.zeros(1, 7))
                gt_labels[i] = gt_labels[i].new_zeros(1)
        per = [self.targets_single(points[b], gt_boxes[b], gt_labels[b], agg_pts[b])
               for b in range(len(gt_labels))]
        (vt, vmask, size_t, ctr_t, dcls, dres, sem, otgt, omask, dist_t, yaw) = \
            [torch.stack(x) for x in zip(*per)]
        oweights = omask / (torch.sum(omask) + 1e-6)                             # :801-802
        bweights = otgt.float() / (torch.sum(otgt).float() + 1e-6)               # :803-804
        return dict(vote_targets=vt, vote_target_masks=vmask, dir_class_targets=dcls,
                    dir_res_targets=dres, mask_targets=sem, objectness_targets=otgt,
                    objectness_weights=oweights, box_loss_weights=bweights,
                    distance_targets=dist_t, dir_targets=yaw, size_targets=size_t,
                    center_targets=ctr_t)

    # :622-712, one decode layer
    def layer_loss(self, common, res, t):
        Lf = self.losses
        w3 = t["box_loss_weights"].unsqueeze(-1).repeat(1, 1, 3)
        onehot = torch.zeros(res["dir_class"].shape[:2] + (self.nb,))
        onehot.scatter_(2, t["dir_class_targets"].unsqueeze(-1), 1)              # :673-676
        out = dict(
            vote_loss=self.vote_module.get_loss(common["seed_points"], common["vote_points"],
                                                common["seed_indices"], t["vote_target_masks"],
                                                t["vote_targets"]),
            objectness_loss=Lf["objectness_loss"](res["obj_scores"].transpose(2, 1),
                                                  t["objectness_targets"],
                                                  weight=t["objectness_weights"]),
            dir_class_loss=Lf["dir_class_loss"](res["dir_class"].transpose(2, 1),
                                                t["dir_class_targets"], weight=t["box_loss_weights"]),
            dir_res_loss=Lf["dir_res_loss"](torch.sum(res["dir_res_norm"] * onehot, -1),
                                            t["dir_res_targets"], weight=t["box_loss_weights"]),
            size_res_loss=Lf["size_res_loss"](res["size"], t["size_targets"], weight=w3),
            center_loss=Lf["center_loss"](res["center"], t["center_targets"], weight=w3),
            semantic_loss=Lf["semantic_loss"](res["sem_scores"].transpose(2, 1), t["mask_targets"],
                                              weight=t["box_loss_weights"]),
            iou_loss=Lf["iou_loss"](self.coder.decode_corners(res["center"], res["size"]),
                                    self.coder.decode_corners(t["center_targets"], t["size_targets"]),
                                    weight=t["box_loss_weights"]))               # :700-707
        return out

    # :596-620
    def loss(self, preds, points, gt_boxes, gt_labels):
        t = self.targets(points, gt_boxes, gt_labels, preds["aggregated_points"])
        per = [self.layer_loss(preds, res, t) for res in preds["decode_res_all"]]
        n = self.num_layers + 1
        return {k: sum(p[k] for p in per) / n for k in per[0]}, t


class OracleDeMF(nn.Module):
    """Backbone + head, fed like DeMFVoteNet.forward_train from demfnet.py:150 on."""

    def __init__(self, cfg):
        super().__init__()
        from demf_amd.config import head_kwargs  # cfg -> the reference's kwargs dict
        b = cfg.backbone
        self.pts_backbone = deps.PointNet2SASSG(
            in_channels=b.in_channels, num_points=b.num_points, radius=b.radius,
            num_samples=b.num_samples, sa_channels=b.sa_channels, fp_channels=b.fp_channels,
            use_xyz=b.use_xyz, normalize_xyz=b.normalize_xyz)
        self.pts_bbox_head = OracleHead(head_kwargs(cfg))

    def forward_head(self, points, img_features, img_metas):
        x = self.pts_backbone(points)                                            # demfnet.py:151
        return self.pts_bbox_head(x["fp_xyz"][-1], x["fp_features"][-1], x["fp_indices"][-1],
                                  img_features, img_metas)                       # demfnet.py:165

    def forward_train(self, points, img_features, img_metas, gt_boxes, gt_labels):
        preds = self.forward_head(points, img_features, img_metas)
        boxes = [deps.DepthInstance3DBoxes(b) for b in gt_boxes]
        losses, targets = self.pts_bbox_head.loss(preds, points, boxes, gt_labels)  # demfnet.py:167
        return losses, preds, targets


# ---------------------------------------------------------------- frozen image stream (a15 / 8f-1)
class OracleEncoder(nn.Module):
    """DeformableDetrEncoder - demf/modeling/layers/deform_detr_encoder.py:12-154."""

    def __init__(self, encoder, positional_encoding, num_feature_levels=4, embed_dims=256):
        super().__init__()
        self.encoder = deps.build_transformer_layer_sequence(encoder)           # :24
        self.positional_encoding = deps.build_positional_encoding(positional_encoding)  # :25-26
        self.level_embeds = nn.Parameter(torch.zeros(num_feature_levels, embed_dims))  # :28-29

    @staticmethod
    def valid_ratio(mask):                                                       # :38-46
        _, H, W = mask.shape
        vh = torch.sum(~mask[:, :, 0], 1).float() / H
        vw = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([vw, vh], -1)

    @staticmethod
    def reference_points(spatial_shapes, valid_ratios):                          # :48-66
        refs = []
        for lvl, (H, W) in enumerate(spatial_shapes):
            ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W),
                                    indexing="ij")
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            refs.append(torch.stack((rx, ry), -1))
        return torch.cat(refs, 1)[:, :, None] * valid_ratios[:, None]

    def forward(self, mlvl_feats, img_metas):                                    # :68-154
        B = mlvl_feats[0].size(0)
        ih, iw = img_metas[0]["batch_input_shape"]
        img_masks = mlvl_feats[0].new_ones((B, ih, iw))
        for i in range(B):
            h, w, _ = img_metas[i]["img_shape"]
            img_masks[i, :h, :w] = 0
        masks = [F.interpolate(img_masks[None], size=f.shape[-2:]).to(torch.bool).squeeze(0)
                 for f in mlvl_feats]
        pos = [self.positional_encoding(m) for m in masks]
        feat_f, mask_f, pos_f, shapes = [], [], [], []
        for lvl, (f, m, p) in enumerate(zip(mlvl_feats, masks, pos)):
            shapes.append(tuple(f.shape[-2:]))
            feat_f.append(f.flatten(2).transpose(1, 2))
            mask_f.append(m.flatten(1))
            pos_f.append(p.flatten(2).transpose(1, 2) + self.level_embeds[lvl].view(1, 1, -1))
        feat_f, mask_f, pos_f = torch.cat(feat_f, 1), torch.cat(mask_f, 1), torch.cat(pos_f, 1)
        ss = torch.as_tensor(shapes, dtype=torch.long)
        lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
        vr = torch.stack([self.valid_ratio(m) for m in masks], 1)
        memory = self.encoder(query=feat_f.permute(1, 0, 2), key=None, value=None,
                              query_pos=pos_f.permute(1, 0, 2), query_key_padding_mask=mask_f,
                              spatial_shapes=ss, reference_points=self.reference_points(shapes, vr),
                              level_start_index=lsi, valid_ratios=vr)
        memory = memory.permute(1, 2, 0)
        outs, start = [], 0
        C = memory.shape[1]
        for h, w in shapes:
            outs.append(memory[:, :, start:start + h * w].reshape(B, C, h, w))
            start += h * w
        return outs


class OracleImageStream(nn.Module):
    """DeMFVoteNet.extract_img_feat (demfnet.py:124-132): backbone -> neck -> encoder, eval."""

    def __init__(self, base=64, blocks=(3, 4, 6, 3), embed_dims=256, num_layers=6, num_heads=8,
                 feedforward_channels=1024, gn_groups=32, num_feats=None):
        super().__init__()
        self.img_backbone = deps.ResNet50((1, 2, 3), base, blocks)
        self.img_neck = deps.ChannelMapper([base * 4 * 2 ** i for i in (1, 2, 3)], embed_dims, 4, gn_groups)
        self.img_encoder = OracleEncoder(
            dict(type="DetrTransformerEncoder", num_layers=num_layers, transformerlayers=dict(
                type="BaseTransformerLayer", attn_cfgs=dict(
                    type="MultiScaleDeformableAttention", embed_dims=embed_dims, num_heads=num_heads),
                feedforward_channels=feedforward_channels, ffn_dropout=0.1,
                operation_order=("self_attn", "norm", "ffn", "norm"))),
            dict(type="SinePositionalEncoding",
                 num_feats=num_feats if num_feats is not None else embed_dims // 2,
                 normalize=True, offset=-0.5), 4, embed_dims)
        self.eval()

    @torch.no_grad()
    def forward(self, img, img_metas):
        return self.img_encoder(self.img_neck(self.img_backbone(img)), img_metas)
