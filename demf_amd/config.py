"""Hyper-parameters of the reference model, as plain dataclasses.

Values are those of configs/demf/demf_votenet.py (cited per field); mmcv's Config
machinery itself is out of scope.
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class BackboneCfg:                      # demf_votenet.py:48-62  (PointNet2SASSG)
    in_channels: int = 4
    num_points: Tuple[int, ...] = (2048, 1024, 512, 256)
    radius: Tuple[float, ...] = (0.2, 0.4, 0.8, 1.2)
    num_samples: Tuple[int, ...] = (64, 32, 16, 16)
    sa_channels: Tuple[Tuple[int, ...], ...] = ((64, 64, 128), (128, 128, 256),
                                                (128, 128, 256), (128, 128, 256))
    fp_channels: Tuple[Tuple[int, ...], ...] = ((256, 256), (256, 256))
    use_xyz: bool = True
    normalize_xyz: bool = True


@dataclass
class HeadCfg:                          # demf_votenet.py:63-163 (DeMFVoteHead)
    num_classes: int = 10
    num_dir_bins: int = 12
    in_channels: int = 256
    shared_conv_channels: Tuple[int, ...] = (128, 128)      # :65-67
    num_decoder_layers: int = 1                              # :70
    embed_dims: int = 256
    num_heads: int = 8                                       # :77,81
    num_levels: int = 4                                      # :82
    num_points: int = 2                                      # :83
    attn_dropout: float = 0.4                                # :78,84
    feedforward_channels: int = 1024                         # :87
    ffn_dropout: float = 0.1                                 # :88
    posembed_input: int = 6                                  # :92-95
    vote_conv_channels: Tuple[int, ...] = (256, 256)         # :146
    gt_per_seed: int = 3                                     # :145
    num_proposal: int = 256                                  # :157
    agg_radius: float = 0.3                                  # :158
    agg_num_sample: int = 16                                 # :159
    agg_mlp_channels: Tuple[int, ...] = (256, 256, 256, 256)  # :160
    pos_distance_thr: float = 0.3                            # :169
    neg_distance_thr: float = 0.6                            # :170
    sample_mod: str = "seed"                                 # :171
    # loss weights, :116-154
    objectness_class_weight: Tuple[float, float] = (0.2, 0.8)
    objectness_loss_weight: float = 5.0
    dir_class_loss_weight: float = 1.0
    dir_res_loss_weight: float = 10.0
    size_res_loss_weight: float = 10.0
    size_res_beta: float = 0.0625
    center_loss_weight: float = 10.0
    center_beta: float = 1.0 / 9.0
    iou_loss_weight: float = 12.0 / 3.0
    semantic_loss_weight: float = 1.0
    vote_loss_dst_weight: float = 10.0


@dataclass
class DeMFCfg:
    backbone: BackboneCfg = field(default_factory=BackboneCfg)
    head: HeadCfg = field(default_factory=HeadCfg)


def head_kwargs(cfg: DeMFCfg):
    """The kwargs dict configs/demf/demf_votenet.py:63-181 passes to DeMFVoteHead."""
    h = cfg.head
    return dict(
        num_classes=h.num_classes,
        bbox_coder=dict(type="DeMFClassAgnosticBBoxCoder", num_dir_bins=h.num_dir_bins, with_rot=True),
        train_cfg=dict(pos_distance_thr=h.pos_distance_thr, neg_distance_thr=h.neg_distance_thr,
                       sample_mod=h.sample_mod),
        test_cfg=dict(sample_mod=h.sample_mod, ensemble_layers=[0, 1], nms_thr=0.25,
                      score_thr=0.05, per_class_proposal=True),
        vote_module_cfg=dict(in_channels=h.in_channels, vote_per_seed=1, gt_per_seed=h.gt_per_seed,
                             conv_channels=h.vote_conv_channels, norm_feats=True,
                             vote_loss=dict(type="ChamferDistance", mode="l1", reduction="none",
                                            loss_dst_weight=h.vote_loss_dst_weight)),
        vote_aggregation_cfg=dict(type="PointSAModule", num_point=h.num_proposal,
                                  radius=h.agg_radius, num_sample=h.agg_num_sample,
                                  mlp_channels=list(h.agg_mlp_channels), use_xyz=True,
                                  normalize_xyz=True),
        pred_layer_cfg=dict(in_channels=h.in_channels, shared_conv_channels=h.shared_conv_channels,
                            bias=True, conv_pred_layers=h.num_decoder_layers + 1),
        decoder=dict(type="DeMFTransformerDecoderLayer", num_layers=h.num_decoder_layers,
                     transformerlayers=dict(
                         type="DetrTransformerDecoderLayer",
                         attn_cfgs=[dict(type="MultiheadAttention", embed_dims=h.embed_dims,
                                         num_heads=h.num_heads, dropout=h.attn_dropout),
                                    dict(type="MultiScaleDeformableAttention",
                                         num_heads=h.num_heads, num_levels=h.num_levels,
                                         num_points=h.num_points, dropout=h.attn_dropout,
                                         embed_dims=h.embed_dims)],
                         feedforward_channels=h.feedforward_channels, ffn_dropout=h.ffn_dropout,
                         operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")),
                     posembed=dict(input_channel=h.posembed_input, num_pos_feats=h.embed_dims)),
        objectness_loss=dict(type="CrossEntropyLoss", class_weight=list(h.objectness_class_weight),
                             reduction="sum", loss_weight=h.objectness_loss_weight),
        dir_class_loss=dict(type="CrossEntropyLoss", reduction="sum", loss_weight=h.dir_class_loss_weight),
        dir_res_loss=dict(type="SmoothL1Loss", reduction="sum", loss_weight=h.dir_res_loss_weight),
        size_class_loss=dict(type="CrossEntropyLoss", reduction="sum", loss_weight=1.0),
        size_res_loss=dict(type="SmoothL1Loss", reduction="sum", loss_weight=h.size_res_loss_weight,
                           beta=h.size_res_beta),
        center_loss=dict(type="SmoothL1Loss", beta=h.center_beta, reduction="sum",
                         loss_weight=h.center_loss_weight),
        iou_loss=dict(type="AxisAlignedIoULoss", reduction="sum", loss_weight=h.iou_loss_weight),
        semantic_loss=dict(type="CrossEntropyLoss", reduction="sum", loss_weight=h.semantic_loss_weight),
    )



# image pyramid of the reference pipeline: Resize((1333,800), keep_ratio) of a
# 530x730 SUN RGB-D frame -> 800x1102, Pad(32) -> 800x1120 (demf_votenet.py:194-197);
# strides 8/16/32/64 (configs/deformdetr/imvotenet_image.py:7,13-20).
IMG_SHAPE = (800, 1102, 3)
BATCH_INPUT_SHAPE = (800, 1120)
PYRAMID_SHAPES = ((100, 140), (50, 70), (25, 35), (13, 18))
