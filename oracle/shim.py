"""CPU ORACLE tooling (runs ONLY in the build container, where /root/reference exists).

Loads the REAL reference source files (demf/modeling/heads/class_agnostic_vote_head.py,
demf/modeling/layers/transformer.py, demf/core/bbox/coders/class_agnostic_bbox_coder.py)
read-only, by registering stand-in ``mmcv*/mmdet*/mmdet3d*`` modules in sys.modules whose
symbols are the oracle's restated dependencies (oracle/deps.py).  This lets the reference's
own in-tree code run on CPU so its outputs can be committed as golden vectors
(oracle/pin_reference.py) - nothing from the reference travels to the GPU box.
"""
import importlib.util
import os
import sys
import types

import torch.nn as nn

from . import deps, torch_ops

REF = os.environ.get("DEMF_REFERENCE", "/root/reference")


class _Registry:
    def __init__(self):
        self.modules = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.modules[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, **extra):
        cfg = dict(cfg)
        t = cfg.pop("type")
        return self.modules[t](**cfg, **extra)


TRANSFORMER_LAYER = _Registry()
HEADS = _Registry()
BBOX_CODERS = _Registry()
TRANSFORMER_LAYER.modules["DetrTransformerDecoderLayer"] = deps.DetrTransformerDecoderLayer


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    passthrough = lambda *a, **k: (lambda f: f)
    _mod("mmcv")
    _mod("mmcv.runner", BaseModule=_BaseModule, force_fp32=passthrough)
    _mod("mmcv.runner.base_module", BaseModule=_BaseModule)
    _mod("mmcv.cnn", xavier_init=deps.xavier_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", TRANSFORMER_LAYER=TRANSFORMER_LAYER)
    _mod("mmcv.cnn.bricks.transformer", build_transformer_layer=TRANSFORMER_LAYER.build,
         MultiScaleDeformableAttention=deps.MultiScaleDeformableAttention,
         build_positional_encoding=deps.build_positional_encoding,
         build_transformer_layer_sequence=deps.build_transformer_layer_sequence)
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn",
         MultiScaleDeformableAttention=deps.MultiScaleDeformableAttention)
    _mod("mmdet")
    _mod("mmdet.core", build_bbox_coder=BBOX_CODERS.build, multi_apply=deps.multi_apply)
    _mod("mmdet.core.bbox")
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=BBOX_CODERS)
    _mod("mmdet.models", HEADS=HEADS)
    _mod("mmdet.models.builder", HEADS=HEADS)
    _mod("mmdet3d")
    _mod("mmdet3d.core")
    _mod("mmdet3d.core.bbox", points_cam2img=deps.points_cam2img)
    _mod("mmdet3d.core.bbox.structures", rotation_3d_in_axis=deps.rotation_3d_in_axis)
    _mod("mmdet3d.core.bbox.coders", PartialBinBasedBBoxCoder=deps.PartialBinBasedBBoxCoder)
    _mod("mmdet3d.models", VoteHead=_BaseModule)
    _mod("mmdet3d.models.losses", chamfer_distance=deps.chamfer_distance)
    _mod("mmdet3d.models.builder", build_loss=deps.build_loss)
    _mod("mmdet3d.models.model_utils", VoteModule=deps.VoteModule)
    _mod("mmdet3d.models.dense_heads")
    _mod("mmdet3d.models.dense_heads.base_conv_bbox_head", BaseConvBboxHead=deps.BaseConvBboxHead)
    _mod("mmdet3d.models.fusion_layers", apply_3d_transformation=deps.apply_3d_transformation,
         coord_2d_transform=deps.coord_2d_transform)
    _mod("mmdet3d.ops", build_sa_module=deps.build_sa_module,
         furthest_point_sample=torch_ops.furthest_point_sample)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None, *a, **k):
        super().__init__()

    def _extract_input(self, feat_dict):
        """mmdet3d VoteHead._extract_input (used at class_agnostic_vote_head.py:408-409)."""
        return feat_dict["seed_points"], feat_dict["seed_features"], feat_dict["seed_indices"]

    def multiclass_nms_single(self, obj_scores, sem_scores, bbox, points, input_meta):
        """mmdet3d VoteHead.multiclass_nms_single (called at class_agnostic_vote_head.py:741)."""
        return deps.multiclass_nms_single(self, obj_scores, sem_scores, bbox, points, input_meta)


def load(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_cache = {}


def reference():
    """-> namespace with the real reference classes."""
    if not _cache:
        assert os.path.isdir(REF), "reference tree not available"
        install()
        coder = load("demf/core/bbox/coders/class_agnostic_bbox_coder.py", "_ref_coder")
        trans = load("demf/modeling/layers/transformer.py", "_ref_transformer")
        head = load("demf/modeling/heads/class_agnostic_vote_head.py", "_ref_head")
        enc = load("demf/modeling/layers/deform_detr_encoder.py", "_ref_encoder")
        _cache.update(coder=coder, transformer=trans, head=head, encoder=enc)
    return types.SimpleNamespace(**_cache)
