"""CPU ORACLE tooling - seeded synthetic inputs / weights shared by the golden-vector
generator (oracle/pin_reference.py), the tests and bench.py's cpu_baseline leg.
Everything is derived from integer seeds so fixtures stay tiny."""
import math

import numpy as np
import torch


class AttrDict(dict):
    """dict with attribute access (stands in for mmcv ConfigDict, which the reference
    reads as ``decoder.num_layers`` at class_agnostic_vote_head.py:386)."""
    __getattr__ = dict.__getitem__


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return type(d)(to_attr(v) for v in d)
    return d


def tiny_cfg():
    """A scaled-down DeMF config (same topology as configs/demf/demf_votenet.py) sized so the
    CPU oracle runs in well under a second."""
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    return DeMFCfg(
        backbone=BackboneCfg(in_channels=4, num_points=(256, 128, 64, 32),
                             radius=(0.4, 0.8, 1.2, 1.6), num_samples=(16, 16, 8, 8),
                             sa_channels=((16, 16, 32), (32, 32, 64), (32, 32, 64), (32, 32, 64)),
                             fp_channels=((64, 64), (64, 64))),
        head=HeadCfg(in_channels=64, shared_conv_channels=(32, 32), embed_dims=64, num_heads=4,
                     num_levels=4, num_points=2, attn_dropout=0.0, feedforward_channels=128,
                     ffn_dropout=0.0, vote_conv_channels=(64, 64), num_proposal=32,
                     agg_radius=0.6, agg_num_sample=8, agg_mlp_channels=(64, 64, 64, 64)))


TINY_PYRAMID = ((16, 22), (8, 11), (4, 6), (2, 3))
TINY_INPUT = (128, 176)


def seed_weights(module, seed=0):
    """Deterministic, well-conditioned weights for every tensor of ``module.state_dict()``
    (sorted-key order, CPU generator) - identical on any machine."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    out = {}
    for k in sorted(sd):
        v = sd[k]
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_mean"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.ones_like(v)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / math.sqrt(fan_in))
        elif k.endswith("sampling_offsets.bias"):
            out[k] = torch.randn(v.shape, generator=g) * 2.0
        elif k.endswith("weight"):  # norm scales
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
    module.load_state_dict(out)
    return out


from demf_amd.synthetic import depth2img, make_scene_batch  # noqa: E402,F401  (seeded scenes)
