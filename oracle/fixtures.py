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


def make_decode_results(seed, B=2, K=48, N=4096, layers=2, num_classes=10, num_dir_bins=12):
    """Synthetic inputs of DeMFVoteHead.get_bboxes (class_agnostic_vote_head.py:714-754): a point
    cloud made of clusters, and per decode layer K proposals whose boxes sit on the clusters
    (overlapping duplicates of equal and different classes, empty boxes, low scores) so that
    every branch of the decode -> points-in-box filter -> aligned NMS -> threshold chain fires.
    -> (points (B,N,4) float32, [dict(obj_scores, sem_scores, center, size, dir_class, dir_res)])"""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_cl = 12
    ctr = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 1.5], size=(B, n_cl, 3))
    pts = (ctr[:, :, None, :] + rng.normal(0, 0.12, size=(B, n_cl, N // n_cl, 3))).reshape(B, -1, 3)
    pad = N - pts.shape[1]
    pts = np.concatenate([pts, rng.uniform(-3, 3, size=(B, pad, 3))], 1)
    points = np.concatenate([pts, pts[..., 2:3]], -1).astype(np.float32)
    out = []
    for _ in range(layers):
        which = rng.integers(0, n_cl + 3, size=(B, K))                 # >= n_cl: box on empty space
        c = np.where((which < n_cl)[..., None], np.take_along_axis(
            ctr, np.minimum(which, n_cl - 1)[..., None].repeat(3, -1), 1),
            rng.uniform([-3, -3, 2.5], [3, 3, 3.0], size=(B, K, 3)))
        center = c + rng.normal(0, 0.05, size=(B, K, 3))
        size = rng.uniform(0.3, 0.9, size=(B, K, 3))
        size[:, ::7] *= -1.0                                            # raw regression can be negative
        obj = rng.normal(0, 2.0, size=(B, K, 2))
        sem = rng.normal(0, 1.0, size=(B, K, num_classes))
        sem[np.arange(B)[:, None], np.arange(K)[None], which % 3] += 4.0   # few distinct classes
        dcl = rng.normal(0, 1.0, size=(B, K, num_dir_bins))
        dres = rng.normal(0, 0.1, size=(B, K, num_dir_bins))
        out.append({k: v.astype(np.float32) for k, v in dict(
            obj_scores=obj, sem_scores=sem, center=center, size=size, dir_class=dcl,
            dir_res=dres).items()})
    return points, out


TINY_IMAGE_STREAM = dict(base=8, blocks=(1, 1, 1, 1), embed_dims=32, num_layers=2, num_heads=4,
                         feedforward_channels=64, gn_groups=8, num_feats=16)


def make_images(seed, B=2, H=64, W=96):
    """Synthetic padded image batch + metas for the image stream: (B,3,H,W) float32, every image
    valid on its own (img_h, img_w) <= (H, W)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    metas = []
    for b in range(B):
        h = int(rng.integers(H * 3 // 4, H + 1)) if b else H
        w = int(rng.integers(W * 3 // 4, W + 1)) if b != 1 else W
        img[b, :, h:, :] = 0
        img[b, :, :, w:] = 0
        metas.append(dict(batch_input_shape=(H, W), img_shape=(h, w, 3)))
    return img, metas


ENC256 = dict(num_layers=2, embed_dims=256, num_heads=8, num_points=4, feedforward_channels=1024, num_feats=128)
ENC256_SHAPES = ((10, 14), (5, 7), (3, 4), (2, 2))
ENC256_INPUT = (80, 112)


def make_encoder_pyramid(seed, B=2, shapes=ENC256_SHAPES, input_hw=ENC256_INPUT, C=256):
    """Synthetic neck pyramid (what ChannelMapper hands DeformableDetrEncoder.forward,
    deform_detr_encoder.py:68) + metas with padded images: [(B,C,h_l,w_l) float32], [meta].  Scene 0 fills
    the padded batch shape; the others are valid on a smaller (img_h, img_w), so padding masks, valid ratios
    and masked value rows are all exercised."""
    import numpy as np
    rng = np.random.default_rng(seed)
    H, W = input_hw
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in shapes]
    metas = []
    for b in range(B):
        h = H if b == 0 else int(rng.integers(H * 2 // 3, H))
        w = W if b == 0 else int(rng.integers(W * 2 // 3, W))
        metas.append(dict(batch_input_shape=(H, W), img_shape=(h, w, 3)))
    return feats, metas


def oracle_encoder(num_layers=6, embed_dims=256, num_heads=8, num_points=4, feedforward_channels=1024,
                   num_feats=128):
    """oracle/model.py's restatement of DeformableDetrEncoder, built from the keyword form the product's
    DeformableDetrEncoder takes."""
    from oracle.model import OracleEncoder
    return OracleEncoder(
        dict(type="DetrTransformerEncoder", num_layers=num_layers, transformerlayers=dict(
            type="BaseTransformerLayer", attn_cfgs=dict(
                type="MultiScaleDeformableAttention", embed_dims=embed_dims, num_heads=num_heads,
                num_levels=4, num_points=num_points),
            feedforward_channels=feedforward_channels, ffn_dropout=0.1,
            operation_order=("self_attn", "norm", "ffn", "norm"))),
        dict(type="SinePositionalEncoding", num_feats=num_feats, normalize=True, offset=-0.5), 4, embed_dims)
