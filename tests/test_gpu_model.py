"""GPU parity of the whole hot path (demf_amd.modules on libdemf_hip.so) against
(1) golden vectors from the REAL reference head (tests/golden/ref_head_*.npz) and
(2) the CPU oracle (oracle/model.py) at growing sizes up to the full reference config.
Tolerance: 1e-4 (north-star), indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures
from oracle.model import OracleDeMF

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-4, atol=1e-4)


def _to_dev(batch, gtb, gtl):
    return (torch.from_numpy(batch["points"]).cuda(),
            [torch.from_numpy(f).cuda() for f in batch["img_features"]],
            [torch.from_numpy(b).cuda() for b in gtb], [torch.from_numpy(l).cuda() for l in gtl])


def _grad_norms(model):
    return {n: p.grad.double().norm().item() for n, p in model.named_parameters()
            if p.grad is not None}


@pytest.mark.parametrize("name,seed,B,n_gt", [("tiny_a", 1, 2, 4), ("tiny_b", 2, 3, 2)])
def test_hot_path_vs_real_reference_goldens(name, seed, B, n_gt, golden_dir):
    from demf_amd.modules import DeMFHotPath
    gold = np.load(os.path.join(golden_dir, f"ref_head_{name}.npz"))
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(B, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=seed, n_gt=n_gt)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    gtb = [gold[f"gt_boxes.{b}"] for b in range(B)]
    gtl = [gold[f"gt_labels.{b}"] for b in range(B)]
    points, feats, gb, gl = _to_dev(batch, gtb, gtl)
    preds = model.forward_head(points, feats, batch["img_metas"])
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(preds[k].cpu().numpy(), gold[k])
    for k in ("seed_points", "vote_points", "vote_offset", "aggregated_points"):
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), gold[k], **TOL, err_msg=k)
    for i, d in enumerate(preds["decode_res_all"]):
        for k, v in d.items():
            np.testing.assert_allclose(v.detach().cpu().numpy(), gold[f"decode{i}.{k}"], **TOL,
                                       err_msg=f"decode{i}.{k}")
    losses = model.pts_bbox_head.loss(preds, points, gb, gl, None, None, batch["img_metas"])
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), gold["loss." + k], rtol=1e-4, err_msg=k)
    sum(losses.values()).backward()
    gn = _grad_norms(model)
    assert sorted(gn) == list(gold["grad_names"])
    np.testing.assert_allclose([gn[n] for n in sorted(gn)], gold["grad_norms"], rtol=1e-3, atol=1e-6)
    small = "pts_bbox_head.decoder.0.layer.attentions.1.attention_weights.bias"
    np.testing.assert_allclose(dict(model.named_parameters())[small].grad.cpu().numpy(),
                               gold["grad." + small], rtol=1e-3, atol=1e-5)


def _ball_margin(xyz, center, r):
    d2 = ((xyz[:, None, :, :].astype(np.float64) - center[:, :, None, :].astype(np.float64)) ** 2).sum(-1)
    return np.abs(d2 - r * r).min() / (r * r)


def _run_pair(cfg, B, N, pyramid, in_shape, img_shape, seed):
    from demf_amd.modules import DeMFHotPath
    batch = fixtures.make_scene_batch(B, N, pyramid, in_shape, cfg.head.embed_dims, seed=seed,
                                      n_gt=5, img_shape=img_shape)
    ref = OracleDeMF(cfg)
    fixtures.seed_weights(ref, seed)
    ref.train()
    pts = torch.from_numpy(batch["points"])
    feats = [torch.from_numpy(f) for f in batch["img_features"]]
    # GT boxes: seeded in-room boxes + boxes on a few oracle proposals (positives exist)
    with torch.no_grad():
        agg = ref.forward_head(pts, feats, batch["img_metas"])["aggregated_points"].numpy()
    rng = np.random.default_rng(seed)
    gtb, gtl = [], []
    for b in range(B):
        pick = rng.choice(agg.shape[1], 3, replace=False)
        dims = rng.uniform(0.6, 1.4, size=(3, 3))
        ctr = agg[b, pick] + rng.normal(0, 0.04, size=(3, 3))
        extra = np.concatenate([ctr - [0, 0, 1] * dims * 0.5, dims, rng.uniform(-3, 3, (3, 1))], 1)
        gtb.append(np.concatenate([batch["gt_boxes"][b], extra.astype(np.float32)], 0))
        gtl.append(np.concatenate([batch["gt_labels"][b], rng.integers(0, 10, 3)]))
    losses_r, preds_r, targets_r = ref.forward_train(pts, feats, batch["img_metas"],
                                                     [torch.from_numpy(b) for b in gtb],
                                                     [torch.from_numpy(l) for l in gtl])
    sum(losses_r.values()).backward()
    margin = _ball_margin(preds_r["vote_points"].detach().numpy(),
                          preds_r["aggregated_points"].detach().numpy(), cfg.head.agg_radius)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    points, f_d, gb, gl = _to_dev(batch, gtb, gtl)
    preds = model.forward_head(points, f_d, batch["img_metas"])
    losses = model.pts_bbox_head.loss(preds, points, gb, gl, None, None, batch["img_metas"])
    sum(losses.values()).backward()
    return dict(ref=ref, model=model, preds_r=preds_r, preds=preds, losses_r=losses_r,
                losses=losses, margin=margin)


def _compare(r):
    preds, preds_r = r["preds"], r["preds_r"]
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(preds[k].cpu().numpy(), preds_r[k].numpy())
    for k in ("seed_points", "vote_points", "aggregated_points", "vote_features"):
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), preds_r[k].detach().numpy(),
                                   **TOL, err_msg=k)
    for i, (d, dr) in enumerate(zip(preds["decode_res_all"], preds_r["decode_res_all"])):
        for k in d:
            np.testing.assert_allclose(d[k].detach().cpu().numpy(), dr[k].detach().numpy(), **TOL,
                                       err_msg=f"decode{i}.{k}")
    for k in r["losses"]:
        np.testing.assert_allclose(r["losses"][k].item(), r["losses_r"][k].item(), rtol=1e-4,
                                   err_msg=k)
    gn, gr = _grad_norms(r["model"]), _grad_norms(r["ref"])
    assert sorted(gn) == sorted(gr)
    for n in sorted(gn):
        np.testing.assert_allclose(gn[n], gr[n], rtol=2e-3, atol=1e-6, err_msg=n)
    # full gradient tensors of the fusion kernel's own projections
    for n in ("pts_bbox_head.decoder.0.layer.attentions.1.sampling_offsets.weight",
              "pts_bbox_head.decoder.0.layer.attentions.1.value_proj.weight",
              "pts_backbone.SA_modules.0.mlps.0.layer0.conv.weight"):
        a = dict(r["model"].named_parameters())[n].grad.cpu().numpy()
        b = dict(r["ref"].named_parameters())[n].grad.numpy()
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-4 * np.abs(b).max(), err_msg=n)


def _seeded_run(cfg, B, N, pyramid, in_shape, img_shape):
    # a neighbour within float round-off of the vote-aggregation ball boundary can land on
    # either side on CPU vs GPU (vote_points come out of GEMMs); pick a seed that has none
    for seed in range(1, 8):
        r = _run_pair(cfg, B, N, pyramid, in_shape, img_shape, seed)
        if r["margin"] > 1e-5:
            return r
    pytest.skip("no boundary-safe seed found")


def test_hot_path_vs_oracle_mid_size():
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
                  head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    _compare(_seeded_run(cfg, 2, 6000, ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560),
                         (400, 551)))


def test_hot_path_vs_oracle_full_config():
    """configs/demf/demf_votenet.py sizes: 20 000 points, 800x1120 pyramid, 256 queries."""
    from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg, HeadCfg
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    _compare(_seeded_run(cfg, 2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2]))
