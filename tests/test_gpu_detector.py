"""The detector-shaped entry points: ``DeMFVoteNet.simple_test`` /
``forward_train`` (demf/modeling/detectors/demfnet.py:134-170, 254-283) on the GPU - image stream ->
hot path -> test-time decode + NMS as ONE chain - against the CPU oracle (oracle/model.py) fed the
same inputs and weights."""
import numpy as np
import pytest
import torch

from oracle import fixtures
from oracle.model import OracleDeMF, OracleImageStream

pytestmark = pytest.mark.gpu

# the image stream at the reference's encoder shape (256 dims, 8 heads, 4 levels, P = 4, FFN a multiple of 128): the
# encoder then runs on this package's kernel path (rows_gemm + raw MSDA; DeformableDetrEncoder._fused_ok), so the
# detector chain below exercises THAT path against the oracle, not the module path
STREAM = dict(base=16, blocks=(1, 1, 1, 1), embed_dims=256, num_layers=2, num_heads=8,
              feedforward_channels=256, gn_groups=32, num_feats=128)
H, W = 128, 192
PYRAMID = ((16, 24), (8, 12), (4, 6), (2, 3))


def _cfg256():
    """fixtures.tiny_cfg widened where the 256-channel image tokens meet the head."""
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    t = fixtures.tiny_cfg()
    b = t.backbone
    return DeMFCfg(
        backbone=BackboneCfg(in_channels=4, num_points=b.num_points, radius=b.radius, num_samples=b.num_samples,
                             sa_channels=b.sa_channels, fp_channels=((64, 64), (64, 256))),
        head=HeadCfg(in_channels=256, shared_conv_channels=(32, 32), embed_dims=256, num_heads=8,
                     num_levels=4, num_points=2, attn_dropout=0.0, feedforward_channels=128,
                     ffn_dropout=0.0, vote_conv_channels=(64, 64), num_proposal=32,
                     agg_radius=0.6, agg_num_sample=8, agg_mlp_channels=(256, 64, 64, 256)))


def _inputs(seed, B=3, N=4096):
    cfg = _cfg256()
    batch = fixtures.make_scene_batch(B, N, PYRAMID, (H, W), cfg.head.embed_dims, seed=seed, n_gt=4)
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    for b, m in enumerate(batch["img_metas"]):
        h, w = m["img_shape"][:2]
        img[b, :, h:, :] = 0
        img[b, :, :, w:] = 0
    return cfg, batch, img


def _models(cfg, seed):
    from demf_amd.modules import DeMFVoteNet
    det = DeMFVoteNet(cfg, **STREAM)
    fixtures.seed_weights(det, seed)
    with torch.no_grad():
        # eval-mode BN on seeded weights gives near-constant, half-negative sizes: bias the size
        # regression and widen the conv heads so that boxes hold points and scores differ
        for i in range(2):
            head = getattr(det.pts_bbox_head, f"conv_pred{i}")
            head.conv_reg.bias[3:6] = 0.8 + 0.1 * i
            head.conv_reg.weight.mul_(4.0)
            head.conv_cls.weight.mul_(8.0)
    ref_img, ref = OracleImageStream(**STREAM), OracleDeMF(cfg)
    sd = det.state_dict()
    ref_img.load_state_dict({k: v for k, v in sd.items() if k.startswith("img_")})
    ref.load_state_dict({k: v for k, v in sd.items() if k.startswith("pts_")})
    return det.cuda(), ref_img, ref


def _assert_same_detections(got, want, tol=1e-3):
    """Same multiset of (box, score, label) up to ``tol``: every expected detection has its own
    partner among the produced ones (order is not part of the contract, and near-equal scores
    make any sort key unstable under 1e-4 noise)."""
    gb, gs, gl = got
    wb, ws, wl = want
    assert gb.shape == wb.shape, f"survivors {gb.shape} vs {wb.shape}"
    assert sorted(gl.tolist()) == sorted(wl.tolist())
    G = np.concatenate([gb, gs[:, None]], 1)
    Wt = np.concatenate([wb, ws[:, None]], 1)
    used = np.zeros(len(G), bool)
    for r, lab in zip(Wt, wl):
        d = np.abs(G - r).max(1)
        d[used | (gl != lab)] = np.inf
        k = int(np.argmin(d))
        assert d[k] <= tol, f"no partner for {r} (label {lab}): nearest is {d[k]:.2e} away"
        used[k] = True


def test_simple_test_vs_oracle_chain():
    cfg, batch, img = _inputs(3)
    det, ref_img, ref = _models(cfg, 3)
    assert det.img_encoder._fused_ok(torch.empty(3, 10, 256, device="cuda")), "encoder kernel path not taken"
    det.eval()
    ref.eval()
    pts = torch.from_numpy(batch["points"])
    x = torch.from_numpy(img)
    got = det.simple_test([p for p in pts.cuda()], batch["img_metas"], x.cuda())
    assert len(got) == pts.shape[0] and set(got[0]) == {"boxes_3d", "scores_3d", "labels_3d"}
    # the convolutional front is library code (fp32 round-off ~1e-3 of scale): the oracle chain
    # starts from the product's own ResNet/ChannelMapper pyramid, so that everything this package
    # computes itself (encoder -> hot path -> decode -> NMS) is what is compared
    with torch.no_grad():
        pyr = [p.cpu() for p in det.img_neck(det.img_backbone(x.cuda()))]
        feats = ref_img.img_encoder(pyr, batch["img_metas"])
        preds = ref.forward_head(pts, feats, batch["img_metas"])
        want = ref.pts_bbox_head.get_bboxes(pts, preds["decode_res_all"])
    total = 0
    for b in range(pts.shape[0]):
        _assert_same_detections(
            (got[b]["boxes_3d"].tensor.numpy(), got[b]["scores_3d"].numpy(), got[b]["labels_3d"].numpy()),
            (want[b][0].numpy(), want[b][1].numpy(), want[b][2].numpy()))
        total += len(want[b][1])
    assert total > 20, "vacuous: no box survived"
    # forward_test = simple_test on lists over augmentations (:172-238)
    again = det.forward_test([[p for p in pts.cuda()]], [batch["img_metas"]], [x.cuda()])
    for a, b in zip(again, got):
        # (fp32 atomics in the backbone's gather/scatter paths: run-to-run differences of a few ulp)
        assert torch.allclose(a["scores_3d"], b["scores_3d"], rtol=1e-4, atol=1e-6)
    with pytest.raises(TypeError):
        det.forward_test(pts.cuda(), [batch["img_metas"]], [x.cuda()])


def test_forward_train_is_the_hot_path_on_the_streams_tokens():
    cfg, batch, img = _inputs(5)
    det, _, _ = _models(cfg, 5)
    det.train()
    assert not det.img_encoder.training and not any(
        p.requires_grad for n, p in det.named_parameters() if n.startswith("img_"))
    pts = torch.from_numpy(batch["points"]).cuda()
    gtb = [torch.from_numpy(b).cuda() for b in batch["gt_boxes"]]
    gtl = [torch.from_numpy(l).cuda() for l in batch["gt_labels"]]
    x = torch.from_numpy(img).cuda()
    sd0 = {k: v.clone() for k, v in det.state_dict().items()}
    from demf_amd import fused
    fused.rng_state(pts.device, seed=3)
    with torch.no_grad():
        losses = det.forward_train(pts, x, batch["img_metas"], gtb, gtl)
    det.load_state_dict(sd0)                       # undo the BN running-stat update
    from demf_amd.modules import DeMFHotPath
    tokens = det.extract_img_feat(x, batch["img_metas"])
    want = DeMFHotPath.forward_train(det, pts, tokens, batch["img_metas"], gtb, gtl)
    for k in want:
        np.testing.assert_allclose(losses[k].item(), want[k].item(), rtol=1e-5)
    want["_total"].backward()
    assert all(p.grad is not None for n, p in det.named_parameters() if n.startswith("pts_") and p.requires_grad)
    assert all(p.grad is None for n, p in det.named_parameters() if n.startswith("img_"))
