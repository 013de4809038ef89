"""Pins the oracle's restatement of the reference's IN-TREE code (oracle/model.py) against
golden vectors produced by the real reference files (oracle/pin_reference.py):
DeMFVoteHead forward / prepare_decoder_inputs / get_targets / loss + backward,
DeMFTransformerDecoderLayer, PositionEmbeddingLearned, DeMFClassAgnosticBBoxCoder."""
import os

import numpy as np
import pytest
import torch

from oracle import deps, fixtures
from oracle.model import Coder, OracleDeMF, PositionEmbeddingLearned

CASES = [("tiny_a", 1, 2, 4), ("tiny_b", 2, 3, 2)]


def run_oracle(name, seed, B, n_gt, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"ref_head_{name}.npz"))
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(B, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=seed, n_gt=n_gt)
    model = OracleDeMF(cfg)
    fixtures.seed_weights(model, seed)
    model.train()
    points = torch.from_numpy(batch["points"])
    feats = [torch.from_numpy(f) for f in batch["img_features"]]
    gtb = [torch.from_numpy(gold[f"gt_boxes.{b}"]) for b in range(B)]
    gtl = [torch.from_numpy(gold[f"gt_labels.{b}"]) for b in range(B)]
    losses, preds, targets = model.forward_train(points, feats, batch["img_metas"], gtb, gtl)
    return gold, model, losses, preds, targets


@pytest.mark.parametrize("name,seed,B,n_gt", CASES)
def test_head_matches_real_reference(name, seed, B, n_gt, golden_dir):
    gold, model, losses, preds, targets = run_oracle(name, seed, B, n_gt, golden_dir)
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(preds[k].numpy(), gold[k])
    for k in ("seed_points", "vote_points", "vote_offset", "aggregated_points"):
        np.testing.assert_allclose(preds[k].detach().numpy(), gold[k], rtol=1e-5, atol=1e-6)
    for i, d in enumerate(preds["decode_res_all"]):
        for k, v in d.items():
            np.testing.assert_allclose(v.detach().numpy(), gold[f"decode{i}.{k}"], rtol=1e-4,
                                       atol=1e-5, err_msg=f"decode{i}.{k}")
    for k, v in targets.items():
        g = gold["target." + k]
        if v.dtype in (torch.long, torch.int32):
            np.testing.assert_array_equal(v.numpy(), g, err_msg=k)
        else:
            np.testing.assert_allclose(v.detach().numpy(), g, rtol=1e-5, atol=1e-6, err_msg=k)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), gold["loss." + k], rtol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    gn = {n: p.grad.double().norm().item() for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(gn) == list(gold["grad_names"])
    np.testing.assert_allclose([gn[n] for n in sorted(gn)], gold["grad_norms"], rtol=2e-4, atol=1e-7)
    small = "pts_bbox_head.decoder.0.layer.attentions.1.attention_weights.bias"
    np.testing.assert_allclose(dict(model.named_parameters())[small].grad.numpy(),
                               gold["grad." + small], rtol=1e-4, atol=1e-6)


def test_decoder_inputs_match_real_reference(golden_dir):
    gold, model, _, preds, _ = run_oracle(*CASES[0], golden_dir)
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=1, n_gt=4)
    feats = [torch.from_numpy(f) for f in batch["img_features"]]
    inp = model.pts_bbox_head.decoder_inputs(preds["aggregated_points"].detach(), feats,
                                             batch["img_metas"])
    np.testing.assert_allclose(inp["reference_points"].numpy(), gold["reference_points"],
                               rtol=1e-5, atol=1e-6)
    assert 0 < (gold["reference_points"] > 0).mean() and (gold["reference_points"] < 1).any()
    np.testing.assert_array_equal(inp["valid_ratios"].numpy(), gold["valid_ratios"])
    np.testing.assert_array_equal(np.packbits(inp["mask_flatten"].numpy(), axis=1),
                                  gold["mask_flatten"])
    np.testing.assert_array_equal(inp["spatial_shapes"].numpy(), gold["spatial_shapes"])
    np.testing.assert_array_equal(inp["level_start_index"].numpy(), gold["level_start_index"])
    np.testing.assert_array_equal(inp["feat_flatten"][::37, :, ::5].numpy(),
                                  gold["feat_flatten_probe"])


def test_coder_and_posembed_match_real_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_glue.npz"))
    coder = Coder(12)
    sp = coder.split_pred(torch.from_numpy(g["coder.in.cls"]), torch.from_numpy(g["coder.in.reg"]),
                          torch.from_numpy(g["coder.in.base"]))
    for k, v in sp.items():
        np.testing.assert_array_equal(v.numpy(), g["coder.split." + k], err_msg=k)
    np.testing.assert_allclose(coder.decode(sp).numpy(), g["coder.decode"], rtol=1e-6)
    np.testing.assert_array_equal(coder.decode_corners(sp["center"], sp["size"].abs()).numpy(),
                                  g["coder.corners"])
    boxes = deps.DepthInstance3DBoxes(torch.from_numpy(g["coder.encode.boxes"]))
    for n, v in zip(("center", "size", "dir_class", "dir_res", "dir"), coder.encode(boxes)):
        np.testing.assert_array_equal(v.numpy(), g["coder.encode." + n], err_msg=n)
    pe = PositionEmbeddingLearned(6, 16)
    fixtures.seed_weights(pe, 3)
    pe.train()
    np.testing.assert_allclose(pe(torch.from_numpy(g["posembed.in"])).detach().numpy(),
                               g["posembed.out"], rtol=1e-5, atol=1e-6)


def test_oracle_get_bboxes_vs_real_reference(golden_dir):
    """oracle/model.py's restatement of DeMFVoteHead.get_bboxes (class_agnostic_vote_head.py:714-754)
    reproduces the real reference's survivors, scores and labels bit for bit."""
    import os
    import numpy as np
    import torch
    from oracle import fixtures
    from oracle.model import OracleDeMF
    gold = np.load(os.path.join(golden_dir, "ref_bboxes.npz"))
    head = OracleDeMF(fixtures.tiny_cfg()).pts_bbox_head
    for seed in (0, 1):
        pts, dec = fixtures.make_decode_results(seed)
        dd = [{k: torch.from_numpy(v) for k, v in d.items()} for d in dec]
        np.testing.assert_array_equal(head.get_bboxes(torch.from_numpy(pts), dd, use_nms=False).numpy(),
                                      gold[f"s{seed}.bbox3d"])
        for b, (bx, sc, lb) in enumerate(head.get_bboxes(torch.from_numpy(pts), dd)):
            np.testing.assert_array_equal(bx.numpy(), gold[f"s{seed}.b{b}.boxes"])
            np.testing.assert_array_equal(sc.numpy(), gold[f"s{seed}.b{b}.scores"])
            np.testing.assert_array_equal(lb.numpy(), gold[f"s{seed}.b{b}.labels"])


def test_oracle_image_stream_vs_real_encoder(golden_dir):
    """oracle/model.py's OracleImageStream (restated ResNet-50 / ChannelMapper + the restatement of
    deform_detr_encoder.py) against the REAL reference DeformableDetrEncoder's output maps
    (tests/golden/ref_encoder.npz, generated by oracle/pin_reference.py): bit-exact."""
    import os
    import numpy as np
    import torch
    from oracle import fixtures
    from oracle.model import OracleImageStream
    gold = np.load(os.path.join(golden_dir, "ref_encoder.npz"))
    m = OracleImageStream(**fixtures.TINY_IMAGE_STREAM)
    fixtures.seed_weights(m, 4)
    img, metas = fixtures.make_images(4)
    pyramid = m.img_neck(m.img_backbone(torch.from_numpy(img)))
    for i, p in enumerate(pyramid):
        np.testing.assert_array_equal(p.detach().numpy(), gold[f"neck{i}"])
    for i, o in enumerate(m(torch.from_numpy(img), metas)):
        np.testing.assert_array_equal(o.numpy(), gold[f"enc{i}"])


def test_oracle_encoder_vs_real_encoder_at_the_reference_shape(golden_dir):
    """The same at the reference's own encoder shape (configs/demf/demf_votenet.py:28-47: 256 dims, 8 heads,
    4 levels, 4 points, FFN 1024): tests/golden/ref_encoder256.npz is the REAL class's output on
    fixtures.make_encoder_pyramid(6); the oracle's restatement reproduces it bit for bit."""
    import os
    import numpy as np
    import torch
    from oracle import fixtures
    gold = np.load(os.path.join(golden_dir, "ref_encoder256.npz"))
    enc = fixtures.oracle_encoder(**fixtures.ENC256)
    assert sorted(enc.state_dict()) == list(gold["state_keys"])
    fixtures.seed_weights(enc, 6)
    enc.eval()
    feats, metas = fixtures.make_encoder_pyramid(6)
    with torch.no_grad():
        outs = enc([torch.from_numpy(f) for f in feats], metas)
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o.numpy(), gold[f"enc{i}"])


def test_oracle_vs_real_head_on_conditioned_weights(golden_dir):
    """tests/golden/ref_head_cond.npz: the REAL reference head on CONDITIONED weights (200 AdamW steps of the fp64
    oracle, oracle/pin_reference.py: conditioned_goldens) and a held-out batch.  The fp32 oracle reproduces its decode
    outputs, losses and EVERY parameter gradient to fp32 round-off - the pin of the checker the GPU test
    (tests/test_gpu_model.py: test_hot_path_on_conditioned_weights) then leans on."""
    import os
    import numpy as np
    import torch
    from oracle import fixtures
    from oracle.model import OracleDeMF
    from oracle.pin_reference import COND as c
    gold = np.load(os.path.join(golden_dir, "ref_head_cond.npz"))
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(c["B"], c["N"], fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=c["eval_seed"], n_gt=c["n_gt"])
    m = OracleDeMF(cfg)
    m.load_state_dict({k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w.")})
    m.train()
    gtb = [torch.from_numpy(gold[f"gt_boxes.{b}"]) for b in range(c["B"])]
    gtl = [torch.from_numpy(gold[f"gt_labels.{b}"]) for b in range(c["B"])]
    losses, preds, _ = m.forward_train(torch.from_numpy(batch["points"]),
                                       [torch.from_numpy(f) for f in batch["img_features"]],
                                       batch["img_metas"], gtb, gtl)
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(preds[k].numpy(), gold[k])
    for i, d in enumerate(preds["decode_res_all"]):
        for k, v in d.items():
            if torch.is_tensor(v):
                np.testing.assert_allclose(v.detach().numpy(), gold[f"decode{i}.{k}"], rtol=0, atol=2e-5, err_msg=k)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), gold["loss." + k], rtol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    top = max(float(np.linalg.norm(gold[k])) for k in gold.files if k.startswith("grad.pts_"))
    n = 0
    for name, p in m.named_parameters():
        if p.grad is None:
            continue
        want = gold["grad." + name].astype(np.float64)
        err = np.linalg.norm(p.grad.double().numpy() - want)
        assert err <= 1e-4 * np.linalg.norm(want) + 1e-9 * top, (name, err, np.linalg.norm(want))
        n += 1
    assert n == len(gold["grad_names"])
