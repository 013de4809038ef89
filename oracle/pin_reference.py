"""Golden-vector generator: runs the REAL reference in-tree code on CPU and commits its
outputs as small fixtures under tests/golden/.

    python -m oracle.pin_reference            (build container only: needs /root/reference)

What is real and what is restated:
  REAL  (imported from /root/reference, read-only, via oracle/shim.py):
        DeMFVoteHead.{forward, transformer_decoder, get_valid_ratio, get_reference_points,
        prepare_decoder_inputs, loss, _loss, get_targets, get_targets_single}
        (demf/modeling/heads/class_agnostic_vote_head.py:335-941),
        DeMFTransformerDecoderLayer + PositionEmbeddingLearned
        (demf/modeling/layers/transformer.py:18-80),
        DeMFClassAgnosticBBoxCoder (demf/core/bbox/coders/class_agnostic_bbox_coder.py:130-251)
  RESTATED (oracle/deps.py, the un-vendored mmdet3d/mmcv/mmdet symbols those files import).
Inputs and weights come from integer seeds (oracle/fixtures.py), so only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import deps, fixtures, shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference_model(cfg):
    """(backbone = restated PointNet2SASSG, head = REAL reference DeMFVoteHead)."""
    from demf_amd.config import head_kwargs
    ref = shim.reference()
    b = cfg.backbone
    backbone = deps.PointNet2SASSG(in_channels=b.in_channels, num_points=b.num_points,
                                   radius=b.radius, num_samples=b.num_samples,
                                   sa_channels=b.sa_channels, fp_channels=b.fp_channels,
                                   use_xyz=b.use_xyz, normalize_xyz=b.normalize_xyz)
    kw = fixtures.to_attr(head_kwargs(cfg))
    kw["bbox_coder"].update(num_sizes=10, mean_sizes=[[1.0, 1.0, 1.0]] * 10)
    head = ref.head.DeMFVoteHead(**kw)
    model = torch.nn.Module()
    model.pts_backbone, model.pts_bbox_head = backbone, head
    return model


def run_reference(cfg, batch, seed, state=None, all_grads=False):
    """``state``: a state dict to load instead of the seeded weights (conditioned_goldens);
    ``all_grads``: store every parameter's gradient, not only the norms."""
    model = build_reference_model(cfg)
    if state is None:
        fixtures.seed_weights(model, seed)
    else:
        model.load_state_dict(state)
    model.train()
    points = torch.from_numpy(batch["points"])
    feats = [torch.from_numpy(f) for f in batch["img_features"]]
    x = model.pts_backbone(points)
    feat_dict = dict(seed_points=x["fp_xyz"][-1], seed_features=x["fp_features"][-1],
                     seed_indices=x["fp_indices"][-1])
    res = model.pts_bbox_head(feat_dict, "seed",
                              dict(img_features=feats, img_metas=batch["img_metas"]))
    out = {}
    for k in ("seed_points", "seed_indices", "vote_points", "vote_offset", "aggregated_points",
              "aggregated_indices"):
        out[k] = res[k].detach().numpy()
    out["vote_features_sum"] = res["vote_features"].detach().double().sum(-1).float().numpy()
    for i, d in enumerate(res["decode_res_all"]):
        for k, v in d.items():
            out[f"decode{i}.{k}"] = v.detach().numpy()
    # decoder inputs as the reference head prepares them
    ff, mf, rp, ss, lsi, vr = model.pts_bbox_head.prepare_decoder_inputs(
        res["aggregated_points"], feats, batch["img_metas"])
    out.update(reference_points=rp.detach().numpy(), valid_ratios=vr.numpy(),
               mask_flatten=np.packbits(mf.numpy(), axis=1), spatial_shapes=ss.numpy(),
               level_start_index=lsi.numpy(),
               feat_flatten_probe=ff[::37, :, ::5].detach().numpy())
    # GT = the seeded in-room boxes + boxes dropped onto a few proposals, so that positive
    # assignments (and therefore every box-loss term) are exercised; stored in the fixture
    rng = np.random.default_rng(seed + 100)
    agg = res["aggregated_points"].detach().numpy()
    gt_boxes, gt_labels = [], []
    for b in range(points.shape[0]):
        pick = rng.choice(agg.shape[1], size=3, replace=False)
        ctr = agg[b, pick] + rng.normal(0, 0.04, size=(3, 3))
        dims = rng.uniform(0.6, 1.4, size=(3, 3))
        yaw = rng.uniform(-np.pi, np.pi, size=(3, 1))
        extra = np.concatenate([ctr - [0, 0, 1] * dims * 0.5, dims, yaw], 1).astype(np.float32)
        gt_boxes.append(np.concatenate([batch["gt_boxes"][b], extra], 0))
        gt_labels.append(np.concatenate([batch["gt_labels"][b], rng.integers(0, 10, size=3)]))
        out[f"gt_boxes.{b}"], out[f"gt_labels.{b}"] = gt_boxes[-1], gt_labels[-1]
    gtb = [deps.DepthInstance3DBoxes(torch.from_numpy(b)) for b in gt_boxes]
    gtl = [torch.from_numpy(l) for l in gt_labels]
    # targets + losses (reference recomputes targets per decode layer; identical)
    preds = dict(res)
    tp = dict(aggregated_points=res["aggregated_points"])
    targets = model.pts_bbox_head.get_targets(points, list(gtb), list(gtl), None, None, tp)
    names = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets",
             "mask_targets", "objectness_targets", "objectness_weights", "box_loss_weights",
             "distance_targets", "dir_targets", "size_targets", "center_targets")
    for n, t in zip(names, targets):
        out["target." + n] = t.detach().numpy()
    losses = model.pts_bbox_head.loss(preds, points, list(gtb), list(gtl), None, None,
                                      batch["img_metas"])
    total = sum(losses.values())
    total.backward()
    for k, v in losses.items():
        out["loss." + k] = v.detach().numpy()
    gn = {n: p.grad.double().norm().item() for n, p in model.named_parameters() if p.grad is not None}
    out["grad_names"] = np.array(sorted(gn))
    out["grad_norms"] = np.array([gn[n] for n in sorted(gn)])
    small = "pts_bbox_head.decoder.0.layer.attentions.1.attention_weights.bias"
    out["grad." + small] = dict(model.named_parameters())[small].grad.numpy()
    if all_grads:
        for n, p in model.named_parameters():
            if p.grad is not None:
                out["grad." + n] = p.grad.numpy()
    return out


def glue_goldens():
    """Small pure-glue vectors from the real classes (coder, posembed, ref-point scaling)."""
    ref = shim.reference()
    g = torch.Generator().manual_seed(7)
    out = {}
    coder = ref.coder.DeMFClassAgnosticBBoxCoder(num_dir_bins=12, with_rot=True, num_sizes=10,
                                                 mean_sizes=[[1.0, 1.0, 1.0]] * 10)
    cls = torch.randn(2, 12, 9, generator=g)
    reg = torch.randn(2, 30, 9, generator=g)
    base = torch.randn(2, 9, 3, generator=g)
    sp = coder.split_pred(cls, reg, base)
    out.update({"coder.in.cls": cls.numpy(), "coder.in.reg": reg.numpy(), "coder.in.base": base.numpy()})
    for k, v in sp.items():
        out["coder.split." + k] = v.numpy()
    out["coder.decode"] = coder.decode(sp).numpy()
    out["coder.corners"] = coder.decode_corners(sp["center"], sp["size"].abs()).numpy()
    boxes = deps.DepthInstance3DBoxes(torch.tensor([[0.1, 2.0, -0.5, 1.0, 2.0, 0.8, 0.4],
                                                    [1.0, 3.0, -0.2, 0.6, 0.7, 1.1, -2.9],
                                                    [-1.0, 4.0, 0.0, 1.5, 0.5, 0.9, 3.1]]))
    enc = coder.encode(boxes, torch.tensor([1, 4, 7]), ret_dir_target=True)
    out["coder.encode.boxes"] = boxes.tensor.numpy()
    for n, v in zip(("center", "size", "dir_class", "dir_res", "dir"), enc):
        out["coder.encode." + n] = v.numpy()
    pe = ref.transformer.PositionEmbeddingLearned(dict(input_channel=6, num_pos_feats=16))
    fixtures.seed_weights(pe, 3)
    pe.train()
    q = torch.randn(3, 10, 6, generator=g)
    out["posembed.in"] = q.numpy()
    out["posembed.out"] = pe(q).detach().numpy()
    return out


def bbox_goldens(seeds=(0, 1)):
    """REAL reference DeMFVoteHead.get_bboxes (class_agnostic_vote_head.py:714-754, with the
    inherited VoteHead.multiclass_nms_single restated in oracle/deps.py) on the synthetic decode
    results of fixtures.make_decode_results(seed): per scene the selected boxes / scores / labels."""
    from demf_amd.config import head_kwargs
    ref = shim.reference()
    kw = fixtures.to_attr(head_kwargs(fixtures.tiny_cfg()))
    kw["bbox_coder"].update(num_sizes=10, mean_sizes=[[1.0, 1.0, 1.0]] * 10)
    head = ref.head.DeMFVoteHead(**kw)
    out = {}
    for seed in seeds:
        pts, dec = fixtures.make_decode_results(seed)
        preds = dict(decode_res_all=[{k: torch.from_numpy(v) for k, v in d.items()} for d in dec])
        metas = [dict(box_type_3d=deps.DepthInstance3DBoxes) for _ in range(pts.shape[0])]
        res = head.get_bboxes(torch.from_numpy(pts), preds, metas)
        for b, (bx, sc, lb) in enumerate(res):
            out[f"s{seed}.b{b}.boxes"] = bx.tensor.numpy()
            out[f"s{seed}.b{b}.scores"] = sc.numpy()
            out[f"s{seed}.b{b}.labels"] = lb.numpy()
        raw = head.get_bboxes(torch.from_numpy(pts), dict(decode_res_all=[
            {k: torch.from_numpy(v) for k, v in d.items()} for d in dec]), metas, use_nms=False)
        out[f"s{seed}.bbox3d"] = raw.numpy()
    return out


def encoder_goldens(seed=4):
    """REAL reference DeformableDetrEncoder (demf/modeling/layers/deform_detr_encoder.py) under the
    shim, fed by the restated ResNet/ChannelMapper on fixtures.make_images(seed): the four encoder
    output maps.  Weights: fixtures.seed_weights on identical state-dict keys."""
    from oracle.model import OracleImageStream
    ref = shim.reference()
    t = fixtures.TINY_IMAGE_STREAM
    stream = OracleImageStream(**t)
    real = ref.encoder.DeformableDetrEncoder(
        encoder=dict(type="DetrTransformerEncoder", num_layers=t["num_layers"], transformerlayers=dict(
            type="BaseTransformerLayer", attn_cfgs=dict(type="MultiScaleDeformableAttention",
                                                        embed_dims=t["embed_dims"], num_heads=t["num_heads"]),
            feedforward_channels=t["feedforward_channels"], ffn_dropout=0.1,
            operation_order=("self_attn", "norm", "ffn", "norm"))),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=t["num_feats"],
                                 normalize=True, offset=-0.5),
        num_feature_levels=4, embed_dims=t["embed_dims"])
    assert sorted(real.state_dict()) == sorted(stream.img_encoder.state_dict())
    stream.img_encoder = real
    fixtures.seed_weights(stream, seed)
    stream.eval()
    img, metas = fixtures.make_images(seed)
    with torch.no_grad():
        pyramid = stream.img_neck(stream.img_backbone(torch.from_numpy(img)))
        outs = real(pyramid, metas)
    out = {f"neck{i}": p.numpy() for i, p in enumerate(pyramid)}
    out.update({f"enc{i}": o.numpy() for i, o in enumerate(outs)})
    return out


def encoder256_goldens(seed=6):
    """REAL reference DeformableDetrEncoder at the reference's own encoder shape (configs/demf/demf_votenet.py:28-47:
    256 dims, 8 heads, 4 levels, 4 points, FFN 1024, post-norm) - two layers on a small padded pyramid
    (fixtures.make_encoder_pyramid).  This is the shape at which demf_amd's encoder runs on its hand-written
    kernels (rows_gemm + raw MSDA), so the golden pins THAT path to the real class."""
    ref = shim.reference()
    t = fixtures.ENC256
    real = ref.encoder.DeformableDetrEncoder(
        encoder=dict(type="DetrTransformerEncoder", num_layers=t["num_layers"], transformerlayers=dict(
            type="BaseTransformerLayer", attn_cfgs=dict(type="MultiScaleDeformableAttention",
                                                        embed_dims=t["embed_dims"], num_heads=t["num_heads"],
                                                        num_levels=4, num_points=t["num_points"]),
            feedforward_channels=t["feedforward_channels"], ffn_dropout=0.1,
            operation_order=("self_attn", "norm", "ffn", "norm"))),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=t["num_feats"],
                                 normalize=True, offset=-0.5),
        num_feature_levels=4, embed_dims=t["embed_dims"])
    fixtures.seed_weights(real, seed)
    real.eval()
    feats, metas = fixtures.make_encoder_pyramid(seed)
    with torch.no_grad():
        outs = real([torch.from_numpy(f) for f in feats], metas)
    out = {f"enc{i}": o.numpy() for i, o in enumerate(outs)}
    out["state_keys"] = np.array(sorted(real.state_dict()))
    return out


COND = dict(steps=200, lr=2e-3, decoder_lr_mult=0.05, weight_decay=0.01, clip=10.0, train_seeds=tuple(range(200, 208)),
            eval_seed=300, B=2, N=1024, n_gt=4)


def train_conditioned_weights(cfg, log=None):
    """Conditioned (trained-like) weights for the tiny config: the fp64 CPU oracle (oracle/model.py) under the
    reference's optimizer settings - AdamW, the decoder group at lr x 0.05, gradient clipping at max-norm 10
    (configs/demf/demf_votenet.py:16-24, configs/_base_/schedules/schedule_3x.py:6) - for COND['steps'] steps over
    eight seeded batches.  -> float32 state dict (parameters + BatchNorm running statistics)."""
    from oracle.model import OracleDeMF
    c = COND
    model = OracleDeMF(cfg)
    fixtures.seed_weights(model, 1)
    model.double().train()
    dec = [p for n, p in model.named_parameters() if ".decoder." in n]
    rest = [p for n, p in model.named_parameters() if ".decoder." not in n]
    opt = torch.optim.AdamW([dict(params=rest, lr=c["lr"]), dict(params=dec, lr=c["lr"] * c["decoder_lr_mult"])],
                            weight_decay=c["weight_decay"])
    batches = []
    for sd in c["train_seeds"]:
        b = fixtures.make_scene_batch(c["B"], c["N"], fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=sd, n_gt=c["n_gt"])
        batches.append((torch.from_numpy(b["points"]).double(), [torch.from_numpy(f).double() for f in b["img_features"]],
                        b["img_metas"], [torch.from_numpy(x).double() for x in b["gt_boxes"]],
                        [torch.from_numpy(x) for x in b["gt_labels"]]))
    for it in range(c["steps"]):
        pts, feats, metas, gtb, gtl = batches[it % len(batches)]
        losses, _, _ = model.forward_train(pts, feats, metas, gtb, gtl)
        total = sum(losses.values())
        opt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(list(model.parameters()), c["clip"])
        opt.step()
        if log is not None and (it % 20 == 0 or it + 1 == c["steps"]):
            log(it, float(total.detach()))
    return {k: (v.detach().float() if v.is_floating_point() else v.detach().clone())
            for k, v in model.state_dict().items()}


def conditioned_goldens():
    """Third head golden (tests/golden/ref_head_cond.npz): the REAL reference head (+ restated backbone) in
    train mode on CONDITIONED weights - train_conditioned_weights above - and a held-out seeded batch: every
    forward output, the targets, the losses and EVERY parameter gradient, plus the weights themselves (``w.<key>``;
    they are the result of an optimisation run, not of a seed).  On a trained-like network the discrete events of
    the untrained 30-BatchNorm one (near-tied max-pools / ReLUs) do not dominate: the GPU path is held to the
    north-star bars here - 1e-4 on every decode output, 1e-3 rel-L2 on every gradient."""
    cfg = fixtures.tiny_cfg()
    c = COND
    state = train_conditioned_weights(cfg, log=lambda it, v: print("  cond step", it, "loss %.4f" % v))
    batch = fixtures.make_scene_batch(c["B"], c["N"], fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=c["eval_seed"], n_gt=c["n_gt"])
    out = run_reference(cfg, batch, c["eval_seed"], state=state, all_grads=True)
    for k, v in state.items():
        out["w." + k] = v.numpy()
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    cfg = fixtures.tiny_cfg()
    for name, seed, B, n_gt in (("tiny_a", 1, 2, 4), ("tiny_b", 2, 3, 2)):
        batch = fixtures.make_scene_batch(B, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                          cfg.head.embed_dims, seed=seed, n_gt=n_gt)
        out = run_reference(cfg, batch, seed)
        np.savez_compressed(os.path.join(GOLD, f"ref_head_{name}.npz"), **out)
        print(name, {k: float(v) for k, v in out.items() if k.startswith("loss.")})
    np.savez_compressed(os.path.join(GOLD, "ref_glue.npz"), **glue_goldens())
    np.savez_compressed(os.path.join(GOLD, "ref_bboxes.npz"), **bbox_goldens())
    np.savez_compressed(os.path.join(GOLD, "ref_encoder.npz"), **encoder_goldens())
    np.savez_compressed(os.path.join(GOLD, "ref_encoder256.npz"), **encoder256_goldens())
    np.savez_compressed(os.path.join(GOLD, "ref_head_cond.npz"), **conditioned_goldens())
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
