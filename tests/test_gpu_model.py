"""GPU parity of the whole hot path (demf_amd.modules on libdemf_hip.so) against
(1) golden vectors from the REAL reference head (tests/golden/ref_head_*.npz) and
(2) the CPU oracle (oracle/model.py) at growing sizes up to the full reference config at B=8.
Forward: 1e-4 (north-star) up to the vote stage, fp32 noise floor beyond; indices bit-exact.
Gradients: EVERY tensor within a stated bound of the fp64 oracle on ONE input per case, chosen by
the oracle alone (tests/parity_tools.py: discrete-event margins) - no retry."""
import os

import numpy as np
import pytest
import torch

import parity_tools as P
from oracle import fixtures

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-4, atol=1e-4)

TARGET_NAMES = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets",
                "mask_targets", "objectness_targets", "objectness_weights", "box_loss_weights",
                "distance_targets", "dir_targets", "size_targets", "center_targets")


def _close_to_gold(a, g, name, tol=5e-4):
    """fp32 CPU golden vs fp32 GPU: both carry the network's fp32 noise floor (measured
    ~1.5e-4 abs at the decode outputs of the tiny config: fp32 vs fp64 oracle)."""
    err = np.abs(a.astype(np.float64) - g.astype(np.float64)).max()
    assert err <= tol * max(1.0, np.abs(g).max()), f"{name}: max err {err:.2e}"


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
            if k.startswith("_"):       # private handles for the fused loss kernel
                continue
            _close_to_gold(v.detach().cpu().numpy(), gold[f"decode{i}.{k}"], f"decode{i}.{k}")
    # the training targets the HIP kernels build (demf_vote_targets / demf_proposal_targets)
    # against the REAL reference's get_targets (class_agnostic_vote_head.py:756-941)
    head = model.pts_bbox_head
    with torch.no_grad():
        tg = head.get_targets(points, gb, gl, {k: v for k, v in preds.items() if k != "decode_res_all"})
    names = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets",
             "mask_targets", "objectness_targets", "objectness_weights", "box_loss_weights",
             "distance_targets", "dir_targets", "size_targets", "center_targets")
    for n, t in zip(names, tg):
        want = gold["target." + n]
        got = t.cpu().numpy()
        if want.dtype.kind in "iu":
            np.testing.assert_array_equal(got, want, err_msg="target." + n)
        else:
            # real-valued targets are functions of the aggregated points (1e-4 parity above)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4, err_msg="target." + n)
    losses = head.loss(preds, points, gb, gl, None, None, batch["img_metas"])
    total = losses.pop("_total")            # what a training loop differentiates
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), gold["loss." + k], rtol=5e-4, err_msg=k)
    np.testing.assert_allclose(total.item(), sum(v.item() for v in losses.values()), rtol=1e-5)
    total.backward()
    gn = _grad_norms(model)
    assert sorted(gn) == list(gold["grad_names"])
    got, want = np.array([gn[n] for n in sorted(gn)]), gold["grad_norms"]
    # BN-shadowed biases: exact 0 + noise.  Per-parameter norms of this tiny model (32..256 points
    # per level) move by up to ~1 % when a single near-tie of a max-pool or a ReLU resolves
    # differently than on the CPU that produced the goldens (the statistics are summed in a
    # different order); the bulk must agree to 5e-3, no parameter may be further than 1.5e-2.
    close = np.abs(got - want) <= 5e-3 * np.abs(want) + 1e-4
    assert close.mean() >= 0.95, [n for n, c in zip(sorted(gn), close) if not c]
    np.testing.assert_allclose(got, want, rtol=1.5e-2, atol=1e-4)
    small = "pts_bbox_head.decoder.0.layer.attentions.1.attention_weights.bias"
    _close_to_gold(dict(model.named_parameters())[small].grad.cpu().numpy(), gold["grad." + small],
                   "grad " + small, tol=2e-3)


@pytest.mark.parametrize("graph", [False, True])
def test_hot_path_on_conditioned_weights(golden_dir, graph):
    """The north-star bars WITHOUT a seed qualification: on conditioned (trained-like) weights - 200 AdamW steps
    of the fp64 oracle at the tiny config, committed with the REAL reference head's outputs on a held-out batch
    (tests/golden/ref_head_cond.npz, oracle/pin_reference.py: conditioned_goldens; the CPU fp32 oracle reproduces
    that golden to round-off, tests/test_oracle_model.py) - the HIP path is held to
        1e-4 (absolute, x max(1, tensor scale)) on every forward tensor up to and including EVERY decode output,
        bit-exact indices and integer targets, 1e-4 relative on every loss term,
        1e-3 rel-L2 on EVERY parameter gradient (BatchNorm-shadowed biases, whose true gradient is 0: an absolute
        floor of 1e-7 of the largest gradient norm).
    class_agnostic_vote_head.py:468-512 (decoder), :596-712 (losses).  ``graph``: the same through one captured
    hipGraph of forward + loss + backward (what bench.py times)."""
    from demf_amd.modules import DeMFHotPath
    from oracle.pin_reference import COND as c
    gold = np.load(os.path.join(golden_dir, "ref_head_cond.npz"))
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(c["B"], c["N"], fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=c["eval_seed"], n_gt=c["n_gt"])
    model = DeMFHotPath(cfg)
    model.load_state_dict({k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w.")})
    model.cuda().train()
    gtb = [gold[f"gt_boxes.{b}"] for b in range(c["B"])]
    gtl = [gold[f"gt_labels.{b}"] for b in range(c["B"])]
    points, feats, gb, gl = _to_dev(batch, gtb, gtl)
    head = model.pts_bbox_head
    params = [p for p in model.parameters() if p.requires_grad]

    def run():
        preds = model.forward_head(points, feats, batch["img_metas"])
        losses = head.loss(preds, points, gb, gl, None, None, batch["img_metas"])
        grads = torch.autograd.grad(losses["_total"], params, allow_unused=True)
        return preds, losses, grads
    if graph:
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        model.load_state_dict(sd0)                  # the warm-up advanced the BatchNorm running statistics only
        g = torch.cuda.CUDAGraph()
        from demf_amd import engine
        with engine._gc_paused(), torch.cuda.graph(g):
            preds, losses, grads = run()
        g.replay()
        torch.cuda.synchronize()
    else:
        preds, losses, grads = run()
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(preds[k].cpu().numpy(), gold[k])
    for k in ("seed_points", "vote_points", "vote_offset", "aggregated_points"):
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), gold[k], **TOL, err_msg=k)
    worst_out = 0.0
    for i, d in enumerate(preds["decode_res_all"]):
        for k, v in d.items():
            if k.startswith("_"):
                continue
            want = gold[f"decode{i}.{k}"]
            err = np.abs(v.detach().cpu().numpy().astype(np.float64) - want).max()
            worst_out = max(worst_out, err / max(1.0, np.abs(want).max()))
            assert err <= 1e-4 * max(1.0, np.abs(want).max()), f"decode{i}.{k}: {err:.2e}"
    with torch.no_grad():
        tg = head.get_targets(points, gb, gl, {k: v for k, v in preds.items() if k != "decode_res_all"})
    for n, t in zip(TARGET_NAMES, tg):
        want = gold["target." + n]
        if want.dtype.kind in "iub":
            np.testing.assert_array_equal(t.cpu().numpy(), want, err_msg="target." + n)
        else:
            np.testing.assert_allclose(t.cpu().numpy(), want, rtol=1e-4, atol=1e-4, err_msg="target." + n)
    for k, v in losses.items():
        if k != "_total":
            np.testing.assert_allclose(v.item(), gold["loss." + k], rtol=1e-4, err_msg=k)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    top = max(float(np.linalg.norm(gold["grad." + n])) for n in names)
    worst = (0.0, None)
    for n, gr in zip(names, grads):
        want = gold["grad." + n].astype(np.float64)
        assert gr is not None, n
        err = float(np.linalg.norm(gr.double().cpu().numpy() - want))
        nrm = float(np.linalg.norm(want))
        if nrm > 1e-6 * top:
            worst = max(worst, (err / nrm, n))
        assert err <= 1e-3 * nrm + 1e-7 * top, f"grad {n}: rel-L2 {err / max(nrm, 1e-30):.2e} (norm {nrm:.2e})"
    print(f"[parity] conditioned weights (graph={graph}): worst decode output {worst_out:.2e} of scale, "
          f"worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")


def _gpu_run(cfg, case):
    from demf_amd.modules import DeMFHotPath
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, case["seed"])
    model.cuda().train()
    points, f_d, gb, gl = _to_dev(case["batch"], case["gtb"], case["gtl"])
    preds = model.forward_head(points, f_d, case["batch"]["img_metas"])
    head = model.pts_bbox_head
    with torch.no_grad():
        tg = head.get_targets(points, gb, gl, {k: v for k, v in preds.items() if k != "decode_res_all"})
    losses = head.loss(preds, points, gb, gl, None, None, case["batch"]["img_metas"])
    losses.pop("_total").backward()
    grads = {n: p.grad.detach() for n, p in model.named_parameters() if p.grad is not None}
    return dict(preds=preds, losses=losses, grads=grads, targets=tg)


def _err(x, t):
    t = t.detach().double().cpu()
    return (x.detach().double().cpu() - t).abs().max().item(), max(t.abs().max().item(), 1.0)


def _check(name, e_gpu, e_cpu, scale, floor=2e-4, cap=2e-2):
    """The HIP path must sit at the fp32 noise floor: no further from the fp64 truth than a few
    times the CPU fp32 path is (or 2e-4 of the tensor scale), and never beyond `cap`.  (8 x: the SA stacks' kernels
    take their operands as two fp16 terms - 2^-22 per operand where the CPU path rounds at 2^-24; these untrained
    full-size networks amplify either by ~1e3.  With DEMF_F16_TERMS=0 - three bf16 terms - 4 x holds.)"""
    assert e_gpu <= max(8 * e_cpu, floor * scale), f"{name}: gpu {e_gpu:.2e} cpu32 {e_cpu:.2e} scale {scale:.2f}"
    assert e_gpu <= cap * scale, f"{name}: gpu err {e_gpu:.2e} vs scale {scale:.2f}"




# Hard ceilings on the per-tensor gradient error (rel-L2 vs the fp64 oracle, in units of the flip
# scale): 2x the worst value measured on MI355X for each case's qualified seed (printed by every run
# as "[parity] worst ...").  The allowance term 2 x risk of the qualification stays; the cap is what
# keeps a 10 % regression from hiding under it.
# Worst observed: mid 7e-3 (r02) / 8.8e-3 (r03); full B=2 3e-2 / 7.8e-3; P=4 4e-2 / 3.7e-3; B=8 4e-2 / 4.8e-2
# (which near-tie flips differs from run to run: atomics order), so each cap is 2x the larger one.
GRAD_CAP = dict(mid=2e-2, full2=6e-2, full2p4=8e-2, full8=1e-1)


def _parity(cfg, B, N, pyramid, in_shape, img_shape, seeds, cap=None, risk_max=P.RISK_MAX):
    """One input (the first seed of ``seeds`` that the oracle qualifies), every check, no retry."""
    case = P.qualified_case(cfg, B, N, pyramid, in_shape, img_shape, seeds=seeds, risk_max=risk_max)
    T, C = case["truth"], case["cpu32"]
    G = _gpu_run(cfg, case)
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(G["preds"][k].cpu().numpy(), T["preds"][k].numpy())
    # up to the vote stage the arithmetic is shallow: hold the north-star 1e-4 outright
    for k in ("seed_points", "vote_points", "aggregated_points", "vote_features"):
        e, s = _err(G["preds"][k], T["preds"][k])
        assert e <= 1e-4 * s, f"{k}: {e:.2e} (scale {s:.2f})"
    for i, d in enumerate(T["preds"]["decode_res_all"]):
        for k in d:
            eg, s = _err(G["preds"]["decode_res_all"][i][k], d[k])
            ec, _ = _err(C["preds"]["decode_res_all"][i][k], d[k])
            _check(f"decode{i}.{k}", eg, ec, s)
    # training targets: the discrete ones exactly (the seed keeps every assignment threshold at a
    # distance), the real-valued ones to 1e-4
    for n, t in zip(TARGET_NAMES, G["targets"]):
        want = T["targets"][n]
        if want.dtype in (torch.int64, torch.int32, torch.bool):
            assert torch.equal(t.cpu(), want), "target " + n
        else:
            e, s = _err(t, want)
            assert e <= 1e-4 * s, f"target {n}: {e:.2e}"
    for k in T["losses"]:
        eg, s = _err(G["losses"][k], T["losses"][k])
        ec, _ = _err(C["losses"][k], T["losses"][k])
        _check("loss." + k, eg, ec, s, floor=1e-4)
    # gradients: EVERY tensor, rel-L2 vs fp64 <= max(1e-3, 4 x the CPU fp32 oracle's own error)
    # + the seed's residual flip allowance: 2 x its worst per-layer flip risk (<= RISK_MAX; a flip
    # can happen on either fp32 path and in more than one layer), printed with the seed
    bad, rows = P.compare_grads(T, C, G["grads"], rtol=1e-3, mult=4.0, allowance=2.0 * case["risk"],
                                cap=cap)
    worst = sorted((r for r in rows if r[3] == r[3]), key=lambda r: -r[4])[:5]
    print("[parity] seed %d, allowance %.2e, cap %s; WORST gradient error in flip-scale units %.3e; worst "
          "tensors (name, |g|, gpu rel err, cpu32 rel err, err / flip scale): %s"
          % (case["seed"], 2.0 * case["risk"], cap, worst[0][4], worst))
    assert not bad, "gradient parity (seed %d, allowance %.1e):\n  " % (case["seed"], 2.0 * case["risk"]) \
        + "\n  ".join(bad)


MID = ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551)


@pytest.mark.no_library_fallback
def test_hot_path_vs_oracle_mid_size():
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
                  head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    _parity(cfg, 2, 6000, *MID, seeds=SEEDS["mid"], cap=GRAD_CAP["mid"])


@pytest.mark.no_library_fallback
def test_hot_path_vs_oracle_full_config():
    """configs/demf/demf_votenet.py sizes: 20 000 points, 800x1120 pyramid, 256 queries."""
    from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg, HeadCfg
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    # a flip the qualification PREDICTS (risk 8.7e-2 at conv_pred0.shared_convs.layer1 for seed 1) is as
    # large as the cap: the input must keep its worst flip below the cap it is held to
    _parity(cfg, 2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2], seeds=SEEDS["full2"],
            cap=GRAD_CAP["full2"], risk_max=5e-2)


@pytest.mark.no_library_fallback
def test_hot_path_vs_oracle_full_config_four_sampling_points():
    """The whole path with P=4 sampling points per level (BASELINE.json's wording of configs[2];
    the reference config has P=2) - bench.py --msda-points 4."""
    from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg, HeadCfg
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0, num_points=4))
    _parity(cfg, 2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2], seeds=SEEDS["full2p4"],
            cap=GRAD_CAP["full2p4"])


@pytest.mark.no_library_fallback
def test_hot_path_vs_oracle_full_config_batch_8():
    """BASELINE configs[2] as benchmarked: 8 scenes x 20 000 points x 18 609 image tokens."""
    from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg, HeadCfg
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    _parity(cfg, 8, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2], seeds=SEEDS["full8"],
            cap=GRAD_CAP["full8"])


# ---- BASELINE configs[3]: the bf16 compute mode against the bf16-EMULATING oracle ------------------------
# The reference has no such variant (fp16_enabled = False, class_agnostic_vote_head.py:384), so the burden
# of proof is here.  Truth = oracle/emulate.py: the same oracle model rounding the operands of every dense
# contraction to bf16 exactly where the kernels do (forward and backward, incl. the product's factored
# first SA layer and sample-then-project attention), in float64 ("emu64").  Its float32 twin ("emu32")
# measures what ANY fp32-accumulating implementation of those rounding points can be expected to deviate:
# round-to-bf16 is discontinuous, an operand within fp32 noise d of a rounding boundary flips by one bf16
# ulp e = 2^-8, so every rounding point turns d ~ 1e-7 into sqrt(d e) ~ 2e-5 of new noise, which the 30
# train-mode BN layers then amplify like any other.  The HIP path must be no further from
# emu64 than a small multiple of emu32 - tensor by tensor, forward and gradient - and emu64 must explain
# most of its distance to the plain fp64 oracle (a wrong rounding point or a wrong-but-finite bf16
# backward fails both).
BF16_MULT = 2.5          # per tensor: err(HIP, emu64) <= max(floor, BF16_MULT x err(emu32, emu64))
LOSS_FLOOR = 7e-2      # see _bf16_parity: the emulated twin's own host-to-host spread on one loss
LOSS_MULT = 4.0        # scalars: a ratio of ONE sample to a scale estimated from seven (a tensor's rel-L2 pools thousands
#                        of elements, a loss is a sum over ~10 positive proposals); observed HIP / twin ratios 0.3 - 2.6
BF16_MEDIAN_MULT = 1.5   # medians over all gradient tensors


def _bf16_parity(cfg, case, label):
    from demf_amd import ops
    E64, E32 = P.emulated_runs(cfg, case)
    T = case["truth"]
    ops.set_compute_dtype("bf16")
    try:
        G = _gpu_run(cfg, case)
    finally:
        ops.set_compute_dtype("f32")
    rel = lambda a, t: ((a.detach().double().cpu() - t.double()).norm() / t.double().norm()).item()
    # coordinate-only stages are mode-independent: FPS / ball query / 3-NN see coordinates only, and with
    # sample_mod = 'seed' so does the proposal FPS
    for k in ("seed_indices", "aggregated_indices"):
        np.testing.assert_array_equal(G["preds"][k].cpu().numpy(), E64["preds"][k].numpy())
    assert rel(G["preds"]["seed_points"], E64["preds"]["seed_points"]) == 0.0
    report, bad = [], []

    def check(name, g, e32, e64, t, floor):
        eg, ec, et = rel(g, e64), rel(e32, e64), rel(g, t)
        report.append((name, eg, ec, et))
        if eg > max(floor, BF16_MULT * ec):
            bad.append("%s: HIP-vs-emu64 %.2e, emu32-vs-emu64 %.2e (x%.1f)" % (name, eg, ec, eg / max(ec, 1e-30)))
    for k in ("vote_points", "vote_features", "aggregated_points"):
        check(k, G["preds"][k], E32["preds"][k], E64["preds"][k], T["preds"][k], 2e-3)
    for i, d in enumerate(E64["preds"]["decode_res_all"]):
        for k in d:
            check("decode%d.%s" % (i, k), G["preds"]["decode_res_all"][i][k], E32["preds"]["decode_res_all"][i][k],
                  d[k], T["preds"]["decode_res_all"][i][k], 2e-3)
    fwd = list(report)
    # losses: scalars, sums over a few positive proposals - ONE sample of the twin's noise per loss is not a
    # scale (measured on the mid input: the twin's dir_class_loss lands 0.8 % from emu64 on the 64-thread GPU
    # host and 6.7 % on the 8-thread build host - fp32 summation order alone; dir_res_loss 1.1 % / 6.1 %).
    # The noise scale is pooled over the losses, with the larger of those two observations as floor.
    rel32 = max(abs(E32["losses"][k].item() - E64["losses"][k].item()) / max(abs(E64["losses"][k].item()), 1e-12)
                for k in E64["losses"])
    loss_tol = max(LOSS_FLOOR, LOSS_MULT * rel32)
    for k in E64["losses"]:
        lg, l64, l32 = G["losses"][k].item(), E64["losses"][k].item(), E32["losses"][k].item()
        if abs(lg - l64) > loss_tol * abs(l64):
            bad.append("loss %s: HIP %.6f emu64 %.6f emu32 %.6f (tolerance %.1e)" % (k, lg, l64, l32, loss_tol))
    gmax = max(v.norm().item() for v in E64["grads"].values())
    assert sorted(E64["grads"]) == sorted(G["grads"])
    grows = []
    for n, g in E64["grads"].items():
        if g.norm().item() < 1e-6 * gmax:               # conv biases shadowed by BatchNorm: exact zeros + noise
            assert G["grads"][n].double().norm().item() <= 1e-3 * gmax, n
            continue
        eg, ec = rel(G["grads"][n], g), rel(E32["grads"][n], g)
        grows.append((n, eg, ec, rel(G["grads"][n], T["grads"][n])))
        if eg > max(1e-2, BF16_MULT * ec):
            bad.append("grad %s: HIP-vs-emu64 %.2e, emu32-vs-emu64 %.2e (x%.1f)" % (n, eg, ec, eg / max(ec, 1e-30)))
    med = [float(np.median([r[i] for r in grows])) for i in (1, 2, 3)]

    def cos(A, B):
        a = torch.cat([A[n].double().cpu().reshape(-1) for n in B])
        b = torch.cat([B[n].double().reshape(-1) for n in B])
        return (a @ b / (a.norm() * b.norm())).item()
    c_ge, c_ce, c_gt = cos(G["grads"], E64["grads"]), cos(E32["grads"], E64["grads"]), cos(G["grads"], T["grads"])
    worst = sorted(grows, key=lambda r: -r[1] / max(r[2], 1e-30))[:5]
    print("[bf16 %s] seed %d: forward (name, HIP-vs-emu64, emu32-vs-emu64, HIP-vs-fp64) %s" %
          (label, case["seed"], [(n, "%.1e" % a, "%.1e" % b_, "%.1e" % c) for n, a, b_, c in fwd[:3]]))
    print("[bf16 %s] gradients: median rel-L2 HIP-vs-emu64 %.2e, emu32-vs-emu64 %.2e, HIP-vs-fp64 %.2e; "
          "whole-gradient cosine HIP.emu64 %.3f, emu32.emu64 %.3f, HIP.fp64 %.3f; worst ratios %s" %
          (label, med[0], med[1], med[2], c_ge, c_ce, c_gt,
           [(n, "%.2e" % a, "%.2e" % b_) for n, a, b_, _ in worst]))
    assert all(torch.isfinite(g).all() for g in G["grads"].values())
    assert not bad, "bf16 parity vs the emulating oracle (%s):\n  " % label + "\n  ".join(bad)
    assert med[0] <= max(1e-2, BF16_MEDIAN_MULT * med[1]), med
    # the direction of the whole gradient survives at least as well as in the fp32-accumulating twin
    assert c_ge >= c_ce - 0.08, (c_ge, c_ce)
    # and the emulation explains the bulk of the mode's deviation from the fp64 oracle
    vp = [r for r in fwd if r[0] == "vote_points"][0]
    assert vp[1] <= 0.5 * vp[3], vp
    # (for the gradients too - unless the fp32-accumulating twin itself is decorrelated from emu64, as at
    # the full size: median rel-L2 ~ sqrt(2) for ANY pair of implementations, nothing left to explain)
    assert med[0] <= 0.75 * med[2] or med[2] <= 1e-2 or med[1] >= 1.0, med


def test_hot_path_bf16_vs_emulating_oracle_mid_size():
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
                  head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    case = P.qualified_case(cfg, 2, 6000, *MID, seeds=SEEDS["mid"])
    _bf16_parity(cfg, case, "mid")


def test_hot_path_bf16_vs_emulating_oracle_full_config_batch_8():
    """BASELINE configs[3] per GPU at full size: 8 scenes x 20 000 points x 18 609 image tokens, on the
    SAME qualified input as the fp32 B=8 test above (the plain oracle runs are shared through
    parity_tools' session cache)."""
    from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg, HeadCfg
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    case = P.qualified_case(cfg, 8, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2],
                            seeds=SEEDS["full8"])
    _bf16_parity(cfg, case, "full B=8")


# Seeds to try, in order.  The first entries were found by running the (oracle-only) qualification
# of tests/parity_tools.py in the build container - tools/qualify_seeds.py - so that the GPU tier
# does not spend minutes of CPU time rejecting seeds; qualification is re-evaluated at test time
# and the search simply continues if the host's BLAS rounds differently.
SEEDS = dict(mid=tuple(range(1, 12)), full2=(4, 5, 6, 8, 11, 1, 2, 3, 7, 9, 10), full2p4=tuple(range(1, 12)),
             full8=tuple(range(1, 12)))
