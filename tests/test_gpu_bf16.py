"""The bf16 compute variant (BASELINE.json configs[3]; SURVEY section 8(b) B2 `T in {float, bf16}`):
dense MFMA kernels on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 storage / statistics /
indices / losses.  What is asserted: (1) the bf16 GEMM is exact on bf16-representable operands (pins
the MFMA fragment layout), (2) the fused shared-MLP stack equals a torch reference that rounds the
GEMM operands to bf16 at the same places, to 2e-3 on the output and every gradient, (3) deviation
from the fp32 path: <= 1e-2 on a 3-layer stack's output; on the whole random-weight hot path
coordinate-only indices stay bit-exact, vote points within 10 % (relative L2), total loss within
30 %, everything finite, and a few bf16 training steps reduce the loss."""
import numpy as np
import pytest
import torch

from oracle import fixtures

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    from demf_amd import ops
    ops.set_compute_dtype("bf16")
    yield
    ops.set_compute_dtype("f32")


def _r(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()


def test_gemm_bf16_is_exact_on_bf16_representable_operands(bf16_mode):
    """Products of bf16 numbers are exact in fp32, so with operands that ARE bf16 numbers the bf16
    MFMA path must agree with an fp64 product up to fp32 accumulation order - this pins the
    fragment layout of v_mfma_f32_32x32x16_bf16 in every staging mode."""
    from demf_amd import fused
    from demf_amd.fused import _p
    q = lambda t: t.bfloat16().float()
    for (M, N, K) in [(200, 96, 64), (2048, 256, 256), (70, 50, 6), (33, 7, 20)]:
        x, w, b = q(_r(M, K, seed=1)), q(_r(N, K, seed=2)), _r(N, seed=3)
        y = torch.empty(M, N, device="cuda")
        fused.gemm(M, N, K, _p(x), (K, 1), _p(w), (K, 1), _p(y), N, bias=_p(b))
        want = x.double() @ w.double().t() + b.double()
        assert (y.double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), (M, N, K)
    # K-strided B (dX form), row-contiguous operands (dW form, split-K)
    M, N, K = 300, 128, 192
    g, w = q(_r(M, K, seed=4)), q(_r(K, N, seed=5))
    y = torch.empty(M, N, device="cuda")
    fused.gemm(M, N, K, _p(g), (K, 1), _p(w), (1, N), _p(y), N)
    assert (y.double() - g.double() @ w.double()).abs().max().item() <= 2e-5 * 50
    R, N, K = 2048, 192, 256
    g, x = q(_r(R, N, seed=8)), q(_r(R, K, seed=9))
    dw = torch.zeros(N, K, device="cuda")
    fused.gemm(N, K, R, _p(g), (1, N), _p(x), (1, K), _p(dw), K, splitk=8)
    want = g.double().t() @ x.double()
    assert (dw.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    # and it really is bf16 arithmetic: an operand that is NOT representable gets rounded
    x = torch.full((64, 64), 1.0 + 2.0 ** -10, device="cuda")
    w = torch.eye(64, device="cuda")
    y = torch.empty(64, 64, device="cuda")
    fused.gemm(64, 64, 64, _p(x), (64, 1), _p(w), (64, 1), _p(y), 64)
    assert torch.all(y == 1.0)
    fused.gemm(64, 64, 64, _p(x), (64, 1), _p(w), (64, 1), _p(y), 64, flags=64)     # DEMF_GEMM_FP32
    assert torch.all(y == 1.0 + 2.0 ** -10)


def _mlp_case(seed, R, ns, chans):
    from demf_amd.modules.layers import RowsMLP
    torch.manual_seed(seed)
    mlp = RowsMLP(chans, dim=2).cuda().train()
    x = _r(R, chans[0], seed=seed).requires_grad_()
    return mlp, x


class _BfMatmul(torch.autograd.Function):
    """y = q(x) q(w)^T with q = round-to-bf16 and fp32 accumulation - what the bf16 MFMA kernels
    compute; backward: dx = q(g) q(w), dw = q(g)^T q(x) (input- and weight-gradient GEMMs run on
    bf16 MFMA as well, fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        q = lambda t: t.bfloat16().double()
        return (q(x) @ q(w).t()).float()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        q = lambda t: t.bfloat16().double()
        return (q(g) @ q(w)).float(), (q(g).t() @ q(x)).float()


def _emulated_stack(x, weights, gammas, betas, ns, eps=1e-5):
    """(conv1x1 -> train-mode BN -> ReLU) x L -> max over ns rows, with the GEMMs rounded exactly
    where the kernels round (operands of forward and input-gradient GEMMs)."""
    h = x
    for w, g, b in zip(weights, gammas, betas):
        y = _BfMatmul.apply(h, w)
        mean, var = y.mean(0), y.var(0, unbiased=False)
        h = torch.relu((y - mean) * torch.rsqrt(var + eps) * g + b)
    return h.view(-1, ns, h.shape[1]).max(1)[0]


def test_shared_mlp_bf16_matches_a_bf16_emulating_reference(bf16_mode):
    """The fused stack in bf16 mode against torch ops that round the GEMM operands to bf16 at the
    same places: outputs and EVERY gradient agree to fp32 accumulation noise (a layout or rounding
    -point bug would show as >= 1e-3).  Sizes keep max-pool / ReLU ties improbable."""
    from demf_amd.modules.layers import RowsMLP
    torch.manual_seed(5)
    chans, ns, R = [64, 64, 128, 256], 16, 1024 * 16
    mlp = RowsMLP(chans, dim=2).cuda().train()
    x = _r(R, chans[0], seed=5).requires_grad_()
    gout = _r(R // ns, chans[-1], seed=6)
    y = mlp.forward_rows(x, ns=ns)
    (y * gout).sum().backward()
    got = [y.detach(), x.grad] + [p.grad for p in mlp.parameters()]
    xr = x.detach().clone().requires_grad_()
    ws = [blk.conv.weight.detach().view(blk.cout, blk.cin).clone().requires_grad_() for blk in mlp]
    gs = [blk.bn.weight.detach().clone().requires_grad_() for blk in mlp]
    bs = [blk.bn.bias.detach().clone().requires_grad_() for blk in mlp]
    yr = _emulated_stack(xr, ws, gs, bs, ns)
    (yr * gout).sum().backward()
    want = [yr.detach(), xr.grad]
    for w, g, b in zip(ws, gs, bs):
        want += [w.grad.view(w.shape[0], w.shape[1], 1, 1), g.grad, b.grad]
    names = ["y", "dx"] + [n for n, _ in mlp.named_parameters()]
    bad = []
    for n, a, b in zip(names, got, want):
        rel = ((a.double() - b.double().view_as(a)).norm() / b.double().norm()).item()
        if rel > 2e-3:
            bad.append((n, rel))
    assert not bad, bad


def test_shared_mlp_bf16_deviation_from_fp32_is_bounded():
    """The same stack, bf16 mode vs fp32 mode: forward within 1e-2 (8-bit mantissa, K <= 128
    contractions, three BN layers), gradients within 0.25 - bf16 noise flips max-pool arg-maxes and
    ReLU masks, and the BN backward amplifies (measured ~0.1; tools/debug_bf16.py)."""
    from demf_amd import ops
    outs = {}
    for mode in ("f32", "bf16"):
        ops.set_compute_dtype(mode)
        try:
            mlp, x = _mlp_case(3, 4096 * 16, 16, [64, 64, 128, 256])
            y = mlp.forward_rows(x, ns=16)
            (y * _r(*y.shape, seed=9)).sum().backward()
            outs[mode] = [y.detach(), x.grad] + [p.grad for p in mlp.parameters()]
        finally:
            ops.set_compute_dtype("f32")
    rels = [((a - b).norm() / b.norm()).item() for a, b in zip(outs["bf16"], outs["f32"])]
    assert 0 < rels[0] < 1e-2, rels[0]
    assert max(rels[1:]) < 0.25, rels


@pytest.mark.parametrize("size", ["tiny", "mid"])
def test_hot_path_bf16_vs_fp32_oracle(size, bf16_mode):
    import parity_tools as P
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    from demf_amd.modules import DeMFHotPath
    if size == "tiny":
        cfg, B, N = fixtures.tiny_cfg(), 2, 1024
        pyr, ins, img = fixtures.TINY_PYRAMID, fixtures.TINY_INPUT, None
    else:
        cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
                      head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
        B, N, pyr, ins, img = 2, 6000, ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551)
    seed = 4
    batch, gtb, gtl = P.make_case(cfg, B, N, pyr, ins, img, seed) or P.make_case(cfg, B, N, pyr, ins, img, seed + 1)
    truth = P.oracle_run(cfg, batch, gtb, gtl, seed, torch.float32, tap=False)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    pts = torch.from_numpy(batch["points"]).cuda()
    feats = [torch.from_numpy(f).cuda() for f in batch["img_features"]]
    preds = model.forward_head(pts, feats, batch["img_metas"])
    # coordinate-only indices: exact
    np.testing.assert_array_equal(preds["seed_indices"].cpu().numpy(), truth["preds"]["seed_indices"].numpy())
    np.testing.assert_array_equal(preds["aggregated_indices"].cpu().numpy(),
                                  truth["preds"]["aggregated_indices"].numpy())
    # Deviation from the fp32 oracle, bounded loosely: with seeded RANDOM weights the 30 train-mode
    # BN layers amplify rounding noise ~400x from the input to the heads (fp32: 1e-7 -> 3e-5,
    # tests/parity_tools.py), so bf16's 4e-3 reaches O(0.1..1) at the decode outputs - a property
    # of the untrained network, not of the kernels (they are pinned by the two tests above).
    for k in ("vote_points", "aggregated_points"):
        got, want = preds[k].detach().cpu(), truth["preds"][k]
        assert ((got - want).norm() / want.norm()).item() < 0.1, k
    for i, d in enumerate(truth["preds"]["decode_res_all"]):
        for k in d:
            assert torch.isfinite(preds["decode_res_all"][i][k]).all(), (i, k)
    losses = model.pts_bbox_head.loss(preds, pts, [torch.from_numpy(b).cuda() for b in gtb],
                                      [torch.from_numpy(l).cuda() for l in gtl], None, None, batch["img_metas"])
    total = losses.pop("_total")
    want_total = sum(v.item() for v in truth["losses"].values())
    assert abs(total.item() - want_total) < 0.3 * want_total, (total.item(), want_total)
    total.backward()
    gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in model.parameters() if p.grad is not None)).item()
    rn = np.sqrt(sum(v.pow(2).sum().item() for v in truth["grads"].values()))
    # (measured over three seeds of this configuration: total loss -1 % .. +14 %, gradient norm
    # -18 % .. +58 % - which side depends on the last bit of the kernels; fp32 modes: 1e-4)
    assert 0.5 * rn < gn < 2.0 * rn, (gn, rn)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_bf16_steps_reduce_the_loss(bf16_mode):
    from demf_amd import engine, synthetic
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, 5)
    model.cuda().train()
    raw = synthetic.make_scene_batch(3, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                     cfg.head.embed_dims, seed=5, n_gt=4)
    batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
                 img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                 img_metas=raw["img_metas"],
                 gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                 gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
    tr = engine.Trainer(model, lr=1e-4)
    losses = [tr.step(batch).item() for _ in range(12)]
    assert all(np.isfinite(losses)) and min(losses[6:]) < losses[0], losses


def test_bf16_training_trajectory_tracks_fp32():
    """Thirty optimizer steps on the same four batches from the same initial weights in the bf16 compute
    mode and - twice - in fp32 (mid-size configuration, dropout off so that all runs see the same
    function).  Training this randomly initialised network is chaotic at the level of single steps even
    between two fp32 runs (see the note under the test), so the yard-stick for the bf16 trajectory is the
    distance between the two fp32 trajectories: its per-step deviation from the nearer fp32 run must be
    within 3x the fp32 runs' own (medians; vote loss and total loss), the first step - same weights -
    must agree to 3 % / 5 %, and the loss level after 30 steps must lie in the fp32 runs' band.  A bf16
    backward that is finite but wrong (a transposed operand, a missing term) fails the level and the
    medians within a few steps."""
    from demf_amd import engine, ops, synthetic
    from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
    from demf_amd.modules import DeMFHotPath
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
                  head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    pyr, ins, img = ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551)

    def fresh_model():
        model = DeMFHotPath(cfg)
        fixtures.seed_weights(model, 21)
        return model.cuda().train()
    probe = fresh_model()
    batches = []
    for i in range(4):
        raw = synthetic.make_scene_batch(4, 6000, pyr, ins, cfg.head.embed_dims, seed=70 + i, n_gt=4, img_shape=img)
        pts = torch.from_numpy(raw["points"]).cuda()
        feats = [torch.from_numpy(f).cuda() for f in raw["img_features"]]
        with torch.no_grad():                      # GT boxes on top of initial proposals: positives exist
            agg = probe.forward_head(pts, feats, raw["img_metas"])["aggregated_points"].cpu().numpy()
        rng = np.random.default_rng(700 + i)
        gtb, gtl = [], []
        for b in range(4):
            pick = rng.choice(agg.shape[1], 4, replace=False)
            dims = rng.uniform(0.8, 1.6, size=(4, 3))
            extra = np.concatenate([agg[b, pick] - [0, 0, 1] * dims * 0.5, dims, rng.uniform(-3, 3, (4, 1))], 1)
            gtb.append(np.concatenate([raw["gt_boxes"][b], extra.astype(np.float32)], 0))
            gtl.append(np.concatenate([raw["gt_labels"][b], rng.integers(0, 10, 4)]))
        batches.append(dict(points=pts, img_features=feats, img_metas=raw["img_metas"],
                            gt_bboxes_3d=[torch.from_numpy(x).cuda() for x in gtb],
                            gt_labels_3d=[torch.from_numpy(x).cuda() for x in gtl]))
    def run(mode):
        ops.set_compute_dtype(mode)
        try:
            model = fresh_model()
            tr = engine.Trainer(model, lr=1e-4)
            rows = []
            for k in range(30):
                b = batches[k % 4]
                tr._arena(True)
                try:
                    losses = model.forward_train(b["points"], b["img_features"], b["img_metas"],
                                                 b["gt_bboxes_3d"], b["gt_labels_3d"])
                    tr.flat.backward_into(losses["_total"])
                finally:
                    tr._arena(False)
                tr._update()
                rows.append((losses["vote_loss"].item(), losses["_total"].item()))
            return np.asarray(rows)
        finally:
            ops.set_compute_dtype("f32")
    a, a2, b = run("f32"), run("f32"), run("bf16")
    dev = lambda x, y: np.abs(x - y) / np.abs(x)
    self_med = np.median(dev(a, a2), axis=0)          # fp32 against itself: (vote, total)
    bf_med = np.median(np.minimum(dev(a, b), dev(a2, b)), axis=0)
    print("[bf16 trajectory] total loss fp32 %.2f -> %.2f / %.2f -> %.2f, bf16 %.2f -> %.2f; per-step deviation "
          "(median; max) fp32-vs-fp32 vote %.3f; %.2f total %.3f; %.2f | bf16-vs-fp32 vote %.3f; %.2f total %.3f; %.2f"
          % (a[:4, 1].mean(), a[-8:, 1].mean(), a2[:4, 1].mean(), a2[-8:, 1].mean(), b[:4, 1].mean(), b[-8:, 1].mean(),
             self_med[0], dev(a, a2)[:, 0].max(), self_med[1], dev(a, a2)[:, 1].max(),
             bf_med[0], dev(a, b)[:, 0].max(), bf_med[1], dev(a, b)[:, 1].max()))
    assert np.isfinite(b).all()
    assert dev(a, b)[0, 1] <= 0.05 and dev(a, b)[0, 0] <= 0.03          # the first step: same weights
    assert bf_med[0] <= max(TRAJ_TOL_VOTE, 3.0 * self_med[0]), (bf_med, self_med)
    assert bf_med[1] <= max(TRAJ_TOL_TOTAL, 3.0 * self_med[1]), (bf_med, self_med)
    lo, hi = min(a[-8:, 1].mean(), a2[-8:, 1].mean()), max(a[-8:, 1].mean(), a2[-8:, 1].mean())
    assert 0.6 * lo <= b[-8:, 1].mean() <= 1.4 * hi, (b[-8:, 1].mean(), lo, hi)


# Measured on MI355X (tools runs of this test's loop, round 4): two fp32 runs of the SAME 30 steps differ
# per step by 1.2-2.9 % (vote loss) and 6-9 % (total loss) in the median, and by up to 4.7x on single
# steps (the number of positive proposals is a discrete function of the predicted votes; the order of
# the fp32 atomics decides on which side of the 0.3 m threshold a proposal lands).  bf16 against fp32:
# 1.6-2.8 % / 10-14 % - the band a second fp32 run would need as well, which is what is asserted.
# (Round 5, six runs of this test on one box: fp32-vs-fp32 vote 0.8-3.5 %, bf16-vs-fp32 vote 2.4 / 3.0 / 3.1 / 4.6 /
# 5.0 / 5.5 % - the 5 % floor failed one run in six while the fp32 runs happened to agree to 0.9 %; floor now 8 %.)
TRAJ_TOL_VOTE = 0.08
TRAJ_TOL_TOTAL = 0.2
