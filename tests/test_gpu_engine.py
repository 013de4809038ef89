"""The training step on the GPU: hipGraph replay == eager, and a few steps reduce the loss."""
import pytest
import torch

from oracle import fixtures

pytestmark = pytest.mark.gpu


def _setup(seed=3, lr=1e-3):
    from demf_amd import engine, synthetic
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    raw = synthetic.make_scene_batch(3, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                     cfg.head.embed_dims, seed=seed, n_gt=4)
    batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
                 img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                 img_metas=raw["img_metas"],
                 gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                 gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
    return engine.Trainer(model, lr=lr), model, batch


def test_graph_replay_matches_eager():
    """Same weights, same batch: gradients and loss out of a hipGraph replay of
    fwd+loss+bwd equal the eager ones (atomics order aside)."""
    ta, ma, batch = _setup()
    tb, mb, _ = _setup()
    la = ta._fwd_bwd(batch)
    for _ in range(2):                      # eager warm-up of the to-be-captured path
        tb._fwd_bwd(batch)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    from demf_amd import engine
    with engine._gc_paused(), torch.cuda.graph(g):
        lb = tb._fwd_bwd(batch)
    tb.flat.flat.fill_(123.0)               # the replay must rewrite every gradient
    g.replay()
    torch.cuda.synchronize()
    assert abs(la.item() - lb.item()) <= 1e-4 * abs(la.item())
    ga, gb = ta.flat.flat.double(), tb.flat.flat.double()
    assert ((ga - gb).norm() / ga.norm()).item() < 1e-3
    # and the packaged capture()/replay() path runs and keeps training
    replay = tb.capture(batch, warmup=1)
    l1 = [replay().item() for _ in range(3)]
    assert all(torch.isfinite(torch.tensor(l1)))


def test_captured_gradient_pack_equals_eager():
    """Trainer.capture packs the gradients with ONE demf_multi_copy launch over an address table
    (uploaded after the capture): with lr = 0 the weights stay put, so the flat gradient buffer
    after a replay must equal the eager one - every entry rewritten."""
    ta, _, batch = _setup(lr=0.0)
    tb, _, _ = _setup(lr=0.0)
    ta._fwd_bwd(batch)
    replay = tb.capture(batch, warmup=1)
    for _ in range(2):
        tb.flat.flat.fill_(123.0)
        replay()
        torch.cuda.synchronize()
        ga, gb = ta.flat.flat.double(), tb.flat.flat.double()
        assert not bool((tb.flat.flat == 123.0).any())
        assert ((ga - gb).norm() / ga.norm()).item() < 1e-3


def test_steps_reduce_the_loss():
    """Small steps on one fixed batch (small enough that the discrete target assignment
    stays put) must walk the loss down."""
    from demf_amd import engine
    tr, model, batch = _setup(seed=5)
    tr = engine.Trainer(model, lr=1e-4)
    losses = [tr.step(batch).item() for _ in range(12)]
    assert losses[-1] < losses[0] and min(losses[6:]) < losses[0] * 0.999, losses


@pytest.mark.parametrize("prefetch", [False, True])
def test_full_size_replays_stay_finite(prefetch):
    """BASELINE-size step (8 scenes x 20000 points, 18609 image tokens) through
    Trainer.capture(): every replay - with and without the pipelined coordinate pre-pass -
    leaves finite gradients, parameters and loss.  Regression test: at::sum's semaphore
    reduction returned inf/NaN bias gradients inside hipGraph replays; the path
    now reduces them with demf_colsum_f32."""
    import bench
    from demf_amd import engine
    from demf_amd.config import DeMFCfg
    from demf_amd.modules import DeMFHotPath
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DeMFHotPath(DeMFCfg()).to(dev).train()
    tr = engine.Trainer(model)
    batch, _ = bench.make_batch(8, seed=1000, device=dev)
    other, _ = bench.make_batch(8, seed=2000, device=dev)
    step = tr.capture(batch, prefetch_geometry=prefetch)
    for it in range(12):
        # alternate two point clouds so that the refreshed geometry is really consumed
        nxt = (other if it % 2 == 0 else batch)["points"] if prefetch else None
        loss = step(nxt) if prefetch else step()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss)), f"replay {it}: loss {float(loss)}"
        assert bool(torch.isfinite(tr.flat.flat).all()), f"replay {it}: non-finite gradient"
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())


def test_flat_adamw_matches_torch_adamw():
    """demf_adamw_f32 (clip folded in) == clip_grad_norm_ + torch.optim.AdamW on the CPU, the
    optimizer the reference's runner uses (schedule_3x.py:6-7), over several steps and both
    parameter groups.  Tolerance 2e-6 absolute on parameters of magnitude <= 1."""
    import copy
    from demf_amd import engine
    torch.manual_seed(0)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(37, 53)
            self.decoder = torch.nn.Linear(53, 11)

        def param_groups(self, lr, weight_decay):
            return [dict(params=list(self.a.parameters()), lr=lr, weight_decay=weight_decay),
                    dict(params=list(self.decoder.parameters()), lr=lr * 0.05, weight_decay=weight_decay)]

    ref = Toy()
    dev = copy.deepcopy(ref).cuda()
    tr = engine.Trainer(dev, lr=0.008, weight_decay=0.01, max_grad_norm=10.0)
    assert tr.fused
    opt = torch.optim.AdamW(ref.param_groups(0.008, 0.01), lr=0.008, weight_decay=0.01)
    for it in range(5):
        grads = [torch.randn_like(p) * (30.0 if it % 2 == 0 else 0.01) for p in ref.parameters()]
        for p, g in zip(ref.parameters(), grads):
            p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(list(ref.parameters()), 10.0)
        opt.step()
        for v, g in zip(tr.flat.views, grads):
            v.copy_(g.cuda())
        tr._update()
    for (n, p), q in zip(ref.named_parameters(), dev.parameters()):
        assert torch.allclose(p, q.cpu(), atol=2e-6, rtol=0), n


def _tiny_batch(seed, n_gt):
    from demf_amd import synthetic
    cfg = fixtures.tiny_cfg()
    raw = synthetic.make_scene_batch(3, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                     cfg.head.embed_dims, seed=seed, n_gt=n_gt)
    return dict(points=torch.from_numpy(raw["points"]).cuda(),
                img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                img_metas=raw["img_metas"],
                gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])


@pytest.mark.parametrize("ahead", [2, 1, 0])
def test_replay_with_changing_batches_matches_eager(ahead):
    """The captured step fed through its static input buffers (load()) with the pipelined
    pre-pass: three different batches (points, image features, metas, GT counts) cycled through
    the graph give the same losses and parameters as eager steps on them - with next_points = the
    next batch (the default one-deep contract), = the batch after the next (the DEMF_GEO_TWO_DEEP
    contract; under the default placement load() then recomputes) and with no prefetch at all
    (load() recomputes the geometry): the tags keep every usage correct, only the overlap differs."""
    batches = [_tiny_batch(11, 4), _tiny_batch(12, 2), _tiny_batch(13, 5)]
    # a small learning rate keeps the two runs from drifting apart through the (legitimately
    # nondeterministic) fp32 atomics: the losses then mainly reflect WHICH batch was consumed
    te, me, _ = _setup(lr=2e-5)
    tg, mg, _ = _setup(lr=2e-5)
    order = [0, 1, 2, 0, 2, 1]
    for _ in range(2):                       # capture() warms up with real steps on its batch
        te.step(batches[order[0]])
    eager = [float(te.step(batches[i])) for i in order]
    replay = tg.capture(batches[order[0]], warmup=2, max_gt=8)
    got = []
    for k, i in enumerate(order):
        if k:
            replay.load(batches[i])
        nxt = batches[order[k + ahead]]["points"] if ahead and k + ahead < len(order) else None
        got.append(float(replay(next_points=nxt)))
    torch.cuda.synchronize()
    assert eager == pytest.approx(got, rel=2e-3), (eager, got)
    assert max(eager) - min(eager) > 0.05 * max(eager)        # the batches really differ
    for (n, p), q in zip(me.named_parameters(), mg.parameters()):
        assert torch.allclose(p, q, rtol=1e-2, atol=2e-4), n


def test_three_training_steps_follow_the_fp64_oracle():
    """End to end over several optimizer steps: forward + loss + backward + gradient clipping (max norm 10,
    configs/_base_/schedules/schedule_3x.py:6-7) + AdamW with the decoder group at lr x 0.05
    (configs/demf/demf_votenet.py:16-24) on the HIP path against the same loop on the fp64 CPU oracle
    (oracle/model.py + torch.optim.AdamW).  The loss of every step and the parameters after the last one
    must agree; train-mode BatchNorm running statistics are part of the state that has to follow."""
    import numpy as np
    from oracle.model import OracleDeMF
    from demf_amd import engine, synthetic
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    seed, lr, steps = 11, 2e-4, 3
    raw = synthetic.make_scene_batch(2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                     cfg.head.embed_dims, seed=seed, n_gt=4)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
                 img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                 img_metas=raw["img_metas"],
                 gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                 gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
    tr = engine.Trainer(model, lr=lr)
    got = [tr.step(batch).item() for _ in range(steps)]
    torch.cuda.synchronize()

    ref = OracleDeMF(cfg)
    fixtures.seed_weights(ref, seed)
    ref.train().double()
    dec = [p for n, p in ref.named_parameters() if "decoder" in n]
    rest = [p for n, p in ref.named_parameters() if "decoder" not in n]
    opt = torch.optim.AdamW([dict(params=rest, lr=lr, weight_decay=0.01),
                             dict(params=dec, lr=lr * 0.05, weight_decay=0.01)], lr=lr, weight_decay=0.01)
    pts = torch.from_numpy(raw["points"]).double()
    feats = [torch.from_numpy(f).double() for f in raw["img_features"]]
    gtb = [torch.from_numpy(b).double() for b in raw["gt_boxes"]]
    gtl = [torch.from_numpy(l) for l in raw["gt_labels"]]
    want = []
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(16, nthreads))                   # (fp64 autograd of small tensors: oversubscription hurts)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        losses, _, _ = ref.forward_train(pts, feats, raw["img_metas"], gtb, gtl)
        total = sum(losses.values())
        total.backward()
        torch.nn.utils.clip_grad_norm_(list(ref.parameters()), 10.0)
        opt.step()
        want.append(total.item())
    torch.set_num_threads(nthreads)
    np.testing.assert_allclose(got, want, rtol=2e-3, err_msg="loss per step")
    assert want[-1] < want[0]                                   # and the loop does train
    sd_ref = ref.state_dict()
    worst = ("", 0.0)
    for k, v in model.state_dict().items():
        if not torch.is_floating_point(v):
            assert int(v) == int(sd_ref[k]), k                  # num_batches_tracked
            continue
        a, b = v.double().cpu(), sd_ref[k].double()
        rel = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
        if rel > worst[1]:
            worst = (k, rel)
    print(f"worst parameter / buffer after {steps} steps: {worst[0]} rel-L2 {worst[1]:.2e}")
    assert worst[1] <= 1.2e-3, worst                             # 2 x the observed 4.6e-4 .. 5.3e-4


def _shape_batches(cfg, n, seed0=40):
    """Batches of two padded image sizes (two pyramids) with differing per-scene GT counts."""
    from demf_amd import synthetic
    # (more tokens than the decoder samples - 32 queries x 4 levels x 2 points x 4 corners = 1 024 - so
    # that the fusion layers take the one-node device path with its counter-based dropout)
    pyr = [((32, 44), (16, 22), (8, 11), (4, 6)), ((32, 40), (16, 20), (8, 10), (4, 5))]
    inp = [(256, 352), (256, 320)]
    order = [0, 1, 0, 1, 1, 0, 0, 1, 0, 1]
    out = []
    for i in range(n):
        w = order[i % len(order)]
        counts = [(3 * i + 2 * b) % 7 for b in range(3)]          # 0 .. 6 boxes, never the same signature
        if i == 6:
            counts[1] = 11                                         # a one-off that needs the 16-slot bucket
        raw = synthetic.make_scene_batch(3, 1024, pyr[w], inp[w], cfg.head.embed_dims, seed=seed0 + i,
                                         gt_counts=counts)
        out.append(dict(points=torch.from_numpy(raw["points"]).cuda(),
                        img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                        img_metas=raw["img_metas"],
                        gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                        gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]]))
    return out


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_shape_bucketed_capture_matches_eager_steps(dropout):
    """Trainer.bucketed(): batches of two padded image sizes and varying GT counts cycle through one
    captured graph per shape (class_agnostic_vote_head.py:556-568 rebuilds masks per batch from
    batch_input_shape; demf_votenet.py:194-197 pads every batch to its own multiple of 32).  The
    sequence of updates must equal that of plain eager steps on a twin: captures are dry (BatchNorm
    statistics and the dropout counter restored), shapes seen once run eagerly, graphs of one cloud
    shape share the pipelined pre-pass."""
    import dataclasses
    from demf_amd import engine
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    cfg = dataclasses.replace(cfg, head=dataclasses.replace(cfg.head, attn_dropout=dropout, ffn_dropout=dropout))
    batches = _shape_batches(cfg, 10)

    def run(bucketed):
        torch.manual_seed(11)
        model = DeMFHotPath(cfg)
        fixtures.seed_weights(model, 9)
        model.cuda().train()
        tr = engine.Trainer(model, lr=2e-4)           # (re)seeds the dropout counter
        stepper = tr.bucketed(max_graphs=3, capture_on=2, warmup=1) if bucketed else None
        losses = []
        for i, b in enumerate(batches):
            nxt = batches[i + 1]["points"] if i + 1 < len(batches) else None
            l = stepper.step(b, next_points=nxt) if bucketed else tr.step(b)
            losses.append(float(l))
        torch.cuda.synchronize()
        bufs = torch.cat([b.detach().double().reshape(-1) for b in model.buffers()])
        return losses, tr.opt.flat.clone().double(), bufs, stepper

    la, pa, ba, _ = run(False)
    lb, pb, bb, sc = run(True)
    # shapes: two pyramids x the 8-slot bucket (+ one 16-slot one-off): first sight eager, then captured
    assert sc.stats["captured"] == 2 and sc.stats["eager"] == 3, sc.stats
    assert sc.stats["replayed"] == len(batches) - sc.stats["eager"]
    assert len(sc.pipes) == 1, "both image sizes share the cloud shape's pre-pass pipeline"
    # (two runs of the SAME path differ through the order of their fp32 atomics, and on this untrained network
    # the difference grows with every update: 2e-3 holds for the first steps, by step 7-8 single runs reach 3.5e-3 -
    # a wrong sequence of updates, e.g. a warm-up pass that was not undone, shows at its first step and is far larger)
    # From step 6 on a near-tie (max-pool / ReLU, see below) may resolve differently between the two engines and move
    # the loss by up to ~0.7 %: the late steps carry a 1.5e-2 ceiling, the early ones the tight bound.
    for i, (x, y) in enumerate(zip(la, lb)):
        tol = 2e-3 * (1 + i / 4) if i < 6 else 1.5e-2
        assert abs(x - y) <= tol * abs(x), (i, x, y)
    # parameters: one missing / extra AdamW update at this learning rate is ~3e-3 of the norm; two correct runs end
    # within 1e-4 of each other, or within ~2.7e-4 when the step-7 near-tie described below resolved differently
    assert ((pa - pb).norm() / pa.norm()).item() < 6e-4
    # BatchNorm statistics: no extra warm-up passes (one leaked pass moves them by ~1e-1).  The run is bimodal: in
    # about one run of four a near-tie (max-pool / ReLU) resolves differently between the two engines at step 7
    # (loss 15.920 vs 15.863, the same two values every time, also at the round-4 commit) and the statistics then
    # differ by 1.2e-3 instead of < 1e-4
    assert ((ba - bb).norm() / ba.norm()).item() < 3e-3


def test_step_cache_evicts_least_recently_used():
    from demf_amd import engine
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    batches = _shape_batches(cfg, 4)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, 9)
    model.cuda().train()
    sc = engine.Trainer(model, lr=1e-4).bucketed(max_graphs=1, capture_on=1, warmup=1)
    for b in batches:                       # shapes 0 1 0 1 with room for one graph: every step re-captures
        assert bool(torch.isfinite(sc.step(b)))
    assert sc.stats["captured"] == 4 and sc.stats["evicted"] == 3 and len(sc.graphs) == 1
