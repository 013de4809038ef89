"""Round 5: the captured step's zero-copy image hand-over."""
import os

import pytest
import torch

from oracle import fixtures

pytestmark = pytest.mark.gpu

PYRAMID = ((32, 44), (16, 22), (8, 11), (4, 6))          # 1 872 tokens > 32 * 4 * 2 * 4 sampled corners
INPUT = (256, 352)


def _batch(seed, n_gt, B=3):
    from demf_amd import synthetic
    cfg = fixtures.tiny_cfg()
    raw = synthetic.make_scene_batch(B, 1024, PYRAMID, INPUT, cfg.head.embed_dims, seed=seed, n_gt=n_gt)
    return dict(points=torch.from_numpy(raw["points"]).cuda(),
                img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                img_metas=raw["img_metas"],
                gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])


def _trainer(lr=2e-5, seed=3):
    from demf_amd import engine
    from demf_amd.modules import DeMFHotPath
    model = DeMFHotPath(fixtures.tiny_cfg())
    fixtures.seed_weights(model, seed)
    model.cuda().train()
    return engine.Trainer(model, lr=lr), model


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_load_converts_the_callers_pyramid_in_place(mode):
    """A step captured on a list of (B,C,H_l,W_l) maps keeps TOKENS as its static image input; ``load`` converts
    each new batch straight out of the caller's maps (DeMFVoteHead.pyramid_tokens -> demf_pyramid_to_tokens, padding
    rows zeroed by the NEW batch's mask, bf16 rows in the bf16 mode) instead of copying the maps into static buffers
    and transposing inside the graph.  Three batches with different images / paddings / GT counts cycled through the
    graph give the losses and parameters of eager steps on them, and of the copying form (DEMF_ZERO_COPY_TOKENS=0)."""
    from demf_amd import ops
    batches = [_batch(21, 4), _batch(22, 2), _batch(23, 5)]
    order = [0, 1, 2, 0, 2, 1]
    ops.set_compute_dtype(mode)
    try:
        te, me = _trainer()
        for _ in range(2):
            te.step(batches[order[0]])
        eager = [float(te.step(batches[i])) for i in order]
        runs = {}
        for zc in ("1", "0"):
            os.environ["DEMF_ZERO_COPY_TOKENS"] = zc
            tg, mg = _trainer()
            replay = tg.capture(batches[order[0]], warmup=2, max_gt=8)
            is_tok = isinstance(replay.static["img_features"], dict)
            assert is_tok == (zc == "1"), "sample-then-project must be active at this shape"
            if is_tok:
                tok = replay.static["img_features"]["tokens"]
                assert tok.dtype == (torch.bfloat16 if mode == "bf16" else torch.float32)
            got = []
            for k, i in enumerate(order):
                if k:
                    replay.load(batches[i])
                nxt = batches[order[k + 1]]["points"] if k + 1 < len(order) else None
                got.append(float(replay(next_points=nxt)))
            torch.cuda.synchronize()
            runs[zc] = (got, [p.detach().clone() for p in mg.parameters()])
    finally:
        os.environ.pop("DEMF_ZERO_COPY_TOKENS", None)
        ops.set_compute_dtype("f32")
    assert max(eager) - min(eager) > 0.02 * max(eager)        # the batches really differ
    for zc, (got, params) in runs.items():
        if mode == "bf16":
            # (bf16 arithmetic on this untrained 30-BatchNorm network: two runs of the SAME code differ by percents
            # even on the first step - atomics order decides near-ties; the trajectory bar is the f32 mode's, here
            # finiteness and the bit-exact token check below)
            assert all(map(lambda v: v == v and abs(v) < 1e6, got))
            continue
        assert eager == pytest.approx(got, rel=2e-3), (zc, eager, got)
        for (n, p), q in zip(me.named_parameters(), params):
            assert torch.allclose(p, q, rtol=1e-2, atol=2e-4), (zc, n)
    # the tokens the graph read last are exactly the conversion of the last batch
    os.environ["DEMF_ZERO_COPY_TOKENS"] = "1"
    ops.set_compute_dtype(mode)
    try:
        tg, mg = _trainer()
        replay = tg.capture(batches[0], warmup=1, max_gt=8)
        replay.load(batches[2])
        want = mg.pts_bbox_head.pyramid_tokens(batches[2]["img_features"], batches[2]["img_metas"])["tokens"]
        assert torch.equal(replay.static["img_features"]["tokens"], want)
        keep = ~mg.pts_bbox_head._meta_tensors(batches[2]["img_metas"], list(PYRAMID), want.device,
                                               torch.float32)["mask_flatten"]
        assert not keep.all() and (want[~keep] == 0).all()      # padded scenes: zeroed rows
        flat = torch.cat([f.flatten(2).transpose(1, 2) for f in batches[2]["img_features"]], 1)
        assert torch.equal(want[keep], flat[keep].to(want.dtype))
    finally:
        os.environ.pop("DEMF_ZERO_COPY_TOKENS", None)
        ops.set_compute_dtype("f32")


def test_overlapped_collective_defers_the_update_but_not_its_result():
    """engine.Trainer with the collective on the communication stream (here a 200 us spin kernel in its place:
    allreduce_stub_us) and norm + AdamW deferred to the start of the next replay: after ``flush()`` parameters and
    optimizer state equal those of the serial order, step for step; ``state_dict()`` and an eager ``step()`` flush
    by themselves."""
    batches = [_batch(31, 4), _batch(32, 3)]
    outs = []
    for overlap in (False, True):
        tr, model = _trainer(lr=2e-5)
        tr.allreduce_stub_us, tr.allreduce_overlap = 200, overlap
        assert tr.allreduce_config() == (1, 200, overlap)
        replay = tr.capture(batches[0], warmup=1, max_gt=8)
        losses = []
        for k in range(5):
            if k:
                replay.load(batches[k % 2])
            losses.append(float(replay(next_points=batches[(k + 1) % 2]["points"])))
            assert bool(getattr(tr, "_pending_update", False)) == overlap
        sd = tr.state_dict()                              # flushes
        assert not getattr(tr, "_pending_update", False) and tr.opt.t == 5 + 1    # (+ the warm-up step)
        tr.step(batches[0])
        torch.cuda.synchronize()
        outs.append((losses, [p.detach().clone() for p in model.parameters()], sd["optimizer"]["exp_avg"]))
    (la, pa, ma), (lb, pb, mb) = outs
    assert la == pytest.approx(lb, rel=2e-3)
    rel = lambda x, y: ((x.double() - y.double()).norm() / x.double().norm().clamp_min(1e-30)).item()
    for x, y in zip(pa, pb):
        assert rel(x, y) <= 1e-3                      # (same updates; fp32 atomics order aside)
    assert rel(ma, mb) <= 5e-2, rel(ma, mb)           # first moments: sums of seven noisy gradients


@pytest.mark.parametrize("mask", [False, True])
def test_linear_on_long_row_sets_has_no_library_gemm(mask):
    """ops.linear beyond 65 536 rows (the value projection of project-then-sample, the image encoder's module
    path): forward and input gradient on demf_rows_gemm_f32, weight gradient on the slab-split dW kernel, bias
    gradient on demf_colsum_f32 - against fp64, with and without the padding-row mask; a shape the long-row kernel
    does not take (N = 96) goes to the strided GEMM of csrc/dense.hip.  (Until round 5 this branch was torch.addmm.)"""
    from demf_amd import ops, _ffi
    g = torch.Generator().manual_seed(4)
    for R, K, N, expect in ((70000, 64, 128, "demf_rows_gemm_f32"), (66000, 64, 96, "demf_gemm_f32")):
        x = torch.randn(R, K, generator=g).cuda().requires_grad_()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_()
        b = torch.randn(N, generator=g).cuda().requires_grad_()
        m = (torch.rand(R, generator=g) < 0.3).cuda() if mask else None
        seen, orig = [], _ffi.call
        _ffi.call = lambda name, *a: (seen.append(name), orig(name, *a))[1]
        try:
            y = ops.linear(x, w, b, row_mask=m)
            go = torch.randn(R, N, generator=g).cuda()
            y.backward(go.clone())
        finally:
            _ffi.call = orig
        assert expect in seen, seen
        xd, wd, bd = x.detach().double(), w.detach().double(), b.detach().double()
        want = xd @ wd.t() + bd
        gd = go.double()
        if mask:
            want[m] = 0
            gd = gd.clone()
            gd[m] = 0
        assert (y.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        for got, ref in ((x.grad, gd @ wd), (w.grad, gd.t() @ xd), (b.grad, gd.sum(0))):
            err = (got.double() - ref).abs().max().item()
            assert err <= 5e-5 * ref.abs().max().item(), (R, N, err, ref.abs().max().item())


def test_side_stream_is_probed_for_real_concurrency():
    """engine.concurrent_stream: the stream the pipelined pre-pass (and the overlapped collective) runs on must not
    share the main stream's hardware queue - two 400 us spin kernels, one per stream, finish in about one length."""
    from demf_amd import _ffi, engine
    main = torch.cuda.current_stream()
    for _ in range(6):                                   # whatever torch's stream pool hands out next
        side = engine.concurrent_stream()
        side.wait_stream(main)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ffi.call("demf_spin_us", 400, main.cuda_stream)
        with torch.cuda.stream(side):
            _ffi.call("demf_spin_us", 400, side.cuda_stream)
        main.wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
        assert e0.elapsed_time(e1) < 0.65, e0.elapsed_time(e1)


def test_double_buffered_step_matches_eager_steps():
    """engine.DoubleBufferedStep: two captured graphs over two sets of static inputs, each batch loaded into the idle
    set on an input stream while the previous step runs.  Six batches (different images, paddings, GT counts) in a
    changing order give the losses and parameters of eager steps - i.e. no step ever read a half-loaded or a stale
    set - with the next cloud handed to the shared pre-pass pipeline, and also without it."""
    batches = [_batch(41, 4), _batch(42, 2), _batch(43, 5), _batch(44, 3)]
    order = [0, 1, 2, 3, 1, 0, 3, 2]
    te, me = _trainer()
    for _ in range(2):
        te.step(batches[order[0]])
    eager = [float(te.step(batches[i])) for i in order]
    for ahead in (True, False):
        tg, mg = _trainer()
        step = tg.capture_double(batches[order[0]], batches[order[1]], warmup=2, max_gt=8)
        assert isinstance(step.static["img_features"], dict)           # zero-copy tokens per set
        got = []
        for k, i in enumerate(order):
            if k:
                step.load(batches[i])
            nxt = batches[order[k + 1]]["points"] if ahead and k + 1 < len(order) else None
            got.append(float(step(next_points=nxt)))
        torch.cuda.synchronize()
        assert eager == pytest.approx(got, rel=2e-3), (ahead, eager, got)
        for (n, p), q in zip(me.named_parameters(), mg.parameters()):
            assert torch.allclose(p, q, rtol=1e-2, atol=2e-4), (ahead, n)
    assert max(eager) - min(eager) > 0.02 * max(eager)
