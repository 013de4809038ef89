"""Round-4 additions to the device path, each against the code it replaces or the torch statement of
the reference behaviour:
  * demf_pad_gt (per-scene GT lists -> padded (B,G,7)/(B,G), class_agnostic_vote_head.py:766-773)
    against the host table path of DeMFVoteHead.pad_gt, for count signatures that never repeat;
  * the ``asum`` fallback of demf_gemm_f32 (bias gradient of a linear layer whose weight-gradient
    launch does not fit the small-tile path: a wider FFN than the reference's 1024, a lowered A/B knob).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gt_lists(counts, seed, box_dim=7):
    g = torch.Generator().manual_seed(seed)
    boxes = [torch.randn(n, box_dim, generator=g) for n in counts]
    labels = [torch.randint(0, 10, (n,), generator=g) for n in counts]
    return boxes, labels


@pytest.mark.parametrize("counts", [(3, 0, 5, 1), (0, 0), (8,) * 8, (1, 2, 3, 4, 5, 6, 7, 0, 9, 2, 4, 6, 1, 1, 0, 3)])
@pytest.mark.parametrize("G", [None, 12])
def test_pad_gt_device_lists_match_the_host_tables(counts, G):
    from demf_amd.modules.head import DeMFVoteHead
    boxes, labels = _gt_lists(counts, seed=sum(counts) + len(counts))
    for slot in (False, True):
        want = DeMFVoteHead.pad_gt(boxes, labels, torch.device("cpu"), with_slot_labels=slot, G=G)
        got = DeMFVoteHead.pad_gt([b.cuda() for b in boxes], [l.cuda() for l in labels], torch.device("cuda"),
                                  with_slot_labels=slot, G=G)
        for w, g_, name in zip(want, got, ("gt", "labels", "valid")):
            assert g_.is_cuda and g_.shape == w.shape and g_.dtype == w.dtype, name
            assert torch.equal(g_.cpu(), w), (name, slot)


def test_pad_gt_issues_no_host_to_device_copy():
    """The per-batch path must not upload anything: count signatures that were never seen before go
    through the same single launch (the table cache only hid the copies for repeats)."""
    from demf_amd.modules.head import DeMFVoteHead
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda")
    lists = []
    for s in range(6):
        counts = tuple(int(c) for c in np.random.default_rng(s).integers(0, 9, size=8))
        b, l = _gt_lists(counts, seed=100 + s)
        lists.append(([x.cuda() for x in b], [x.cuda() for x in l]))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for b, l in lists:
            DeMFVoteHead.pad_gt(b, l, dev, with_slot_labels=True, G=8)
        torch.cuda.synchronize()
    names = [e.name.lower() for e in prof.events()]
    assert not any("memcpy" in n and "htod" in n for n in names), [n for n in names if "memcpy" in n]


def test_pad_gt_rejects_overfull_scene():
    from demf_amd.modules.head import DeMFVoteHead
    boxes, labels = _gt_lists((3, 9), 0)
    with pytest.raises(ValueError, match="ground-truth boxes"):
        DeMFVoteHead.pad_gt([b.cuda() for b in boxes], [l.cuda() for l in labels], torch.device("cuda"), G=8)


@pytest.mark.parametrize("F", [1024, 2048])
def test_linear_bias_gradient_beyond_the_small_tile_path(F, monkeypatch):
    """ops.linear backward with a (F x 256) weight on 4 096 rows: F = 2048 is 2 048 tiles - past the
    small-tile limit the in-kernel row sums need; the library then runs the product without them and
    takes the bias gradient with demf_colsum_f32 instead of failing."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(F)
    x = torch.randn(4096, 256, generator=g).cuda().requires_grad_()     # split-K 16: 1 024 / 2 048 tiles
    w = (torch.randn(F, 256, generator=g) * 0.05).cuda().requires_grad_()
    b = torch.randn(F, generator=g).cuda().requires_grad_()
    dy = torch.randn(4096, F, generator=g).cuda()
    y = ops.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
    x64, w64, b64 = (t.detach().double() for t in (x, w, b))
    torch.testing.assert_close(y.double(), x64 @ w64.t() + b64, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gb.double(), dy.double().sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(gw.double(), dy.double().t() @ x64, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(gx.double(), dy.double() @ w64, rtol=1e-4, atol=1e-3)


def test_gemm_asum_fallback_through_the_c_abi():
    """demf_gemm_f32 with ``asum`` on a launch shape that is NOT the small-tile reduction-strided form:
    (a) more tiles than the limit, (b) the grouped entry point's single-launch fallback."""
    from demf_amd import fused
    g = torch.Generator().manual_seed(7)
    R, N, K = 4096, 2304, 64          # dW (N,K) = dy^T x : 36 x 1 tiles of 64 x 64, x split-K 16 x ... > 1024
    dy = torch.randn(R, N, generator=g).cuda()
    x = torch.randn(R, K, generator=g).cuda()
    want_w = dy.double().t() @ x.double()
    want_b = dy.double().sum(0)
    for grouped in (False, True):
        dw = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        grp = [] if grouped else None
        fused.gemm(N, K, R, dy.data_ptr(), (1, N), x.data_ptr(), (1, K), dw.data_ptr(), K,
                   splitk=64, asum=db.data_ptr(), group=grp)
        if grouped:
            fused.gemm_group(grp)
        torch.testing.assert_close(dw.double(), want_w, rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(db.double(), want_b, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("Dh", [32, 6])          # the 4-channel-per-lane kernels and the any-width form
def test_msda_bf16_value_rows_are_the_fp32_operator_on_widened_rows(Dh):
    """demf_msda_{fwd,bwd}_bf16: value rows stored as bf16 (2-byte elements) - BIT-identical to the fp32 entry
    points on the same rows widened to fp32 (forward; the backward's scalar gradients likewise), because the
    widening is exact and everything after the load is the same fp32 code."""
    from demf_amd import _ffi
    B, Q, H, L, P = 2, 100, 4, 4, 2
    shapes = [(20, 28), (10, 14), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(Dh)
    v16 = torch.randn(B, S, H, Dh, generator=g).bfloat16().cuda()
    v32 = v16.float()
    ss = torch.tensor(shapes, dtype=torch.long).cuda()
    lsi = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    loc = torch.rand(B, Q, H, L, P, 2, generator=g).cuda() * 1.2 - 0.1
    w = torch.softmax(torch.randn(B, Q, H, L * P, generator=g), -1).view(B, Q, H, L, P).cuda()
    go = torch.randn(B, Q, H * Dh, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    outs = {}
    for name, v in (("f32", v32), ("bf16", v16)):
        out = torch.empty(B, Q, H * Dh, device="cuda")
        gv, gl, gw = torch.zeros(B, S, H, Dh, device="cuda"), torch.empty_like(loc), torch.empty_like(w)
        _ffi.call("demf_msda_fwd_" + name, B, S, H, Dh, L, Q, P, p(v), p(ss), p(lsi), p(loc), p(w), p(out), st)
        _ffi.call("demf_msda_bwd_" + name, B, S, H, Dh, L, Q, P, p(v), p(ss), p(lsi), p(loc), p(w), p(go), p(gv), p(gl),
                  p(gw), st)
        outs[name] = (out, gl, gw, gv)
    for a, b, what in zip(outs["f32"][:3], outs["bf16"][:3], ("out", "grad_loc", "grad_weight")):
        assert torch.equal(a, b), what
    torch.testing.assert_close(outs["f32"][3], outs["bf16"][3], rtol=1e-5, atol=1e-5)      # (atomics order)
    assert outs["f32"][0].abs().max() > 0


def test_compute_mode_is_a_parameter_per_thread():
    """demf_ctx: two host threads run the same dense call in DIFFERENT modes at the same time (their own
    streams), each through its own context - results equal the single-threaded runs under the process default
    set to that mode, and the process default itself is never touched."""
    import ctypes
    import threading
    from demf_amd import _ffi, ops
    lib = _ffi.load()

    class Ctx(ctypes.Structure):
        _fields_ = [("compute_mode", ctypes.c_int), ("reserved", ctypes.c_int * 7)]

    R, K, N = 4096, 64, 128
    g = torch.Generator().manual_seed(3)
    X, Wt = torch.randn(R, K, generator=g).cuda(), (torch.randn(N, K, generator=g) / 8).cuda()

    def run(mode, use_ctx, stream, reps=1):
        Y = torch.empty(R, N, device="cuda")
        with torch.cuda.stream(stream):
            for _ in range(reps):
                if use_ctx:
                    c = Ctx(mode, (ctypes.c_int * 7)())
                    _ffi.call("demf_mlp_gemm_fwd_ctx", ctypes.addressof(c), R, K, N, K, X.data_ptr(), None, Wt.data_ptr(),
                              Y.data_ptr(), None, stream.cuda_stream)
                else:
                    _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, X.data_ptr(), None, Wt.data_ptr(), Y.data_ptr(), None,
                              stream.cuda_stream)
        stream.synchronize()
        return Y
    s0 = torch.cuda.Stream()
    torch.cuda.synchronize()
    want = {}
    for mode, name in ((1, "bf16"), (2, "f32")):
        ops.set_compute_dtype(name)
        want[mode] = run(mode, False, s0)
    ops.set_compute_dtype("f32")
    default = lib.demf_get_compute_dtype()
    assert not torch.equal(want[1], want[2])                     # the modes really differ
    got, errs = {}, []

    def worker(mode):
        try:
            got[mode] = run(mode, True, torch.cuda.Stream(), reps=200)
            # the scoped form: every dense call of THIS thread between push and pop
            c = Ctx(mode, (ctypes.c_int * 7)())
            _ffi.call("demf_ctx_push", ctypes.addressof(c))
            try:
                assert lib.demf_get_compute_dtype() == mode
                got[("scoped", mode)] = run(mode, False, torch.cuda.Stream(), reps=50)
            finally:
                _ffi.call("demf_ctx_pop")
        except Exception as e:          # noqa: BLE001 - reported below
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(m,)) for m in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for m in (1, 2):
        assert torch.equal(got[m], want[m]) and torch.equal(got[("scoped", m)], want[m]), m
    assert lib.demf_get_compute_dtype() == default
    with pytest.raises(RuntimeError, match="no context pushed"):
        _ffi.call("demf_ctx_pop")


def test_prepass_bookkeeping_kernels_match_torch():
    """demf_sa_index_chain / demf_split_points (the pre-pass's arange + int64 conversions + gathers, and the two
    strided input copies) against the torch statements they replace."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(1)
    B, N = 3, 5000
    sizes = [2048, 1024, 512, 256]
    idx, prev = [], N
    for m in sizes:
        idx.append(torch.stack([torch.randperm(prev, generator=g)[:m] for _ in range(B)]).int().cuda())
        prev = m
    got = ops.sa_index_chain(N, idx)
    want = [torch.arange(N).unsqueeze(0).repeat(B, 1).cuda()]
    for t in idx:
        want.append(torch.gather(want[-1], 1, t.long()))
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.dtype == torch.int64 and torch.equal(a, b)
    for C in (1, 0, 5):
        pts = torch.randn(B, N, 3 + C, generator=g).cuda()
        xyz, feat = ops.split_points(pts)
        assert torch.equal(xyz, pts[..., :3]) and xyz.is_contiguous()
        assert (feat is None) if C == 0 else (torch.equal(feat, pts[..., 3:]) and feat.is_contiguous())


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_grouped_weight_gradients_equal_the_single_launches(mode):
    """demf_mlp_gemm_bwd_dw_group: many dW = dZ^T A products of different shapes (dense and pooled upstream
    gradients, raw and BN+ReLU'd inputs, a strided destination) in a handful of launches - the same numbers as
    one demf_mlp_gemm_bwd_dw_ld each (fp32 atomics order aside)."""
    import ctypes
    from demf_amd import _ffi, ops
    g = torch.Generator().manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    shapes = [(2048, 128, 128, 1, True), (8192, 256, 256, 1, True), (4096, 256, 512, 1, False), (2048, 12, 128, 1, True),
              (32768, 128, 128, 16, True), (8192, 256, 384, 1, True), (2048, 32, 128, 1, False), (65536, 128, 64, 32, True),
              (1024, 64, 64, 1, True), (16384, 256, 256, 1, True), (20000, 128, 128, 1, True), (512, 256, 128, 1, True),
              (4096, 128, 256, 1, True), (2048, 128, 128, 1, True)]
    ops.set_compute_dtype(mode)
    try:
        jobs, keep, want, got = [], [], [], []
        for (R, N, K, ns, pro) in shapes:
            Y = torch.randn(R, N, generator=g).cuda()
            X = torch.randn(R, K, generator=g).cuda()
            vec = torch.randn(5 * N, generator=g).cuda()
            pss = torch.randn(2 * K, generator=g).cuda() if pro else None
            if ns > 1:
                G, dP = None, torch.randn(R // ns, N, generator=g).cuda()
                arg = torch.randint(0, ns, (R // ns, N), generator=g).int().cuda()
            else:
                G, dP, arg = torch.randn(R, N, generator=g).cuda(), None, None
            ld = K + 4
            w1, w2 = torch.zeros(N, ld, device="cuda"), torch.zeros(N, ld, device="cuda")
            _ffi.call("demf_mlp_gemm_bwd_dw_ld", R, N, K, K, p(G), p(dP), p(arg), ns, p(Y), p(vec), p(X), p(pss), p(w1),
                      ld, st)
            jobs.append(_ffi.DwJob(R, N, K, K, p(G), p(dP), p(arg), ns, p(Y), p(vec), p(X), p(pss), p(w2), ld))
            keep.append((Y, X, vec, pss, G, dP, arg))
            want.append(w1)
            got.append(w2)
        arr = (_ffi.DwJob * len(jobs))(*jobs)
        _ffi.call("demf_mlp_gemm_bwd_dw_group", len(jobs), ctypes.addressof(arr), st)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype("f32")
    for i, (a, b) in enumerate(zip(want, got)):
        scale = a.abs().max().item()
        assert scale > 0 and (a - b).abs().max().item() <= 2e-5 * scale, (i, shapes[i], (a - b).abs().max().item(), scale)
        assert torch.equal(a[:, shapes[i][2]:], b[:, shapes[i][2]:])         # the padding columns stay untouched


def test_deferred_weight_gradients_reach_every_parameter():
    """engine.FlatGrads.backward_into queues the few-row stacks' dW products (ops.deferred_weight_grads) and issues
    them as grouped launches at the end of the backward: every parameter's gradient equals the immediate form -
    including first-layer weights that reach the kernel through a torch.cat (padded columns), whose gradient
    autograd slices the moment the node returns and which therefore must NOT be queued."""
    from oracle import fixtures
    from demf_amd import engine, ops, synthetic
    from demf_amd.modules import DeMFHotPath

    def run(defer):
        old, ops._DEFER_DW = ops._DEFER_DW, defer
        try:
            cfg = fixtures.tiny_cfg()
            model = DeMFHotPath(cfg)
            fixtures.seed_weights(model, 7)
            model.cuda().train()
            raw = synthetic.make_scene_batch(2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT, cfg.head.embed_dims,
                                             seed=50, n_gt=4)
            batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
                         img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                         img_metas=raw["img_metas"], gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                         gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
            tr = engine.Trainer(model, lr=1e-3)
            tr._fwd_bwd(batch)
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            ops._DEFER_DW = old
    a, b = run(0), run(1)
    assert a.keys() == b.keys() and not ops.DEFER.jobs and not ops.DEFER.keep
    for n in a:
        s = a[n].abs().max().item()
        assert (a[n] - b[n]).abs().max().item() <= 1e-4 * max(s, 1e-6), n


def test_deferred_weight_gradients_with_a_weight_used_by_two_nodes():
    """A shared MLP called twice inside ops.deferred_weight_grads(): the autograd engine sums the two weight
    gradients the moment the second node returns, so both products must have been ISSUED by then (a repeated
    parameter flushes the queue and runs at once - ops.dw_job); nested contexts join the outer queue."""
    from demf_amd import ops
    from demf_amd.modules.layers import RowsMLP
    torch.manual_seed(2)
    mlp = RowsMLP([64, 64, 64], dim=1).cuda().train()
    xa, xb = torch.randn(512, 64, device="cuda"), torch.randn(768, 64, device="cuda")

    def run(defer):
        old, ops._DEFER_DW = ops._DEFER_DW, defer
        try:
            for p in mlp.parameters():
                p.grad = None
            with ops.deferred_weight_grads():
                with ops.deferred_weight_grads():          # re-entrant: the inner exit neither flushes nor switches off
                    pass
                assert ops.DEFER.on
                loss = mlp.forward_rows(xa).square().sum() + 0.5 * mlp.forward_rows(xb).sum()
                loss.backward()
            assert not ops.DEFER.on and not ops.DEFER.jobs and not ops.DEFER.leaves
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in mlp.named_parameters() if p.grad is not None}
        finally:
            ops._DEFER_DW = old
    a, b = run(0), run(1)
    assert a.keys() == b.keys()
    for n in a:
        s = a[n].abs().max().item()
        assert s > 0 and (a[n] - b[n]).abs().max().item() <= 1e-4 * s, n
