"""Round 6: the optimizer update as nodes of the step's hipGraph (device-resident step count / learning-rate
factor / gradient norm), the engine fixes that came with it."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import fixtures

pytestmark = pytest.mark.gpu

PYRAMID = ((32, 44), (16, 22), (8, 11), (4, 6))
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


def test_sumsq_and_state_adamw_kernels():
    """demf_sumsq_f32 / demf_multi_copy_sumsq accumulate the squared norm into the device state, and
    demf_adamw_state_f32 (every group in one launch, step count and lr factor read from the device) reproduces
    clip_grad_norm_ + torch.optim.AdamW over steps with and without active clipping and an lr-factor change."""
    from demf_amd import _ffi, ops
    torch.manual_seed(1)
    n0, n1 = 5003, 777
    n = n0 + n1
    p = torch.randn(n, device="cuda")
    ref_p = [p[:n0].clone().cpu().requires_grad_(), p[n0:].clone().cpu().requires_grad_()]
    opt = torch.optim.AdamW([dict(params=[ref_p[0]], lr=0.008, weight_decay=0.01),
                             dict(params=[ref_p[1]], lr=0.0004, weight_decay=0.02)])
    g = torch.zeros(n, device="cuda")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(64, dtype=torch.uint8, device="cuda")
    state.view(torch.float32)[5] = 1.0
    seg = ((ctypes.c_longlong * 2)(0, n0), (ctypes.c_longlong * 2)(n0, n1), (ctypes.c_float * 2)(0.008, 0.0004),
           (ctypes.c_float * 2)(0.01, 0.02))
    a = [ctypes.cast(x, ctypes.c_void_p) for x in seg]
    st = torch.cuda.current_stream().cuda_stream
    for it in range(6):
        scale = 30.0 if it % 2 == 0 else 0.01
        grads = torch.randn(n) * scale
        if it == 4:
            state.view(torch.float32)[5] = 0.1
            for grp in opt.param_groups:
                grp["lr"] *= 0.1
        ref_p[0].grad, ref_p[1].grad = grads[:n0].clone(), grads[n0:].clone()
        torch.nn.utils.clip_grad_norm_(ref_p, 10.0)
        opt.step()
        if it % 2 == 0:
            g.copy_(grads.cuda())
            _ffi.call("demf_sumsq_f32", n, g.data_ptr(), state.data_ptr(), st)
        else:
            # the pack form: two unaligned source pieces copied into g, norm taken on the way
            src = grads.cuda()
            k = 1001
            pieces = [src[:k].clone(), src[k:].clone()]
            g.zero_()
            tab = torch.tensor([[t.data_ptr() for t in pieces], [g.data_ptr(), g.data_ptr() + 4 * k],
                                [k, n - k]], dtype=torch.int64, device="cuda")
            _ffi.call("demf_multi_copy_sumsq", 2, tab.data_ptr(), 4, state.data_ptr(), st)
            assert torch.equal(g, src)
        got = float(state.view(torch.float64)[0])
        assert got == pytest.approx(float(grads.double().pow(2).sum()), rel=1e-6)
        _ffi.call("demf_adamw_state_f32", 2, a[0], a[1], a[2], a[3], p.data_ptr(), g.data_ptr(), m.data_ptr(),
                  v.data_ptr(), state.data_ptr(), 10.0, 1.0, 0.9, 0.999, 1e-8, st)
        torch.cuda.synchronize()
        assert int(state.view(torch.int64)[1]) == it + 1
        assert float(state.view(torch.float64)[0]) == 0.0 and int(state.view(torch.int32)[4]) == 0
    want = torch.cat([ref_p[0].detach(), ref_p[1].detach()])
    assert torch.allclose(p.cpu(), want, atol=3e-6, rtol=0)


def test_update_inside_the_graph_equals_the_eager_update():
    """A step captured with norm + clip + AdamW as nodes of its graph (the default on one rank) walks the same
    trajectory as the same capture with the update left eager behind the graph, batch for batch; the optimizer's
    step count lives on the device and survives state_dict / load_state_dict; set_epoch reaches captured replays."""
    batches = [_batch(41, 4), _batch(42, 3), _batch(43, 5)]
    outs = []
    for in_graph in (True, False):
        # (lr 3e-4: at 1e-3 this tiny untrained network hits a loss spike (14 -> 111) at the fourth batch, where the
        #  fp32 atomics' order alone decides between two trajectories (111.57 / 118.81) in either update mode)
        tr, model = _trainer(lr=3e-4)
        replay = tr.capture(batches[0], warmup=1, max_gt=8, update_in_graph=in_graph)
        assert replay.update_in_graph == in_graph
        losses = []
        for k in range(6):
            if k:
                replay.load(batches[k % 3])
            if k == 3:
                tr.set_epoch(24)                       # lr x 0.1 from here on
            losses.append(float(replay(next_points=batches[(k + 1) % 3]["points"])))
        torch.cuda.synchronize()
        assert tr.opt.t == 6 + 1
        sd = tr.state_dict()
        assert sd["optimizer"]["t"] == 7 and sd["optimizer"]["lr_factor"] == pytest.approx(0.1)
        outs.append((losses, [p.detach().clone() for p in model.parameters()], sd))
    (la, pa, sda), (lb, pb, sdb) = outs
    assert la == pytest.approx(lb, rel=2e-3)
    rel = lambda x, y: ((x.double() - y.double()).norm() / x.double().norm().clamp_min(1e-30)).item()
    for x, y in zip(pa, pb):
        assert rel(x, y) <= 1e-3
    # resume: a fresh trainer loaded from the in-graph run continues with step 8's bias corrections
    tr2, _ = _trainer(lr=3e-4)
    tr2.load_state_dict(sda)
    assert tr2.opt.t == 7 and tr2.opt.lr_factor == pytest.approx(0.1)
    tr2.step(batches[0])
    assert tr2.opt.t == 8


def test_captured_update_refuses_a_changed_collective_setup():
    tr, _ = _trainer()
    b = _batch(44, 2)
    replay = tr.capture(b, warmup=1, max_gt=8)
    assert replay.update_in_graph
    replay()
    tr.allreduce_stub_us = 50
    with pytest.raises(RuntimeError, match="update inside the graph"):
        replay()


def test_captured_step_is_this_librarys_kernels():
    """The captured step is the WHOLE step (arena fill ... AdamW) and launches kernels of this library: no
    at::native reduce / multi-tensor node and no runtime memset: the arena fill, the gradient pack with the clip
    norm and the one-launch AdamW are kernels of the library inside the graph.  (What is left of the framework in
    the full-size step are the autograd engine's own gradient sums of tensors with several consumers and the
    position embedding's 6 -> 8 column weight pad: tools/op_sites.py lists them.)"""
    from torch.profiler import ProfilerActivity, profile
    tr, _ = _trainer()
    b = _batch(45, 3)
    replay = tr.capture(b, warmup=2, max_gt=8)
    assert replay.update_in_graph
    for _ in range(2):
        replay()
    torch.cuda.synchronize()
    p0 = tr.opt.flat.clone()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr._graph.replay()
        torch.cuda.synchronize()
    assert not torch.equal(p0, tr.opt.flat), "the graph alone must have updated the parameters"
    names = {str(e.key): int(e.count) for e in prof.key_averages()}
    kernels = {n: c for n, c in names.items() if "(" in n or n.startswith("__amd")}
    if not any("demf::" in n for n in kernels):
        pytest.skip("the profiler recorded no kernels of the graph replay on this runtime")
    joined = "\n".join(sorted(kernels))
    for must in ("demf::zero_f32_k", "demf::multi_copy_sumsq_k", "demf::adamw_state_k"):
        assert must in joined, (must, joined)
    for never in ("reduce_kernel", "multi_tensor_apply", "fillBuffer", "NormTwoOps"):
        assert never not in joined, (never, joined)


def test_double_buffered_load_waits_for_the_producer_stream():
    """DoubleBufferedStep.load runs on an input stream: it must see a batch that the CALLER's stream is still
    producing.  The batch's points are written by a long chain on the current stream right before load."""
    tr, _ = _trainer()
    a, b2 = _batch(46, 2), _batch(47, 2)
    step = tr.capture_double(a, b2, warmup=1, max_gt=8)
    step(next_points=b2["points"])
    src = _batch(48, 3)
    want = src["points"].clone()
    late = dict(src)
    buf = torch.zeros_like(want)
    from demf_amd import _ffi
    _ffi.call("demf_spin_us", 3000, torch.cuda.current_stream().cuda_stream)      # the producer is slow...
    buf.copy_(want)                                                               # ...and writes the cloud late
    late["points"] = buf
    step.load(late)
    torch.cuda.synchronize()
    assert torch.equal(step.static["points"], want)


def test_library_fallbacks_raise_unless_allowed():
    """The module-path MultiheadAttention (torch.bmm) and the library convolutions of the image stream refuse to run on
    the GPU unless DEMF_ALLOW_LIBRARY_FALLBACK=1 / ops.LIBRARY_FALLBACK is set."""
    from demf_amd import ops
    from demf_amd.modules.transformer import MultiheadAttention
    mha = MultiheadAttention(64, 4).cuda()
    q = torch.randn(5, 2, 64, device="cuda")
    prev = ops.LIBRARY_FALLBACK
    try:
        ops.LIBRARY_FALLBACK = False
        with pytest.raises(RuntimeError, match="library"):
            mha(q)
        ops.LIBRARY_FALLBACK = True
        assert mha(q).shape == q.shape
    finally:
        ops.LIBRARY_FALLBACK = prev


def test_replicated_accumulator_ring_wraps_cleanly():
    """The row reductions of the BatchNorm backward add into one of 8 copies of a zeroed block taken from a ring of 128
    (csrc/bn_fin.h accum_slot) and their last workgroup folds and clears the copies.  300 launches in a row - more than
    two laps of the ring - must each return the sums of THEIR launch: a block that was not left zeroed, or a fold that
    missed a copy, shows as the previous user's sums in a later result."""
    from demf_amd import _ffi
    torch.manual_seed(7)
    R, N = 4096, 128
    st = torch.cuda.current_stream().cuda_stream
    g12 = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
    gamma = (1.0 + 0.1 * torch.randn(N)).cuda()
    for it in range(300):
        y = torch.randn(R, N, device="cuda")
        dp = torch.randn(R, N, device="cuda") * (1.0 + it % 7)
        mean, var = y.mean(0), y.var(0, unbiased=False)
        invstd = torch.rsqrt(var + 1e-5)
        ss = torch.cat([gamma * invstd, 0.05 - mean * gamma * invstd]).contiguous()
        mi = torch.cat([mean, invstd]).contiguous()
        arg = torch.zeros(R, N, dtype=torch.int32, device="cuda")
        vec6 = torch.empty(5 * N, device="cuda")
        dgamma, dbeta = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        _ffi.call("demf_bn_bwd_reduce_vectors", R, N, 1, dp.data_ptr(), arg.data_ptr(), y.data_ptr(), y.data_ptr(),
                  ss.data_ptr(), mi.data_ptr(), g12.data_ptr(), gamma.data_ptr(), vec6.data_ptr(), dgamma.data_ptr(),
                  dbeta.data_ptr(), 0, st)
        if it % 37 == 0 or it >= 296:
            on = (y * ss[:N] + ss[N:]) > 0
            dz = torch.where(on, dp, torch.zeros_like(dp)).double()
            xhat = ((y - mean) * invstd).double()
            want_b, want_g = dz.sum(0), (dz * xhat).sum(0)
            assert torch.allclose(dbeta.double(), want_b, rtol=1e-5, atol=1e-3), it
            assert torch.allclose(dgamma.double(), want_g, rtol=1e-5, atol=1e-3), it
    torch.cuda.synchronize()
    assert float(g12.abs().max()) == 0.0


@pytest.mark.parametrize("R,K,N", [(8192, 256, 259), (4096, 128, 7), (1024, 256, 259)])
def test_linear_backward_with_an_unaligned_output_width(R, K, N):
    """ops.linear whose output width is not a multiple of 4 (the vote module's conv_out: 3 + 256 columns): from 4 096
    rows on, the backward pads the gradient and the weight to a 4-aligned width instead of staging them element by
    element; input, weight and bias gradients against fp64 autograd, with and without the padding."""
    from demf_amd import ops
    torch.manual_seed(R + N)
    x = torch.randn(R, K, dtype=torch.float64)
    w = torch.randn(N, K, dtype=torch.float64) / K ** 0.5
    b = torch.randn(N, dtype=torch.float64)
    go = torch.randn(R, N, dtype=torch.float64)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    torch.nn.functional.linear(xr, wr, br).backward(go)
    xg, wg, bg = (t.float().cuda().requires_grad_() for t in (x, w, b))
    y = ops.linear(xg, wg, bg)
    y.backward(go.float().cuda())
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    assert rel(y.detach(), torch.nn.functional.linear(x, w, b)) <= 2e-6
    assert rel(xg.grad, xr.grad) <= 2e-6 and rel(wg.grad, wr.grad) <= 2e-6 and rel(bg.grad, br.grad) <= 2e-6
    assert wg.grad.shape == (N, K) and bg.grad.shape == (N,)
