"""csrc/mlp_bwd.hip: the one-pass backward of a shared-MLP layer (dX + dW + the BN-backward sums of
the layer below from one staged dY tile; W resident in registers; operands split into bf16 planes
once) against (a) the two-launch path it replaces (demf_mlp_gemm_bwd_dx_red / _dx_first +
demf_mlp_gemm_bwd_dw) on identical inputs and (b) an fp64 torch reference of the op chain
(Conv1x1 -> train-mode BatchNorm -> ReLU, x L, max over ns: mmdet3d PointSAModule as the reference
builds it, configs/demf/demf_votenet.py:48-62), in the fp32-grade and the bf16 compute mode."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # (groups, ns, ld, channels, x_grad)  - the layers the fused kernel takes are noted
    (700, 64, 4, (64, 64, 128), False),       # SA1: L3 (128,64) sparse, L2 (64,64) FIRST
    (333, 64, 4, (64, 64, 128), False),       # ... ragged last slab (R % 64 != 0 is impossible with ns=64; 333*64 rows)
    (520, 32, 64, (128, 128, 128), True),     # L3 (128,128) sparse ns=32, L2 (128,128) dense
    (300, 16, 36, (128, 128, 256), True),     # SA3/SA4-like: L3 (256,128) sparse ns=16, L2 (128,128) dense
    (1030, 16, 128, (256, 256, 256), True),   # vote aggregation (256 -> 256 on >= 16384 rows): L3 sparse and L2 dense
                                              # per 128-column chunk (demf_mlp_bwd_fused_cols), L1 (256,128) dense
    (45, 16, 64, (128, 128, 128), True),      # vote-aggregation-like, few rows (720: ragged 32-row slabs)
    (1000, 4, 32, (64, 64, 64), True),        # (64,64) RED sparse ns=4 and dense
    (2000, 1, 64, (128, 128, 64), True),      # ns = 1 (unpooled tail, two-launch path for the last layer); L2 (128,128) dense
]


def _make(Rp, ns, ld, chans, seed):
    g = torch.Generator().manual_seed(seed)
    R = Rp * ns
    x = torch.randn(R, ld, generator=g, dtype=torch.float64) * 0.7 + 0.1
    layers, k = [], ld
    for n in chans:
        layers.append((torch.randn(n, k, generator=g, dtype=torch.float64) / np.sqrt(k),
                       1.0 + 0.2 * torch.randn(n, generator=g, dtype=torch.float64),
                       0.1 * torch.randn(n, generator=g, dtype=torch.float64)))
        k = n
    layers[1][1][3] = -0.6                     # a negative BN scale below the fused layer
    go = torch.randn(Rp, chans[-1], generator=g, dtype=torch.float64)
    return x, layers, go


def _run(x, layers, go, ns, xgrad):
    from demf_amd import ops
    xd = x.float().cuda().requires_grad_(xgrad)
    ls = []
    for W, gm, bt in layers:
        n = W.shape[0]
        ls.append((W.float().cuda().requires_grad_(), gm.float().cuda().requires_grad_(),
                   bt.float().cuda().requires_grad_(), torch.zeros(n, device="cuda"),
                   torch.ones(n, device="cuda")))
    out = ops.shared_mlp_pool(xd, ns, ls, True)
    out.backward(go.float().cuda())
    grads = [t.grad for l in ls for t in l[:3]]
    if xgrad:
        grads.append(xd.grad)
    return out.detach(), grads


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("Rp,ns,ld,chans,xgrad", CASES)
def test_fused_backward_equals_the_two_launch_path(Rp, ns, ld, chans, xgrad, mode):
    from demf_amd import ops
    x, layers, go = _make(Rp, ns, ld, chans, seed=Rp + ns)
    ops.set_compute_dtype(mode)
    calls = []
    from demf_amd import _ffi
    orig = _ffi.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    cols_min = ops._FUSED_COLS_MIN_R
    try:
        _ffi.call = spy
        ops._NO_BWD_FUSE = False
        # (the no-store pooled last layer re-associates the backward: in the bf16 mode it rounds A and
        # W^T diag(a) W where this comparison assumes dY and W - it has its own test below)
        ops._POOL_NOY = mode == "f32"
        ops._SA1_X4 = False                    # (rebuilds layer 0's output with an fma chain: not bit-equal to the MFMA's)
        ops._FUSED_COLS_MIN_R = 16384          # (opt-in: DEMF_FUSED_COLS_MIN_R, measured neutral on the step)
        out_f, g_f = _run(x, layers, go, ns, xgrad)
        n_fused = calls.count("demf_mlp_bwd_fused")
        n_cols = calls.count("demf_mlp_bwd_fused_cols")
        calls.clear()
        ops._NO_BWD_FUSE = True
        out_u, g_u = _run(x, layers, go, ns, xgrad)
        assert "demf_mlp_bwd_fused" not in calls and "demf_mlp_bwd_fused_cols" not in calls
    finally:
        _ffi.call = orig
        ops._NO_BWD_FUSE = False
        ops._POOL_NOY = True
        ops._SA1_X4 = True
        ops._FUSED_COLS_MIN_R = cols_min
        ops.set_compute_dtype("f32")
    assert n_fused + n_cols >= 1, "the case must exercise the fused kernel"
    assert n_cols == (4 if chans == (256, 256, 256) else 0)       # two layers x two column chunks
    assert torch.equal(out_f, out_u)
    # same operands, same roundings (bf16 mode rounds dY, act(Y_{l-1}) and W at the same places in both
    # paths); only the fp32 accumulation order differs
    tol = 2e-5 if mode == "f32" else 3e-3   # (bf16: a re-rounded operand moves by one bf16 ulp)
    for i, (a, b) in enumerate(zip(g_f, g_u)):
        scale = max(1e-6, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= tol * scale, (i, err, scale)


@pytest.mark.parametrize("Rp,ns,ld,chans,xgrad", CASES[:6])
def test_fused_backward_vs_fp64_reference(Rp, ns, ld, chans, xgrad):
    """fp32-grade mode against fp64 autograd of the same chain.  Gradients relative L2 (a single
    ReLU / max-pool near-tie resolving differently in fp32 moves whole tensors by ~1e-3, see
    tests/test_gpu_mlp.py, which does the on-branch comparison for every kernel variant)."""
    from demf_amd import ops
    x, layers, go = _make(Rp, ns, ld, chans, seed=Rp + ns)
    xr = x.clone().requires_grad_(xgrad)
    lr_ = [(W.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_())
           for W, g, b in layers]
    h = xr
    for W, g, b in lr_:
        h = F.relu(F.batch_norm(F.linear(h, W), None, None, g, b, True, 0.1, 1e-5))
    ref = h.view(Rp, ns, -1).max(1)[0]
    ref.backward(go)
    want = [t.grad for l in lr_ for t in l] + ([xr.grad] if xgrad else [])
    ops._NO_BWD_FUSE = False
    cols_min, ops._FUSED_COLS_MIN_R = ops._FUSED_COLS_MIN_R, 16384      # also the per-column-chunk form
    try:
        out, got = _run(x, layers, go, ns, xgrad)
    finally:
        ops._FUSED_COLS_MIN_R = cols_min
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4 * max(1.0, float(ref.detach().abs().max()))
    for i, (a, b) in enumerate(zip(got, want)):
        rel = float((a.double().cpu() - b).norm() / b.norm())
        assert rel <= 5e-3, (i, rel)


@pytest.mark.parametrize("case", [0, 2, 3, 4])
def test_two_fp16_terms_against_three_bf16_terms(case):
    """The default f32 mode runs the SA stacks' forward and fused backward kernels on TWO fp16 terms per operand
    (three products, the gradient operand scaled by a power of two per slab: csrc/common.h, csrc/mlp_bwd.hip) where
    "f32x3" runs three bf16 terms (six products).  Same inputs, both modes: outputs within 2e-6 of scale, every
    gradient within 2e-5 relative L2 (measured 3e-7 ... 1.5e-6: the size of fp32's own accumulation noise), and
    both equally far from fp64 autograd.  Gradient magnitudes of 1e-7 (a loss averaged over many rows) must not
    matter: the upstream gradient is also run scaled by 2^-20."""
    from demf_amd import ops
    Rp, ns, ld, chans, xgrad = CASES[case]
    x, layers, go = _make(Rp, ns, ld, chans, seed=4000 + case)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    try:
        for gscale in (1.0, 2.0 ** -20):
            ops.set_compute_dtype("f32x3")
            out3, g3 = _run(x, layers, go * gscale, ns, xgrad)
            ops.set_compute_dtype("f32h2")
            out2, g2 = _run(x, layers, go * gscale, ns, xgrad)
            scale = float(out3.abs().max())
            assert float((out2 - out3).abs().max()) <= 2e-6 * scale
            worst = max(rel(a, b) for a, b in zip(g2, g3))
            print("case %d, upstream gradient x %g: worst gradient rel-L2 two fp16 terms vs three bf16 terms %.2e"
                  % (case, gscale, worst))
            assert worst <= 2e-5, worst
            assert all(bool(torch.isfinite(g).all()) for g in g2)
    finally:
        ops.set_compute_dtype("f32")


def test_bf16_row_storage_of_the_sa1_stack():
    """bf16 compute mode (BASELINE configs[3]): an SA1-shaped stack keeps the raw outputs of layers 1 / 2
    and the gradient between their backwards as bf16 ROWS in HBM (demf_mlp_gemm_fwd_bn_st /
    _pool_bn_st, demf_mlp_bwd_fused store_flags) - against the same stack with fp32 rows: the only
    difference is one extra bf16 rounding of those rows (4e-3 relative per element)."""
    from demf_amd import _ffi, ops
    Rp, ns, ld, chans = 512, 64, 4, (64, 64, 128)
    x, layers, go = _make(Rp, ns, ld, chans, seed=77)
    layers[2][1][5] = -0.8          # pooled layer: a negative scale (min selected) ...
    layers[2][1][9] = 0.0           # ... and a zero scale (the sparse reduce gathers the row from bf16 Y)
    ops.set_compute_dtype("bf16")
    calls = []
    orig = _ffi.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    no_store_default = ops._NO_BF16_STORE
    try:
        _ffi.call = spy
        ops._NO_BF16_STORE = False
        out_s, g_s = _run(x, layers, go, ns, False)
        st_calls = [c for c in calls if c.endswith("_st")]
        calls.clear()
        ops._NO_BF16_STORE = True
        ops._POOL_NOY = False            # fp32 rows, stored: the form the bf16 rows differ from by one rounding
        out_f, g_f = _run(x, layers, go, ns, False)
        assert not [c for c in calls if c.endswith("_st")]
    finally:
        _ffi.call = orig
        ops._NO_BF16_STORE = no_store_default
        ops._POOL_NOY = True
        ops.set_compute_dtype("f32")
    assert sorted(st_calls) == ["demf_mlp_gemm_fwd_bn_st", "demf_mlp_gemm_fwd_pool_bn_st"], st_calls
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(out_s, out_f) <= 2e-2, rel(out_s, out_f)
    worst = max(rel(a, b) for a, b in zip(g_s, g_f))
    print("bf16 row storage vs fp32 rows (bf16 compute): output rel-L2 %.2e, worst gradient rel-L2 %.2e"
          % (rel(out_s, out_f), worst))
    assert worst <= 0.15, worst
    assert all(torch.isfinite(g).all() for g in g_s)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("Rp", [256, 700])
def test_pooled_last_layer_without_its_output(Rp, mode):
    """SA1's last layer (R x 64 -> 128, BN, ReLU, max over 64 rows) WITHOUT its (R x 128) output: the
    forward does not store it (demf_mlp_gemm_fwd_pool_bn_st, store_flags 4), the backward
    (demf_mlp_bwd_pool) is  dA = (gi dZ).W + A.(W^T diag(a) W) + b^T W,  dW = (gi dZ)^T.A + diag(a) W (A^T A)
    + b (x) colsum(A) - algebraically the layer's backward, re-associated.  Against the stored-output path
    (same forward arithmetic: identical outputs; gradients to fp32 re-association) and, in the fp32-grade
    mode, against fp64 autograd.  Pooled-layer scales: positive, negative (minimum selected) and ZERO
    (constant channel: slot 0 is the selected row and its raw value feeds dgamma)."""
    from demf_amd import _ffi, ops
    ns, ld, chans = 64, 4, (64, 64, 128)
    x, layers, go = _make(Rp, ns, ld, chans, seed=900 + Rp)
    layers[2][1][5] = -0.8
    layers[2][1][9] = 0.0
    layers[2][1][77] = 0.0
    # (the forms are compared with each other: one arithmetic for all of them - in the default f32 mode the kernels that
    #  have a two-fp16-term form and those that do not differ in the last bits, and a ReLU / max-pool near-tie may flip)
    ops.set_compute_dtype("f32x3" if mode == "f32" else mode)
    calls = []
    orig = _ffi.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    try:
        _ffi.call = spy
        ops._POOL_NOY, ops._SA1_X4 = True, False
        out_n, g_n = _run(x, layers, go, ns, False)
        assert calls.count("demf_mlp_bwd_pool") == 1 and "demf_pool_select_slot0" in calls
        calls.clear()
        ops._POOL_NOY = False
        out_s, g_s = _run(x, layers, go, ns, False)
        assert "demf_mlp_bwd_pool" not in calls
        # + the first layer without ITS output (statistics from the rows' moments, consumers rebuild it)
        calls.clear()
        ops._POOL_NOY, ops._SA1_X4 = True, True
        out_x, g_x = _run(x, layers, go, ns, False)
        assert {"demf_mlp_first_stats", "demf_mlp_gemm_fwd_bn_x4", "demf_mlp_bwd_fused_x4"} <= set(calls)
    finally:
        _ffi.call = orig
        ops._POOL_NOY, ops._SA1_X4 = True, True
        ops.set_compute_dtype("f32")
    assert torch.equal(out_n, out_s)
    relx = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    tolx = 2e-5 if mode == "f32" else 2e-2
    assert relx(out_x, out_s) <= tolx, relx(out_x, out_s)
    worst_x = max(relx(a, b) for a, b in zip(g_x, g_s))
    print("first layer without its output vs stored (%s): output rel-L2 %.2e, worst gradient rel-L2 %.2e"
          % (mode, relx(out_x, out_s), worst_x))
    assert worst_x <= (2e-4 if mode == "f32" else 3e-2), worst_x
    names = [f"layer{l}.{n}" for l in range(3) for n in ("W", "gamma", "beta")]
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    tol = 1e-4 if mode == "f32" else 2e-2      # bf16: M = W^T diag(a) W and A are rounded where dY and W were
    worst = max((rel(a, b), n) for n, a, b in zip(names, g_n, g_s))
    print("no-store pooled layer vs stored (%s): worst gradient rel-L2 %.2e at %s" % (mode, *worst))
    for n, a, b in zip(names, g_n, g_s):
        assert rel(a, b) <= tol, (n, rel(a, b))
    if mode == "f32":
        xr = x.clone()
        lr_ = [(W.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_())
               for W, g, b in layers]
        h = xr
        for W, g, b in lr_:
            h = F.relu(F.batch_norm(F.linear(h, W), None, None, g, b, True, 0.1, 1e-5))
        ref = h.view(Rp, ns, -1).max(1)[0]
        ref.backward(go)
        want = [t.grad for l in lr_ for t in l]
        assert float((out_n.double().cpu() - ref.detach()).abs().max()) <= 1e-4 * max(1.0, float(ref.detach().abs().max()))
        for n, a, b in zip(names, g_n, want):
            assert rel(a.cpu(), b) <= 5e-3, (n, rel(a.cpu(), b))
        for n, a, b in zip(names, g_x, want):
            assert rel(a.cpu(), b) <= 5e-3, ("x4 " + n, rel(a.cpu(), b))
