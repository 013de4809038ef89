"""Operator layer: the Python API the reference imports from mmdet3d.ops / mmcv.ops,
re-implemented on libdemf_hip.so (gfx950 HIP kernels).

Signatures, argument meaning and error behaviour mirror the upstream operators the
reference calls (cited per function); tensors stay PyTorch-ROCm tensors and only
raw device pointers + the current HIP stream cross the C ABI.  There is no CPU or
eager fallback: a CPU tensor or a missing library raises.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _ffi

__all__ = [
    "set_compute_dtype", "get_compute_dtype", "gt_prep", "target_weights", "vote_combine", "furthest_point_sample", "ball_query", "grouping_operation", "gather_points",
    "three_nn", "three_nn_weights", "three_interpolate", "three_interpolate_cat_cl", "MultiScaleDeformableAttnFunction",
    "group_concat_cl", "gather_rows_cl", "three_interpolate_cl", "maxpool_ns", "shared_mlp_pool",
]


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Library (rocBLAS / MIOpen) fall-backs of the MODULE paths - torch.bmm in the seq-first MultiheadAttention module,
# F.conv2d for image-stream widths the kernels of csrc/conv.hip do not take - are off the captured hot path (the
# profile shows no library kernel) and must not be taken silently: they raise unless this is set
# (DEMF_ALLOW_LIBRARY_FALLBACK=1, or ops.LIBRARY_FALLBACK = True in a test).
LIBRARY_FALLBACK = bool(int(os.environ.get("DEMF_ALLOW_LIBRARY_FALLBACK", "0") or 0))


def library_fallback(what):
    """Call where a GPU module path is about to use a library GEMM / convolution instead of this package's kernels."""
    if not LIBRARY_FALLBACK:
        raise RuntimeError(what + ": this path would run a library (rocBLAS / MIOpen) kernel instead of the kernels of "
                           "libdemf_hip.so; set DEMF_ALLOW_LIBRARY_FALLBACK=1 to allow it")


class _DeferredDW:
    """Weight-gradient products of the few-row stacks, queued and issued together.  Nothing in a backward depends
    on a layer's dW (only the optimizer does), but one at a time each is a node of the step's dependency chain:
    15-20 launches of ~20 us whose ramps and tails do not overlap.  Inside ``deferred_weight_grads()`` the
    callers queue their job (pointers + sizes of demf_mlp_gemm_bwd_dw_ld) and keep its operands referenced; the
    context's exit issues everything as demf_mlp_gemm_bwd_dw_group on the current stream: one launch per kernel
    variant.  Outside the context a job is launched at once."""

    def __init__(self):
        self.on, self.jobs, self.keep, self.leaves = False, [], [], set()


DEFER = _DeferredDW()
_DEFER_DW = int(os.environ.get("DEMF_DEFER_DW", "1") or 0)


def _plain_parameter(w):
    """True if ``w`` is a leaf or a contiguous view of one (conv.weight.view(N, K)): its gradient reaches the
    parameter through alias-only backward nodes."""
    try:
        if w.is_leaf:
            return True
        return bool(w._is_view() and w._base is not None and w._base.is_leaf and w.is_contiguous() and
                    w._base.is_contiguous() and w.numel() == w._base.numel())
    except Exception:          # noqa: BLE001 - unknown tensor kinds are simply not deferred
        return False


def _leaf_key(w):
    """Identity of the parameter a weight operand aliases (see _plain_parameter), or None."""
    try:
        return id(w if w.is_leaf else w._base)
    except Exception:          # noqa: BLE001
        return None


def dw_job(R, N, K, ldx, G, dP, arg, ns, Y, vec6, xprev, pss, dW, lddw, dw_off=0, defer=True, leaf=None):
    """One dW = dZ^T A product (arguments of demf_mlp_gemm_bwd_dw_ld; tensors, ``dw_off`` floats into dW).
    ``leaf``: _leaf_key of the parameter this gradient belongs to.  A parameter consumed by TWO nodes (a shared
    MLP called twice, tied heads) has its two gradients summed by the autograd engine as soon as the second node
    returns - both products must have been issued by then, so a repeated leaf flushes the queue and runs at once."""
    j = _ffi.DwJob(R, N, K, ldx, _p(G), _p(dP), _p(arg), ns, _p(Y), _p(vec6), _p(xprev), _p(pss),
                   dW.data_ptr() + 4 * dw_off, lddw)
    if DEFER.on and _DEFER_DW and defer and leaf is not None and leaf in DEFER.leaves:
        flush_dw()
        defer = False
    if DEFER.on and _DEFER_DW and defer:
        if leaf is not None:
            DEFER.leaves.add(leaf)
        DEFER.jobs.append(j)
        DEFER.keep.extend(t for t in (G, dP, arg, Y, vec6, xprev, pss, dW) if t is not None)
        return
    _ffi.call("demf_mlp_gemm_bwd_dw_group", 1, ctypes.addressof(j), _stream())


def flush_dw():
    if DEFER.jobs:
        arr = (_ffi.DwJob * len(DEFER.jobs))(*DEFER.jobs)
        DEFER.jobs = []
        try:
            _ffi.call("demf_mlp_gemm_bwd_dw_group", len(arr), ctypes.addressof(arr), _stream())
        finally:
            DEFER.keep.clear()


class deferred_weight_grads:
    """Re-entrant: a nested context joins the outer one's queue (flushed at the OUTERMOST exit)."""

    def __enter__(self):
        self._prev = DEFER.on
        DEFER.on = True
        return self

    def __exit__(self, *exc):
        DEFER.on = self._prev
        if self._prev:
            return False
        if exc[0] is None:
            flush_dw()
        else:
            DEFER.jobs, DEFER.keep = [], []
        DEFER.leaves.clear()
        return False



class MultiCopy:
    """dst[i].copy_(src[i]) (src[i] None: dst[i].zero_()) for a FIXED list of contiguous tensor pairs
    as one kernel launch (demf_multi_copy); the address table is built once."""

    def __init__(self, dst, src):
        assert len(dst) == len(src) and len(dst) > 0
        tab = [[], [], []]
        for d, s in zip(dst, src):
            assert d.is_contiguous() and d.is_cuda and (d.numel() * d.element_size()) % 4 == 0
            if s is not None:
                assert s.is_contiguous() and s.dtype == d.dtype and s.numel() == d.numel()
            tab[0].append(0 if s is None else s.data_ptr())
            tab[1].append(d.data_ptr())
            tab[2].append(d.numel() * d.element_size() // 4)
        self.keep = (list(dst), list(src))             # the table holds raw addresses
        self.n = len(dst)
        self.table = torch.tensor(tab, dtype=torch.int64, device=dst[0].device)
        self.blocks = max(1, min(64, max(tab[2]) // 4096))

    def __call__(self):
        _ffi.call("demf_multi_copy", self.n, self.table.data_ptr(), self.blocks, _stream())


class _ZeroArena:
    """Zero-filled fp32 scratch for ONE training step, handed out by a bump pointer and re-zeroed
    by a single fill at the start of the next step - instead of one memset launch per accumulated
    output (weight-gradient workspaces of every fused stack, scatter targets of the gather /
    interpolation backwards, loss accumulators: ~30 per step).  Only active between ``begin`` and
    ``end`` (engine.Trainer brackets forward + backward with it); outside, ``zeros`` is torch.zeros."""

    def __init__(self):
        self.buf, self.off, self.high, self.active = None, 0, 0, False
        self._captured = []       # buffers whose address is baked into a captured hipGraph: kept alive

    def begin(self, device):
        capturing = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
        if self.buf is None or self.buf.device != device:
            if capturing:
                raise RuntimeError("zero arena must be sized by an eager warm-up step before capture")
            self.buf = torch.zeros(1 << 22, dtype=torch.float32, device=device)
        elif capturing:
            # the recorded fill covers the WHOLE buffer: a piece first taken while capturing (beyond
            # the warm-up steps' watermark) is re-zeroed on every replay too
            _ffi.call("demf_zero_f32", self.buf.numel(), self.buf.data_ptr(), _stream())
        elif self.high:
            _ffi.call("demf_zero_f32", self.high, self.buf.data_ptr(), _stream())
        if capturing and not any(b is self.buf for b in self._captured):
            # the graph holds this buffer's address (its fill node, every weight-gradient workspace and
            # loss accumulator): it must outlive any later growth of the arena
            self._captured.append(self.buf)
        self.off, self.active = 0, True

    def end(self):
        self.active = False

    def take(self, numel):
        n = (numel + 63) // 64 * 64                   # 256-byte aligned pieces
        if self.off + n > self.buf.numel():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("zero arena overflow during capture")
            # grow: pieces already handed out stay valid views of the old buffer (still zero-filled);
            # a buffer that a captured graph uses stays referenced in ``_captured`` and keeps serving
            # that graph's replays, eager steps move on to the new one
            self.buf = torch.zeros(max(2 * self.buf.numel(), self.off + n), dtype=torch.float32,
                                   device=self.buf.device)
        out = self.buf[self.off:self.off + numel]
        self.off += n
        self.high = max(self.high, self.off)
        return out


ARENA = _ZeroArena()


def zeros(shape, device):
    """fp32 zeros: a piece of the step's zero arena when one is open, else torch.zeros."""
    if isinstance(shape, int):
        shape = (shape,)
    if ARENA.active and ARENA.buf is not None and ARENA.buf.device == torch.device(device):
        n = 1
        for d in shape:
            n *= int(d)
        return ARENA.take(n).view(*shape)
    return torch.zeros(shape, dtype=torch.float32, device=device)


_COMPUTE_DTYPE = "f32"


_COMPUTE_MODES = {"f32_native": 0, "bf16": 1, "f32x3": 2, "f32h2": 2}
# the library's mode (demf_set_compute_dtype); its initial value follows the same environment switch
_COMPUTE_MODE = 0 if int(os.environ.get("DEMF_F32_NATIVE", "0") or 0) else 2


def set_compute_dtype(name):
    """"f32" (default), "f32_native", "f32x3" or "bf16" (BASELINE.json configs[3]).
    f32x3: the shared-MLP GEMMs (forward / input-gradient) split every fp32 operand EXACTLY into
    three bf16 terms and accumulate the six significant products in fp32 on the bf16 MFMA, which
    gfx950 runs 16x faster than its fp32 MFMA: fp32-grade results (error against fp64 measured equal
    to the fp32 MFMA's, tests/test_gpu_split.py) at 3/8 of the issue time.  f32_native: the same
    kernels on v_mfma_f32_32x32x2_f32.  "f32" = f32x3, or f32_native with DEMF_F32_NATIVE=1.
    bf16: the dense MFMA kernels - shared-MLP GEMMs forward / input-gradient / weight-gradient, the
    decoder layer's GEMMs, the linear heads - round their operands to bf16 on the way into LDS and
    run v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
    f32h2 (what "f32" selects unless DEMF_F16_TERMS=0): as f32x3, but the kernels that have the form (the SA
    stacks' forward and fused backward kernels) take every operand as TWO fp16 terms and three products - half the
    matrix work at ~2^-22 relative, the size of fp32's own accumulation noise (csrc/common.h); "f32x3" switches
    that form off (pure three-term arithmetic: the mode in which two kernel forms of one layer agree bit for bit).
    Tensors in memory, BN statistics, indices, sampling and losses stay fp32 in every mode.
    Process-wide (demf_set_compute_dtype, demf_set_f16_terms)."""
    global _COMPUTE_DTYPE
    if name == "f32":
        mode = 0 if int(os.environ.get("DEMF_F32_NATIVE", "0") or 0) else 2
    elif name in _COMPUTE_MODES:
        mode = _COMPUTE_MODES[name]
    else:
        raise ValueError("compute dtype must be 'f32' or one of %s" % sorted(_COMPUTE_MODES))
    _ffi.call("demf_set_compute_dtype", mode)
    if mode == 2:
        h2 = name == "f32h2" or (name == "f32" and int(os.environ.get("DEMF_F16_TERMS", "1") or 0) != 0)
        _ffi.call("demf_set_f16_terms", int(h2))
    global _COMPUTE_MODE
    _COMPUTE_DTYPE, _COMPUTE_MODE = name, mode


def get_compute_dtype():
    return _COMPUTE_DTYPE


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must be a GPU (HIP) tensor: demf_amd operators have no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")  # upstream: assert x.is_contiguous()
    return t


def _p(t):
    return t.data_ptr() if t is not None else None


# --------------------------------------------------------------------------
# PointNet++ operators (mmdet3d.ops API)
# --------------------------------------------------------------------------
class _FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, points_xyz, num_points):
        _chk(points_xyz, "points_xyz")
        B, N, three = points_xyz.shape
        assert three == 3
        idx = torch.empty((B, num_points), dtype=torch.int32, device=points_xyz.device)
        # the register-resident kernel covers 64 <= N <= 24576; outside it the library
        # needs the (B,N) running-distance scratch the upstream ABI always carries
        temp = None
        if N < 64 or N > 24 * 1024 or (N >= 4096 and num_points >= 64 and not (N <= 4 * num_points)):
            # (large clouds: the scratch holds the Hilbert-cell order of the exact box-pruned kernel)
            temp = torch.empty((B, N), dtype=torch.float32, device=points_xyz.device)
        elif 2 <= num_points <= 1024 and N <= 4 * num_points and not _NO_FPS_CHECK:
            # scratch of the ordered-input check (every SA level after the first samples a cloud that already
            # is in FPS order): B flags + B tickets + the B x M nearest-earlier-sample distances of its
            # chip-wide form (demf_fps_ws_f32)
            temp = torch.empty((B, num_points + 2), dtype=torch.float32, device=points_xyz.device)
        _ffi.call("demf_fps_ws_f32", B, N, num_points, _p(points_xyz), _p(temp),
                  0 if temp is None else temp.numel(), _p(idx), _stream())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g=None):
        return None, None


def furthest_point_sample(points_xyz, num_points):
    """(B,N,3) f32 -> (B,num_points) i32.  Reference use:
    demf/modeling/heads/class_agnostic_vote_head.py:13,429-430."""
    return _FurthestPointSampling.apply(points_xyz, num_points)


_BQ_GRID_MIN_N = int(os.environ.get("DEMF_BQ_GRID_MIN_N", "8192"))     # smaller clouds: the all-pairs scan


class _BallQuery(Function):
    @staticmethod
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        _chk(xyz, "xyz")
        _chk(center_xyz, "center_xyz")
        assert min_radius < max_radius
        B, N, _ = xyz.shape
        M = center_xyz.shape[1]
        idx = torch.empty((B, M, sample_num), dtype=torch.int32, device=xyz.device)
        if min_radius == 0 and _BQ_GRID_MIN_N <= N <= 32768 and M > 0 and B > 0:
            # large cloud (SA1): hashed-grid search, same hits in the same order (csrc/ball_query.hip)
            n_start, n_cells = ctypes.c_longlong(), ctypes.c_longlong()
            _ffi.call("demf_ball_query_grid_ws", B, N, ctypes.addressof(n_start), ctypes.addressof(n_cells))
            ws_start = torch.empty(n_start.value, dtype=torch.int32, device=xyz.device)
            ws_cells = torch.empty(n_cells.value, dtype=torch.float32, device=xyz.device)
            _ffi.call("demf_ball_query_grid_f32", B, N, M, float(max_radius), int(sample_num),
                      _p(center_xyz), _p(xyz), _p(idx), _p(ws_start), _p(ws_cells), _stream())
            ctx.mark_non_differentiable(idx)
            return idx
        _ffi.call("demf_ball_query_f32", B, N, M, float(min_radius), float(max_radius),
                  int(sample_num), _p(center_xyz), _p(xyz), _p(idx), _stream())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g=None):
        return None, None, None, None, None


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    """-> idx (B,M,sample_num) i32.  Reference use: QueryAndGroup inside
    build_sa_module (class_agnostic_vote_head.py:383,455)."""
    return _BallQuery.apply(min_radius, max_radius, sample_num, xyz, center_xyz)


class _GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, indices):
        _chk(features, "features")
        _chk(indices, "indices", torch.int32)
        B, C, N = features.shape
        _, M, ns = indices.shape
        out = torch.empty((B, C, M, ns), dtype=features.dtype, device=features.device)
        _ffi.call("demf_group_points_fwd", B, C, N, M, ns, _p(features), _p(indices), _p(out),
                  _stream())
        ctx.save_for_backward(indices)
        ctx.N = N
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (indices,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, M, ns = grad_out.shape
        grad = torch.zeros((B, C, ctx.N), dtype=grad_out.dtype, device=grad_out.device)
        _ffi.call("demf_group_points_bwd", B, C, ctx.N, M, ns, _p(grad_out), _p(indices),
                  _p(grad), _stream())
        return grad, None


def grouping_operation(features, indices):
    """features (B,C,N), indices (B,M,ns) i32 -> (B,C,M,ns); differentiable wrt features."""
    return _GroupingOperation.apply(features, indices)


class _GatherPoints(Function):
    @staticmethod
    def forward(ctx, features, indices):
        _chk(features, "features")
        _chk(indices, "indices", torch.int32)
        B, C, N = features.shape
        M = indices.shape[1]
        out = torch.empty((B, C, M), dtype=features.dtype, device=features.device)
        _ffi.call("demf_gather_points_fwd", B, C, N, M, _p(features), _p(indices), _p(out),
                  _stream())
        ctx.save_for_backward(indices)
        ctx.N = N
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (indices,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, M = grad_out.shape
        grad = torch.zeros((B, C, ctx.N), dtype=grad_out.dtype, device=grad_out.device)
        _ffi.call("demf_gather_points_bwd", B, C, ctx.N, M, _p(grad_out), _p(indices), _p(grad),
                  _stream())
        return grad, None


def gather_points(features, indices):
    """features (B,C,N), indices (B,M) i32 -> (B,C,M); differentiable wrt features."""
    return _GatherPoints.apply(features, indices)


class _ThreeNN(Function):
    @staticmethod
    def forward(ctx, target, source):
        _chk(target, "target")
        _chk(source, "source")
        B, n, _ = target.shape
        m = source.shape[1]
        dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=target.device)
        idx = torch.empty((B, n, 3), dtype=torch.int32, device=target.device)
        _ffi.call("demf_three_nn_f32", B, n, m, _p(target), _p(source), _p(dist2), _p(idx),
                  _stream())
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


def three_nn(target, source):
    """-> (dist (B,n,3) [sqrt applied], idx (B,n,3) i32).  PointFPModule."""
    return _ThreeNN.apply(target, source)


@torch.no_grad()
def three_nn_weights(target, source):
    """-> (idx (B,n,3) i32, weight (B,n,3)): the 3-NN search and the inverse-distance weights
    PointFPModule.forward derives from it (1/(dist+1e-8), normalised over the three), one launch.
    Coordinates only: no gradient (the reference's three_nn is not differentiable either)."""
    _chk(target, "target")
    _chk(source, "source")
    B, n, _ = target.shape
    m = source.shape[1]
    dist = torch.empty((B, n, 3), dtype=torch.float32, device=target.device)
    weight = torch.empty((B, n, 3), dtype=torch.float32, device=target.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=target.device)
    _ffi.call("demf_three_nn_weights_f32", B, n, m, _p(target), _p(source), _p(dist), _p(idx),
              _p(weight), _stream())
    return idx, weight


class _ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, indices, weight):
        _chk(features, "features")
        _chk(indices, "indices", torch.int32)
        _chk(weight, "weight")
        B, C, m = features.shape
        n = indices.shape[1]
        out = torch.empty((B, C, n), dtype=features.dtype, device=features.device)
        _ffi.call("demf_three_interpolate_fwd", B, C, m, n, _p(features), _p(indices),
                  _p(weight), _p(out), _stream())
        ctx.save_for_backward(indices, weight)
        ctx.m = m
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        indices, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, n = grad_out.shape
        grad = torch.zeros((B, C, ctx.m), dtype=grad_out.dtype, device=grad_out.device)
        _ffi.call("demf_three_interpolate_bwd", B, C, n, ctx.m, _p(grad_out), _p(indices),
                  _p(weight), _p(grad), _stream())
        return grad, None, None


def three_interpolate(features, indices, weight):
    """features (B,C,m), indices/weight (B,n,3) -> (B,C,n); differentiable wrt features."""
    return _ThreeInterpolate.apply(features, indices, weight)


# --------------------------------------------------------------------------
# Multi-scale deformable attention (mmcv.ops API)
# --------------------------------------------------------------------------
class MultiScaleDeformableAttnFunction(Function):
    """Same call contract as mmcv.ops.multi_scale_deform_attn.
    MultiScaleDeformableAttnFunction (used by the fusion cross-attention reached from
    demf/modeling/layers/transformer.py:73): ``apply(value, value_spatial_shapes,
    value_level_start_index, sampling_locations, attention_weights, im2col_step)``.
    ``im2col_step`` only chunks the batch upstream; it does not change results and is
    accepted and ignored here (one launch covers the batch)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step=64):
        _chk(value, "value")
        _chk(value_spatial_shapes, "value_spatial_shapes", torch.int64)
        _chk(value_level_start_index, "value_level_start_index", torch.int64)
        _chk(sampling_locations, "sampling_locations")
        _chk(attention_weights, "attention_weights")
        B, S, H, Dh = value.shape
        _, Q, _, L, P, _ = sampling_locations.shape
        out = torch.empty((B, Q, H * Dh), dtype=value.dtype, device=value.device)
        _ffi.call("demf_msda_fwd_f32", B, S, H, Dh, L, Q, P, _p(value),
                  _p(value_spatial_shapes), _p(value_level_start_index),
                  _p(sampling_locations), _p(attention_weights), _p(out), _stream())
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attw = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        B, S, H, Dh = value.shape
        _, Q, _, L, P, _ = loc.shape
        # no gradient wrt value wanted (e.g. frozen image tokens): the scatter is skipped
        grad_value = torch.zeros_like(value) if ctx.needs_input_grad[0] else None
        grad_loc = torch.empty_like(loc)
        grad_attw = torch.empty_like(attw)
        _ffi.call("demf_msda_bwd_f32", B, S, H, Dh, L, Q, P, _p(value), _p(shapes), _p(lsi),
                  _p(loc), _p(attw), _p(grad_output), _p(grad_value), _p(grad_loc),
                  _p(grad_attw), _stream())
        return grad_value, None, None, grad_loc, grad_attw, None


# --------------------------------------------------------------------------
# Point-major ("channels-last") fused variants used by demf_amd.modules
# --------------------------------------------------------------------------
class _GroupConcatCL(Function):
    @staticmethod
    def forward(ctx, xyz, center, feat, idx, radius, normalize_xyz, ldo, xyz_col, feat_col,
                inv_off, inv_rows):
        _chk(xyz, "xyz")
        _chk(center, "center")
        _chk(idx, "idx", torch.int32)
        B, N, _ = xyz.shape
        _, M, ns = idx.shape
        C = 0
        if feat is not None:
            _chk(feat, "feat")
            C = feat.shape[2]
        out = torch.empty((B, M, ns, ldo), dtype=xyz.dtype, device=xyz.device)
        _ffi.call("demf_group_concat_cl_fwd", B, N, M, ns, C, ldo, xyz_col, feat_col,
                  float(radius), int(bool(normalize_xyz)), _p(xyz), _p(center), _p(feat),
                  _p(idx), _p(out), _stream())
        gather = inv_off is not None and C >= 4 and C % 4 == 0 and ldo % 4 == 0 and feat_col % 4 == 0
        if gather:
            _chk(inv_off, "inv_off", torch.int32)
            _chk(inv_rows, "inv_rows", torch.int32)
            assert inv_off.shape == (B, N + 1) and inv_rows.shape == (B, M * ns)
            ctx.save_for_backward(idx, inv_off, inv_rows)
        else:
            ctx.save_for_backward(idx)
        ctx.gather = gather
        ctx.dims = (B, N, M, ns, C, ldo, xyz_col, feat_col, float(radius),
                    int(bool(normalize_xyz)))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx = ctx.saved_tensors[0]
        B, N, M, ns, C, ldo, xyz_col, feat_col, radius, norm = ctx.dims
        want_feat = C > 0 and ctx.needs_input_grad[2]
        want_xyz = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gfeat = gxyz = gcenter = None
        if want_feat or want_xyz:
            grad_out = grad_out.contiguous()
            kw = dict(dtype=grad_out.dtype, device=grad_out.device)
            if want_feat and ctx.gather:
                # atomic-free: every source point sums the rows that gathered it
                _, inv_off, inv_rows = ctx.saved_tensors
                gfeat = torch.empty((B, N, C), **kw)
                _ffi.call("demf_group_concat_cl_bwd_gather", B, N, M * ns, C, ldo, feat_col,
                          _p(grad_out), _p(inv_off), _p(inv_rows), _p(gfeat), _stream())
            elif want_feat:
                gfeat = torch.zeros((B, N, C), **kw)
            if want_xyz:
                gxyz = torch.zeros((B, N, 3), **kw)
                gcenter = torch.zeros((B, M, 3), **kw)
            scatter_feat = gfeat if not ctx.gather else None
            if scatter_feat is not None or want_xyz:
                _ffi.call("demf_group_concat_cl_bwd", B, N, M, ns, C, ldo, xyz_col, feat_col,
                          radius, norm, _p(grad_out), _p(idx), _p(scatter_feat), _p(gxyz),
                          _p(gcenter), _stream())
        return (gxyz if ctx.needs_input_grad[0] else None,
                gcenter if ctx.needs_input_grad[1] else None,
                gfeat, None, None, None, None, None, None, None, None)


def invert_index(idx, num_source):
    """Inverse neighbour lists of idx (B,M,ns) int32 over ``num_source`` points:
    (off (B,N+1), rows (B,M*ns)) - see demf_invert_index."""
    _chk(idx, "idx", torch.int32)
    B = idx.shape[0]
    E = idx[0].numel()
    off = torch.empty((B, num_source + 1), dtype=torch.int32, device=idx.device)
    rows = torch.empty((B, E), dtype=torch.int32, device=idx.device)
    # (long entry lists - SA1 - are inverted chip-wide: that form needs B * E ints of scratch)
    ws = torch.empty((B, E), dtype=torch.int32, device=idx.device) if E >= 32768 else None
    _ffi.call("demf_invert_index_ws", B, num_source, E, _p(idx), _p(off), _p(rows), _p(ws), _stream())
    return off, rows


def group_concat_cl(xyz, center, feat, idx, radius, normalize_xyz, ldo=None, xyz_col=None,
                    feat_col=0, inverse=None):
    """Fused QueryAndGroup on point-major features.
    xyz (B,N,3), center (B,M,3), feat (B,N,C)|None, idx (B,M,ns) -> (B,M,ns,ldo) rows
    ``[feat | (xyz_j - center)/radius | 0-pad]`` (column order chosen by the caller).
    ``inverse`` = invert_index(idx, N): the feature gradient is then gathered instead of
    scattered with atomics."""
    C = 0 if feat is None else feat.shape[2]
    if xyz_col is None:
        xyz_col = feat_col + C
    if ldo is None:
        ldo = max(xyz_col + 3, feat_col + C)
    inv_off, inv_rows = inverse if inverse is not None else (None, None)
    return _GroupConcatCL.apply(xyz, center, feat, idx, radius, normalize_xyz, ldo, xyz_col,
                                feat_col, inv_off, inv_rows)


class _GatherRowsCL(Function):
    @staticmethod
    def forward(ctx, feat, idx):
        _chk(feat, "feat")
        _chk(idx, "idx", torch.int32)
        B, N, C = feat.shape
        M = idx.shape[1]
        out = torch.empty((B, M, C), dtype=feat.dtype, device=feat.device)
        _ffi.call("demf_gather_rows_cl_fwd", B, N, M, C, _p(feat), _p(idx), _p(out), _stream())
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, M, C = grad_out.shape
        g = zeros((B, ctx.N, C), grad_out.device)
        _ffi.call("demf_gather_rows_cl_bwd", B, ctx.N, M, C, _p(grad_out), _p(idx), _p(g),
                  _stream())
        return g, None


def gather_rows_cl(feat, idx):
    """feat (B,N,C), idx (B,M) i32 -> (B,M,C); differentiable wrt feat."""
    return _GatherRowsCL.apply(feat, idx)


class _ThreeInterpolateCL(Function):
    @staticmethod
    def forward(ctx, feat, idx, weight):
        _chk(feat, "feat")
        _chk(idx, "idx", torch.int32)
        _chk(weight, "weight")
        B, m, C = feat.shape
        n = idx.shape[1]
        out = torch.empty((B, n, C), dtype=feat.dtype, device=feat.device)
        _ffi.call("demf_three_interpolate_cl_fwd", B, m, n, C, C, 0, _p(feat), _p(idx),
                  _p(weight), _p(out), _stream())
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, n, C = grad_out.shape
        g = zeros((B, ctx.m, C), grad_out.device)
        _ffi.call("demf_three_interpolate_cl_bwd", B, ctx.m, n, C, C, 0, _p(grad_out), _p(idx),
                  _p(weight), _p(g), _stream())
        return g, None, None


class _ThreeInterpolateCatCL(Function):
    @staticmethod
    def forward(ctx, feat, idx, weight, skip):
        _chk(feat, "feat")
        _chk(idx, "idx", torch.int32)
        _chk(weight, "weight")
        _chk(skip, "skip")
        B, m, C = feat.shape
        n, Cs = idx.shape[1], skip.shape[2]
        out = torch.empty((B, n, C + Cs), dtype=feat.dtype, device=feat.device)
        _ffi.call("demf_three_interpolate_cat_cl_fwd", B, m, n, C, Cs, _p(feat), _p(idx), _p(weight),
                  _p(skip), _p(out), _stream())
        ctx.save_for_backward(idx, weight)
        ctx.dims = (m, C, Cs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        m, C, Cs = ctx.dims
        grad_out = grad_out.contiguous()
        B, n, _ = grad_out.shape
        g = zeros((B, m, C), grad_out.device)
        _ffi.call("demf_three_interpolate_cl_bwd", B, m, n, C, C + Cs, 0, _p(grad_out), _p(idx),
                  _p(weight), _p(g), _stream())
        return g, None, None, grad_out[..., C:]


def three_interpolate_cat_cl(feat, idx, weight, skip):
    """[three_interpolate_cl(feat, idx, weight) | skip (B,n,Cs)] -> (B,n,C+Cs) in one launch (the
    interpolation + concatenation of PointFPModule.forward); differentiable wrt feat and skip."""
    return _ThreeInterpolateCatCL.apply(feat, idx, weight, skip)


def three_interpolate_cl(feat, idx, weight):
    """feat (B,m,C), idx/weight (B,n,3) -> (B,n,C); differentiable wrt feat."""
    return _ThreeInterpolateCL.apply(feat, idx, weight)


class _MaxPoolNS(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, "x")
        R, ns, C = x.shape
        out = torch.empty((R, C), dtype=x.dtype, device=x.device)
        arg = torch.empty((R, C), dtype=torch.int32, device=x.device)
        _ffi.call("demf_maxpool_ns_fwd", R, ns, C, _p(x), _p(out), _p(arg), _stream())
        ctx.save_for_backward(arg)
        ctx.ns = ns
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        R, C = grad_out.shape
        gx = torch.empty((R, ctx.ns, C), dtype=grad_out.dtype, device=grad_out.device)
        _ffi.call("demf_maxpool_ns_bwd", R, ctx.ns, C, _p(grad_out), _p(arg), _p(gx), _stream())
        return gx


def maxpool_ns(x):
    """x (R,ns,C) -> (R,C): max over the neighbour axis (F.max_pool2d(kernel=[1,ns])
    of the reference's PointSAModule, in point-major layout)."""
    return _MaxPoolNS.apply(x)


_IDENTITY_DY = {}


def _identity_dy_vectors(n, device):
    """The 5 per-channel vectors (scale, shift, gi, a, b) that make the dY prologue of
    demf_mlp_gemm_bwd_dw the identity: mask 0*y+1 > 0, dY = 1*dZ + 0*y + 0."""
    key = (n, str(device))
    if key not in _IDENTITY_DY:
        v = torch.zeros(5, n, dtype=torch.float32, device=device)
        v[1] = 1.0
        v[2] = 1.0
        _IDENTITY_DY[key] = v.reshape(-1).contiguous()
    return _IDENTITY_DY[key]


class _LinearRows(Function):
    """y = x W^T + b on rows [, rows selected by ``row_mask`` zeroed].  No library GEMM: few rows (heads, vote
    module, position embedding) run on the strided GEMM of csrc/dense.hip (``fused.gemm``); long row sets with
    GEMM-friendly widths (the value projection of project-then-sample, the image encoder's module path: K % 32 == 0,
    N % 128 == 0) on the long-row kernel of csrc/rows_gemm.hip with the mask fused; the weight gradient of a long
    reduction on the slab-split dW kernel of the shared-MLP path, the bias gradient through demf_colsum_f32 / the
    GEMM's row sums.  The row mask is applied in place on the incoming gradient: no autograd view+in-place
    machinery."""

    @staticmethod
    def _rows_ok(R, K, N, x):
        return R > _OWN_GEMM_ROWS and K % 32 == 0 and N % 128 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0

    @staticmethod
    def forward(ctx, x, weight, bias, row_mask):
        from . import fused
        ctx.has_bias = bias is not None
        R, K = x.shape
        N = weight.shape[0]
        if x.stride(1) != 1:
            x = x.contiguous()
        if bias is not None and not bias.is_contiguous():
            bias = bias.contiguous()
        y = torch.empty((R, N), dtype=torch.float32, device=x.device)
        planes = 1 if get_compute_dtype() == "bf16" else 3
        if _LinearRows._rows_ok(R, K, N, x):
            rows_gemm(x, split_planes(weight, planes), bias.detach() if bias is not None else None, y,
                      row_mask=row_mask, mask_col0=0)
        else:
            fused.gemm(R, N, K, _p(x), (x.stride(0), 1), _p(weight), weight.stride(), _p(y), N, bias=_p(bias))
            if row_mask is not None:
                y.masked_fill_(row_mask.unsqueeze(-1), 0.0)
        if row_mask is not None:
            ctx.save_for_backward(x, weight, row_mask)
        else:
            ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import fused
        x, weight = ctx.saved_tensors[:2]
        g = g.contiguous()
        R, K = x.shape
        N = weight.shape[0]
        if len(ctx.saved_tensors) == 3:
            # the incoming gradient is a temporary of the producer (the MSDA backward's
            # grad_value); it is masked in place rather than cloned (152 MB at the bench size)
            g.masked_fill_(ctx.saved_tensors[2].unsqueeze(-1), 0.0)
        gx = gw = gb = None
        # An output width that is not a multiple of 4 (the vote module's 3 + 256 = 259 columns on 8 192 seeds) leaves the
        # gradient rows unaligned: both backward products then stage their operands element by element (33 + 44 us, and
        # 23 us for the column sums).  The gradient and the weight are copied once into zero-padded buffers of 4-aligned
        # width instead (two small copies): the products run on the vectorised staging and the bias gradient rides in
        # the weight-gradient launch; what is returned are the leading rows of the padded results.
        Np, gq, wq = N, g, weight
        if N % 4 != 0 and R >= 4096 and len(ctx.saved_tensors) == 2:
            Np = (N + 3) // 4 * 4
            gq = zeros((R, Np), g.device)
            gq[:, :N].copy_(g)
            wq = zeros((Np, K), g.device)
            wq[:N].copy_(weight)
        if ctx.needs_input_grad[0]:
            gx = torch.empty((R, K), dtype=torch.float32, device=g.device)
            if Np != N:
                fused.gemm(R, K, Np, _p(gq), (Np, 1), _p(wq), (1, K), _p(gx), K)
            elif _LinearRows._rows_ok(R, N, K, g):
                planes = 1 if get_compute_dtype() == "bf16" else 3
                rows_gemm(g, split_planes(weight.detach().t().contiguous(), planes), None, gx)
            else:
                fused.gemm(R, K, N, _p(g), (N, 1), _p(weight), (weight.stride(1), weight.stride(0)), _p(gx), K)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            # one zero-filled workspace for dW | db (split-K atomics, column sums)
            ws = zeros(Np * K + Np, g.device)
            gw = ws[:Np * K].view(Np, K)
            gb = ws[Np * K:] if ctx.has_bias else None
            in_gemm = False
            if Np != N:
                in_gemm = gb is not None
                fused.gemm(Np, K, R, _p(gq), (1, Np), _p(x), (1, x.stride(0)), _p(gw), K,
                           splitk=fused._splitk(R), asum=_p(gb) if in_gemm else None)
                if gb is not None and not in_gemm:
                    _ffi.call("demf_colsum_f32", R, Np, Np, _p(gq), _p(gb), _stream())
                return gx, gw[:N], (gb[:N] if gb is not None else None), None
            if R >= 32768 and N % 4 == 0 and K % 4 == 0 and x.is_contiguous():
                # long reduction (the value projection: 149 k tokens -> 256x256): the slab-split
                # dW kernel of the shared-MLP path, run with an identity dY prologue
                _ffi.call("demf_mlp_gemm_bwd_dw", R, N, K, K, _p(g), None, None, 1, _p(g),
                          _p(_identity_dy_vectors(N, g.device)), _p(x), None, _p(gw), _stream())
            else:
                # (+ the bias gradient = row sums of the A operand, taken by the same launch)
                in_gemm = gb is not None and N % 4 == 0 and N >= 4
                fused.gemm(N, K, R, _p(g), (1, N), _p(x), (1, x.stride(0)), _p(gw), K,
                           splitk=fused._splitk(R), asum=_p(gb) if in_gemm else None)
            if gb is not None and not in_gemm:
                _ffi.call("demf_colsum_f32", R, N, N, _p(g), _p(gb), _stream())
        return gx, gw, gb, None


def linear(x, weight, bias=None, row_mask=None):
    """F.linear(x (..., K), weight (N, K), bias (N)) for device tensors of the hot path;
    ``row_mask`` (...) bool: output rows to zero (the padding mask of the value projection)."""
    if not x.is_cuda:
        raise RuntimeError("x must be a GPU (HIP) tensor: demf_amd operators have no CPU path")
    lead = x.shape[:-1]
    y = _LinearRows.apply(x.reshape(-1, x.shape[-1]), weight, bias,
                          None if row_mask is None else row_mask.reshape(-1))
    return y.view(*lead, weight.shape[0])


def split_planes(w, planes=3):
    """(N,K) fp32 weight -> (planes, N, K) bf16: planes = 3 is w = h + m + l EXACTLY (three 8-bit significand
    slices, the operand form of the fp32-grade bf16-MFMA kernels), planes = 1 the rounded weight."""
    w = w.detach().float().contiguous()
    h = w.to(torch.bfloat16)
    if planes == 1:
        return h.unsqueeze(0).contiguous()
    r = w - h.float()
    m = r.to(torch.bfloat16)
    low = (r - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, low]).contiguous()


def rows_gemm(x, w_planes, bias, out, a2=None, a2_cols=0, relu=False, ln=None, row_mask=None, mask_col0=0,
              a2_replace=False):
    """out (R,N) = epi((x [+ a2 on output columns < a2_cols]) . W^T + bias) on demf_rows_gemm_f32
    (csrc/rows_gemm.hip): ``w_planes`` from ``split_planes``; ``relu``; ``ln`` = (residual (R,N), gamma, beta,
    eps): LayerNorm(residual + .) in the epilogue (N == 256); ``row_mask`` (R) bool: rows zeroed in columns
    >= mask_col0; ``a2_replace``: ``a2`` is a second operand that feeds those columns INSTEAD of x (e.g. the
    pre-added x + pos).  Forward only (the frozen image stream)."""
    R, K = x.shape
    planes, N, Kw = w_planes.shape
    if not (x.is_cuda and out.is_cuda and w_planes.is_cuda):
        raise RuntimeError("rows_gemm: operands must be GPU (HIP) tensors: demf_amd operators have no CPU path")
    assert x.dtype == torch.float32 and out.dtype == torch.float32, "rows_gemm: fp32 rows in, fp32 rows out"
    assert Kw == K and out.shape == (R, N) and x.stride(1) == 1 and out.stride(1) == 1
    assert w_planes.dtype == torch.bfloat16 and w_planes.is_contiguous()
    mode = 2 if ln is not None else (1 if relu else 0)
    res, g, b, eps = ln if ln is not None else (None, None, None, 0.0)
    if a2 is not None:
        assert a2.shape == x.shape and a2.stride() == x.stride() and a2.is_cuda and a2.dtype == torch.float32
    if row_mask is not None:
        assert row_mask.dtype == torch.bool and row_mask.numel() == R and row_mask.is_contiguous() and row_mask.is_cuda
    for name, v, n in (("bias", bias, N), ("gamma", g, N), ("beta", b, N)):
        assert v is None or (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == n), \
            "rows_gemm: %s must be a contiguous fp32 GPU vector of %d elements" % (name, n)
    if res is not None:
        assert res.is_cuda and res.dtype == torch.float32 and tuple(res.shape) == (R, N) and res.stride(1) == 1 and \
            res.stride(0) % 4 == 0 and out.stride(0) % 4 == 0, "rows_gemm: residual must be fp32 (R,N) rows"
    _ffi.call("demf_rows_gemm_f32", R, N, K, _p(x), x.stride(0), _p(a2), int(a2_cols), int(bool(a2_replace)),
              _p(w_planes), planes,
              _p(bias), mode, _p(row_mask), int(mask_col0), _p(res), res.stride(0) if res is not None else 0,
              _p(g), _p(b), float(eps), _p(out), out.stride(0), _stream())
    return out


def _gpu_f32(name, *ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(name + ": operands must be GPU (HIP) tensors: demf_amd operators have no CPU path")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(name + ": contiguous fp32 tensors required")


def conv_weight_planes(weight, planes=3, scale=None):
    """(Cout,Cin,KH,KW) convolution weight [x per-output-channel ``scale``: a folded frozen BatchNorm] ->
    (planes, Cout, KH*KW*Cin) bf16 planes in the reduction order (kh, kw, c) of demf_conv_nhwc_f32."""
    w = weight.detach().float()
    if scale is not None:
        w = w * scale.detach().float().view(-1, 1, 1, 1)
    return split_planes(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), planes)


def stem_weight_planes(weight, planes=3, scale=None):
    """(Cout,3,7,7) stem weight -> (planes, Cout, 7*32): element [kh*32 + kw*4 + c], zeros at kw == 7 / c == 3
    (demf_conv_stem7_nhwc4_f32 reads one kernel row = 8 pixels x 4 channels per reduction step)."""
    w = weight.detach().float()
    assert tuple(w.shape[1:]) == (3, 7, 7)
    if scale is not None:
        w = w * scale.detach().float().view(-1, 1, 1, 1)
    full = w.new_zeros(w.shape[0], 7, 8, 4)
    full[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    return split_planes(full.reshape(w.shape[0], 224), planes)


def conv_nhwc(x, w_planes, bias, KH, KW, stride=1, pad=0, resid=None, relu=False, out=None, ksplit=None):
    """y (B,Ho,Wo,Cout) = [relu](conv(x (B,H,W,Cin)) + bias [+ resid]) on channels-last activations
    (demf_conv_nhwc_f32, csrc/conv.hip: implicit GEMM, nothing unfolded).  ``w_planes`` from conv_weight_planes.
    ``ksplit`` (default: chosen here): a plain convolution (no bias / residual / ReLU) with few output pixels and
    a long reduction is split over the reduction, partial tiles added into a zeroed output.
    Forward only (the frozen image stream)."""
    B, H, W, Cin = x.shape
    planes, Cout, K = w_planes.shape
    _gpu_f32("conv_nhwc", x, bias, resid, out)
    assert K == KH * KW * Cin and w_planes.dtype == torch.bfloat16 and w_planes.is_contiguous() and w_planes.is_cuda
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (B, Ho, Wo, Cout) and (bias is None or bias.numel() == Cout)
    assert resid is None or tuple(resid.shape) == (B, Ho, Wo, Cout)
    plain = bias is None and resid is None and not relu
    if ksplit is None:
        # few tiles and a long reduction (the neck's 3x3 level: 30 tiles x 576 steps): slices until ~768 workgroups,
        # at least 8 reduction steps each.  (With more tiles the atomic adds cost more than the idle CUs: the
        # 2048 -> 256 1x1 level, 110 tiles, measured 0.165 ms split six ways against 0.114 ms whole.)
        tiles = -(-(B * Ho * Wo) // 128) * (Cout // (128 if Cout % 128 == 0 else 64))
        ksplit = max(1, min(768 // max(tiles, 1), (K // 32) // 8, 64)) if plain and tiles <= 64 else 1
    if ksplit > 1:
        assert plain, "conv_nhwc: split-K needs a plain convolution"
        out.zero_()
    _ffi.call("demf_conv_nhwc_f32", B, H, W, Cin, Cout, KH, KW, stride, pad, _p(x), _p(w_planes), planes, _p(bias),
              _p(resid), int(bool(relu)), int(ksplit), _p(out), _stream())
    return out


def conv_stem7(img, w_planes, bias, relu=True):
    """ResNet stem: (B,3,H,W) image -> relu(conv7x7 s2 p3 + bias) as (B,H/2,W/2,Cout) channels-last rows
    (the image passes through an NHWC4 copy: demf_nchw3_to_nhwc4_f32 + demf_conv_stem7_nhwc4_f32)."""
    B, C, H, W = img.shape
    planes, Cout, K = w_planes.shape
    _gpu_f32("conv_stem7", img, bias)
    assert C == 3 and K == 224 and w_planes.dtype == torch.bfloat16 and w_planes.is_contiguous() and w_planes.is_cuda
    x4 = torch.empty((B, H, W, 4), dtype=torch.float32, device=img.device)
    _ffi.call("demf_nchw3_to_nhwc4_f32", B, H, W, _p(img), _p(x4), _stream())
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), dtype=torch.float32, device=img.device)
    _ffi.call("demf_conv_stem7_nhwc4_f32", B, H, W, Cout, _p(x4), _p(w_planes), planes, _p(bias), int(bool(relu)),
              _p(out), _stream())
    return out


def maxpool3x3s2_nhwc(x):
    B, H, W, C = x.shape
    _gpu_f32("maxpool3x3s2_nhwc", x)
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    _ffi.call("demf_maxpool3x3s2_nhwc_f32", B, H, W, C, _p(x), _p(out), _stream())
    return out


def groupnorm_nhwc_into(x, groups, gamma, beta, eps, tokens, row0):
    """GroupNorm of x (B,h,w,256) channels-last, written to rows [row0, row0 + h*w) of ``tokens`` (B,S,256)."""
    B, h, w, C = x.shape
    _gpu_f32("groupnorm_nhwc", x, gamma, beta, tokens)
    assert tokens.shape[0] == B and tokens.shape[2] == C and row0 + h * w <= tokens.shape[1]
    sums = torch.empty(2 * B * groups, dtype=torch.float64, device=x.device)
    _ffi.call("demf_groupnorm_nhwc_f32", B, h * w, C, groups, float(eps), _p(x), _p(gamma), _p(beta), _p(sums),
              tokens.data_ptr() + 4 * row0 * C, tokens.shape[1] * C, _stream())
    return tokens


_MSDA_HEAD = int(os.environ.get("DEMF_MSDA_HEAD", "1") or 0)        # A/B switch: the (scene, head)-major form


def msda_fwd_raw(raw, value_col0, off_col0, lgt_col0, ref, spatial_shapes, level_start_index, B, S, H, Dh, P, out,
                 level_sizes=None):
    """Self-attention form of the multi-scale deformable attention (the encoder: queries = the S tokens): raw
    offsets / logits / projected value are column ranges of ``raw`` (B*S, ld); softmax and
    loc = ref + offset / (W_l, H_l) happen inside the kernel (demf_msda_fwd_raw_f32).  -> out (B*S, H*Dh)
    ``level_sizes`` (host ints, tokens per level): with 8 heads x 32 channels x 4 levels and enough queries the
    coarse levels' value rows of one (scene, head) that fit LDS are kept there (demf_msda_fwd_raw_head_f32)."""
    L = spatial_shapes.shape[0]
    assert raw.stride(1) == 1 and ref.is_contiguous() and out.is_contiguous()
    if (_MSDA_HEAD and level_sizes is not None and H == 8 and Dh == 32 and L == 4 and len(level_sizes) == 4
            and P in (2, 4) and S >= 2048 and sum(level_sizes) == S):
        budget = 160 * 1024 - 8 * 8 * (4 * P) * 20          # (csrc/msda.hip: MSDA_HW waves x 8 queries x samples x 20 B)
        for first in (2, 3):
            staged = int(sum(level_sizes[first:]))
            if 1 <= staged < S and staged * 128 <= budget:
                _ffi.call("demf_msda_fwd_raw_head_f32", B, S, S, P, raw.data_ptr() + 4 * value_col0, raw.stride(0),
                          _p(spatial_shapes), _p(level_start_index), _p(raw), raw.stride(0), int(off_col0),
                          int(lgt_col0), _p(ref), _p(out), first, staged, _stream())
                return out
    _ffi.call("demf_msda_fwd_raw_f32", B, S, H, Dh, L, S, P, raw.data_ptr() + 4 * value_col0, raw.stride(0),
              _p(spatial_shapes), _p(level_start_index), _p(raw), raw.stride(0), int(off_col0), int(lgt_col0),
              _p(ref), _p(out), _stream())
    return out


# --------------------------------------------------------------------------
# Fused shared MLP: (1x1 conv -> train-mode BN -> ReLU) x L [-> max over ns]
# --------------------------------------------------------------------------
_ACCUM64 = {}
_OWN_GEMM_ROWS = int(__import__('os').environ.get('DEMF_OWN_GEMM_ROWS', '65536'))    # A/B switch
_NO_FPS_CHECK = bool(int(__import__('os').environ.get('DEMF_NO_FPS_CHECK', '0')))       # A/B switch
_NO_RED_FUSE = bool(int(__import__('os').environ.get('DEMF_NO_RED_FUSE', '0')))        # A/B switch
_NO_FIRST_FUSE = bool(int(__import__('os').environ.get('DEMF_NO_FIRST_FUSE', '0')))    # A/B switch
_NO_BWD_FUSE = bool(int(__import__('os').environ.get('DEMF_NO_BWD_FUSE', '0')))        # A/B switch
# bf16 ROW storage of SA1's stack in the bf16 compute mode: built and parity-tested, measured neutral on
# the step (5.26 vs 5.23 ms: these kernels are not HBM-bound), so opt-in (DEMF_BF16_STORE=1)
_NO_BF16_STORE = not bool(int(__import__('os').environ.get('DEMF_BF16_STORE', '0')))
_VEC_FIN = int(__import__('os').environ.get('DEMF_VEC_FIN', '7'))   # A/B: bit 0 sparse reduce, 1 fused, 2 dx_red


def _bwd_fused_ok(N, K, ns, sparse, first=False):
    """Shapes / modes demf_mlp_bwd_fused covers (csrc/mlp_bwd.hip): one pass over a layer's saved
    output for dX, dW and the sums of the layer below, instead of a dx and a dW launch."""
    if _NO_BWD_FUSE or _COMPUTE_MODE not in (1, 2):
        return False
    if sparse and (ns < 4 or ns % 4):
        return False
    if first:
        return N == 64 and K == 64 and not sparse
    return (N, K) in ((128, 64), (128, 128), (64, 64), (256, 128))
# rows from which a 256-output layer with a 256 / 384 / 512-channel input takes the one-pass backward per
# 128-column chunk; 0 = never (default: measured neutral on the step - 5.67 vs 5.67 ms - because the
# (256,128) kernel walks a 32-row slab in ~8 us and the vote aggregation's 32 768 rows are 4 slabs per block)
_FUSED_COLS_MIN_R = int(os.environ.get("DEMF_FUSED_COLS_MIN_R", "0"))


def _bwd_fused_cols_ok(R, N, K, ns, sparse):
    """demf_mlp_bwd_fused_cols: the one-pass backward on 128-column chunks of a 256-output layer whose
    input has K = 256 / 384 / 512 channels (the vote aggregation stack), on enough rows to fill the
    persistent grid."""
    if _NO_BWD_FUSE or _COMPUTE_MODE not in (1, 2) or _FUSED_COLS_MIN_R <= 0 or R < _FUSED_COLS_MIN_R:
        return False
    if sparse and (ns < 4 or ns % 4):
        return False
    return N == 256 and K > 128 and K % 128 == 0 and K <= 512


# SA1's pooled last layer without its (R x 128) output: the forward does not store it, the backward is
# written in terms of the layer's INPUT activations (csrc/mlp_bwd.hip mlp_bwd_pool_kernel).  A/B switch.
_POOL_NOY = bool(int(os.environ.get("DEMF_POOL_NOY", "1")))


# SA1's first layer (4-float rows -> 64 channels) without its (R x 64) output: statistics from the rows'
# second moments, the consumers rebuild the values they need (csrc/mlp.hip mlp_first_stats_k, ST bit 3 of
# mlp_fwd_res_kernel; csrc/mlp_bwd.hip ST bit 2).  A/B switch.
_SA1_X4 = bool(int(os.environ.get("DEMF_SA1_X4", "1")))


def _x3_forward_on():
    """The no-store forwards exist in the bf16 / three-term kernels only: in mode 2 the A/B switch DEMF_X3_MASK
    without bit 0 sends forward launches to the native-fp32 kernel (csrc/mlp.hip launch_gemm), which stores."""
    return _COMPUTE_MODE == 1 or (_COMPUTE_MODE == 2 and (int(os.environ.get("DEMF_X3_MASK", "7") or 0) & 1))


def _sa1_x4_ok(R, ld, shapes, training, x_grad, ns):
    return (_SA1_X4 and training and not x_grad and _x3_forward_on() and ld == 4 and len(shapes) >= 3
            and shapes[0] == (64, 4) and shapes[1] == (64, 64) and R >= 16384 and not _NO_BWD_FUSE
            and not _NO_FIRST_FUSE and int(os.environ.get("DEMF_FWD_RES", "1") or 0) and
            not int(os.environ.get("DEMF_STATIC_TILES", "0") or 0))


def _pool_noy_ok(R, ns, shapes, training, fuse_pool):
    """Shapes / modes of the no-store pooled last layer (N = 128 <- K = 64, 64-row groups, enough rows for
    the weight-resident forward)."""
    return (_POOL_NOY and training and fuse_pool and _x3_forward_on() and ns == 64 and R >= 16384
            and R % 64 == 0 and len(shapes) >= 2 and shapes[-1] == (128, 64) and shapes[-2][0] == 64
            and not _NO_BWD_FUSE and (_VEC_FIN & 1))


_NO_GROUP_FIRST = bool(int(__import__('os').environ.get('DEMF_NO_GROUP_FIRST', '0')))  # A/B switch
_NO_FUSED_POOL = bool(int(__import__('os').environ.get('DEMF_NO_FUSED_POOL', '0')))   # A/B switch


def _accum64(n, device):
    """A persistent, zero-initialised fp64 accumulator of >= n elements per device.  The kernels
    that consume it (demf_bn_finalize, demf_bn_bwd_vectors) leave it zeroed again, so the step
    issues no fill for it.  Never (re)allocated inside a graph capture."""
    key = str(device)
    buf = _ACCUM64.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(n, dtype=torch.float64, device=device)
        buf = torch.zeros(max(n, 4096), dtype=torch.float64, device=device)
        _ACCUM64[key] = buf
    return buf


def reset_accumulators():
    """Zero the persistent fp64 accumulators.  The kernels that consume them leave them zeroed ("self-cleaning"), which
    holds only if every launch sequence runs to its end: an exception between a layer's statistics launch and the
    launch that consumes them (a bad argument further down the forward, a refused fall-back) leaves sums behind that
    the NEXT step would add to.  engine.Trainer calls this when a step raises; tests call it between cases."""
    for buf in _ACCUM64.values():
        buf.zero_()


class _SharedMLPPool(Function):
    """x (R, ld) rows -> pooled (R/ns, C_L).  Per layer l the tensors are
    (W_l (N_l, K_l), gamma_l, beta_l, running_mean_l, running_var_l, conv_bias_l|None);
    K_0 == ld.  A conv bias in front of a train-mode BN cancels in the output, so it never
    reaches the kernels; its gradient is returned as exact zeros.
    Forward never materialises a BN/ReLU output: only the raw conv outputs Y_l are stored
    (needed for backward), statistics come out of the GEMM epilogue, and the last
    BN+ReLU is fused with the max over the ``ns`` neighbours (ns == 1: plain activation)."""

    @staticmethod
    def forward(ctx, x, ns, training, eps, momentum, geo, geo_xyz, geo_center, *tensors):
        _chk(x, "x")
        R, ld = x.shape
        L = len(tensors) // 7
        if geo is not None:
            # x = per-point feature rows (B*N, C); the grouped rows never exist (csrc/group_first.hip)
            g_xyz, g_center, g_idx, g_off, g_rows, g_radius, g_norm = geo
            _chk(g_xyz, "xyz")
            _chk(g_center, "center")
            _chk(g_idx, "idx", torch.int32)
            gB, gN, _ = g_xyz.shape
            _, gM, gns = g_idx.shape
            assert gns == ns and R == gB * gN and L >= 2 and tensors[0].shape[1] == ld + 3
            R = gB * gM * ns
        assert len(tensors) == 7 * L and R % ns == 0
        dev = x.device
        st = _stream()
        Ys, sss, mis = [], [], []
        cur, cur_ld, pro = x, ld, None
        fuse_pool = False
        # bf16 compute mode (BASELINE configs[3]): SA1-shaped stacks (4-float rows -> 64 -> 64 -> 128, the
        # 1 M-row layers that carry most of the step's HBM traffic) keep the raw outputs of layers 1 and 2
        # - and the gradient that flows between their backwards - as bf16 rows; every kernel on that path
        # is a round-3 kernel that loads / stores them directly (csrc/mlp.hip mlp_fwd_res_kernel,
        # csrc/mlp_bwd.hip).  Statistics, pooled extrema, vectors and weight gradients stay fp32.
        store16 = (_COMPUTE_MODE == 1 and not _NO_BF16_STORE and training and geo is None and L == 3
                   and ld == 4 and not x.requires_grad and R >= 16384 and R % 64 == 0
                   and ns in (16, 32, 64) and not _NO_FUSED_POOL and not _NO_BWD_FUSE and not _NO_FIRST_FUSE
                   and [tuple(tensors[7 * l].shape) for l in range(3)] == [(64, 4), (64, 64), (128, 64)])
        noy = geo is None and not store16 and L >= 2 and ld % 4 == 0 and _pool_noy_ok(
            R, ns, [tuple(tensors[7 * l].shape) for l in range(L)], training,
            training and not _NO_FUSED_POOL and ns in (16, 32, 64))
        x4 = geo is None and not store16 and _sa1_x4_ok(
            R, ld, [tuple(tensors[7 * l].shape) for l in range(L)], training, x.requires_grad, ns)
        # the self-cleaning fp64 accumulator holds every layer's statistics (no fill launches)
        ws = _accum64(2 * sum(tensors[7 * l].shape[0] for l in range(L)), dev) if training else None
        woff = 0
        for l in range(L):
            W, gamma, beta, rmean, rvar = tensors[7 * l:7 * l + 5]
            nbt = tensors[7 * l + 6]
            _chk(W, "weight")
            N, K = W.shape
            first_geo = geo is not None and l == 0
            assert first_geo or K == cur_ld, \
                f"layer {l}: weight has {K} input columns, rows have {cur_ld}"
            # (x4: layer 0's output is never written - an empty placeholder keeps the saved-tensor layout)
            Y = torch.empty((0 if (x4 and l == 0) else R, N),
                            dtype=torch.bfloat16 if (store16 and l >= 1) else torch.float32, device=dev)
            ss = torch.empty(2 * N, dtype=torch.float32, device=dev)
            mi = torch.empty(2 * N, dtype=torch.float32, device=dev)
            fuse_pool = training and l == L - 1 and l > 0 and not _NO_FUSED_POOL and \
                ns in (16, 32, 64)
            if first_geo:
                # reference column order [xyz(3) | feat(C)]: U = feat . Wf^T once per source point
                U = torch.empty((x.shape[0], N), dtype=torch.float32, device=dev)
                from . import fused
                fused.gemm(x.shape[0], N, ld, _p(x), (ld, 1), W.data_ptr() + 12, (K, 1), _p(U), N)
                ctx.geo_U = U if training else None     # the backward forms y again from it instead of reading Y_0
                stats = None
                if training:
                    stats = ws[woff:woff + 2 * N]
                    woff += 2 * N
                # (the xyz columns of W are read in place: w_ld = K)
                # (train mode: + the BatchNorm bookkeeping, by the launch's last workgroup)
                fin = (_p(gamma), _p(beta), float(eps), float(momentum), _p(rmean), _p(rvar), _p(nbt), _p(ss), _p(mi),
                       _p(tensors[7 * l + 5])) if training else (None, None, 0.0, 0.0, None, None, None, None, None, None)
                _ffi.call("demf_group_first_fwd", gB, gN, gM, ns, N, float(g_radius),
                          int(bool(g_norm)), _p(g_xyz), _p(g_center), _p(g_idx), _p(U), _p(W), K,
                          _p(Y), _p(stats), *fin, st)
                if not training:
                    invstd = torch.rsqrt(rvar + eps)
                    ss[:N] = gamma * invstd
                    ss[N:] = beta - rmean * gamma * invstd
                    mi[:N] = rmean
                    mi[N:] = invstd
            elif fuse_pool:
                # last layer: the max over the ns neighbours rides in the GEMM epilogue
                stats = ws[woff:woff + 2 * N]
                woff += 2 * N
                # only the extremum the sign of gamma selects (pmin / amin = NULL)
                pm = torch.empty((R // ns, N), dtype=torch.float32, device=dev)
                am = torch.empty((R // ns, N), dtype=torch.int32, device=dev)
                # (+ the BN bookkeeping, in the GEMM's last workgroup)
                if noy:
                    # the raw (R x N) output is never written: its backward needs the layer's INPUT only
                    Y = torch.empty((0, N), dtype=torch.float32, device=dev)
                    _ffi.call("demf_mlp_gemm_fwd_pool_bn_st", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), None,
                              _p(stats), ns, _p(pm), _p(am), _p(gamma), _p(beta), float(eps),
                              float(momentum), _p(rmean), _p(rvar), _p(nbt), _p(ss), _p(mi),
                              _p(tensors[7 * l + 5]), 4, st)
                elif store16:
                    _ffi.call("demf_mlp_gemm_fwd_pool_bn_st", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), _p(Y),
                              _p(stats), ns, _p(pm), _p(am), _p(gamma), _p(beta), float(eps),
                              float(momentum), _p(rmean), _p(rvar), _p(nbt), _p(ss), _p(mi),
                              _p(tensors[7 * l + 5]), 3, st)
                else:
                    _ffi.call("demf_mlp_gemm_fwd_pool_bn", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), _p(Y),
                              _p(stats), ns, _p(pm), None, _p(am), None, _p(gamma),
                              _p(beta), float(eps), float(momentum), _p(rmean), _p(rvar), _p(nbt), _p(ss),
                              _p(mi), _p(tensors[7 * l + 5]), st)
            elif training and x4 and l == 0:
                # statistics of y = x.W0^T from the second moments of the 16-byte rows; no output
                stats = ws[woff:woff + 2 * N]
                woff += 2 * N
                _ffi.call("demf_mlp_first_stats", R, N, _p(cur), _p(W), _p(stats), _p(gamma), _p(beta), float(eps),
                          float(momentum), _p(rmean), _p(rvar), _p(nbt), _p(ss), _p(mi), _p(tensors[7 * l + 5]), st)
            elif training and x4 and l == 1:
                stats = ws[woff:woff + 2 * N]
                woff += 2 * N
                _ffi.call("demf_mlp_gemm_fwd_bn_x4", R, N, _p(x), _p(tensors[0]), _p(pro), _p(W), _p(Y), _p(stats),
                          _p(gamma), _p(beta), float(eps), float(momentum), _p(rmean), _p(rvar), _p(nbt),
                          _p(ss), _p(mi), _p(tensors[7 * l + 5]), st)
            elif training:
                stats = ws[woff:woff + 2 * N]
                woff += 2 * N
                if store16 and l == 1:
                    _ffi.call("demf_mlp_gemm_fwd_bn_st", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), _p(Y),
                              _p(stats), _p(gamma), _p(beta), float(eps), float(momentum), _p(rmean),
                              _p(rvar), _p(nbt), _p(ss), _p(mi), _p(tensors[7 * l + 5]), 2, st)
                else:
                    _ffi.call("demf_mlp_gemm_fwd_bn", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), _p(Y),
                              _p(stats), _p(gamma), _p(beta), float(eps), float(momentum), _p(rmean),
                              _p(rvar), _p(nbt), _p(ss), _p(mi), _p(tensors[7 * l + 5]), st)
            else:
                _ffi.call("demf_mlp_gemm_fwd", R, K, N, cur_ld, _p(cur), _p(pro), _p(W), _p(Y),
                          None, st)
                invstd = torch.rsqrt(rvar + eps)
                ss[:N] = gamma * invstd
                ss[N:] = beta - rmean * gamma * invstd
                mi[:N] = rmean
                mi[N:] = invstd
            Ys.append(Y)
            sss.append(ss)
            mis.append(mi)
            cur, cur_ld, pro = Y, N, ss
        C = Ys[-1].shape[1]
        out = torch.empty((R // ns, C), dtype=torch.float32, device=dev)
        arg = torch.empty((R // ns, C), dtype=torch.int32, device=dev)
        yraw = None
        if fuse_pool:
            # + the raw output at the selected rows: the sparse BN-backward reduce reads it
            # instead of gathering 2 M scattered values from the (R x C) output
            yraw = torch.empty((R // ns, C), dtype=torch.float32, device=dev)
            if noy:
                _ffi.call("demf_pool_select_slot0", R // ns, C, _p(pm), _p(am), _p(sss[-1]), _p(out), _p(arg),
                          _p(yraw), st)
            else:
                _ffi.call("demf_pool_select", R // ns, C, _p(pm), _p(pm), _p(am), _p(am),
                          _p(sss[-1]), _p(out), _p(arg), _p(yraw), st)
        else:
            _ffi.call("demf_bnrelu_maxpool_fwd", R // ns, ns, C, _p(Ys[-1]), _p(sss[-1]), _p(out),
                      _p(arg), st)
        ctx.save_for_backward(x, arg, *Ys, *sss, *mis, *[tensors[7 * l] for l in range(L)],
                              *[tensors[7 * l + 1] for l in range(L)],
                              *([yraw] if yraw is not None else []))
        ctx.bias_shapes = [None if tensors[7 * l + 5] is None else tensors[7 * l + 5].shape
                           for l in range(L)]
        ctx.meta = (R, ld, ns, L, training)
        # a layer's dW may be queued (dw_job) only if nothing but view ops sits between this node and the
        # parameter: a torch.cat / index op in between (padded first-layer weights) would read the still empty
        # gradient when the node returns
        ctx.defer_ok = [_plain_parameter(tensors[7 * l]) for l in range(L)]
        ctx.leaf_keys = [_leaf_key(tensors[7 * l]) if ok else None for l, ok in enumerate(ctx.defer_ok)]
        ctx.store16 = store16
        ctx.noy = noy
        ctx.x4 = x4
        ctx.mode = _COMPUTE_MODE          # the no-store forms keep no Y to fall back on: the backward needs this mode
        ctx.geo = None
        if geo is not None:
            ctx.geo = (g_xyz, g_center, g_off, g_rows, float(g_radius), int(bool(g_norm)))
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        R, ld, ns, L, training = ctx.meta
        if not training:
            raise RuntimeError("shared_mlp_pool backward is only defined in training mode")
        if (ctx.noy or ctx.x4) and _COMPUTE_MODE != ctx.mode:
            raise RuntimeError("shared_mlp_pool: the forward ran in compute mode %d and did not store the rows the "
                               "backward of mode %d reads (no-store forms); keep ops.set_compute_dtype unchanged "
                               "between a forward and its backward" % (ctx.mode, _COMPUTE_MODE))
        saved = ctx.saved_tensors
        x, arg = saved[0], saved[1]
        Ys, sss, mis = saved[2:2 + L], saved[2 + L:2 + 2 * L], saved[2 + 2 * L:2 + 3 * L]
        Ws, gammas = saved[2 + 3 * L:2 + 4 * L], saved[2 + 4 * L:2 + 5 * L]
        yraw = saved[2 + 5 * L] if len(saved) > 2 + 5 * L else None
        dev, st = x.device, _stream()
        dP = grad_out.contiguous()
        G = None
        grads = [None] * (7 * L)
        dx = dxyz = dcenter = None
        # BN reductions go through the self-cleaning fp64 accumulator; the weight gradients of all
        # layers share one zero-filled fp32 workspace
        # SA1-like stacks (4-float input rows, no input gradient, >= 3 layers): layer 0's backward is
        # taken from the output tile of layer 1's dx GEMM, whose (R x N0) result is never stored
        fuse_first = (ctx.geo is None and L >= 3 and not ctx.needs_input_grad[0] and ld == 4
                      and Ws[0].shape[0] <= 64 and Ws[0].shape[0] % 4 == 0 and not _NO_FIRST_FUSE)
        n64 = 2 * sum(W.shape[0] for W in Ws) + (10 * Ws[0].shape[0] + 4 if fuse_first else 0)
        if ctx.geo is not None:
            n64 = max(n64, 24 * Ws[0].shape[0])      # (the factored first layer's dWx totals: every sum above is consumed by then)
        ws64 = _accum64(n64, dev)
        # (+ room for the exact-zero gradients of conv biases shadowed by BatchNorm)
        nbias = sum(W.shape[0] for W, bs in zip(Ws, ctx.bias_shapes) if bs is not None)
        ws32 = zeros(sum(W.numel() for W in Ws) + nbias, dev)
        o64 = o32 = 0
        g12_pending = None
        vec_ready = None      # layer l's sums taken by the dx launch of layer l+1 (RED epilogue), whose last
        #                       workgroup also formed the backward vectors (vec6, dgamma, dbeta)
        for l in range(L - 1, -1, -1):
            W = Ws[l]
            N, K = W.shape
            if vec_ready is not None:
                # formed by the last workgroup of the launch that produced this layer's sums
                vec6, dgamma, dbeta = vec_ready
                vec_ready = None
                if g12_pending is not None:          # (A/B: the producer left the sums, vectors as a launch)
                    _ffi.call("demf_bn_bwd_vectors", N, R, _p(g12_pending), _p(gammas[l]), _p(sss[l]),
                              _p(mis[l]), _p(vec6), _p(dgamma), _p(dbeta), st)
                    g12_pending = None
            else:
                vec6 = torch.empty(5 * N, dtype=torch.float32, device=dev)
                dgamma = torch.empty(N, dtype=torch.float32, device=dev)
                dbeta = torch.empty(N, dtype=torch.float32, device=dev)
                g12 = ws64[o64:o64 + 2 * N]
                o64 += 2 * N
                if G is None and (_VEC_FIN & 1):
                    # pooled last layer: sums + vectors in one launch
                    _ffi.call("demf_bn_bwd_reduce_vectors", R, N, ns, _p(dP), _p(arg),
                              _p(Ys[l]) if Ys[l].numel() else None, _p(yraw),
                              _p(sss[l]), _p(mis[l]), _p(g12), _p(gammas[l]), _p(vec6), _p(dgamma),
                              _p(dbeta), int(Ys[l].dtype == torch.bfloat16), st)
                else:
                    _ffi.call("demf_bn_bwd_reduce", R, N, ns, _p(G), _p(dP if G is None else None),
                              _p(arg if G is None else None), _p(Ys[l]), _p(yraw if G is None else None),
                              _p(sss[l]), _p(mis[l]), _p(g12), st)
                    _ffi.call("demf_bn_bwd_vectors", N, R, _p(g12), _p(gammas[l]), _p(sss[l]), _p(mis[l]),
                              _p(vec6), _p(dgamma), _p(dbeta), st)
            if l == 0 and ctx.geo is not None:
                # factored first layer: dU per source point through the inverse lists, then the
                # feature half of dW and the input gradient are GEMMs over the source points
                g_xyz, g_center, g_off, g_rows, g_radius, g_norm = ctx.geo
                if g_off is None:
                    raise RuntimeError("shared_mlp_pool: backward of the factored first layer "
                                       "needs the inverse neighbour lists")
                gB, gN, _ = g_xyz.shape
                gM = g_center.shape[1]
                dU = torch.empty((gB * gN, N), dtype=torch.float32, device=dev)
                # both halves of dW0 (N, K) = [xyz columns | feature columns] land in place in the
                # zero-filled workspace: no transposed copies, no concatenation
                dW0 = ws32[o32:o32 + N * K].view(N, K)
                o32 += N * K
                dxyz = dcenter = None
                if ctx.needs_input_grad[6] or ctx.needs_input_grad[7]:
                    # the coordinates carry a gradient too (vote aggregation)
                    dxyz = torch.empty((gB, gN, 3), dtype=torch.float32, device=dev)
                    dcenter = zeros((gB, gM, 3), dev)
                _ffi.call("demf_group_first_bwd", gB, gN, gM, ns, N, g_radius, g_norm, _p(g_xyz),
                          _p(g_center), _p(G), _p(Ys[0]), _p(getattr(ctx, "geo_U", None)), _p(vec6), _p(g_off),
                          _p(g_rows), _p(dU),
                          _p(dW0), K, _p(W), K, _p(dxyz), _p(dcenter), _p(ws64[:24 * N]), st)
                # dWf = dU^T . feat: long thin reduction -> the slab-split dW kernel, identity prologue
                C0 = K - 3
                dw_job(gB * gN, N, C0, C0, dU, None, None, 1, dU, _identity_dy_vectors(N, dev), x, None, dW0, K,
                       dw_off=3, defer=ctx.defer_ok[0], leaf=ctx.leaf_keys[0])
                grads[0] = dW0
                grads[1], grads[2] = dgamma, dbeta
                if ctx.bias_shapes[0] is not None:
                    grads[5] = ws32[o32:o32 + N].view(ctx.bias_shapes[0])
                    o32 += N
                if ctx.needs_input_grad[0]:
                    from . import fused
                    dx = torch.empty((gB * gN, C0), dtype=torch.float32, device=dev)
                    fused.gemm(gB * gN, C0, N, _p(dU), (N, 1), W.data_ptr() + 12, (1, K), _p(dx), C0)
                break
            xprev = Ys[l - 1] if l > 0 else x
            ldx = xprev.shape[1]
            dW = ws32[o32:o32 + N * K].view(N, K)
            o32 += N * K
            sparse = G is None
            first_here = l == 1 and fuse_first
            s16 = getattr(ctx, "store16", False)
            if l == L - 1 and getattr(ctx, "noy", False):
                # pooled last layer whose output was never stored: dX, dW and layer l-1's sums from its
                # INPUT activations and the sparse pooled gradient (csrc/mlp_bwd.hip mlp_bwd_pool_kernel)
                grads[7 * l], grads[7 * l + 1], grads[7 * l + 2] = dW, dgamma, dbeta
                if ctx.bias_shapes[l] is not None:
                    grads[7 * l + 5] = ws32[o32:o32 + N].view(ctx.bias_shapes[l])
                    o32 += N
                dX = torch.empty((R, K), dtype=torch.float32, device=dev)
                g12p = ws64[o64:o64 + 2 * K]
                o64 += 2 * K
                vec_ready = (torch.empty(5 * K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev))
                nws = ctypes.c_longlong()
                _ffi.call("demf_mlp_bwd_pool_ws", R, ctypes.addressof(nws))
                wsp = torch.empty(nws.value, dtype=torch.float32, device=dev)
                _ffi.call("demf_mlp_bwd_pool", R, N, K, ns, _p(dP), _p(arg), _p(yraw), _p(vec6), _p(W),
                          _p(Ys[l - 1]), _p(sss[l - 1]), _p(mis[l - 1]), _p(dX), _p(dW), _p(g12p),
                          _p(gammas[l - 1]) if (_VEC_FIN & 2) else None, _p(vec_ready[0]), _p(vec_ready[1]),
                          _p(vec_ready[2]), _p(wsp), st)
                if not (_VEC_FIN & 2):
                    g12_pending = g12p
                G = dX
                continue
            if l > 0 and ldx == K and _bwd_fused_ok(N, K, ns, sparse, first_here):
                # ONE pass over Y_l: dX (or, for SA1's layer 1, the raw sums of layer 0's whole
                # backward instead of dX), dW and layer l-1's BN-backward sums (csrc/mlp_bwd.hip)
                grads[7 * l], grads[7 * l + 1], grads[7 * l + 2] = dW, dgamma, dbeta
                if ctx.bias_shapes[l] is not None:
                    grads[7 * l + 5] = ws32[o32:o32 + N].view(ctx.bias_shapes[l])
                    o32 += N
                if first_here:
                    N0 = K
                    sums = ws64[o64:o64 + 10 * N0 + 4]
                    o64 += 10 * N0 + 4
                    if getattr(ctx, "x4", False):
                        _ffi.call("demf_mlp_bwd_fused_x4", R, N, K, _p(G), _p(Ys[l]), _p(vec6), _p(W), _p(x),
                                  _p(Ws[0]), _p(sss[0]), _p(mis[0]), _p(dW), _p(sums), st)
                    else:
                        _ffi.call("demf_mlp_bwd_fused", R, N, K, _p(G), None, None, ns, _p(Ys[l]), _p(vec6),
                                  _p(W), _p(Ys[0]), _p(sss[0]), _p(mis[0]), None, _p(dW), None, _p(x),
                                  _p(sums), None, None, None, None, 2 if s16 else 0, st)
                    dW0 = ws32[o32:o32 + N0 * 4].view(N0, 4)
                    o32 += N0 * 4
                    dgamma0 = torch.empty(N0, dtype=torch.float32, device=dev)
                    dbeta0 = torch.empty(N0, dtype=torch.float32, device=dev)
                    _ffi.call("demf_mlp_first_finish", N0, R, _p(sums), _p(gammas[0]), _p(mis[0]),
                              _p(dW0), _p(dgamma0), _p(dbeta0), st)
                    grads[0], grads[1], grads[2] = dW0, dgamma0, dbeta0
                    if ctx.bias_shapes[0] is not None:
                        grads[5] = ws32[o32:o32 + N0].view(ctx.bias_shapes[0])
                        o32 += N0
                    break
                dX = torch.empty((R, K), dtype=torch.bfloat16 if s16 else torch.float32, device=dev)
                g12p = ws64[o64:o64 + 2 * K]
                o64 += 2 * K
                vec_ready = (torch.empty(5 * K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev))
                _ffi.call("demf_mlp_bwd_fused", R, N, K, _p(G), _p(dP if sparse else None),
                          _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(W), _p(Ys[l - 1]),
                          _p(sss[l - 1]), _p(mis[l - 1]), _p(dX), _p(dW), _p(g12p), None, None,
                          _p(gammas[l - 1]) if (_VEC_FIN & 2) else None, _p(vec_ready[0]), _p(vec_ready[1]),
                          _p(vec_ready[2]), 3 if s16 else 0, st)
                if not (_VEC_FIN & 2):
                    g12_pending = g12p
                G = dX
                continue
            if l > 0 and ldx == K and not first_here and not s16 and _bwd_fused_cols_ok(R, N, K, ns, sparse):
                # the one-pass backward per 128-column chunk of a wider layer l-1 (vote aggregation:
                # 256 -> 256 on 32 768 rows): each launch yields its columns of dX and dW, the sums and
                # (last workgroup) the vectors of those channels
                grads[7 * l], grads[7 * l + 1], grads[7 * l + 2] = dW, dgamma, dbeta
                if ctx.bias_shapes[l] is not None:
                    grads[7 * l + 5] = ws32[o32:o32 + N].view(ctx.bias_shapes[l])
                    o32 += N
                dX = torch.empty((R, K), dtype=torch.float32, device=dev)
                g12p = ws64[o64:o64 + 2 * K]
                o64 += 2 * K
                vec_ready = (torch.empty(5 * K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev),
                             torch.empty(K, dtype=torch.float32, device=dev))
                for c0 in range(0, K, 128):
                    _ffi.call("demf_mlp_bwd_fused_cols", R, N, K, c0, 128, _p(G), _p(dP if sparse else None),
                              _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(W), _p(Ys[l - 1]),
                              _p(sss[l - 1]), _p(mis[l - 1]), _p(dX), _p(dW), _p(g12p),
                              _p(gammas[l - 1]) if (_VEC_FIN & 2) else None, _p(vec_ready[0]),
                              _p(vec_ready[1]), _p(vec_ready[2]), st)
                if not (_VEC_FIN & 2):
                    g12_pending = g12p
                G = dX
                continue
            dw_job(R, N, K, ldx, G, dP if sparse else None, arg if sparse else None, ns, Ys[l], vec6, xprev,
                   sss[l - 1] if l > 0 else None, dW, K, defer=ctx.defer_ok[l], leaf=ctx.leaf_keys[l])
            grads[7 * l], grads[7 * l + 1], grads[7 * l + 2] = dW, dgamma, dbeta
            if ctx.bias_shapes[l] is not None:
                grads[7 * l + 5] = ws32[o32:o32 + N].view(ctx.bias_shapes[l])
                o32 += N
            if l == 1 and fuse_first:
                N0 = K
                sums = ws64[o64:o64 + 10 * N0 + 4]
                o64 += 10 * N0 + 4
                _ffi.call("demf_mlp_gemm_bwd_dx_first", R, N, N0, _p(G), _p(Ys[1]), _p(vec6), _p(W),
                          _p(x), _p(Ys[0]), _p(sss[0]), _p(mis[0]), _p(sums), st)
                dW0 = ws32[o32:o32 + N0 * 4].view(N0, 4)
                o32 += N0 * 4
                dgamma0 = torch.empty(N0, dtype=torch.float32, device=dev)
                dbeta0 = torch.empty(N0, dtype=torch.float32, device=dev)
                _ffi.call("demf_mlp_first_finish", N0, R, _p(sums), _p(gammas[0]), _p(mis[0]),
                          _p(dW0), _p(dgamma0), _p(dbeta0), st)
                grads[0], grads[1], grads[2] = dW0, dgamma0, dbeta0
                if ctx.bias_shapes[0] is not None:
                    grads[5] = ws32[o32:o32 + N0].view(ctx.bias_shapes[0])
                    o32 += N0
                break
            if l > 0 or ctx.needs_input_grad[0]:
                dX = torch.empty((R, K), dtype=torch.float32, device=dev)
                if K % 4 == 0 and l > 0 and not _NO_RED_FUSE:
                    # + layer l-1's BN-backward sums from the output tiles (no separate reduce pass)
                    g12p = ws64[o64:o64 + 2 * K]
                    o64 += 2 * K
                    vec_ready = (torch.empty(5 * K, dtype=torch.float32, device=dev),
                                 torch.empty(K, dtype=torch.float32, device=dev),
                                 torch.empty(K, dtype=torch.float32, device=dev))
                    if _VEC_FIN & 4:
                        _ffi.call("demf_mlp_gemm_bwd_dx_red_v", R, N, K, K, _p(G), _p(dP if sparse else None),
                                  _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(W), _p(dX),
                                  _p(Ys[l - 1]), _p(sss[l - 1]), _p(mis[l - 1]), _p(g12p), _p(gammas[l - 1]),
                                  _p(vec_ready[0]), _p(vec_ready[1]), _p(vec_ready[2]), st)
                    else:
                        _ffi.call("demf_mlp_gemm_bwd_dx_red", R, N, K, K, _p(G), _p(dP if sparse else None),
                                  _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(W), _p(dX),
                                  _p(Ys[l - 1]), _p(sss[l - 1]), _p(mis[l - 1]), _p(g12p), st)
                        g12_pending = g12p
                elif K % 4 == 0:
                    # W read as it is: the kernel transposes the slab on its way into LDS
                    _ffi.call("demf_mlp_gemm_bwd_dx_w", R, N, K, K, _p(G), _p(dP if sparse else None),
                              _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(W), _p(dX), st)
                else:
                    Wtt = W.t().contiguous()
                    _ffi.call("demf_mlp_gemm_bwd_dx", R, N, K, K, _p(G), _p(dP if sparse else None),
                              _p(arg if sparse else None), ns, _p(Ys[l]), _p(vec6), _p(Wtt), _p(dX),
                              st)
                if l > 0:
                    G = dX
                else:
                    dx = dX
        return (dx, None, None, None, None, None,
                dxyz if ctx.needs_input_grad[6] else None,
                dcenter if ctx.needs_input_grad[7] else None, *grads)


def shared_mlp_pool(x, ns, layers, training=True, eps=1e-5, momentum=0.1, geo=None):
    """Fused (conv1x1 -> BN -> ReLU) x L -> max over ``ns`` consecutive rows (ns=1: none).
    ``layers`` = [(weight (N,K), gamma, beta, running_mean, running_var[, conv_bias
    [, num_batches_tracked]]), ...].
    A conv bias in front of a train-mode BN cancels in the normalised output (and its gradient
    is identically zero); it only shifts the running mean, which is applied here.
    ``num_batches_tracked`` (int64 scalar tensor) is incremented by the statistics kernel.
    ``geo`` = (xyz (B,N,3), centres (B,M,3), idx (B,M,ns), inv_off|None, inv_rows|None, radius,
    normalize_xyz): x is then the per-point feature rows (B*N, C) and layer 0 (weight (C1, 3+C) in
    the reference column order [xyz | feat]) is applied WITHOUT building the grouped rows:
    y = (feat . Wf^T)[idx] + rel_xyz . Wx^T  (linearity of the convolution; csrc/group_first.hip)."""
    flat = []
    for layer in layers:
        W, gamma, beta, rmean, rvar = layer[:5]
        bias = layer[5] if len(layer) > 5 else None
        nbt = layer[6] if len(layer) > 6 and training else None
        if not training and bias is not None:
            rmean = rmean - bias.detach()      # eval: BN sees y + bias
        flat += [W, gamma, beta, rmean, rvar, bias, nbt]
    gx = gc = None
    if geo is not None and torch.is_grad_enabled():
        gx = geo[0] if geo[0].requires_grad else None
        gc = geo[1] if geo[1].requires_grad else None
    # (a conv bias in front of the BN only shifts the running mean: folded into demf_bn_finalize)
    return _SharedMLPPool.apply(x, ns, training, eps, momentum, geo, gx, gc, *flat)


class _L2NormRows(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, "x")
        R, C = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(R, dtype=torch.float32, device=x.device)
        _ffi.call("demf_l2norm_rows_fwd", R, C, _p(x), _p(y), _p(norm), _stream())
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _ffi.call("demf_l2norm_rows_bwd", y.shape[0], y.shape[1], _p(y), _p(norm), _p(dy), _p(dx), _stream())
        return dx


def l2norm_rows(x):
    """x (R,C) -> x / ||x||_2 per row (VoteModule norm_feats), one kernel each way."""
    return _L2NormRows.apply(x)


class _VoteCombine(Function):
    @staticmethod
    def forward(ctx, rows, votes, seed_xyz):
        _chk(rows, "rows")
        _chk(votes, "votes")
        _chk(seed_xyz, "seed_xyz")
        R, C = rows.shape
        assert votes.shape == (R, C + 3) and seed_xyz.numel() == 3 * R
        vote_xyz = torch.empty_like(seed_xyz)
        y = torch.empty_like(rows)
        norm = torch.empty(R, dtype=torch.float32, device=rows.device)
        _ffi.call("demf_vote_combine_fwd", R, C, _p(rows), _p(votes), _p(seed_xyz), _p(vote_xyz), _p(y),
                  _p(norm), _stream())
        ctx.save_for_backward(y, norm)
        return vote_xyz, y

    @staticmethod
    @once_differentiable
    def backward(ctx, dxyz, dy):
        y, norm = ctx.saved_tensors
        R, C = y.shape
        dxyz = None if dxyz is None else dxyz.contiguous()
        dy = None if dy is None else dy.contiguous()
        dvotes = torch.empty((R, C + 3), dtype=torch.float32, device=y.device)
        drows = torch.empty_like(y)
        _ffi.call("demf_vote_combine_bwd", R, C, _p(y), _p(norm), _p(dy), _p(dxyz), _p(dvotes),
                  _p(drows), _stream())
        return drows, dvotes, (dxyz if ctx.needs_input_grad[2] else None)


def vote_combine(rows, votes, seed_xyz):
    """VoteModule tail in one kernel each way: rows (R,C) seed features, votes (R,3+C) conv_out rows,
    seed_xyz (B,N,3) -> (vote_xyz = seed_xyz + votes[:, :3], l2-normalised rows of rows + votes[:, 3:])."""
    return _VoteCombine.apply(rows.contiguous(), votes.contiguous(), seed_xyz.contiguous())


# --------------------------------------------------------------------------
# Fused head losses (DeMFVoteHead._loss on the raw conv-head rows)
# --------------------------------------------------------------------------
import ctypes as _ct

HEAD_LOSS_NAMES = ("objectness_loss", "dir_class_loss", "dir_res_loss", "size_res_loss",
                   "center_loss", "semantic_loss", "iou_loss")


class _HeadLoss(Function):
    @staticmethod
    def forward(ctx, cls_rows, reg_rows, base, hyper, center_t, size_t, dir_class_t, dir_res_t,
                sem_t, obj_t, obj_w, box_w):
        for t, n in ((cls_rows, "cls"), (reg_rows, "reg"), (base, "base"), (center_t, "center_t"),
                     (size_t, "size_t"), (dir_res_t, "dir_res_t"), (obj_w, "obj_w"), (box_w, "box_w")):
            _chk(t, n)
        for t, n in ((dir_class_t, "dir_class_t"), (sem_t, "sem_t"), (obj_t, "obj_t")):
            _chk(t, n, torch.int64)
        R = cls_rows.shape[0]
        assert cls_rows.shape[1] == 12 and reg_rows.shape[1] == 30
        hp = (_ct.c_float * 12)(*hyper)
        out = zeros(7, cls_rows.device)
        _ffi.call("demf_head_loss_fwd", R, 12, 10, hp, _p(cls_rows), _p(reg_rows), _p(base),
                  _p(center_t), _p(size_t), _p(dir_class_t), _p(dir_res_t), _p(sem_t), _p(obj_t),
                  _p(obj_w), _p(box_w), _p(out), _stream())
        ctx.save_for_backward(cls_rows, reg_rows, base, center_t, size_t, dir_class_t, dir_res_t,
                              sem_t, obj_t, obj_w, box_w)
        ctx.hyper = tuple(hyper)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        cls_rows, reg_rows, base = ctx.saved_tensors[:3]
        rest = ctx.saved_tensors[3:]
        R = cls_rows.shape[0]
        hp = (_ct.c_float * 12)(*ctx.hyper)
        gout = grad_out.contiguous()
        gc, gr, gb = torch.empty_like(cls_rows), torch.empty_like(reg_rows), torch.empty_like(base)
        _ffi.call("demf_head_loss_bwd", R, 12, 10, hp, _p(cls_rows), _p(reg_rows), _p(base),
                  *[_p(t) for t in rest], _p(gout), _p(gc), _p(gr), _p(gb), _stream())
        return (gc, gr, gb) + (None,) * 9


def head_loss(cls_rows, reg_rows, base_xyz, hyper12, center_t, size_t, dir_class_t, dir_res_t,
              sem_t, obj_t, obj_w, box_w):
    """-> (7,) tensor of reduction='sum' losses in HEAD_LOSS_NAMES order; rows are (B*Q, .)."""
    return _HeadLoss.apply(cls_rows, reg_rows, base_xyz, hyper12, center_t, size_t, dir_class_t,
                           dir_res_t, sem_t, obj_t, obj_w, box_w)


class _VoteLoss(Function):
    @staticmethod
    def forward(ctx, vote_points, seed_points, seed_indices, masks, vote_targets, gt_per_seed,
                dst_weight):
        _chk(vote_points, "vote_points")
        _chk(seed_points, "seed_points")
        _chk(seed_indices, "seed_indices", torch.int64)
        _chk(masks, "vote_target_masks", torch.int64)
        _chk(vote_targets, "vote_targets")
        B, S, _ = seed_points.shape
        N = masks.shape[1]
        # (the denominator gather(masks, 1, seed_indices).sum() is counted inside the kernel)
        msum = torch.empty(1, dtype=torch.float32, device=vote_points.device)
        out = zeros(1, vote_points.device)
        _ffi.call("demf_vote_loss_fwd", B, S, N, int(gt_per_seed), float(dst_weight), _p(seed_points),
                  _p(vote_points), _p(seed_indices), _p(masks), _p(vote_targets), _p(msum),
                  _p(out), _stream())
        ctx.save_for_backward(vote_points, seed_points, seed_indices, masks, vote_targets, msum)
        ctx.meta = (B, S, N, int(gt_per_seed), float(dst_weight))
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        vote_points, seed_points, seed_indices, masks, vote_targets, msum = ctx.saved_tensors
        B, S, N, gps, w = ctx.meta
        g = grad_out.reshape(1).contiguous()
        gv = torch.empty_like(vote_points)
        _ffi.call("demf_vote_loss", B, S, N, gps, w, _p(seed_points), _p(vote_points),
                  _p(seed_indices), _p(masks), _p(vote_targets), _p(msum), _p(g), None, _p(gv),
                  _stream())
        return gv, None, None, None, None, None, None


def vote_loss(vote_points, seed_points, seed_indices, vote_target_masks, vote_targets, gt_per_seed,
              dst_weight):
    """VoteModule.get_loss (vote_per_seed == 1) as one kernel each way."""
    return _VoteLoss.apply(vote_points.contiguous(), seed_points.contiguous(), seed_indices.contiguous(),
                           vote_target_masks.contiguous(), vote_targets.contiguous(), gt_per_seed,
                           dst_weight)


def gt_prep(gt, labels_padded, num_dir_bins):
    """Everything the target kernels need from the padded ground truth alone - gt (B,G,7), labels
    (B,G) int64 with -1 on padding slots - in ONE launch (demf_gt_prep) instead of ~20 element-wise
    ones: dict(cs, sn = cos / sin(-yaw); dir_class, dir_res = bbox_coder.angle2class(yaw);
    valid (bool view of valid_u8), lab = labels clamped at 0, center = gravity centres (B,G,3))."""
    _chk(gt, "gt")
    _chk(labels_padded, "labels", torch.int64)
    B, G = gt.shape[:2]
    dev = gt.device
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    out = dict(cs=f(B, G), sn=f(B, G), dir_class=torch.empty((B, G), dtype=torch.int64, device=dev),
               dir_res=f(B, G), valid_u8=torch.empty((B, G), dtype=torch.uint8, device=dev),
               lab=torch.empty((B, G), dtype=torch.int64, device=dev), center=f(B, G, 3))
    _ffi.call("demf_gt_prep", B, G, int(num_dir_bins), _p(gt), _p(labels_padded), _p(out["cs"]),
              _p(out["sn"]), _p(out["dir_class"]), _p(out["dir_res"]), _p(out["valid_u8"]),
              _p(out["lab"]), _p(out["center"]), _stream())
    out["valid"] = out["valid_u8"].view(torch.bool)
    return out


def pad_gt_lists(boxes, labels, G, out=None):
    """list of (n_b, >=7) fp32 device box rows, list of (n_b,) int64 device labels -> (gt (B,G,7),
    labels (B,G) int64 with -1 on padding slots, valid (B,G) bool): demf_pad_gt - one launch, the per-
    scene pointers and counts travel by value, nothing is uploaded.  An empty scene gets the reference's
    all-zero fake box with label 0 (class_agnostic_vote_head.py:766-773).  B <= 32.
    ``out`` = (gt, labels) of an earlier call: written in place (the static buffers of a captured step)."""
    B = len(boxes)
    dev = boxes[0].device
    keep = []
    for i in range(B):
        b, l = boxes[i], labels[i]
        if b.dtype != torch.float32 or not b.is_contiguous():
            b = b.float().contiguous()
        if l.dtype != torch.int64 or not l.is_contiguous():
            l = l.long().contiguous()
        keep.append((b, l))
    counts = (ctypes.c_int * B)(*[int(b.shape[0]) for b, _ in keep])
    dims = (ctypes.c_int * B)(*[int(b.shape[1]) if b.dim() == 2 else 7 for b, _ in keep])
    bp = (ctypes.c_void_p * B)(*[b.data_ptr() if b.numel() else None for b, _ in keep])
    lp = (ctypes.c_void_p * B)(*[l.data_ptr() if l.numel() else None for _, l in keep])
    if out is not None:
        gt, lab = out
        assert tuple(gt.shape) == (B, G, 7) and tuple(lab.shape) == (B, G) and gt.is_contiguous() and \
            lab.is_contiguous() and gt.dtype == torch.float32 and lab.dtype == torch.int64
    else:
        gt = torch.empty((B, G, 7), dtype=torch.float32, device=dev)
        lab = torch.empty((B, G), dtype=torch.int64, device=dev)
    valid = torch.empty((B, G), dtype=torch.uint8, device=dev)
    _ffi.call("demf_pad_gt", B, int(G), ctypes.addressof(counts), ctypes.addressof(dims), ctypes.addressof(bp),
              ctypes.addressof(lp), _p(gt), _p(lab), _p(valid), _stream())
    return gt, lab, valid.view(torch.bool)


@torch.no_grad()
def query_pos_rows(reg_rows, base_xyz):
    """The decoder layer's position-embedding input from a prediction head's raw regression rows:
    reg_rows (B,Q,nreg) point-major conv_reg output, base_xyz (B,Q,3) -> (B*Q, 8) rows
    [base + reg[:3] (centre) | reg[3:6] (size) | 0 0], detached - what the reference builds as
    ``torch.cat([decode_res['center'], decode_res['size']], -1).detach().clone()``
    (class_agnostic_vote_head.py:497-498), zero-padded to the 8 columns the first GEMM stages.  One launch."""
    _chk(base_xyz, "base_xyz")
    B, Q, nreg = reg_rows.shape
    reg = reg_rows if reg_rows.is_contiguous() else reg_rows.contiguous()
    _chk(reg, "reg_rows")
    out = torch.empty((B * Q, 8), dtype=torch.float32, device=reg.device)
    _ffi.call("demf_query_pos_rows", B * Q, nreg, _p(reg), _p(base_xyz), _p(out), _stream())
    return out


class _LossTotal(Function):
    @staticmethod
    def forward(ctx, vote, *vecs):
        n = len(vecs)
        vecs = [v.contiguous() for v in vecs]
        out = torch.empty(8, dtype=torch.float32, device=vecs[0].device)
        tab = (ctypes.c_void_p * n)(*[v.data_ptr() for v in vecs])
        _ffi.call("demf_loss_total", n, ctypes.addressof(tab), _p(vote), _p(out), _stream())
        ctx.n, ctx.has_vote = n, vote is not None
        return out

    @staticmethod
    def backward(ctx, g8):
        g8 = g8.contiguous()
        gv = torch.empty((ctx.n, 7), dtype=torch.float32, device=g8.device)
        gvote = torch.empty(1, dtype=torch.float32, device=g8.device) if ctx.has_vote else None
        _ffi.call("demf_loss_total_bwd", ctx.n, _p(g8), _p(gv), _p(gvote), _stream())
        return (gvote,) + tuple(gv[i] for i in range(ctx.n))


def loss_total(vecs, vote=None):
    """Tail of DeMFVoteHead.loss: (7,) loss vectors of the decode results -> (8,) = [their mean | sum of the
    mean + vote loss]; one launch each way instead of add / div / sum / add and their backward nodes."""
    assert 1 <= len(vecs) <= 4
    for v in vecs:
        _chk(v, "loss vector")
    return _LossTotal.apply(None if vote is None else vote.reshape(1), *vecs)


def target_weights(objectness_masks, objectness_targets):
    """-> (objectness_weights, box_loss_weights): each tensor divided by (its sum + 1e-6)
    (class_agnostic_vote_head.py:797-816), one launch."""
    _chk(objectness_masks, "objectness_masks")
    _chk(objectness_targets, "objectness_targets", torch.int64)
    ow = torch.empty_like(objectness_masks)
    bw = torch.empty_like(objectness_masks)
    _ffi.call("demf_target_weights", objectness_masks.numel(), _p(objectness_masks),
              _p(objectness_targets), _p(ow), _p(bw), _stream())
    return ow, bw


def vote_targets(points, gt, valid, prep=None):
    """points (B,N,>=3), gt (B,G,7), valid (B,G) bool -> (vote_targets (B,N,9), masks (B,N) int64):
    the per-point half of DeMFVoteHead.get_targets (class_agnostic_vote_head.py:828-858).
    ``prep``: the result of ``gt_prep`` on the same boxes (saves the element-wise launches)."""
    _chk(points, "points")
    _chk(gt, "gt")
    B, N, stride = points.shape
    G = gt.shape[1]
    if prep is not None:
        cs, sn, v = prep["cs"], prep["sn"], prep["valid_u8"]
    else:
        yaw = gt[..., 6]
        cs, sn = torch.cos(-yaw).contiguous(), torch.sin(-yaw).contiguous()
        v = valid.to(torch.uint8).contiguous()
    vt = torch.empty((B, N, 9), dtype=torch.float32, device=points.device)
    mask = torch.empty((B, N), dtype=torch.int64, device=points.device)
    _ffi.call("demf_vote_targets", B, N, stride, G, _p(points), _p(gt), _p(cs), _p(sn), _p(v),
              _p(vt), _p(mask), _stream())
    return vt, mask


def proposal_targets(agg, gt, lab, valid, dir_class, dir_res, with_rot, pos_thr, neg_thr, res_scale,
                     prep=None):
    """The per-proposal half of DeMFVoteHead.get_targets (class_agnostic_vote_head.py:877-934) in
    one kernel.  agg (B,Q,3), gt (B,G,7), lab / valid (B,G), dir_class / dir_res = angle2class(yaw).
    -> dict(center, size, dir_class, dir_res, dir, mask, distance, objectness, objectness_masks)"""
    _chk(agg, "agg")
    _chk(gt, "gt")
    B, Q, _ = agg.shape
    G = gt.shape[1]
    dev = agg.device
    if prep is not None:
        cs, sn, valid_u8 = prep["cs"], prep["sn"], prep["valid_u8"]
    else:
        yaw = gt[..., 6]
        cs, sn = torch.cos(-yaw).contiguous(), torch.sin(-yaw).contiguous()
        valid_u8 = valid.to(torch.uint8).contiguous()
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    l = lambda *s: torch.empty(s, dtype=torch.int64, device=dev)
    out = dict(center=f(B, Q, 3), size=f(B, Q, 3), dir_class=l(B, Q), dir_res=f(B, Q), dir=f(B, Q),
               mask=l(B, Q), distance=f(B, Q, 6), objectness=l(B, Q), objectness_masks=f(B, Q))
    _ffi.call("demf_proposal_targets", B, Q, G, int(bool(with_rot)), float(pos_thr), float(neg_thr),
              float(res_scale), _p(agg), _p(gt), _p(cs), _p(sn), _p(dir_class.contiguous()),
              _p(dir_res.contiguous()), _p(lab.contiguous()), _p(valid_u8),
              _p(out["center"]), _p(out["size"]), _p(out["dir_class"]), _p(out["dir_res"]),
              _p(out["dir"]), _p(out["mask"]), _p(out["distance"]), _p(out["objectness"]),
              _p(out["objectness_masks"]), _stream())
    return out


def box_extent_count(points, boxes7):
    """points (B,N,>=3), boxes7 (B,K,7) gravity-centre boxes -> (boxes_bottom (B,K,7),
    extent (B,K,6), count (B,K) int32): see demf_box_extent_count."""
    _chk(points, "points")
    _chk(boxes7, "boxes7")
    B, N, stride = points.shape
    K = boxes7.shape[1]
    yaw = boxes7[..., 6]
    c, s = torch.cos(yaw).contiguous(), torch.sin(yaw).contiguous()
    out = torch.empty_like(boxes7)
    ext = torch.empty((B, K, 6), dtype=torch.float32, device=points.device)
    cnt = torch.empty((B, K), dtype=torch.int32, device=points.device)
    _ffi.call("demf_box_extent_count", B, N, stride, K, _p(points), _p(boxes7), _p(c), _p(s),
              _p(out), _p(ext), _p(cnt), _stream())
    return out, ext, cnt


def aligned_nms(extent, scores, classes, valid, iou_thr):
    """Class-aware greedy NMS on axis-aligned extents (B,K,6) per scene -> keep (B,K) bool."""
    _chk(extent, "extent")
    _chk(scores, "scores")
    _chk(classes, "classes", torch.int64)
    B, K = scores.shape
    keep = torch.empty((B, K), dtype=torch.uint8, device=scores.device)
    _ffi.call("demf_aligned_nms", B, K, float(iou_thr), _p(extent), _p(scores), _p(classes),
              _p(valid.to(torch.uint8).contiguous()), _p(keep), _stream())
    return keep.bool()


def sa_index_chain(N, level_indices):
    """[arange(N) per scene] + every SA level's samples as int64 indices into the input cloud (the
    ``sa_indices`` of PointNet2SASSG.forward) from the levels' int32 FPS indices, one launch."""
    B, dev = level_indices[0].shape[0], level_indices[0].device
    for t in level_indices:
        _chk(t, "fps indices", torch.int32)
    outs = [torch.empty((B, N), dtype=torch.int64, device=dev)] + \
        [torch.empty(tuple(t.shape), dtype=torch.int64, device=dev) for t in level_indices]
    n = len(level_indices)
    idx = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in level_indices])
    ms = (ctypes.c_int * max(n, 1))(*[t.shape[1] for t in level_indices])
    op = (ctypes.c_void_p * (n + 1))(*[t.data_ptr() for t in outs])
    _ffi.call("demf_sa_index_chain", B, N, n, ctypes.addressof(idx), ctypes.addressof(ms), ctypes.addressof(op),
              _stream())
    return outs


def split_points(points):
    """(B,N,3+C) -> (xyz (B,N,3), feature rows (B,N,C) or None), contiguous, one launch."""
    _chk(points, "points")
    B, N, D = points.shape
    C = D - 3
    pts = points if points.is_contiguous() else points.contiguous()
    xyz = torch.empty((B, N, 3), dtype=torch.float32, device=points.device)
    feat = torch.empty((B, N, C), dtype=torch.float32, device=points.device) if C > 0 else None
    _ffi.call("demf_split_points", B * N, C, _p(pts), _p(xyz), _p(feat), _stream())
    return xyz, feat


def pyramid_to_tokens(mlvl_feats, zero_mask=None, bf16=False, out=None):
    """list of (B,C,H_l,W_l) -> (B, sum H_l W_l, C) channels-last tokens (no gradient: the image
    pyramid is an input of the hot path).  ``zero_mask`` (B,S) bool: tokens to write as zeros
    (image padding), fused into the transposes.  ``bf16``: bf16 token rows (the bf16 compute mode's
    sample-then-project attention gathers them with demf_msda_*_bf16: half the bytes).  ``out``: an
    existing token buffer of that shape / dtype to write into (the static input of a captured step)."""
    B, C = mlvl_feats[0].shape[:2]
    sizes = [f.shape[2] * f.shape[3] for f in mlvl_feats]
    S = sum(sizes)
    bf16 = bool(bf16) and C % 4 == 0 and len(sizes) <= 8
    if out is None:
        out = torch.empty((B, S, C), dtype=torch.bfloat16 if bf16 else torch.float32, device=mlvl_feats[0].device)
    elif not (out.is_contiguous() and tuple(out.shape) == (B, S, C) and out.device == mlvl_feats[0].device
              and out.dtype == (torch.bfloat16 if bf16 else torch.float32)):
        raise ValueError("pyramid_to_tokens: `out` must be a contiguous (B,S,C) tensor of the token dtype")
    m = None if zero_mask is None else \
        (zero_mask if zero_mask.dtype == torch.uint8 else zero_mask.to(torch.uint8)).contiguous()
    for f in mlvl_feats:
        _chk(f, "feature map")
    if len(sizes) <= 8:                  # every level in one launch
        srcs = (ctypes.c_void_p * len(sizes))(*[f.data_ptr() for f in mlvl_feats])
        hws = (ctypes.c_int * len(sizes))(*sizes)
        _ffi.call("demf_pyramid_to_tokens_bf16" if bf16 else "demf_pyramid_to_tokens", B, C, S, len(sizes),
                  ctypes.addressof(srcs), ctypes.addressof(hws), _p(m), _p(out), _stream())
        return out
    row0 = 0
    for f, hw in zip(mlvl_feats, sizes):
        _ffi.call("demf_nchw_to_tokens", B, C, hw, S, row0, _p(f), _p(m), _p(out), _stream())
        row0 += hw
    return out


def msda_sample_then_project(tokens, keep4, spatial_shapes, level_start_index, sampling_locations,
                             attention_weights, weight, bias):
    """Multi-scale deformable attention with the value projection applied AFTER sampling:

        sum_s w_s * bilinear(keep * (W x + b))[s]  ==  W_h (sum_s w_s * bilinear(keep * x)[s])
                                                       + b_h  sum_s w_s * bilinear(keep)[s]

    (everything is linear in x).  ``tokens`` (B,S,C) = keep * x, ``keep4`` (B,S,4) = [keep,0,0,0],
    locations (B,Q,H,L,P,2), weights (B,Q,H,L,P), ``weight`` (H*Dh, C) / ``bias`` (H*Dh) = the
    module's value_proj.  With Q*L*P*4 sampled corners << S tokens this replaces a (B*S, C) x (C, C)
    GEMM, its masking passes, the (B,S,C) value / grad-value buffers and the fp32 atomics of the
    backward by two gathers on the same kernels (H=1, Dh=C) plus H small batched GEMMs.
    No gradient flows to the tokens (they are the frozen image stream's output).  -> (B,Q,H*Dh)"""
    B, S, C = tokens.shape
    _, Q, H, L, P, _ = sampling_locations.shape
    Dh = weight.shape[0] // H
    loc = sampling_locations.reshape(B, Q * H, 1, L, P, 2)
    aw = attention_weights.reshape(B, Q * H, 1, L, P)
    if tokens.dtype == torch.bfloat16:            # (the unfused module path gathers fp32 rows)
        tokens = tokens.float()
    z = MultiScaleDeformableAttnFunction.apply(tokens.view(B, S, 1, C), spatial_shapes,
                                               level_start_index, loc, aw)          # (B,Q*H,C)
    ksum = MultiScaleDeformableAttnFunction.apply(keep4.view(B, S, 1, 4), spatial_shapes,
                                                  level_start_index, loc, aw)[..., 0]  # (B,Q*H)
    zh = z.view(B * Q, H, C).transpose(0, 1)                                  # (H, B*Q, C)
    out = torch.baddbmm(                                                      # (H, B*Q, Dh)
        (ksum.view(B * Q, H).t().unsqueeze(-1) * bias.view(H, 1, Dh)),
        zh, weight.view(H, Dh, C).transpose(1, 2))
    return out.transpose(0, 1).reshape(B, Q, H * Dh)
