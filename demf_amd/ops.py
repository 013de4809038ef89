"""Operator layer: the Python API the reference imports from mmdet3d.ops / mmcv.ops,
re-implemented on libdemf_hip.so (gfx950 HIP kernels).

Signatures, argument meaning and error behaviour mirror the upstream operators the
reference calls (cited per function); tensors stay PyTorch-ROCm tensors and only
raw device pointers + the current HIP stream cross the C ABI.  There is no CPU or
eager fallback: a CPU tensor or a missing library raises.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _ffi

__all__ = [
    "furthest_point_sample", "ball_query", "grouping_operation", "gather_points",
    "three_nn", "three_interpolate", "MultiScaleDeformableAttnFunction",
    "group_concat_cl", "gather_rows_cl", "three_interpolate_cl", "maxpool_ns",
]


def _stream():
    return torch.cuda.current_stream().cuda_stream


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
        if N < 64 or N > 24 * 1024:
            temp = torch.empty((B, N), dtype=torch.float32, device=points_xyz.device)
        _ffi.call("demf_fps_f32", B, N, num_points, _p(points_xyz), _p(temp), _p(idx),
                  _stream())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g=None):
        return None, None


def furthest_point_sample(points_xyz, num_points):
    """(B,N,3) f32 -> (B,num_points) i32.  Reference use:
    demf/modeling/heads/class_agnostic_vote_head.py:13,429-430."""
    return _FurthestPointSampling.apply(points_xyz, num_points)


class _BallQuery(Function):
    @staticmethod
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        _chk(xyz, "xyz")
        _chk(center_xyz, "center_xyz")
        assert min_radius < max_radius
        B, N, _ = xyz.shape
        M = center_xyz.shape[1]
        idx = torch.empty((B, M, sample_num), dtype=torch.int32, device=xyz.device)
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
        grad_value = torch.zeros_like(value)
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
    def forward(ctx, xyz, center, feat, idx, radius, normalize_xyz, ldo, xyz_col, feat_col):
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
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, M, ns, C, ldo, xyz_col, feat_col, float(radius),
                    int(bool(normalize_xyz)))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        B, N, M, ns, C, ldo, xyz_col, feat_col, radius, norm = ctx.dims
        want_feat = C > 0 and ctx.needs_input_grad[2]
        want_xyz = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gfeat = gxyz = gcenter = None
        if want_feat or want_xyz:
            grad_out = grad_out.contiguous()
            kw = dict(dtype=grad_out.dtype, device=grad_out.device)
            if want_feat:
                gfeat = torch.zeros((B, N, C), **kw)
            if want_xyz:
                gxyz = torch.zeros((B, N, 3), **kw)
                gcenter = torch.zeros((B, M, 3), **kw)
            _ffi.call("demf_group_concat_cl_bwd", B, N, M, ns, C, ldo, xyz_col, feat_col,
                      radius, norm, _p(grad_out), _p(idx), _p(gfeat), _p(gxyz), _p(gcenter),
                      _stream())
        return (gxyz if ctx.needs_input_grad[0] else None,
                gcenter if ctx.needs_input_grad[1] else None,
                gfeat, None, None, None, None, None, None)


def group_concat_cl(xyz, center, feat, idx, radius, normalize_xyz, ldo=None, xyz_col=None,
                    feat_col=0):
    """Fused QueryAndGroup on point-major features.
    xyz (B,N,3), center (B,M,3), feat (B,N,C)|None, idx (B,M,ns) -> (B,M,ns,ldo) rows
    ``[feat | (xyz_j - center)/radius | 0-pad]`` (column order chosen by the caller)."""
    C = 0 if feat is None else feat.shape[2]
    if xyz_col is None:
        xyz_col = feat_col + C
    if ldo is None:
        ldo = max(xyz_col + 3, feat_col + C)
    return _GroupConcatCL.apply(xyz, center, feat, idx, radius, normalize_xyz, ldo, xyz_col,
                                feat_col)


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
        g = torch.zeros((B, ctx.N, C), dtype=grad_out.dtype, device=grad_out.device)
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
        g = torch.zeros((B, ctx.m, C), dtype=grad_out.dtype, device=grad_out.device)
        _ffi.call("demf_three_interpolate_cl_bwd", B, ctx.m, n, C, C, 0, _p(grad_out), _p(idx),
                  _p(weight), _p(g), _stream())
        return g, None, None


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
