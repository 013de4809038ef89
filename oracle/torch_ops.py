"""CPU ORACLE (test infrastructure) - torch.autograd wrappers over oracle/kernels.py with
the mmdet3d.ops / mmcv.ops call signatures, so CPU restatements of the reference modules
(and the real reference files under the stub shim) can run end-to-end on the oracle."""
import numpy as np
import torch
from torch.autograd import Function

from . import kernels as K


def _np(t):
    return t.detach().cpu().numpy()


class _FPS(Function):
    @staticmethod
    def forward(ctx, xyz, m):
        idx = torch.from_numpy(K.fps(_np(xyz.float()), int(m)))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g=None):
        return None, None


def furthest_point_sample(points_xyz, num_points):
    return _FPS.apply(points_xyz, num_points)


class _BallQuery(Function):
    @staticmethod
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        idx = torch.from_numpy(K.ball_query(min_radius, max_radius, sample_num, _np(xyz.float()),
                                            _np(center_xyz.float())))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, g=None):
        return None, None, None, None, None


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    return _BallQuery.apply(min_radius, max_radius, sample_num, xyz, center_xyz)


class _Group(Function):
    @staticmethod
    def forward(ctx, features, indices):
        ctx.save_for_backward(indices)
        ctx.N = features.shape[2]
        return torch.from_numpy(K.group_points_fwd(_np(features), _np(indices)))

    @staticmethod
    def backward(ctx, grad_out):
        (indices,) = ctx.saved_tensors
        return torch.from_numpy(K.group_points_bwd(_np(grad_out), _np(indices), ctx.N)), None


def grouping_operation(features, indices):
    if features.dtype == torch.float64:  # noise-floor runs: exact gather, autograd by torch
        B, C, N = features.shape
        flat = indices.reshape(B, 1, -1).expand(-1, C, -1).long()
        return torch.gather(features, 2, flat).reshape(B, C, *indices.shape[1:])
    return _Group.apply(features, indices)


def gather_points(features, indices):
    return grouping_operation(features, indices.unsqueeze(-1)).squeeze(-1)


class _ThreeNN(Function):
    @staticmethod
    def forward(ctx, target, source):
        d2, idx = K.three_nn(_np(target), _np(source))
        idx = torch.from_numpy(idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(torch.from_numpy(d2)), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


def three_nn(target, source):
    if target.dtype == torch.float64:
        d, idx = _ThreeNN.apply(target.float(), source.float())
        B, n, _ = target.shape
        nb = torch.gather(source.unsqueeze(1).expand(-1, n, -1, -1), 2,
                          idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return (nb - target.unsqueeze(2)).pow(2).sum(-1).sqrt(), idx
    return _ThreeNN.apply(target, source)


class _Interp(Function):
    @staticmethod
    def forward(ctx, features, indices, weight):
        ctx.save_for_backward(indices, weight)
        ctx.m = features.shape[2]
        return torch.from_numpy(K.three_interpolate_fwd(_np(features), _np(indices), _np(weight)))

    @staticmethod
    def backward(ctx, grad_out):
        indices, weight = ctx.saved_tensors
        g = K.three_interpolate_bwd(_np(grad_out), _np(indices), _np(weight), ctx.m)
        return torch.from_numpy(g), None, None


def three_interpolate(features, indices, weight):
    if features.dtype == torch.float64:
        g = grouping_operation(features, indices)          # (B,C,n,3)
        return (g * weight.unsqueeze(1)).sum(-1)
    return _Interp.apply(features, indices, weight)


class MultiScaleDeformableAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, shapes, lsi, loc, attw, im2col_step=64):
        ctx.save_for_backward(value, shapes, lsi, loc, attw)
        dt = np.float64 if value.dtype == torch.float64 else np.float32
        return torch.from_numpy(K.msda_fwd(_np(value), _np(shapes), _np(lsi), _np(loc), _np(attw), dt))

    @staticmethod
    def backward(ctx, grad_out):
        value, shapes, lsi, loc, attw = ctx.saved_tensors
        dt = np.float64 if value.dtype == torch.float64 else np.float32
        gv, gl, ga = K.msda_bwd(_np(value), _np(shapes), _np(lsi), _np(loc), _np(attw),
                                _np(grad_out), dt)
        return torch.from_numpy(gv), None, None, torch.from_numpy(gl), torch.from_numpy(ga), None
