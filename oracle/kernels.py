"""CPU ORACLE (test infrastructure, NOT a product path) - numpy front-end.

ctypes bindings of ``oracle/csrc/demf_oracle.c``: the C restatement of the native
operators the reference path reaches through mmdet3d.ops / mmcv.ops (see that
file's header for the per-operator citations and the parity-pinning status).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdemf_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the C oracle with the committed Makefile (gcc, seconds)."""
    import hashlib
    src = os.path.join(_HERE, "csrc", "demf_oracle.c")
    with open(src, "rb") as f, open(os.path.join(_HERE, "Makefile"), "rb") as m:
        digest = hashlib.sha256(f.read() + m.read()).hexdigest()
    stamp = _SO + ".sha256"
    stale = not (os.path.exists(_SO) and os.path.exists(stamp)) or open(stamp).read().strip() != digest
    if force or stale:
        subprocess.run(["make", "-B", "-C", _HERE], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(digest)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def fps(xyz, m):
    """xyz (B,N,3) f32 -> idx (B,m) i32.  class_agnostic_vote_head.py:429-430."""
    xyz = _c(xyz, np.float32)
    B, N, _ = xyz.shape
    idx = np.zeros((B, m), np.int32)
    lib().oracle_fps(B, N, m, _p(xyz, _f32p), _p(idx, _i32p))
    return idx


def ball_query(min_radius, max_radius, nsample, xyz, center):
    """-> idx (B,M,nsample) i32.  QueryAndGroup, class_agnostic_vote_head.py:383."""
    xyz, center = _c(xyz, np.float32), _c(center, np.float32)
    B, N, _ = xyz.shape
    M = center.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    lib().oracle_ball_query(B, N, M, ctypes.c_float(min_radius), ctypes.c_float(max_radius),
                            nsample, _p(center, _f32p), _p(xyz, _f32p), _p(idx, _i32p))
    return idx


def group_points_fwd(feat, idx):
    """feat (B,C,N), idx (B,M,ns) -> (B,C,M,ns)."""
    feat, idx = _c(feat, np.float32), _c(idx, np.int32)
    B, C, N = feat.shape
    J = int(np.prod(idx.shape[1:]))
    out = np.empty((B, C) + idx.shape[1:], np.float32)
    lib().oracle_group_points_fwd(B, C, N, J, _p(feat, _f32p), _p(idx, _i32p), _p(out, _f32p))
    return out


def group_points_bwd(gout, idx, N):
    gout, idx = _c(gout, np.float32), _c(idx, np.int32)
    B, C = gout.shape[:2]
    J = int(np.prod(idx.shape[1:]))
    gfeat = np.zeros((B, C, N), np.float32)
    lib().oracle_group_points_bwd(B, C, N, J, _p(gout, _f32p), _p(idx, _i32p), _p(gfeat, _f32p))
    return gfeat


def three_nn(target, source):
    """-> (dist2 (B,n,3) SQUARED, idx (B,n,3) i32).  PointFPModule, demf_votenet.py:56."""
    target, source = _c(target, np.float32), _c(source, np.float32)
    B, n, _ = target.shape
    m = source.shape[1]
    d2 = np.empty((B, n, 3), np.float32)
    idx = np.empty((B, n, 3), np.int32)
    lib().oracle_three_nn(B, n, m, _p(target, _f32p), _p(source, _f32p), _p(d2, _f32p),
                          _p(idx, _i32p))
    return d2, idx


def three_interpolate_fwd(feat, idx, w):
    feat, idx, w = _c(feat, np.float32), _c(idx, np.int32), _c(w, np.float32)
    B, C, m = feat.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), np.float32)
    lib().oracle_three_interpolate_fwd(B, C, m, n, _p(feat, _f32p), _p(idx, _i32p),
                                       _p(w, _f32p), _p(out, _f32p))
    return out


def three_interpolate_bwd(gout, idx, w, m):
    gout, idx, w = _c(gout, np.float32), _c(idx, np.int32), _c(w, np.float32)
    B, C, n = gout.shape
    gfeat = np.zeros((B, C, m), np.float32)
    lib().oracle_three_interpolate_bwd(B, C, n, m, _p(gout, _f32p), _p(idx, _i32p),
                                       _p(w, _f32p), _p(gfeat, _f32p))
    return gfeat


def _msda_args(value, shapes, lsi, loc, attw, dtype):
    value, loc, attw = _c(value, dtype), _c(loc, dtype), _c(attw, dtype)
    shapes, lsi = _c(shapes, np.int64), _c(lsi, np.int64)
    B, S, H, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    return value, shapes, lsi, loc, attw, (B, S, H, Dh, L, Q, P)


def msda_fwd(value, shapes, lsi, loc, attw, dtype=np.float32):
    """value (B,S,H,Dh), loc (B,Q,H,L,P,2), attw (B,Q,H,L,P) -> (B,Q,H*Dh).
    transformer.py:73 -> mmcv MultiScaleDeformableAttnFunction."""
    value, shapes, lsi, loc, attw, dims = _msda_args(value, shapes, lsi, loc, attw, dtype)
    B, S, H, Dh, L, Q, P = dims
    out = np.empty((B, Q, H * Dh), dtype)
    fp = _f32p if dtype == np.float32 else _f64p
    fn = lib().oracle_msda_f32_fwd if dtype == np.float32 else lib().oracle_msda_f64_fwd
    fn(B, S, H, Dh, L, Q, P, _p(value, fp), _p(shapes, _i64p), _p(lsi, _i64p), _p(loc, fp),
       _p(attw, fp), _p(out, fp))
    return out


def msda_bwd(value, shapes, lsi, loc, attw, gout, dtype=np.float32):
    """-> (grad_value, grad_loc, grad_attw)."""
    value, shapes, lsi, loc, attw, dims = _msda_args(value, shapes, lsi, loc, attw, dtype)
    B, S, H, Dh, L, Q, P = dims
    gout = _c(gout, dtype)
    gvalue = np.zeros_like(value)
    gloc = np.zeros_like(loc)
    gattw = np.zeros_like(attw)
    fp = _f32p if dtype == np.float32 else _f64p
    fn = lib().oracle_msda_f32_bwd if dtype == np.float32 else lib().oracle_msda_f64_bwd
    fn(B, S, H, Dh, L, Q, P, _p(value, fp), _p(shapes, _i64p), _p(lsi, _i64p), _p(loc, fp),
       _p(attw, fp), _p(gout, fp), _p(gvalue, fp), _p(gloc, fp), _p(gattw, fp))
    return gvalue, gloc, gattw
