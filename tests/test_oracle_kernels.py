"""Pins the C oracle (oracle/csrc/demf_oracle.c) against independent implementations.

The reference ships no tests or fixtures for these operators and its dependencies are
not installed, so the oracle is pinned against (a) brute-force numpy restatements of
the published algorithms on grid-snapped inputs where every fp32 operation is exact
(any evaluation order gives the same bits, and ties are plentiful), (b) size-independent
properties on random inputs, and (c) third-party pure-PyTorch implementations of
deformable attention (torch grid_sample; transformers' MultiScaleDeformableAttention,
itself the same formulation as mmcv's CPU fallback for this operator).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import PYRAMID, TINY_PYRAMID, msda_inputs, scene_points
from oracle import kernels as ok


# ----------------------------------------------------------------- FPS
def _np_fps(xyz, m):
    """Published algorithm + upstream tie rule, in float64 (exact on grid inputs)."""
    n = xyz.shape[0]
    bs = 1
    while bs * 2 <= n and bs < 1024:
        bs *= 2
    p = xyz.astype(np.float64)
    temp = np.full(n, 1e10)
    key = (np.arange(n) % bs) * (n + 1) + np.arange(n)  # (k mod bs, k) lexicographic
    out = [0]
    for _ in range(1, m):
        d = ((p - p[out[-1]]) ** 2).sum(1)
        temp = np.minimum(temp, d)
        cand = np.flatnonzero(temp == temp.max())
        out.append(int(cand[np.argmin(key[cand])]))
    return np.array(out, np.int32)


@pytest.mark.parametrize("n,m", [(64, 16), (100, 37), (512, 128), (1500, 200), (2500, 64)])
def test_fps_grid_exact(n, m):
    xyz = scene_points(2, n, seed=n, grid=8)  # coarse grid: many exact ties + duplicates
    got = ok.fps(xyz, m)
    for b in range(2):
        np.testing.assert_array_equal(got[b], _np_fps(xyz[b], m))


def test_fps_random_is_farthest():
    xyz = scene_points(1, 3000, seed=3)
    idx = ok.fps(xyz, 64)[0]
    p = xyz[0].astype(np.float64)
    assert idx[0] == 0 and len(set(idx.tolist())) == 64
    temp = np.full(len(p), np.inf)
    for j in range(1, 64):
        temp = np.minimum(temp, ((p - p[idx[j - 1]]) ** 2).sum(1))
        assert temp[idx[j]] >= temp.max() * (1 - 1e-5)


def test_fps_edge_cases():
    one = np.zeros((1, 1, 3), np.float32)
    assert ok.fps(one, 1).tolist() == [[0]]
    dup = np.ones((1, 70, 3), np.float32)  # all duplicates: every round is an all-way tie
    np.testing.assert_array_equal(ok.fps(dup, 5)[0], _np_fps(dup[0], 5))


# ----------------------------------------------------------------- ball query
def _np_ball_query(min_r, max_r, ns, xyz, center):
    p, c = xyz.astype(np.float64), center.astype(np.float64)
    out = np.zeros((len(c), ns), np.int32)
    for m in range(len(c)):
        d2 = ((p - c[m]) ** 2).sum(1)
        hit = np.flatnonzero((d2 == 0) | ((d2 >= min_r ** 2) & (d2 < max_r ** 2)))[:ns]
        if len(hit):
            out[m] = hit[0]
            out[m, :len(hit)] = hit
    return out


@pytest.mark.parametrize("min_r,max_r,ns", [(0.0, 0.5, 16), (0.0, 0.25, 64), (0.25, 0.75, 8)])
def test_ball_query_grid_exact(min_r, max_r, ns):
    xyz = scene_points(2, 700, seed=5, grid=8)
    center = xyz[:, ::9].copy()
    center[:, -1] = 50.0  # an empty ball -> zeros
    got = ok.ball_query(min_r, max_r, ns, xyz, center)
    for b in range(2):
        np.testing.assert_array_equal(got[b], _np_ball_query(min_r, max_r, ns, xyz[b], center[b]))


# ----------------------------------------------------------------- three_nn / interpolate
def test_three_nn_grid_exact():
    tgt = scene_points(2, 300, seed=7, grid=4)
    src = scene_points(2, 90, seed=8, grid=4)
    d2, idx = ok.three_nn(tgt, src)
    for b in range(2):
        dd = ((tgt[b][:, None].astype(np.float64) - src[b][None].astype(np.float64)) ** 2).sum(-1)
        order = np.argsort(dd, axis=1, kind="stable")[:, :3]  # earliest source wins ties
        np.testing.assert_array_equal(idx[b], order)
        np.testing.assert_array_equal(d2[b], np.take_along_axis(dd, order, 1).astype(np.float32))


def test_three_nn_fewer_than_three_sources():
    tgt = scene_points(1, 5, seed=1)
    src = scene_points(1, 2, seed=2)
    d2, idx = ok.three_nn(tgt, src)
    assert np.isinf(d2[0, :, 2]).all() and (idx[0, :, 2] == 0).all()


def test_group_gather_interpolate_vs_numpy():
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((2, 5, 40)).astype(np.float32)
    idx = rng.integers(0, 40, size=(2, 7, 3)).astype(np.int32)
    out = ok.group_points_fwd(feat, idx)
    for b in range(2):
        np.testing.assert_array_equal(out[b], feat[b][:, idx[b]])
    g = rng.standard_normal(out.shape).astype(np.float32)
    gf = ok.group_points_bwd(g, idx, 40)
    ref = np.zeros((2, 5, 40), np.float64)
    for b in range(2):
        for c in range(5):
            np.add.at(ref[b, c], idx[b].ravel(), g[b, c].ravel())
    np.testing.assert_allclose(gf, ref, rtol=1e-5, atol=1e-6)
    w = rng.uniform(0, 1, size=(2, 7, 3)).astype(np.float32)
    o = ok.three_interpolate_fwd(feat, idx, w)
    ref = np.einsum("bcnk,bnk->bcn", np.stack([feat[b][:, idx[b]] for b in range(2)]), w)
    np.testing.assert_allclose(o, ref, rtol=1e-5, atol=1e-6)
    go = rng.standard_normal(o.shape).astype(np.float32)
    gi = ok.three_interpolate_bwd(go, idx, w, 40)
    ref = np.zeros((2, 5, 40), np.float64)
    for b in range(2):
        for c in range(5):
            np.add.at(ref[b, c], idx[b].ravel(), (go[b, c][:, None] * w[b]).ravel())
    np.testing.assert_allclose(gi, ref, rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------- MSDA
def _grid_sample_msda(value, shapes, loc, attw):
    """Deformable-DETR's published formulation on torch.grid_sample."""
    B, S, H, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for l, (h, w) in enumerate(shapes):
        v = vals[l].flatten(2).transpose(1, 2).reshape(B * H, Dh, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros",
                                     align_corners=False))
    a = attw.transpose(1, 2).reshape(B * H, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, H * Dh, Q)
    return out.transpose(1, 2).contiguous()


@pytest.mark.parametrize("shapes,P,Dh", [(TINY_PYRAMID, 2, 8), (TINY_PYRAMID, 4, 32),
                                         ([(6, 7)], 3, 4), (PYRAMID, 2, 32)])
def test_msda_forward_vs_grid_sample_and_transformers(shapes, P, Dh):
    B, Q, H = 2, 19, 4
    value, shp, lsi, loc, attw = msda_inputs(B, Q, H, Dh, shapes, P, seed=11, dtype=np.float64)
    ref = _grid_sample_msda(torch.from_numpy(value), shapes, torch.from_numpy(loc),
                            torch.from_numpy(attw)).numpy()
    got64 = ok.msda_fwd(value, shp, lsi, loc, attw, dtype=np.float64)
    np.testing.assert_allclose(got64, ref, rtol=1e-10, atol=1e-10)
    got32 = ok.msda_fwd(value, shp, lsi, loc, attw, dtype=np.float32)
    np.testing.assert_allclose(got32, ref, rtol=1e-4, atol=1e-5)
    m = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
    hf = m.MultiScaleDeformableAttention()(torch.from_numpy(value), torch.from_numpy(shp),
                                           [tuple(s) for s in shapes], torch.from_numpy(lsi),
                                           torch.from_numpy(loc), torch.from_numpy(attw), 64)
    np.testing.assert_allclose(got64, hf.numpy(), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("shapes,P,Dh", [(TINY_PYRAMID, 2, 8), ([(5, 4), (3, 3)], 4, 32)])
def test_msda_backward_vs_autograd(shapes, P, Dh):
    B, Q, H = 2, 11, 3
    value, shp, lsi, loc, attw = msda_inputs(B, Q, H, Dh, shapes, P, seed=13, dtype=np.float64)
    # keep samples off pixel boundaries where bilinear is not differentiable
    tv, tl, ta = (torch.from_numpy(x).requires_grad_() for x in (value, loc, attw))
    out = _grid_sample_msda(tv, shapes, tl, ta)
    go = torch.from_numpy(np.random.default_rng(1).standard_normal(out.shape))
    out.backward(go)
    gv, gl, ga = ok.msda_bwd(value, shp, lsi, loc, attw, go.numpy(), dtype=np.float64)
    np.testing.assert_allclose(gv, tv.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ga, ta.grad.numpy(), rtol=1e-9, atol=1e-9)
    # grad wrt location: identical except exactly on pixel boundaries (measure zero; the
    # pinned border samples in msda_inputs hit them, where one-sided derivatives differ)
    h_w = np.array([[w, h] for h, w in shapes], np.float64)[None, None, None, :, None, :]
    pix = loc * h_w - 0.5
    smooth = (np.abs(pix - np.round(pix)) > 1e-9).all(-1)
    np.testing.assert_allclose(gl[smooth], tl.grad.numpy()[smooth], rtol=1e-8, atol=1e-8)
    # fp32 oracle agrees with the fp64 one to fp32 round-off
    gv32, gl32, ga32 = ok.msda_bwd(value, shp, lsi, loc, attw, go.numpy(), dtype=np.float32)
    np.testing.assert_allclose(gv32, gv, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ga32, ga, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gl32[smooth], gl[smooth], rtol=1e-3, atol=1e-3)
