"""Shared synthetic-input builders for the test-suite (seeded, size-parametrised)."""
import numpy as np

# reference pyramid for an 800x1120 padded input
PYRAMID = [(100, 140), (50, 70), (25, 35), (13, 18)]
TINY_PYRAMID = [(8, 10), (4, 5), (2, 3), (1, 2)]


def scene_points(B, N, seed=0, grid=None, clustered=False):
    """Points in a 6x6x3 m room.  grid=k snaps coordinates to multiples of 1/k so every
    fp32 distance is exact (and ties are plentiful)."""
    rng = np.random.default_rng(seed)
    if clustered:
        centres = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 2.5], size=(B, 20, 3))
        which = rng.integers(0, 20, size=(B, N))
        pts = np.take_along_axis(centres, which[..., None].repeat(3, -1), 1)
        pts = pts + rng.normal(0, 0.3, size=(B, N, 3))
    else:
        pts = rng.uniform([-3, -3, 0], [3, 3, 3], size=(B, N, 3))
    if grid:
        pts = np.round(pts * grid) / grid
    return pts.astype(np.float32)


def level_start_index(shapes):
    sizes = [h * w for h, w in shapes]
    return np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)


def msda_inputs(B, Q, H, Dh, shapes, P, seed=0, spread=0.25, dtype=np.float32):
    """value/loc/weights with a share of out-of-image and exactly-on-border locations."""
    rng = np.random.default_rng(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((B, S, H, Dh))
    loc = rng.uniform(-spread, 1 + spread, size=(B, Q, H, L, P, 2))
    # pin some samples to exact borders / pixel centres
    flat = loc.reshape(-1, 2)
    n = flat.shape[0]
    flat[0:n:17] = 0.0
    flat[1:n:19] = 1.0
    flat[2:n:23, 0] = 0.5
    attw = rng.uniform(0.1, 1.0, size=(B, Q, H, L * P))
    attw = (attw / attw.sum(-1, keepdims=True)).reshape(B, Q, H, L, P)
    shp = np.asarray(shapes, np.int64)
    return (value.astype(dtype), shp, level_start_index(shapes), loc.astype(dtype),
            attw.astype(dtype))
