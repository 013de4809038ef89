"""Seeded synthetic SUN RGB-D-like batches (there is no dataset access in the build or GPU
environments): points, image pyramid, img_metas with 3-D augmentation flows, GT boxes.
Shapes follow configs/demf/demf_votenet.py:184-216 (20 000 points with a height channel,
800x1120 padded image -> 4-level pyramid).  Used by bench.py, smoke() and the tests."""
import numpy as np


def depth2img():
    """SUN RGB-D-like K @ Rt (original 530x730 frame): depth coords (x right, y forward,
    z up) -> camera (x right, y down, z forward), 8 degree tilt."""
    fx = fy = 529.5
    cx, cy = 365.0, 265.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Rt = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
    tilt = np.deg2rad(8.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
    return (K @ Rx @ Rt).astype(np.float32)


def make_scene_batch(B, N, pyramid, batch_input_shape, channels, seed=0, n_gt=4,
                     img_shape=None, scale_factor=None, cloud="uniform", gt_counts=None):
    """Synthetic SUN RGB-D-like batch: points (B,N,4) (xyz + height) in front of the camera,
    image pyramid ~N(0,1), img_metas alternating an identity 3-D flow with a
    flip+rot+scale+trans flow (and a smaller valid image, so padding masks are exercised),
    and n_gt-ish rotated GT boxes per scene.
    ``cloud``: "uniform" in the room volume, or "clustered" = 20 Gaussian blobs (sigma 0.3 m) per scene
    (a depth sensor sees surfaces, not a volume - the SA1 balls then
    saturate at 64 neighbours and the neighbour lists are unbalanced).
    ``gt_counts``: explicit number of GT boxes per scene (default: n_gt - b % 3)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-2.5, 0.8, -1.2]), np.array([2.5, 6.0, 1.6])
    if cloud == "clustered":
        centres = rng.uniform(lo + 0.3, hi - 0.3, size=(B, 20, 3))
        which = rng.integers(0, 20, size=(B, N))
        xyz = np.take_along_axis(centres, which[..., None].repeat(3, -1), 1) + rng.normal(0, 0.3, size=(B, N, 3))
        xyz = np.clip(xyz, lo - 0.5, hi + 0.5)
    elif cloud == "uniform":
        xyz = rng.uniform(lo, hi, size=(B, N, 3))
    else:
        raise ValueError("cloud must be 'uniform' or 'clustered'")
    height = xyz[..., 2:3] - xyz[..., 2:3].min(axis=1, keepdims=True)
    points = np.concatenate([xyz, height], -1).astype(np.float32)
    feats = [rng.standard_normal((B, channels, h, w)).astype(np.float32) for h, w in pyramid]
    in_h, in_w = batch_input_shape
    if img_shape is None:
        img_shape = (in_h, in_w - in_w // 60)
    if scale_factor is None:
        scale_factor = img_shape[0] / 530.0
    metas, boxes, labels = [], [], []
    for b in range(B):
        aug = b % 2 == 1
        ang = 0.3 if aug else 0.0
        rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        shp = (img_shape[0] - (in_h // 16 if aug else 0), img_shape[1] - (in_w // 10 if aug else 0), 3)
        metas.append(dict(
            img_shape=shp, batch_input_shape=(in_h, in_w), depth2img=depth2img(),
            scale_factor=np.full(4, scale_factor, np.float32), flip=False,
            pcd_horizontal_flip=aug, pcd_vertical_flip=False,
            pcd_rotation=rot.astype(np.float32), pcd_scale_factor=1.07 if aug else 1.0,
            pcd_trans=np.array([0.05, -0.1, 0.02], np.float32) if aug else np.zeros(3, np.float32),
            transformation_3d_flow=["HF", "R", "S", "T"] if aug else []))
        k = max(0, n_gt - b % 3) if gt_counts is None else int(gt_counts[b])
        ctr = rng.uniform([-2.0, 1.5, -1.0], [2.0, 5.0, 0.0], size=(k, 3))
        dims = rng.uniform([0.5, 0.5, 0.4], [1.8, 1.6, 1.2], size=(k, 3))
        yaw = rng.uniform(-np.pi, np.pi, size=(k, 1))
        boxes.append(np.concatenate([ctr, dims, yaw], 1).astype(np.float32))
        labels.append(rng.integers(0, 10, size=(k,)).astype(np.int64))
    return dict(points=points, img_features=feats, img_metas=metas, gt_boxes=boxes, gt_labels=labels)
