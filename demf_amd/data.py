"""Data path of the hot path: the on-disk formats and the 3-D augmentations
whose metadata ``DeMFVoteHead.get_reference_points`` (class_agnostic_vote_head.py:524-547) undoes.

The reference delegates all of this to mmdet3d 0.18.1 pipelines (configs/demf/demf_votenet.py:184-216,
configs/_base_/datasets/sunrgbd-3d-10class.py); they are not in the reference tree, so what follows
restates their published behaviour ("dep-recall", parity with upstream binaries unpinned).  What IS
checked (tests/test_data_path.py): the metadata written here and the head's projection are mutually
consistent - an augmented point projects to the pixel of its un-augmented original.
"""
import numpy as np

# mmdet3d SUNRGBDDataset.get_data_info: depth2img = K @ (AXIS @ Rt^T), AXIS = depth -> camera axes
_DEPTH_TO_CAM = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])


def load_points_bin(path, load_dim=6, use_dim=(0, 1, 2), shift_height=True):
    """LoadPointsFromFile(coord_type='DEPTH', shift_height=True, load_dim=6, use_dim=[0,1,2])
    (demf_votenet.py:186-190): float32 records -> (N, len(use_dim) [+1]) with the height above
    the 0.99-percentile floor appended."""
    pts = np.fromfile(path, dtype=np.float32).reshape(-1, load_dim)[:, list(use_dim)]
    return add_height(pts) if shift_height else pts


def add_height(points):
    floor = np.percentile(points[:, 2], 0.99)
    return np.concatenate([points, (points[:, 2] - floor)[:, None]], 1).astype(np.float32)


def depth2img_from_calib(K, Rt):
    """sunrgbd_infos_*.pkl 'calib' -> the 3x3 ``depth2img`` the head consumes."""
    return (np.asarray(K, np.float64).reshape(3, 3) @ (_DEPTH_TO_CAM @ np.asarray(Rt, np.float64).reshape(3, 3).T)
            ).astype(np.float32)


def sample_points(points, num_points, rng):
    """IndoorPointSample(num_points=20000): without replacement when enough points exist."""
    n = points.shape[0]
    choice = rng.choice(n, num_points, replace=n < num_points)
    return points[choice], choice


def rotation_z(angle):
    """Row-vector rotation used by mmdet3d's ``points.rotate(angle)``: p' = p @ R^T, counter-clockwise
    about z; the matrix stored as ``pcd_rotation`` is R^T."""
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def augment_3d(points, boxes, meta, rng, flip_ratio=0.5, rot_range=(-np.pi / 6, np.pi / 6),
               scale_range=(0.85, 1.15), translation_std=(0.0, 0.0, 0.0), sync_2d=False):
    """RandomFlip3D(sync_2d, flip_ratio_bev_horizontal) + GlobalRotScaleTrans (demf_votenet.py:
    198-206) on depth-coordinate points (N,>=3) and boxes (n,7); returns the augmented copies and
    ``meta`` extended with the fields the head's inverse reads: ``flip``, ``pcd_horizontal_flip``,
    ``pcd_rotation``, ``pcd_scale_factor``, ``pcd_trans``, ``transformation_3d_flow``.

    Box yaw follows mmdet3d 0.18.1's depth convention (the one ``geometry.DepthBoxes.points_in_boxes``
    and the head's target kernels use): ``DepthInstance3DBoxes.rotate`` turns the centres by +angle
    and does ``yaw -= angle``; ``flip('horizontal')`` does ``yaw = pi - yaw``.  ``sync_2d`` defaults to
    False as in the reference config (RandomFlip3D(sync_2d=False), image flip_ratio 0.0,
    demf_votenet.py:198-202): the image is NOT mirrored with the cloud, so ``meta['flip']`` stays
    whatever the 2-D pipeline wrote.  With ``sync_2d=True`` the caller must mirror the image (or its
    feature pyramid) itself when ``meta['flip']`` comes back True."""
    pts, bx = points.copy(), boxes.copy()
    meta = dict(meta)
    flow = []
    flip = bool(rng.random() < flip_ratio)
    if sync_2d:
        meta["flip"] = flip                      # the caller mirrors the image together with the cloud
    else:
        meta.setdefault("flip", False)
    meta["pcd_horizontal_flip"] = flip
    meta["pcd_vertical_flip"] = False
    if flip:                                     # DepthPoints.flip('horizontal'): x -> -x
        pts[:, 0] = -pts[:, 0]
        bx[:, 0] = -bx[:, 0]
        bx[:, 6] = -bx[:, 6] + np.pi
    flow.append("HF")
    angle = rng.uniform(*rot_range)
    rot_t = rotation_z(angle).T                   # p @ rot_t rotates by +angle
    pts[:, :3] = pts[:, :3] @ rot_t
    bx[:, :3] = bx[:, :3] @ rot_t
    bx[:, 6] -= angle                             # DepthInstance3DBoxes.rotate (0.18.1)
    meta["pcd_rotation"] = rot_t.astype(np.float32)
    flow.append("R")
    scale = rng.uniform(*scale_range)
    pts[:, :3] *= scale
    bx[:, :6] *= scale
    meta["pcd_scale_factor"] = float(scale)
    flow.append("S")
    trans = rng.normal(scale=np.asarray(translation_std, np.float64), size=3)
    pts[:, :3] += trans
    bx[:, :3] += trans
    meta["pcd_trans"] = trans.astype(np.float32)
    flow.append("T")
    meta["transformation_3d_flow"] = flow
    if pts.shape[1] > 3:                          # the height channel scales with the cloud
        pts[:, 3] *= scale
    return pts.astype(np.float32), bx.astype(np.float32), meta


def resize_meta(meta, ori_shape, img_scale, pad_divisor=32):
    """Resize(keep_ratio) + Pad(size_divisor=32) bookkeeping (demf_votenet.py:192-197): image of
    ``ori_shape`` (h,w) scaled to fit ``img_scale`` (long, short) -> img_shape, scale_factor and the
    padded ``batch_input_shape`` of a batch of one."""
    h, w = ori_shape
    long_e, short_e = max(img_scale), min(img_scale)
    s = min(long_e / max(h, w), short_e / min(h, w))
    nh, nw = int(h * s + 0.5), int(w * s + 0.5)
    meta = dict(meta)
    meta["ori_shape"] = (h, w, 3)
    meta["img_shape"] = (nh, nw, 3)
    meta["scale_factor"] = np.array([nw / w, nh / h, nw / w, nh / h], np.float32)
    meta["batch_input_shape"] = (int(np.ceil(nh / pad_divisor)) * pad_divisor,
                                 int(np.ceil(nw / pad_divisor)) * pad_divisor)
    return meta


def remap_checkpoint(state_dict):
    """``DeMFVoteNet._load_from_state_dict`` (demf/modeling/detectors/demfnet.py:85-101) as a pure
    function: a stage-1 checkpoint (the Deformable-DETR image detector of configs/deformdetr/*)
    carries the image encoder under ``img_bbox_head.transformer.encoder.*`` and
    ``img_bbox_head.transformer.level_embeds``; those keys move to ``img_encoder.*`` and EVERY other
    ``img_bbox_head.*`` key (decoder, fc_cls, reg branches, query embedding ...) is dropped.  All
    other keys pass through unchanged (``pts_backbone.* / pts_bbox_head.* / img_backbone.* /
    img_neck.* / img_encoder.*`` use the reference's names in this package).  Like the reference the
    test is a substring test on the whole key.  -> new dict (the input is not modified)."""
    out = {}
    for key, v in state_dict.items():
        if not key.startswith("img_bbox_head"):
            out[key] = v
        elif "encoder" in key or "level_embeds" in key:
            out[key.replace("img_bbox_head.transformer", "img_encoder")] = v
    return out


def split_checkpoint(state_dict):
    """remap_checkpoint + split by consumer: (hot-path state for ``DeMFHotPath`` - ``pts_backbone.*``
    / ``pts_bbox_head.*`` -, image-stream state for ``ImageStream`` - ``img_backbone.* / img_neck.* /
    img_encoder.*``)."""
    sd = remap_checkpoint(state_dict)
    hot = {k: v for k, v in sd.items() if k.startswith(("pts_backbone.", "pts_bbox_head."))}
    img = {k: v for k, v in sd.items() if k.startswith(("img_backbone.", "img_neck.", "img_encoder."))}
    return hot, img
