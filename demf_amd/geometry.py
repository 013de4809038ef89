"""Box / rotation helpers the DeMF head takes from mmdet3d (restated; [dep-recall] for
mmdet3d==0.18.1 conventions - none of these sources are in the reference tree).
Used by the target generation (class_agnostic_vote_head.py:818-941).  The point -> image
projection of get_reference_points (:524-547) is composed on the host in modules/head.py.
"""
import torch


def rotation_3d_in_axis_z(points, angles):
    """mmdet3d 0.18.1 ``rotation_3d_in_axis(points (N,M,3), angles (N,), axis=2)``:
    points @ [[cos,-sin,0],[sin,cos,0],[0,0,1]] (per-box matrix)."""
    c, s = torch.cos(angles), torch.sin(angles)
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    c, s = c[:, None], s[:, None]
    return torch.stack([x * c + y * s, -x * s + y * c, z], dim=-1)


class DepthBoxes:
    """The fields of mmdet3d DepthInstance3DBoxes the head reads: tensor (n,7) =
    (x, y, z_bottom, dx, dy, dz, yaw)."""

    def __init__(self, tensor):
        self.tensor = tensor

    def to(self, device):
        return DepthBoxes(self.tensor.to(device))

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], dim=1)

    @property
    def dims(self):
        return self.tensor[:, 3:6]

    @property
    def yaw(self):
        return self.tensor[:, 6]

    def points_in_boxes(self, points):
        """(N,3+) -> (N, n) membership, points_in_boxes_batch semantics: |z-cz| <= dz/2 and
        the box-frame x,y strictly inside the half extents."""
        p = points[:, :3]
        c = self.gravity_center
        rel = p[:, None, :] - c[None, :, :]                     # (N,n,3)
        yaw = self.yaw
        cs, sn = torch.cos(-yaw)[None], torch.sin(-yaw)[None]
        lx = rel[..., 0] * cs + rel[..., 1] * sn                # rotation_3d_in_axis(-yaw)
        ly = -rel[..., 0] * sn + rel[..., 1] * cs
        half = self.dims[None] * 0.5
        inside = (rel[..., 2].abs() <= half[..., 2]) & (lx.abs() < half[..., 0]) & \
                 (ly.abs() < half[..., 1])
        return inside


def level_masks(img_metas, spatial_shapes):
    """Padding masks of the image pyramid (class_agnostic_vote_head.py:559-568 and
    deform_detr_encoder.py:70-82): the (B,Hpad,Wpad) mask that is 0 on ``img_shape`` resized to
    every level with nearest-neighbour ``F.interpolate`` - an index lookup, evaluated on the host.
    -> list of numpy bool (B,h,w), True = padding."""
    import numpy as np
    hw = np.asarray([m["img_shape"][:2] for m in img_metas], dtype=np.int64)
    in_h, in_w = img_metas[0]["batch_input_shape"]
    masks = []
    for h, w in spatial_shapes:
        ys = np.floor(np.arange(h, dtype=np.float32) * np.float32(in_h / h)).astype(np.int64)
        xs = np.floor(np.arange(w, dtype=np.float32) * np.float32(in_w / w)).astype(np.int64)
        masks.append((ys[None, :, None] >= hw[:, 0, None, None]) | (xs[None, None, :] >= hw[:, 1, None, None]))
    return masks
