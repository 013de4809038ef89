"""Coordinate helpers the DeMF head calls into mmdet3d for (restated; [dep-recall] for
mmdet3d==0.18.1 conventions - none of these sources are in the reference tree).

Used by DeMFVoteHead.get_reference_points (class_agnostic_vote_head.py:524-547) and the
target generation (class_agnostic_vote_head.py:818-941).
"""
import torch


def rotation_3d_in_axis_z(points, angles):
    """mmdet3d 0.18.1 ``rotation_3d_in_axis(points (N,M,3), angles (N,), axis=2)``:
    points @ [[cos,-sin,0],[sin,cos,0],[0,0,1]] (per-box matrix)."""
    c, s = torch.cos(angles), torch.sin(angles)
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    c, s = c[:, None], s[:, None]
    return torch.stack([x * c + y * s, -x * s + y * c, z], dim=-1)


def apply_3d_transformation_reverse(pcd, img_meta):
    """``apply_3d_transformation(pcd, 'DEPTH', img_meta, reverse=True)``: undo the 3-D
    augmentation flow recorded in ``img_meta['transformation_3d_flow']``."""
    dtype, device = pcd.dtype, pcd.device
    rot = (torch.as_tensor(img_meta["pcd_rotation"], dtype=dtype, device=device)
           if "pcd_rotation" in img_meta else torch.eye(3, dtype=dtype, device=device))
    scale = img_meta.get("pcd_scale_factor", 1.0)
    trans = (torch.as_tensor(img_meta["pcd_trans"], dtype=dtype, device=device)
             if "pcd_trans" in img_meta else torch.zeros(3, dtype=dtype, device=device))
    hflip = img_meta.get("pcd_horizontal_flip", False)
    vflip = img_meta.get("pcd_vertical_flip", False)
    flow = list(img_meta.get("transformation_3d_flow", []))[::-1]
    pcd = pcd.clone()
    inv_rot = rot.inverse()
    for op in flow:
        if op == "T":
            pcd = pcd - trans
        elif op == "S":
            pcd = pcd * (1.0 / scale)
        elif op == "R":
            pcd = pcd @ inv_rot
        elif op == "HF":
            if hflip:  # DepthPoints.flip('horizontal'): x -> -x
                pcd = pcd * pcd.new_tensor([-1.0, 1.0, 1.0])
        elif op == "VF":
            if vflip:  # DepthPoints.flip('vertical'): y -> -y
                pcd = pcd * pcd.new_tensor([1.0, -1.0, 1.0])
        else:
            raise KeyError(op)
    return pcd


def points_cam2img(points_3d, proj_mat):
    """``points_cam2img(points, proj_mat, with_depth=False)`` with a 3x3 / 3x4 / 4x4 matrix."""
    d1, d2 = proj_mat.shape[:2]
    if d1 == 3:
        full = torch.eye(4, dtype=proj_mat.dtype, device=proj_mat.device)
        full[:d1, :d2] = proj_mat
        proj_mat = full
    ones = points_3d.new_ones(points_3d.shape[:-1] + (1,))
    p = torch.cat([points_3d, ones], dim=-1) @ proj_mat.T
    return p[..., :2] / p[..., 2:3]


def coord_2d_transform(img_meta, coord_2d):
    """``coord_2d_transform(img_meta, coord_2d, is_orig2new=True)``: image-space scale,
    crop offset and horizontal flip of the 2-D pipeline."""
    img_h, img_w = img_meta["img_shape"][:2]
    sf = img_meta.get("scale_factor", [1.0, 1.0])
    sf = torch.as_tensor(sf, dtype=coord_2d.dtype, device=coord_2d.device)[:2]
    off = torch.as_tensor(img_meta.get("img_crop_offset", [0.0, 0.0]), dtype=coord_2d.dtype,
                          device=coord_2d.device)
    out = coord_2d * sf + off
    if img_meta.get("flip", False):
        out = torch.stack([img_w - out[:, 0], out[:, 1]], dim=-1)
    return out


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
