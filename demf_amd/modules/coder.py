"""DeMFClassAgnosticBBoxCoder - mirrors
demf/core/bbox/coders/class_agnostic_bbox_coder.py:130-251 (DeMF variant) and the
angle helpers of mmdet3d's PartialBinBasedBBoxCoder it inherits."""
import numpy as np
import torch


class _Preds(dict):
    """split_pred's result dict.  Two of the reference's entries are materialised on first access instead
    of by split_pred itself: ``dir_res`` = dir_res_norm * (pi / num_dir_bins) (coder.py:233) and ``center``
    = base_xyz + reg[..., 0:3] (coder.py:214).  Nothing on the fused training path reads them (the loss
    kernels take the raw rows, the decoder's position embedding is built by ops.query_pos_rows), so they
    would only cost a launch + an autograd node per prediction head and step.
    Every read access of the dict contract sees the keys - ``d[k]``, ``.get``, ``in``, iteration,
    ``keys / items / values``, ``dict(d)`` / ``{**d}`` and ``.copy()`` (which keeps the recipes)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lazy = {}                 # key -> zero-argument callable

    def _lazy(self, only=None):
        for key in ([only] if only is not None else list(self.lazy)):
            fn = self.lazy.get(key)
            if fn is not None and not dict.__contains__(self, key):
                dict.__setitem__(self, key, fn())

    def __missing__(self, key):
        self._lazy(key)
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        self._lazy(key)
        return dict.get(self, key, default)

    def __contains__(self, key):
        self._lazy(key)
        return dict.__contains__(self, key)

    def __iter__(self):
        self._lazy()
        return dict.__iter__(self)

    def __len__(self):
        self._lazy()
        return dict.__len__(self)

    def keys(self):
        self._lazy()
        return dict.keys(self)

    def items(self):
        self._lazy()
        return dict.items(self)

    def values(self):
        self._lazy()
        return dict.values(self)

    def copy(self):
        out = _Preds(dict.items(self))
        out.lazy = dict(self.lazy)
        return out


class DeMFClassAgnosticBBoxCoder:
    def __init__(self, num_dir_bins, with_rot=True, **unused):
        self.num_dir_bins = num_dir_bins
        self.with_rot = with_rot
        self.num_sizes = 0

    # ---- PartialBinBasedBBoxCoder helpers (upstream, [dep-recall]) ----
    def angle2class(self, angle):
        angle = angle % (2 * np.pi)
        per = 2 * np.pi / float(self.num_dir_bins)
        shifted = (angle + per / 2) % (2 * np.pi)
        cls = shifted // per
        res = shifted - (cls * per + per / 2)
        return cls.long(), res

    def class2angle(self, angle_cls, angle_res, limit_period=True):
        per = 2 * np.pi / float(self.num_dir_bins)
        angle = angle_cls.float() * per + angle_res
        if limit_period:
            angle = torch.where(angle > np.pi, angle - 2 * np.pi, angle)
        return angle

    # ---- coder.py:142-166 ----
    def encode(self, gravity_center, dims, yaw, labels, ret_dir_target=False):
        """Takes the box fields the reference reads off gt_bboxes_3d
        (.gravity_center, .dims, .yaw)."""
        center_target = gravity_center
        size_res_target = dims
        if self.with_rot:
            dir_class_target, dir_res_target = self.angle2class(yaw)
            dir_target = yaw
        else:
            dir_class_target = labels.new_zeros(labels.shape[0])
            dir_res_target = dims.new_zeros(labels.shape[0])
            dir_target = dims.new_zeros(labels.shape[0])
        if ret_dir_target:
            return center_target, size_res_target, dir_class_target, dir_res_target, dir_target
        return center_target, size_res_target, dir_class_target, dir_res_target

    # ---- coder.py:168-194 ----
    def decode(self, bbox_out, mode="rpn"):
        assert mode == "rpn"
        center, bbox_size = bbox_out["center"], bbox_out["size"]
        B, N, _ = center.shape
        if self.with_rot:
            dir_class = torch.argmax(bbox_out["dir_class"], -1).detach()
            dir_res = torch.gather(bbox_out["dir_res"], -1, dir_class.unsqueeze(-1)).squeeze(-1)
            dir_angle = self.class2angle(dir_class, dir_res).reshape(B, N, 1)
            dir_angle = dir_angle % (2 * np.pi)
        else:
            dir_angle = center.new_zeros(B, N, 1)
        return torch.cat([center, bbox_size, dir_angle], dim=-1)

    # ---- coder.py:196-240 ----
    def split_pred(self, cls_preds, reg_preds, base_xyz):
        results = _Preds()
        cls_t = cls_preds.transpose(2, 1)
        reg_t = reg_preds.transpose(2, 1)
        with_sem = cls_t.shape[-1] > 2
        nb = self.num_dir_bins
        # (the reference materialises every slice with .contiguous(); here they stay views of the
        # raw conv rows - same values, no copy kernels; consumers that need dense memory copy)
        # "center" = base_xyz + reg[..., 0:3]: materialised on first access (see _Preds)
        results.lazy["center"] = lambda: base_xyz + reg_t[..., 0:3]
        results["size"] = reg_t[..., 3:6]
        results["dir_class"] = reg_t[..., 6:6 + nb]
        dir_res_norm = reg_t[..., 6 + nb:6 + 2 * nb]
        results["dir_res_norm"] = dir_res_norm
        # "dir_res" = dir_res_norm * (pi / nb) is only read by decode(): materialised on first access
        results.lazy["dir_res"] = lambda: dir_res_norm * (np.pi / nb)
        results["obj_scores"] = cls_t[..., 0:2]
        if with_sem:
            results["sem_scores"] = cls_t[..., 2:]
        return results

    # ---- coder.py:242-251 ----
    def decode_corners(self, center, size):
        half = size / 2.0
        return torch.cat([center - half, center + half], dim=-1)
