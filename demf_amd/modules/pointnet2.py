"""PointNet++ set abstraction / feature propagation on the gfx950 operators.

Mirrors the mmdet3d modules the reference instantiates: ``build_sa_module`` ->
PointSAModule (demf/modeling/heads/class_agnostic_vote_head.py:383,455) and the
PointNet2SASSG backbone (configs/demf/demf_votenet.py:48-62) consumed through
DeMFVoteNet.extract_pts_feat (demf/modeling/detectors/demfnet.py:151-152).

Interface (shapes, argument names, returned tuple) and state-dict names follow the
upstream modules; internally features are kept point-major (B,N,C) so every gather
moves whole rows and the shared MLP is a GEMM.  Feature tensors handed across module
boundaries are (B,C,N) *views* of that storage, so the reference's layout contract
holds without transposition copies.
"""
import torch
import torch.nn as nn

from .. import ops
from .layers import RowsMLP


def _rows(features_bcn):
    """(B,C,N) reference-layout tensor -> point-major (B,N,C) contiguous storage.
    Free when the input is already a view of point-major storage."""
    return features_bcn.transpose(1, 2).contiguous()


def _pad4(n):
    return (n + 3) // 4 * 4


class PointSAModule(nn.Module):
    """Single-scale set abstraction: D-FPS -> ball query -> group -> shared MLP -> max.

    forward(points_xyz (B,N,3), features (B,C,N)|None, indices (B,M)|None,
            target_xyz (B,M,3)|None) -> (new_xyz (B,M,3), new_features (B,C',M),
            indices (B,M) i32)          [upstream PointSAModule.forward contract]
    """

    def __init__(self, num_point, radius, num_sample, mlp_channels, use_xyz=True,
                 normalize_xyz=False, pool_mod="max", fps_mod=("D-FPS",), **unused):
        super().__init__()
        assert pool_mod == "max", "the reference config uses pool_mod='max'"
        assert tuple(fps_mod) == ("D-FPS",), "the reference config uses D-FPS"
        self.num_point, self.radius, self.num_sample = num_point, radius, num_sample
        self.use_xyz, self.normalize_xyz = use_xyz, normalize_xyz
        ch = list(mlp_channels)
        if use_xyz:
            ch[0] += 3
        self.in_feat = ch[0] - (3 if use_xyz else 0)
        # upstream: self.mlps = ModuleList([Sequential(layer0.., ConvModule(Conv2d,BN2d))])
        self.mlps = nn.ModuleList([RowsMLP(ch, dim=2, bias=False)])

    def _first_weight(self, ld):
        """Reference column order is [xyz(3), feat(C)]; grouped rows here are
        [feat(C), xyz(3), 0-pad] so the feature copy is 16-byte aligned."""
        layer0 = self.mlps[0][0]
        w = layer0.weight2d()
        C = self.in_feat
        parts = [w[:, 3:], w[:, :3]] if self.use_xyz else [w]
        pad = ld - w.shape[1]
        if pad:
            parts.append(w.new_zeros(w.shape[0], pad))
        return torch.cat(parts, dim=1) if len(parts) > 1 else w

    def index_geometry(self, points_xyz, indices=None, with_inverse=False):
        """Everything of this level that depends on coordinates only: D-FPS indices, the sampled
        centres, the ball-query neighbour lists and (``with_inverse``: levels whose input features
        carry a gradient) their inverse lists.  -> (indices, new_xyz, group_idx[, inv_off, inv_rows])"""
        if indices is None:
            indices = ops.furthest_point_sample(points_xyz, self.num_point)
        new_xyz = ops.gather_rows_cl(points_xyz, indices)
        group_idx = ops.ball_query(0.0, self.radius, self.num_sample, points_xyz, new_xyz)
        if with_inverse and points_xyz.shape[1] <= 16384:
            return (indices, new_xyz, group_idx) + ops.invert_index(group_idx, points_xyz.shape[1])
        return indices, new_xyz, group_idx

    def forward(self, points_xyz, features=None, indices=None, target_xyz=None, group_idx=None,
                group_inv=None, new_xyz=None):
        B, N, _ = points_xyz.shape
        if indices is not None:
            assert indices.shape[1] == self.num_point
            if new_xyz is None:
                new_xyz = ops.gather_rows_cl(points_xyz, indices)
        elif target_xyz is not None:
            new_xyz = target_xyz.contiguous()
        else:
            indices = ops.furthest_point_sample(points_xyz, self.num_point)
            new_xyz = ops.gather_rows_cl(points_xyz, indices)
        M = new_xyz.shape[1]
        feat = _rows(features) if features is not None else None
        C = 0 if feat is None else feat.shape[2]
        idx = group_idx if group_idx is not None else \
            ops.ball_query(0.0, self.radius, self.num_sample, points_xyz, new_xyz)
        assert self.use_xyz or feat is not None
        ld = _pad4(C + 3) if self.use_xyz else C
        needs_grad = feat is not None and torch.is_grad_enabled() and \
            (feat.requires_grad or points_xyz.requires_grad or new_xyz.requires_grad)
        if group_inv is None and needs_grad and C % 4 == 0 and N <= 16384:
            group_inv = ops.invert_index(idx, N)
        mlp = self.mlps[0]
        if self.use_xyz and feat is not None and len(mlp) >= 2 and C % 4 == 0 \
                and mlp[0].cout in (64, 128, 256) and not ops._NO_GROUP_FIRST \
                and (group_inv is not None or not needs_grad):
            # first layer per SOURCE point: y = (feat . Wf^T)[idx] + rel_xyz . Wx^T, the grouped
            # rows (B*M*ns, 3+C) are never built (csrc/group_first.hip)
            inv_off, inv_rows = group_inv if group_inv is not None else (None, None)
            x = mlp.forward_rows(feat.view(B * N, C), ns=self.num_sample,
                                 geo=(points_xyz, new_xyz, idx, inv_off, inv_rows, self.radius,
                                      self.normalize_xyz))
        else:
            # Rows without padding (SA1: xyz + one feature = 4 floats) are written in the REFERENCE column
            # order [xyz | feat]: the first conv's weight is then used as stored - no concatenation of its
            # permuted columns per step, no slice-and-copy of its gradient in the backward.
            ref_order = self.use_xyz and C + 3 == ld and C < 4
            grouped = ops.group_concat_cl(points_xyz, new_xyz, feat, idx, self.radius,
                                          self.normalize_xyz, ldo=ld,
                                          xyz_col=(0 if ref_order else C) if self.use_xyz else 0,
                                          feat_col=3 if ref_order else 0,
                                          inverse=group_inv) \
                if self.use_xyz else ops.gather_rows_cl(
                    feat, idx.view(B, M * self.num_sample)).view(B, M, self.num_sample, C)
            x = grouped.view(B * M * self.num_sample, ld)
            # shared MLP + BN + ReLU + max over the ns neighbours: one fused chain (csrc/mlp.hip)
            x = mlp.forward_rows(x, None if ref_order else self._first_weight(ld), ns=self.num_sample)
        new_features = x.view(B, M, -1).transpose(1, 2)  # (B,C',M) view of point-major
        return new_xyz, new_features, indices


def build_sa_module(cfg):
    """``build_sa_module(cfg)`` of mmdet3d.ops for the cfg dicts the reference uses
    (type='PointSAModule'; class_agnostic_vote_head.py:383)."""
    cfg = dict(cfg)
    t = cfg.pop("type", "PointSAModule")
    assert t == "PointSAModule", f"unsupported SA module type {t}"
    return PointSAModule(**cfg)


class PointFPModule(nn.Module):
    """Feature propagation: 3-NN inverse-distance interpolation + shared MLP.

    forward(target (B,n,3), source (B,m,3), target_feats (B,C1,n), source_feats
            (B,C2,m)) -> (B,C',n)                       [upstream PointFPModule.forward]
    """

    def __init__(self, mlp_channels):
        super().__init__()
        self.mlps = RowsMLP(list(mlp_channels), dim=2, bias=False)

    @staticmethod
    def index_geometry(target, source):
        """3-NN indices and inverse-distance weights (coordinates only)."""
        if target.is_cuda and not (target.requires_grad or source.requires_grad):
            return ops.three_nn_weights(target.contiguous(), source.contiguous())
        dist, idx = ops.three_nn(target, source)
        dist_recip = 1.0 / (dist + 1e-8)
        return idx, (dist_recip / dist_recip.sum(dim=2, keepdim=True)).contiguous()

    def forward(self, target, source, target_feats, source_feats, nn=None):
        B, n, _ = target.shape
        idx, weight = nn if nn is not None else self.index_geometry(target, source)
        if target_feats is not None and source_feats.is_cuda:
            x = ops.three_interpolate_cat_cl(_rows(source_feats), idx, weight.contiguous(),
                                             _rows(target_feats).contiguous())
        else:
            interp = ops.three_interpolate_cl(_rows(source_feats), idx, weight.contiguous())
            x = torch.cat([interp, _rows(target_feats)], dim=2) if target_feats is not None else interp
        x = self.mlps.forward_rows(x.view(B * n, -1))
        return x.view(B, n, -1).transpose(1, 2)


class PointNet2SASSG(nn.Module):
    """PointNet++ single-scale-grouping backbone (4 SA + 2 FP in the reference config).

    forward(points (B,N,3+C)) -> dict(fp_xyz, fp_features, fp_indices, sa_xyz,
    sa_features, sa_indices) exactly as upstream.
    """

    def __init__(self, in_channels=4, num_points=(2048, 1024, 512, 256),
                 radius=(0.2, 0.4, 0.8, 1.2), num_samples=(64, 32, 16, 16),
                 sa_channels=((64, 64, 128), (128, 128, 256), (128, 128, 256), (128, 128, 256)),
                 fp_channels=((256, 256), (256, 256)), use_xyz=True, normalize_xyz=True,
                 **unused):
        super().__init__()
        self.num_sa, self.num_fp = len(sa_channels), len(fp_channels)
        self.SA_modules = nn.ModuleList()
        sa_in = in_channels - 3
        skip = [sa_in]
        for i in range(self.num_sa):
            ch = [sa_in] + list(sa_channels[i])
            self.SA_modules.append(PointSAModule(num_points[i], radius[i], num_samples[i], ch,
                                                 use_xyz=use_xyz, normalize_xyz=normalize_xyz))
            sa_in = ch[-1]
            skip.append(sa_in)
        self.FP_modules = nn.ModuleList()
        fp_src, fp_tgt = skip.pop(), skip.pop()
        for i in range(self.num_fp):
            ch = [fp_src + fp_tgt] + list(fp_channels[i])
            self.FP_modules.append(PointFPModule(ch))
            if i != self.num_fp - 1:
                fp_src, fp_tgt = ch[-1], skip.pop()

    @torch.no_grad()
    def index_geometry(self, points):
        """The coordinate-only pre-pass of the whole backbone: per SA level (fps indices, centres,
        ball-query lists), per FP level (3-NN indices, weights).  It depends on nothing but the
        input cloud, so a training loop can run it for batch k+1 on a side stream while batch k
        trains (demf_amd/engine.py) - FPS is a latency-bound chain that occupies only B of the
        256 CUs."""
        on_dev = points.is_cuda and points.dtype == torch.float32
        if on_dev:
            xyz, feat_rows = ops.split_points(points)          # one launch instead of two strided copies
        else:
            xyz = points[..., 0:3].contiguous()
            feat_rows = points[..., 3:].contiguous() if points.shape[-1] > 3 else None
        B, N = xyz.shape[:2]
        sa, cur = [], xyz
        for i, m in enumerate(self.SA_modules):
            lvl = m.index_geometry(cur, with_inverse=i > 0)   # level 0 gathers the raw input
            sa.append(lvl)
            cur = lvl[1]
        if on_dev and len(sa) <= 8:
            chain = ops.sa_index_chain(N, [lvl[0] for lvl in sa])     # indices into the input cloud, one launch
        else:
            chain = [torch.arange(N, device=xyz.device).unsqueeze(0).repeat(B, 1).long()]
            for lvl in sa:
                chain.append(torch.gather(chain[-1], 1, lvl[0].long()))
        sa_xyz = [xyz] + [t[1] for t in sa]
        fp = [PointFPModule.index_geometry(sa_xyz[self.num_sa - i - 1], sa_xyz[self.num_sa - i])
              for i in range(self.num_fp)]
        geo = dict(sa=sa, fp=fp, xyz=xyz, sa_indices=chain)
        if feat_rows is not None:
            geo["feat_rows"] = feat_rows                           # (B,N,C0) point-major input features
        return geo

    def forward(self, points, geometry=None):
        if geometry is not None:
            xyz = geometry["xyz"]
            features = geometry["feat_rows"].transpose(1, 2) if "feat_rows" in geometry else None
            indices = geometry["sa_indices"][0]
        else:
            xyz = points[..., 0:3].contiguous()
            features = points[..., 3:].transpose(1, 2) if points.shape[-1] > 3 else None
            B, N = xyz.shape[:2]
            indices = torch.arange(N, device=xyz.device).unsqueeze(0).repeat(B, 1).long()
        sa_xyz, sa_features, sa_indices = [xyz], [features], [indices]
        for i in range(self.num_sa):
            if geometry is not None:
                lvl = geometry["sa"][i]
                cur_xyz, cur_feat, cur_idx = self.SA_modules[i](
                    sa_xyz[i], sa_features[i], indices=lvl[0], target_xyz=None, group_idx=lvl[2],
                    group_inv=tuple(lvl[3:5]) if len(lvl) >= 5 else None, new_xyz=lvl[1])
                sa_indices.append(geometry["sa_indices"][i + 1])
            else:
                cur_xyz, cur_feat, cur_idx = self.SA_modules[i](sa_xyz[i], sa_features[i])
                sa_indices.append(torch.gather(sa_indices[-1], 1, cur_idx.long()))
            sa_xyz.append(cur_xyz)
            sa_features.append(cur_feat)
        fp_xyz, fp_features, fp_indices = [sa_xyz[-1]], [sa_features[-1]], [sa_indices[-1]]
        for i in range(self.num_fp):
            fp_features.append(self.FP_modules[i](sa_xyz[self.num_sa - i - 1],
                                                  sa_xyz[self.num_sa - i],
                                                  sa_features[self.num_sa - i - 1],
                                                  fp_features[-1],
                                                  nn=geometry["fp"][i] if geometry is not None else None))
            fp_xyz.append(sa_xyz[self.num_sa - i - 1])
            fp_indices.append(sa_indices[self.num_sa - i - 1])
        return dict(fp_xyz=fp_xyz, fp_features=fp_features, fp_indices=fp_indices,
                    sa_xyz=sa_xyz, sa_features=sa_features, sa_indices=sa_indices)
