"""DeMFVoteHead on the gfx950 operators.

Mirrors demf/modeling/heads/class_agnostic_vote_head.py:335-941 (DeMFVoteHead): same
constructor kwargs, ``forward(feat_dict, sample_mod, img_dict)`` contract, result keys
and state-dict names (``vote_module``, ``vote_aggregation``, ``decoder.{i}``,
``conv_pred{i}``).  Differences are in HOW, not WHAT:
  * target generation is batched over scenes and GT boxes with static shapes (no Python
    loop over boxes, no nonzero()/host sync) so the whole step can be captured in a
    hipGraph, and is computed once per step instead of once per decode layer
    (the reference recomputes identical targets at :604-612);
  * the per-scene reference-point projection is composed on the host into one 4x4 per
    scene and applied as a single batched matmul.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..geometry import DepthBoxes, level_masks, rotation_3d_in_axis_z
from . import losses as L
from .coder import DeMFClassAgnosticBBoxCoder
from .pointnet2 import build_sa_module
from .transformer import DeMFTransformerDecoderLayer
from .vote import BaseConvBboxHead, VoteModule


BF16_TOKENS = True     # bf16 compute mode: image tokens as bf16 rows (False: fp32 rows as in the fp32 modes)


def compose_projection(img_meta):
    """Host-side (float64) composition of DeMFVoteHead.get_reference_points
    (class_agnostic_vote_head.py:524-547) for one scene:
    reverse 3-D aug flow -> depth2img -> 2-D scale/crop/flip -> /(W-1, H-1).
    Returns (M (4,4): [p,1] @ M.T = (x, y, w, .), au, bu, av, bv) so that
    u = x/w*au + bu and v = y/w*av + bv are the normalised coordinates before clamping."""
    A = np.eye(3)
    t = np.zeros(3)
    rot = np.asarray(img_meta.get("pcd_rotation", np.eye(3)), np.float64)
    scale = float(img_meta.get("pcd_scale_factor", 1.0))
    trans = np.asarray(img_meta.get("pcd_trans", np.zeros(3)), np.float64)
    for op in list(img_meta.get("transformation_3d_flow", []))[::-1]:
        if op == "T":
            t = t - trans
        elif op == "S":
            A, t = A / scale, t / scale
        elif op == "R":
            inv = np.linalg.inv(rot)
            A, t = A @ inv, t @ inv
        elif op == "HF":
            if img_meta.get("pcd_horizontal_flip", False):
                D = np.diag([-1.0, 1.0, 1.0])
                A, t = A @ D, t @ D
        elif op == "VF":
            if img_meta.get("pcd_vertical_flip", False):
                D = np.diag([1.0, -1.0, 1.0])
                A, t = A @ D, t @ D
        else:
            raise KeyError(op)
    P = np.asarray(img_meta["depth2img"], np.float64)
    P4 = np.eye(4)
    P4[:P.shape[0], :P.shape[1]] = P
    T = np.eye(4)           # column-vector form of p' = p @ A + t
    T[:3, :3] = A.T
    T[:3, 3] = t
    M = P4 @ T
    img_h, img_w = img_meta["img_shape"][:2]
    sf = np.asarray(img_meta.get("scale_factor", [1.0, 1.0]), np.float64)[:2]
    off = np.asarray(img_meta.get("img_crop_offset", [0.0, 0.0]), np.float64)
    au, bu, av, bv = sf[0], off[0], sf[1], off[1]
    if img_meta.get("flip", False):
        au, bu = -au, img_w - bu
    return M, au / (img_w - 1), bu / (img_w - 1), av / (img_h - 1), bv / (img_h - 1)


class DeMFVoteHead(nn.Module):
    def __init__(self, num_classes, bbox_coder, train_cfg=None, test_cfg=None,
                 vote_module_cfg=None, vote_aggregation_cfg=None, pred_layer_cfg=None,
                 conv_cfg=None, norm_cfg=None, objectness_loss=None, center_loss=None,
                 dir_class_loss=None, dir_res_loss=None, size_class_loss=None,
                 size_res_loss=None, semantic_loss=None, iou_loss=None, decoder=None,
                 init_cfg=None):
        super().__init__()
        self.num_classes = num_classes
        self.train_cfg, self.test_cfg = train_cfg or {}, test_cfg or {}
        self.gt_per_seed = vote_module_cfg["gt_per_seed"]
        self.num_proposal = vote_aggregation_cfg["num_point"]
        # loss hyper-parameters (cfg dicts of demf_votenet.py:116-141)
        self.loss_cfg = dict(objectness=objectness_loss or {}, center=center_loss or {},
                             dir_class=dir_class_loss or {}, dir_res=dir_res_loss or {},
                             size_res=size_res_loss or {}, semantic=semantic_loss,
                             iou=iou_loss)
        bc = dict(bbox_coder)
        bc.pop("type", None)
        self.bbox_coder = DeMFClassAgnosticBBoxCoder(**bc)
        self.num_dir_bins = self.bbox_coder.num_dir_bins
        vm = dict(vote_module_cfg)
        vl = vm.pop("vote_loss", {}) or {}
        self.vote_module = VoteModule(**vm, vote_loss_dst_weight=vl.get("loss_dst_weight", 1.0))
        self.vote_aggregation = build_sa_module(vote_aggregation_cfg)
        self.fp16_enabled = False
        dec = dict(decoder)
        self.num_decoder_layers = self.num_fusion_layers = dec.pop("num_layers")
        dec.pop("type", None)
        self.decoder = nn.ModuleList(
            [DeMFTransformerDecoderLayer(**dec) for _ in range(self.num_decoder_layers)])
        pl = dict(pred_layer_cfg)
        self.conv_pred_layers = pl.pop("conv_pred_layers")
        assert self.conv_pred_layers == self.num_decoder_layers + 1
        self.conv_preds = []
        ncls = num_classes + 2 if semantic_loss is not None else 2
        nreg = 6 + self.num_dir_bins * 2
        for i in range(self.conv_pred_layers):
            m = BaseConvBboxHead(**pl, num_cls_out_channels=ncls, num_reg_out_channels=nreg)
            self.add_module("conv_pred" + str(i), m)
            self.conv_preds.append(m)

    # ---- :405-466 ------------------------------------------------------------
    def forward(self, feat_dict, sample_mod, img_dict):
        assert sample_mod in ["vote", "seed", "random", "spec"]
        seed_points = feat_dict["seed_points"]
        seed_features = feat_dict["seed_features"]
        seed_indices = feat_dict["seed_indices"]
        img_features, img_metas = img_dict["img_features"], img_dict["img_metas"]
        vote_points, vote_features, vote_offset = self.vote_module(seed_points, seed_features)
        results = dict(seed_points=seed_points, seed_indices=seed_indices,
                       vote_points=vote_points, vote_features=vote_features,
                       vote_offset=vote_offset)
        if sample_mod == "vote":
            agg_in = dict(points_xyz=vote_points, features=vote_features)
        elif sample_mod == "seed":
            sample_indices = feat_dict.get("sample_indices")   # coordinate-only: may be prefetched
            if sample_indices is None:
                sample_indices = ops.furthest_point_sample(seed_points, self.num_proposal)
            agg_in = dict(points_xyz=vote_points, features=vote_features, indices=sample_indices)
        elif sample_mod == "random":
            B, num_seed = seed_points.shape[:2]
            sample_indices = torch.randint(0, num_seed, (B, self.num_proposal),
                                           dtype=torch.int32, device=seed_points.device)
            agg_in = dict(points_xyz=vote_points, features=vote_features, indices=sample_indices)
        else:  # 'spec'
            agg_in = dict(points_xyz=seed_points, features=seed_features, target_xyz=vote_points)
        aggregated_points, features, aggregated_indices = self.vote_aggregation(**agg_in)
        results["aggregated_points"] = aggregated_points
        results["aggregated_indices"] = aggregated_indices
        results["decode_res_all"] = self.transformer_decoder(
            features, aggregated_points, img_features, img_metas, img_dict.get("image_inputs"))
        return results

    # ---- :468-512 ------------------------------------------------------------
    def transformer_decoder(self, features, aggregated_points, img_features, img_metas,
                            image_inputs=None):
        decode_res_all = []
        cls_p, reg_p = self.conv_preds[0](features)
        decode_res = self._split(cls_p, reg_p, aggregated_points)
        decode_res_all.append(decode_res)
        if image_inputs is None:
            image_inputs = self.prepare_image_inputs(img_features, img_metas)
        feat_flatten, mask_flatten = image_inputs["feat_flatten"], image_inputs["mask_flatten"]
        spatial_shapes = image_inputs["spatial_shapes"]
        level_start_index, valid_ratios = image_inputs["level_start_index"], image_inputs["valid_ratios"]
        if features.is_cuda and image_inputs.get("value_tokens") is not None:
            # device path: each fusion layer is one autograd node on batch-major rows
            # (demf_amd/fused.py); the reference-point projection rides in its sampling kernel
            B, E, Q = features.shape
            mt = self._meta_tensors(img_metas, image_inputs["spatial"], features.device, features.dtype)
            rows = features.transpose(1, 2).reshape(B * Q, E)
            pts = aggregated_points.reshape(B * Q, 3)
            for i in range(self.num_decoder_layers):
                # (center | size | zero columns up to a multiple of 4: the row width the position
                # embedding's first GEMM stages, written by this one concatenation)
                query_pos = ops.query_pos_rows(decode_res["_rows"][1], aggregated_points)
                rows = self.decoder[i].forward_rows(
                    rows, query_pos, pts, image_inputs["value_tokens"], spatial_shapes,
                    level_start_index, (mt["M"], mt["ab"]), valid_ratios, B, layer_index=i)
                cls_p, reg_p = self.conv_preds[i + 1](rows.view(B, Q, E).transpose(1, 2))
                decode_res = self._split(cls_p, reg_p, aggregated_points)
                decode_res_all.append(decode_res)
            return decode_res_all
        reference_points = self.get_reference_points(aggregated_points, img_metas,
                                                     image_inputs["spatial"])
        query = features.permute(2, 0, 1)
        for i in range(self.num_decoder_layers):
            query_pos = torch.cat([decode_res["center"], decode_res["size"]], dim=-1).detach().clone()
            query = self.decoder[i](query=query, key=None, value=feat_flatten, query_pos=query_pos,
                                    key_padding_mask=mask_flatten,
                                    reference_points=reference_points,
                                    spatial_shapes=spatial_shapes,
                                    level_start_index=level_start_index,
                                    valid_ratios=valid_ratios,
                                    value_projected=(image_inputs["value_projected"][i]
                                                     if image_inputs["value_projected"] is not None else None),
                                    value_tokens=image_inputs.get("value_tokens"))
            cls_p, reg_p = self.conv_preds[i + 1](query.permute(1, 2, 0))
            decode_res = self._split(cls_p, reg_p, aggregated_points)
            decode_res_all.append(decode_res)
        return decode_res_all

    def _zero_cols(self, B, Q, n, like):
        key = (B, Q, n, like.device, like.dtype)
        cache = self.__dict__.setdefault("_zero_cols_cache", {})
        if key not in cache:
            cache[key] = torch.zeros((B, Q, n), dtype=like.dtype, device=like.device)
        return cache[key]

    def _split(self, cls_p, reg_p, base_xyz):
        """split_pred + private handles on the raw conv-head rows ((B*Q, 12) / (B*Q, 30) - the
        point-major storage behind cls_p / reg_p) for the fused loss kernel."""
        res = self.bbox_coder.split_pred(cls_p, reg_p, base_xyz)
        res["_rows"] = (cls_p.transpose(1, 2), reg_p.transpose(1, 2), base_xyz)
        return res

    # ---- :514-522 ------------------------------------------------------------
    def get_valid_ratio(self, mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def refresh_metas(self, static_metas, new_metas):
        """A hipGraph-replayed step reads the device constants cached for the ``static_metas``
        object; this overwrites them IN PLACE with the constants of ``new_metas`` (same batch size
        and padded image size), so the next replay sees the new batch's calibration / masks."""
        cache = self.__dict__.get("_meta_cache", {})
        found = False
        for (mid, shapes, dev), entry in list(cache.items()):
            if mid != id(static_metas):
                continue
            found = True
            fresh, layout = self._pack_meta_arrays(self._build_meta_arrays(new_metas, shapes, entry["M"].dtype))
            if [(k, o, n, str(d), tuple(sh)) for k, o, n, d, sh in layout] != entry["_layout"]:
                raise RuntimeError("refresh_metas: the new metas do not have the captured batch's shapes")
            if not entry["_blob"].is_cuda:
                entry["_blob"].copy_(torch.as_tensor(fresh))
                continue
            # Device entries are views of one byte blob: ONE upload on a SIDE stream into a fresh device
            # tensor, then ONE device-to-device copy on the current stream.  A pageable host -> device copy
            # makes the host wait for everything queued on ITS stream; on the stream of the training step that
            # is the step in flight, and the per-batch input path would run behind the GPU instead of
            # underneath it (measured: the host spent 5.3 ms per load in these copies; pinned staging +
            # non-blocking copies on the step's own stream were worse still - the DMA copies between graph
            # launches cost milliseconds).  (Round 3 moved every array on its own: 24 copy launches per batch.)
            cur = torch.cuda.current_stream()
            up = self.__dict__.get("_meta_upload_stream")
            if up is None:
                up = self.__dict__.setdefault("_meta_upload_stream", torch.cuda.Stream())
            with torch.cuda.stream(up):
                dev_fresh = torch.as_tensor(fresh, device=entry["_blob"].device)
            cur.wait_stream(up)
            dev_fresh.record_stream(cur)
            entry["_blob"].copy_(dev_fresh)
        if not found:
            raise RuntimeError("refresh_metas: no cached device constants for this metas object "
                               "(it was never used in a forward, or its entry was evicted)")

    def unpin_metas(self, img_metas):
        """Undo ``pin_metas`` (the captured graph that read this entry has been dropped: engine.StepCache)."""
        pinned = self.__dict__.get("_meta_pinned", {})
        n = pinned.get(id(img_metas), 0)
        if n > 1:
            pinned[id(img_metas)] = n - 1          # another live graph still reads this entry
            return
        pinned.pop(id(img_metas), None)
        keep = self.__dict__.get("_meta_pinned_keep", [])
        self.__dict__["_meta_pinned_keep"] = [m for m in keep if m is not img_metas]

    def pin_metas(self, img_metas):
        """Entries of ``img_metas`` are never evicted from the cache: a captured hipGraph holds raw
        pointers to their tensors (Trainer.capture calls this for its static metas)."""
        pinned = self.__dict__.setdefault("_meta_pinned", {})          # id -> number of graphs holding it
        pinned[id(img_metas)] = pinned.get(id(img_metas), 0) + 1
        if pinned[id(img_metas)] == 1:
            self.__dict__.setdefault("_meta_pinned_keep", []).append(img_metas)

    def _meta_tensors(self, img_metas, mlvl_shapes, dev, dt):
        """Device-side constants derived from the (host) img_metas, cached per metas object:
        the step then issues no host->device copy and can be captured in a hipGraph."""
        key = (id(img_metas), tuple(mlvl_shapes), str(dev))
        cache = self.__dict__.setdefault("_meta_cache", {})
        if key not in cache:
            pinned = self.__dict__.get("_meta_pinned", ())
            loose = [k for k in cache if k[0] not in pinned]
            while len(loose) >= 8:           # LRU over the un-pinned entries (dict keeps use order)
                del cache[loose.pop(0)]
            cache[key] = self._build_meta_tensors(img_metas, mlvl_shapes, dev, dt)
            cache[key]["keep"] = img_metas   # keep the keyed object alive so its id stays unique
        else:
            cache[key] = cache.pop(key)      # most recently used goes last
        return cache[key]

    @staticmethod
    def _build_meta_arrays(img_metas, mlvl_shapes, dt):
        """The per-batch constants as host arrays in their final dtypes (name -> ndarray or None)."""
        fdt = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16}.get(dt, np.float32)
        comp = [compose_projection(m) for m in img_metas]
        sizes = [h * w for h, w in mlvl_shapes]
        # padding masks (:559-568): nearest-neighbour resize of the (B,Hpad,Wpad) mask is an
        # index lookup, and the valid ratios (:514-522) count its first column / row - both
        # functions of the metas alone, so they are built here once per metas object
        hw = np.asarray([m["img_shape"][:2] for m in img_metas], dtype=np.int64)
        masks, ratios = [], []
        for (h, w), m in zip(mlvl_shapes, level_masks(img_metas, mlvl_shapes)):
            masks.append(m.reshape(len(img_metas), h * w))
            valid_h = (~m[:, :, 0]).sum(1).astype(np.float32)
            valid_w = (~m[:, 0, :]).sum(1).astype(np.float32)
            ratios.append(np.stack([valid_w / np.float32(w), valid_h / np.float32(h)], -1))
        empty = len(mlvl_shapes) == 0
        flat = None if empty else np.concatenate(masks, 1)
        return dict(
            M=np.stack([c[0] for c in comp]).astype(fdt),
            ab=np.asarray([c[1:] for c in comp]).astype(fdt),
            hw=hw,
            mask_flatten=None if empty else np.ascontiguousarray(flat),
            # (the same mask as bytes: what the token transposes read - converted here once per metas object
            # instead of by a launch per step)
            mask_u8=None if empty else np.ascontiguousarray(flat).astype(np.uint8),
            # (converted to float in numpy: torch's bool -> float32 copy of this 0.6 M-element array costs
            # 60-75 ms on the host, numpy's 0.5 ms - it is on the per-batch path of replay.load)
            keep4=None if empty else np.stack([(~flat).astype(fdt)] + [np.zeros(flat.shape, fdt)] * 3, -1),
            valid_ratios=None if empty else np.stack(ratios, 1).astype(fdt),
            spatial_shapes=np.asarray(list(mlvl_shapes), dtype=np.int64).reshape(-1, 2),
            level_start_index=np.asarray([0] + list(np.cumsum(sizes)[:-1]), dtype=np.int64))

    @staticmethod
    def _pack_meta_arrays(arrays):
        """name -> ndarray as ONE byte blob (64-byte aligned pieces) + its layout [(name, offset, nbytes,
        dtype, shape)]: a batch's constants travel as one upload and one device copy instead of two per array."""
        layout, off = [], 0
        for k, v in arrays.items():
            if v is None:
                continue
            v = np.ascontiguousarray(v)
            layout.append((k, off, v.nbytes, v.dtype, v.shape))
            off += (v.nbytes + 63) // 64 * 64
        blob = np.zeros(max(off, 64), np.uint8)
        for (k, o, n, _, _) in layout:
            blob[o:o + n] = np.ascontiguousarray(arrays[k]).reshape(-1).view(np.uint8)
        return blob, layout

    def _build_meta_tensors(self, img_metas, mlvl_shapes, dev, dt):
        arrays = self._build_meta_arrays(img_metas, mlvl_shapes, dt)
        blob, layout = self._pack_meta_arrays(arrays)
        dblob = torch.as_tensor(blob, device=dev)
        out = {k: None for k in arrays}
        for (k, o, n, ndt, shp) in layout:
            tdt = torch.as_tensor(np.zeros(0, ndt)).dtype
            out[k] = dblob[o:o + n].view(tdt).view(shp)
        out["_blob"], out["_layout"] = dblob, [(k, o, n, str(ndt), tuple(shp)) for k, o, n, ndt, shp in layout]
        return out

    # ---- :524-547 ------------------------------------------------------------
    def get_reference_points(self, seeds_3d_batch, img_metas, mlvl_shapes=()):
        dev, dt = seeds_3d_batch.device, seeds_3d_batch.dtype
        mt = self._meta_tensors(img_metas, mlvl_shapes, dev, dt)
        M, ab = mt["M"], mt["ab"]                                   # (B,4,4), (B,4)
        ones = seeds_3d_batch.new_ones(seeds_3d_batch.shape[:-1] + (1,))
        p = torch.cat([seeds_3d_batch, ones], dim=-1) @ M.transpose(1, 2)
        uv = p[..., :2] / p[..., 2:3]
        uv = uv * ab[:, None, 0::2] + ab[:, None, 1::2]   # (au, av) scale, (bu, bv) offset
        return torch.clamp(uv, 0, 1)

    # ---- :549-594 ------------------------------------------------------------
    def _sample_first(self, on_gpu, C0, S):
        """The decoder samples far fewer corners than there are tokens: keep the tokens unprojected
        (padding rows zeroed) and project after sampling (ops.msda_sample_then_project); no
        (B,S,C) value tensor per decoder layer."""
        att = self.decoder[0].layer.attentions[1]
        samples = self.num_proposal * att.num_levels * att.num_points * 4
        return bool(on_gpu and samples < S and C0 % 4 == 0 and C0 <= 256)

    def pyramid_tokens(self, mlvl_feats, img_metas, out=None):
        """The image pyramid [(B,C,H_l,W_l)] as the tokens ``prepare_image_inputs`` would build from it
        (:570-591: flatten + concat; here one tiled-transpose launch with the padding rows zeroed, bf16 rows in
        the bf16 compute mode) in the dict form that ``forward`` / ``prepare_image_inputs`` accept in place of
        the pyramid - or None where that form does not apply (CPU tensors, a pyramid that needs a gradient,
        project-then-sample).  ``out``: the dict of an earlier call - its token buffer is overwritten in place:
        a captured step (engine.Trainer.capture) converts each new batch straight out of the caller's tensors
        into its static token buffer instead of copying the 152 MB pyramid into static maps first and
        converting inside the graph."""
        feats = list(mlvl_feats)
        on_gpu = feats[0].is_cuda and not any(f.requires_grad for f in feats) and all(f.is_contiguous() for f in feats)
        spatial = [tuple(f.shape[-2:]) for f in feats]
        S = sum(h * w for h, w in spatial)
        if not self._sample_first(on_gpu, feats[0].shape[1], S) or feats[0].dtype != torch.float32:
            return None
        mt = self._meta_tensors(img_metas, spatial, feats[0].device, feats[0].dtype)
        bf16 = BF16_TOKENS and ops.get_compute_dtype() == "bf16"
        if out is not None and list(out["spatial"]) != spatial:
            raise ValueError("pyramid_tokens: `out` holds levels %s, the pyramid has %s" % (out["spatial"], spatial))
        tokens = ops.pyramid_to_tokens(feats, mt["mask_u8"], bf16=bf16, out=None if out is None else out["tokens"])
        return out if out is not None else dict(tokens=tokens, spatial=spatial, padding_zeroed=True)

    def prepare_image_inputs(self, mlvl_feats, img_metas):
        """The part of prepare_decoder_inputs (:556-594) that depends only on the image pyramid:
        padding masks, flattened tokens, valid ratios - plus the per-layer value projection of
        the fusion attention.  Independent of the point stream, so the detector runs it on a
        side stream while furthest-point sampling occupies 8 of the 256 CUs."""
        prepared = False
        if isinstance(mlvl_feats, dict):
            # channels-last tokens (B,S,C) straight from demf_amd.modules.ImageStream.tokens():
            # no flatten + concat copy of the pyramid (:570-591)
            spatial, feat_flatten = list(mlvl_feats["spatial"]), mlvl_feats["tokens"]
            on_gpu, C0 = feat_flatten.is_cuda and not feat_flatten.requires_grad, feat_flatten.shape[2]
            prepared = bool(mlvl_feats.get("padding_zeroed"))       # built by pyramid_tokens
        else:
            spatial = [tuple(f.shape[-2:]) for f in mlvl_feats]
            feat_flatten = None
            on_gpu = mlvl_feats[0].is_cuda and not any(f.requires_grad for f in mlvl_feats) and \
                all(f.is_contiguous() for f in mlvl_feats)
            C0 = mlvl_feats[0].shape[1]
        dev = mlvl_feats["tokens"].device if isinstance(mlvl_feats, dict) else mlvl_feats[0].device
        dt = mlvl_feats["tokens"].dtype if isinstance(mlvl_feats, dict) else mlvl_feats[0].dtype
        if dt == torch.bfloat16:                 # bf16 token rows (bf16 compute mode): the constants stay fp32
            dt = torch.float32
        mt = self._meta_tensors(img_metas, spatial, dev, dt)
        mask_flatten, valid_ratios = mt["mask_flatten"], mt["valid_ratios"]
        S = mask_flatten.shape[1]
        sample_first = self._sample_first(on_gpu, C0, S)
        if prepared and not sample_first:
            raise ValueError("tokens with zeroed padding rows serve the sample-then-project attention only")
        if feat_flatten is None:
            if on_gpu:           # tiled transposes; the padding mask rides along when wanted
                # (bf16 compute mode, sample-then-project: the tokens are an operand of nothing but the gather
                # in front of the bf16 value projection - bf16 rows, half of the 152 MB)
                feat_flatten = ops.pyramid_to_tokens(mlvl_feats, mt["mask_u8"] if sample_first else None,
                                                     bf16=sample_first and BF16_TOKENS and
                                                     ops.get_compute_dtype() == "bf16")
            else:
                feat_flatten = torch.cat([f.flatten(2).transpose(1, 2) for f in mlvl_feats], 1)
        elif sample_first and not prepared:
            feat_flatten = feat_flatten.masked_fill(mask_flatten.unsqueeze(-1), 0.0)
        value_tokens = (feat_flatten, mt["keep4"]) if sample_first else None
        value_projected = None
        feat_flatten = feat_flatten.permute(1, 0, 2)
        if value_tokens is None:
            value_projected = [layer.layer.attentions[1].project_value(feat_flatten, mask_flatten)
                               for layer in self.decoder]
        return dict(feat_flatten=feat_flatten, mask_flatten=mask_flatten, spatial=spatial,
                    spatial_shapes=mt["spatial_shapes"], level_start_index=mt["level_start_index"],
                    valid_ratios=valid_ratios, value_projected=value_projected,
                    value_tokens=value_tokens)

    def prepare_decoder_inputs(self, seeds_3d, mlvl_feats, img_metas):
        ii = self.prepare_image_inputs(mlvl_feats, img_metas)
        reference_points = self.get_reference_points(seeds_3d, img_metas, ii["spatial"])
        return ii["feat_flatten"], ii["mask_flatten"], reference_points, ii["spatial_shapes"], \
            ii["level_start_index"], ii["valid_ratios"]

    # ---- loss: :596-712 ------------------------------------------------------
    def loss(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask=None,
             pts_instance_mask=None, img_metas=None, gt_bboxes_ignore=None, vote_pack=None):
        bbox_preds = dict(bbox_preds)
        decode_res_all = bbox_preds.pop("decode_res_all")
        targets = self.get_targets(points, gt_bboxes_3d, gt_labels_3d, bbox_preds, vote_pack)
        assert self.num_fusion_layers + 1 == len(decode_res_all)
        if all(self._fusable(d) for d in decode_res_all):
            # device path: one 7-vector per decode layer, averaged as a vector; the vote loss does
            # not depend on the decode layer, so it is evaluated once ((v+v)/2 == v exactly).
            # "_total" (sum of all eight) is what a training loop should differentiate: it avoids
            # the per-entry select/stack nodes of the dict.
            # (the fused kernels read the raw rows + the vote / seed tensors of the outer dict only: no
            # merged dict, which would materialise the lazy "dir_res" of every decode result)
            vecs, vote = zip(*[self._loss_fused(bbox_preds, targets, d["_rows"],
                                                 with_vote=(i == 0))
                               for i, d in enumerate(decode_res_all)])
            if vecs[0].is_cuda and len(vecs) <= 4:
                out8 = ops.loss_total(list(vecs), vote[0])            # one launch each way
                mean7, total = out8[:7], out8[7]
            else:
                mean7 = vecs[0]
                for v in vecs[1:]:
                    mean7 = mean7 + v
                mean7 = mean7 / len(vecs)
                total = mean7.sum() + vote[0]
            losses = dict(vote_loss=vote[0])
            for i, name in enumerate(ops.HEAD_LOSS_NAMES):
                losses[name] = mean7[i]
            losses["_total"] = total
            return losses
        losses_all = [self._loss({**bbox_preds, **d}, targets) for d in decode_res_all]
        return {k: sum(l[k] for l in losses_all) / (self.num_fusion_layers + 1)
                for k in losses_all[0]}

    def _fusable(self, decode_res):
        rows = decode_res.get("_rows")
        c = self.loss_cfg
        return rows is not None and rows[0].is_cuda and rows[0].shape[-1] == 12 and \
            rows[1].shape[-1] == 30 and c["semantic"] is not None and bool(c["iou"])

    def _loss(self, bbox_preds, targets):
        (vote_targets, vote_target_masks, dir_class_targets, dir_res_targets, mask_targets,
         objectness_targets, objectness_weights, box_loss_weights, distance_targets,
         dir_targets, size_targets, center_targets) = targets
        c = self.loss_cfg
        if self._fusable(bbox_preds):
            seven, vote = self._loss_fused(bbox_preds, targets, bbox_preds["_rows"])
            losses = dict(vote_loss=vote)
            for i, name in enumerate(ops.HEAD_LOSS_NAMES):
                losses[name] = seven[i]
            return losses
        vote_loss = self.vote_module.get_loss(bbox_preds["seed_points"], bbox_preds["vote_points"],
                                              bbox_preds["seed_indices"], vote_target_masks,
                                              vote_targets)
        ocw = c["objectness"].get("class_weight")
        if ocw is not None:
            dev = bbox_preds["obj_scores"].device
            cw = self.__dict__.setdefault("_ocw", {})
            if str(dev) not in cw:
                cw[str(dev)] = torch.tensor(ocw, dtype=torch.float32, device=dev)
            ocw = cw[str(dev)]
        objectness_loss = L.cross_entropy_sum(bbox_preds["obj_scores"].transpose(2, 1),
                                              objectness_targets, objectness_weights, ocw,
                                              c["objectness"].get("loss_weight", 1.0))
        w3 = box_loss_weights.unsqueeze(-1).repeat(1, 1, 3)
        size_reg_loss = L.smooth_l1_sum(bbox_preds["size"], size_targets, w3,
                                        c["size_res"].get("beta", 1.0),
                                        c["size_res"].get("loss_weight", 1.0))
        center_loss = L.smooth_l1_sum(bbox_preds["center"], center_targets, w3,
                                      c["center"].get("beta", 1.0),
                                      c["center"].get("loss_weight", 1.0))
        dir_class_loss = L.cross_entropy_sum(bbox_preds["dir_class"].transpose(2, 1),
                                             dir_class_targets, box_loss_weights, None,
                                             c["dir_class"].get("loss_weight", 1.0))
        one_hot = F.one_hot(dir_class_targets, self.num_dir_bins).to(vote_targets.dtype)
        dir_res_norm = torch.sum(bbox_preds["dir_res_norm"] * one_hot, -1)
        dir_res_loss = L.smooth_l1_sum(dir_res_norm, dir_res_targets, box_loss_weights,
                                       c["dir_res"].get("beta", 1.0),
                                       c["dir_res"].get("loss_weight", 1.0))
        losses = dict(vote_loss=vote_loss, objectness_loss=objectness_loss,
                      dir_class_loss=dir_class_loss, dir_res_loss=dir_res_loss,
                      size_res_loss=size_reg_loss, center_loss=center_loss)
        if c["semantic"] is not None:
            losses["semantic_loss"] = L.cross_entropy_sum(
                bbox_preds["sem_scores"].transpose(2, 1), mask_targets, box_loss_weights, None,
                c["semantic"].get("loss_weight", 1.0))
        if c["iou"]:
            corners_pred = self.bbox_coder.decode_corners(bbox_preds["center"], bbox_preds["size"])
            corners_target = self.bbox_coder.decode_corners(center_targets, size_targets)
            losses["iou_loss"] = L.axis_aligned_iou_loss_sum(corners_pred, corners_target,
                                                             box_loss_weights,
                                                             c["iou"].get("loss_weight", 1.0))
        return losses

    def _loss_fused(self, bbox_preds, targets, rows, with_vote=True):
        """The same eight losses through csrc/loss.hip: one kernel for the seven per-proposal
        reductions and one for the vote loss (instead of ~100 small kernels each way).
        -> (seven (7,) in ops.HEAD_LOSS_NAMES order, vote loss | None)"""
        (vote_targets, vote_target_masks, dir_class_targets, dir_res_targets, mask_targets,
         objectness_targets, objectness_weights, box_loss_weights, distance_targets,
         dir_targets, size_targets, center_targets) = targets
        c = self.loss_cfg
        cw = c["objectness"].get("class_weight") or [1.0, 1.0]
        hyper = (cw[0], cw[1], c["objectness"].get("loss_weight", 1.0),
                 c["dir_class"].get("loss_weight", 1.0), c["dir_res"].get("loss_weight", 1.0),
                 c["size_res"].get("loss_weight", 1.0), c["center"].get("loss_weight", 1.0),
                 c["semantic"].get("loss_weight", 1.0), c["iou"].get("loss_weight", 1.0),
                 c["dir_res"].get("beta", 1.0), c["size_res"].get("beta", 1.0),
                 c["center"].get("beta", 1.0))
        cls_rows, reg_rows, base = rows
        R = cls_rows.shape[0] * cls_rows.shape[1]
        seven = ops.head_loss(
            cls_rows.reshape(R, 12), reg_rows.reshape(R, 30), base.reshape(R, 3).contiguous(), hyper,
            center_targets.reshape(R, 3), size_targets.reshape(R, 3),
            dir_class_targets.reshape(R), dir_res_targets.reshape(R).contiguous(),
            mask_targets.reshape(R), objectness_targets.reshape(R),
            objectness_weights.reshape(R).contiguous(), box_loss_weights.reshape(R).contiguous())
        vote = ops.vote_loss(bbox_preds["vote_points"], bbox_preds["seed_points"],
                             bbox_preds["seed_indices"], vote_target_masks, vote_targets,
                             self.gt_per_seed, self.vote_module.vote_loss_dst_weight) \
            if with_vote else None
        return seven, vote

    # ---- test-time decode + NMS: :714-754 --------------------------------------------
    @torch.no_grad()
    def get_bboxes(self, points, bbox_preds, input_metas, rescale=False, use_nms=True):
        """Same contract as the reference: the decode results of ``test_cfg.ensemble_layers`` are
        concatenated, boxes holding <= 5 points are dropped, class-aware aligned NMS
        (csrc/postprocess.hip) and the score threshold select the survivors, and every survivor is
        reported once per class (``per_class_proposal``).  -> list of (boxes, scores, labels) per
        scene, or the raw (B,K,7) boxes when ``use_nms`` is False."""
        decode_res_all = bbox_preds["decode_res_all"]
        tc = self.test_cfg
        obj, sem, box = [], [], []
        for i in tc["ensemble_layers"]:
            d = decode_res_all[i]
            obj.append(F.softmax(d["obj_scores"], dim=-1)[..., -1])
            sem.append(F.softmax(d["sem_scores"], dim=-1))
            box.append(self.bbox_coder.decode(d))
        obj, sem = torch.cat(obj, 1).contiguous(), torch.cat(sem, 1)
        box = torch.cat(box, 1).contiguous()
        if not use_nms:
            return box
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        bottom, extent, count = ops.box_extent_count(points.contiguous(), box)
        classes = torch.argmax(sem, -1)
        keep = ops.aligned_nms(extent, obj, classes, count > 5, tc["nms_thr"])
        selected = keep & (obj > tc["score_thr"])
        results = []
        for b in range(box.shape[0]):
            sel = selected[b]
            bx, sc, cl = bottom[b][sel], obj[b][sel], classes[b][sel]
            if tc["per_class_proposal"]:
                C = sem.shape[-1]
                ss = sem[b][sel]                                           # (n, C)
                bx = bx.repeat(C, 1)
                sc = (sc[None, :] * ss.t()).reshape(-1)
                cl = torch.arange(C, device=cl.device, dtype=cl.dtype).repeat_interleave(int(sel.sum()))
            wrap = input_metas[b].get("box_type_3d") if input_metas is not None else None
            results.append((wrap(bx, box_dim=bx.shape[-1], with_yaw=self.bbox_coder.with_rot)
                            if wrap is not None else DepthBoxes(bx), sc, cl))
        return results

    # ---- targets: :756-941, batched ----------------------------------------------
    _PAD_CACHE = {}

    @staticmethod
    def pad_gt(gt_bboxes_3d, gt_labels_3d, device, with_slot_labels=False, G=None):
        """list[DepthBoxes|(n,7) tensor], list[(n,) long] -> padded (B,G,7), (B,G), valid (B,G).
        An empty scene gets the reference's single all-zero fake box (:766-773).
        One concatenation + one row gather per tensor: the rows are [0-row | all boxes] and
        [-1, 0 | all labels]; the slot -> row table and the valid mask depend only on the per-scene
        counts and are cached per count signature.  ``with_slot_labels``: the labels keep -1 in the
        padding slots (the form ``ops.gt_prep`` reads ``valid`` from) instead of 0."""
        boxes = [b.tensor if isinstance(b, DepthBoxes) else b for b in gt_bboxes_3d]
        counts = tuple(int(b.shape[0]) for b in boxes)
        if G is None:
            G = max(1, max(counts))                      # (a captured step passes its static slot count)
        elif max(counts) > G:
            raise ValueError(f"a scene has {max(counts)} ground-truth boxes, the padded form holds {G}")
        B = len(boxes)
        if B <= 32 and torch.device(device).type == "cuda" and all(b.is_cuda for b in boxes) \
                and all(l.is_cuda for l in gt_labels_3d):
            # device lists: ONE launch, pointers + counts by value in the kernel arguments - no index
            # tables, no host -> device copy whatever the per-scene counts are (ops.pad_gt_lists)
            gt, slot, valid = ops.pad_gt_lists(boxes, gt_labels_3d, G)
            if with_slot_labels:
                return gt, slot, valid
            return gt, slot.clamp_(min=0), valid
        key = (counts, G, str(device))
        cache = DeMFVoteHead._PAD_CACHE
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            valid = np.zeros((B, G), dtype=bool)
            box_row = np.zeros((B, G), dtype=np.int64)          # row 0 of the box rows: zeros
            lab_row = np.zeros((B, G), dtype=np.int64)          # row 0 of the label rows: -1 (padding)
            at = 0
            for i, n in enumerate(counts):
                valid[i, :max(n, 1)] = True
                box_row[i, :n] = 1 + at + np.arange(n)
                lab_row[i, :n] = 2 + at + np.arange(n)
                if n == 0:
                    lab_row[i, 0] = 1                           # row 1: label 0 of the fake box
                at += n
            cache[key] = (torch.as_tensor(box_row.reshape(-1), device=device),
                          torch.as_tensor(lab_row.reshape(-1), device=device),
                          torch.as_tensor(valid, device=device),
                          torch.zeros((1, 7), dtype=torch.float32, device=device),
                          torch.tensor([-1, 0], dtype=torch.long, device=device))
        box_row, lab_row, valid, zero_row, lab_head = cache[key]
        gt = torch.cat([zero_row] + [b.to(device).float() for b in boxes if b.shape[0]]).index_select(0, box_row)
        slot = torch.cat([lab_head] + [l.to(device) for l, n in zip(gt_labels_3d, counts) if n]) \
            .index_select(0, lab_row).view(B, G)
        if with_slot_labels:
            return gt.view(B, G, 7), slot, valid
        return gt.view(B, G, 7), slot.clamp_(min=0), valid

    @torch.no_grad()
    def vote_targets(self, points, gt_bboxes_3d, gt_labels_3d):
        """The proposal-independent half of get_targets (:828-858): per-point vote targets from
        box membership (first / second / last containing box).  Only needs the inputs, so the
        detector computes it on a side stream.  -> dict incl. the padded GT."""
        if isinstance(points, (list, tuple)):
            points = torch.stack(points)
        if isinstance(gt_bboxes_3d, (list, tuple)) and points.is_cuda:
            # per-scene lists on the device: pad to (B,G,7) / (B,G) with -1 in the padding slots and
            # take the padded path below (2 x (cat + row gather), then gt_prep + vote_targets_k)
            gt_bboxes_3d, gt_labels_3d, _ = self.pad_gt(gt_bboxes_3d, gt_labels_3d, points.device,
                                                        with_slot_labels=True)
        if isinstance(gt_bboxes_3d, (list, tuple)):
            gt, lab, valid = self.pad_gt(gt_bboxes_3d, gt_labels_3d, points.device)
        else:
            # already padded (B,G,7) / (B,G): label -1 marks a padding slot (static-shape loops)
            gt = gt_bboxes_3d
            if gt.is_cuda and gt.is_contiguous() and gt_labels_3d.is_contiguous() and \
                    gt_labels_3d.dtype == torch.int64 and points.is_contiguous() and gt.shape[1] <= 64:
                # device path of the captured step: the GT-only quantities (cos / sin(-yaw),
                # angle2class, valid, clamped labels, gravity centres) in one launch, then
                # vote_targets_k; the torch code below is the specification of both
                prep = ops.gt_prep(gt, gt_labels_3d, self.num_dir_bins)
                vote_targets, masks = ops.vote_targets(points, gt, prep["valid"], prep=prep)
                return dict(gt=gt, lab=prep["lab"], valid=prep["valid"], center=prep["center"],
                            dims=gt[..., 3:6], yaw=gt[..., 6], vote_targets=vote_targets,
                            vote_target_masks=masks, prep=prep)
            valid = gt_labels_3d >= 0
            lab = gt_labels_3d.clamp(min=0)
        B, G = gt.shape[:2]
        p = points[..., :3]
        center = torch.cat([gt[..., :2], gt[..., 2:3] + gt[..., 5:6] * 0.5], dim=-1)  # gravity
        dims, yaw = gt[..., 3:6], gt[..., 6]
        if points.is_cuda and G <= 64 and points.is_contiguous() and gt.is_contiguous():
            # one kernel (csrc/loss.hip: vote_targets_k) instead of ~45 broadcast kernels; the torch
            # code below is its specification (and what the CPU golden tests run)
            vote_targets, masks = ops.vote_targets(points, gt, valid)
            return dict(gt=gt, lab=lab, valid=valid, center=center, dims=dims, yaw=yaw,
                        vote_targets=vote_targets, vote_target_masks=masks)
        rel = p[:, :, None, :] - center[:, None, :, :]                        # (B,N,G,3)
        cs, sn = torch.cos(-yaw)[:, None], torch.sin(-yaw)[:, None]
        lx = rel[..., 0] * cs + rel[..., 1] * sn
        ly = -rel[..., 0] * sn + rel[..., 1] * cs
        half = dims[:, None] * 0.5
        inb = (rel[..., 2].abs() <= half[..., 2]) & (lx.abs() < half[..., 0]) & \
              (ly.abs() < half[..., 1]) & valid[:, None, :]
        # running count of containing boxes: a (G,G) triangular matmul instead of an int64 scan
        tri = torch.triu(torch.ones(G, G, dtype=p.dtype, device=p.device))
        cnt = torch.matmul(inb.to(p.dtype), tri).round().long()
        total = cnt[..., -1:]
        votes = -rel                                                          # centre - point
        pick = lambda m: (votes * m.unsqueeze(-1).to(votes.dtype)).sum(2)
        v1 = pick(inb & (cnt == 1))
        v2 = pick(inb & (cnt == 2))
        vl = pick(inb & (cnt == total))
        s1 = torch.where(total >= 2, v2, v1)
        s2 = torch.where(total >= 3, vl, v1)
        has = total > 0
        vote_targets = torch.cat([v1, s1, s2], dim=-1) * has.to(votes.dtype)
        return dict(gt=gt, lab=lab, valid=valid, center=center, dims=dims, yaw=yaw,
                    vote_targets=vote_targets, vote_target_masks=has.squeeze(-1).long())

    @torch.no_grad()
    def get_targets(self, points, gt_bboxes_3d, gt_labels_3d, bbox_preds, vote_pack=None):
        vp = vote_pack if vote_pack is not None else \
            self.vote_targets(points, gt_bboxes_3d, gt_labels_3d)
        lab, valid, center, dims, yaw = vp["lab"], vp["valid"], vp["center"], vp["dims"], vp["yaw"]
        vote_targets, vote_target_masks = vp["vote_targets"], vp["vote_target_masks"]
        agg = bbox_preds["aggregated_points"]

        # -- proposal targets (:877-934)
        prep = vp.get("prep")
        if prep is not None:
            dir_class_t, dir_res_t = prep["dir_class"], prep["dir_res"]
        else:
            dir_class_t, dir_res_t = self.bbox_coder.angle2class(yaw)
        pos_thr = self.train_cfg["pos_distance_thr"]
        neg_thr = self.train_cfg["neg_distance_thr"]
        if agg.is_cuda and "gt" in vp and vp["gt"].is_contiguous():
            # one kernel (csrc/loss.hip: proposal_targets_k); the torch code below is its spec
            t = ops.proposal_targets(agg.contiguous(), vp["gt"], lab, valid, dir_class_t, dir_res_t,
                                     self.bbox_coder.with_rot, pos_thr, neg_thr,
                                     np.pi / self.num_dir_bins, prep=prep)
            objectness_masks, objectness_targets = t["objectness_masks"], t["objectness"]
            objectness_weights, box_loss_weights = ops.target_weights(objectness_masks,
                                                                      objectness_targets)
            return (vote_targets, vote_target_masks, t["dir_class"], t["dir_res"], t["mask"],
                    objectness_targets, objectness_weights, box_loss_weights, t["distance"],
                    t["dir"], t["size"], t["center"])
        d = ((agg[:, :, None, :] - center[:, None, :, :]) ** 2).sum(-1)       # (B,Q,G) mse-sum
        d = d.masked_fill(~valid[:, None, :], float("inf"))
        distance1, assignment = d.min(-1)
        euc = torch.sqrt(distance1 + 1e-6)
        objectness_masks = ((euc < pos_thr) | (euc > neg_thr)).to(agg.dtype)
        g3 = assignment.unsqueeze(-1).expand(-1, -1, 3)
        center_targets = torch.gather(center, 1, g3)
        size_targets = torch.gather(dims, 1, g3)
        dir_class_targets = torch.gather(dir_class_t, 1, assignment)
        dir_res_targets = torch.gather(dir_res_t, 1, assignment) / (np.pi / self.num_dir_bins)
        dir_targets = torch.gather(yaw, 1, assignment)
        mask_targets = torch.gather(lab, 1, assignment).long()
        canonical = agg - center_targets
        if self.bbox_coder.with_rot:
            Bq = canonical.shape[0] * canonical.shape[1]
            canonical = rotation_3d_in_axis_z(canonical.reshape(Bq, 1, 3),
                                              -dir_targets.reshape(Bq)).reshape(agg.shape)
        half_t = size_targets / 2.0
        distance_targets = torch.cat([half_t - canonical, half_t + canonical], dim=-1)
        inside = (distance_targets >= 0.0).all(dim=-1)
        objectness_targets = ((euc < pos_thr) & inside).long()

        # -- batch normalisation of weights (:797-816)
        objectness_weights = objectness_masks / (objectness_masks.sum() + 1e-6)
        box_loss_weights = objectness_targets.float() / (objectness_targets.sum().float() + 1e-6)
        return (vote_targets, vote_target_masks, dir_class_targets, dir_res_targets, mask_targets,
                objectness_targets, objectness_weights, box_loss_weights, distance_targets,
                dir_targets, size_targets, center_targets)
