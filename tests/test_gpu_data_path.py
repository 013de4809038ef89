"""SURVEY 8(f) rank 4 on the GPU: what ``demf_amd/data.py`` writes - ``.bin`` points with the height
channel, IndoorPointSample, RandomFlip3D + GlobalRotScaleTrans (flip + rotation + scale + translation)
with their ``img_meta`` flow fields, Resize + Pad bookkeeping (configs/demf/demf_votenet.py:184-216) -
is handed to the HIP kernels that consume it and compared with oracle/model.py on the same sample:

  * ``demf_msda_prep_fwd`` (get_reference_points, class_agnostic_vote_head.py:524-547: undo the 3-D
    flow, project with depth2img, redo the 2-D flow, normalise, clamp; x valid ratios,
    transformer.py:62-68) against ``OracleHead.reference_points`` x ``decoder_inputs`` valid ratios;
  * ``demf_gt_prep`` / ``demf_vote_targets`` / ``demf_proposal_targets`` / ``demf_target_weights``
    (get_targets, :756-941) on the AUGMENTED boxes (yaw convention of mmdet3d 0.18.1 depth boxes)
    against ``OracleHead.targets``;
  * the whole forward on that batch (metas -> fused decoder layer -> decode outputs) against the
    fp64 oracle.
"""
import numpy as np
import pytest
import torch

from oracle import deps, fixtures
from oracle.model import OracleDeMF

pytestmark = pytest.mark.gpu

ORI = (530, 730)


def _dataset_sample(rng, tmp_path, i, n_raw=3000, n_pts=1024):
    """One SUN RGB-D-like sample through data.py: raw 6-float records on disk -> points with height,
    sampled, augmented; boxes that own points; the img_meta of the 2-D and 3-D pipelines."""
    from demf_amd import data, synthetic
    xyz = rng.uniform([-2.5, 0.8, -1.2], [2.5, 6.0, 1.6], size=(n_raw, 3))
    ctr = rng.uniform([-1.5, 1.8, -0.8], [1.5, 4.5, 0.2], size=(3, 3))
    dims = rng.uniform([0.6, 0.6, 0.5], [1.6, 1.4, 1.0], size=(3, 3))
    yaw = rng.uniform(-np.pi, np.pi, size=(3, 1))
    # a cluster of points inside every box (so that vote targets exist), in the box frame
    for c, d, a in zip(ctr, dims, yaw[:, 0]):
        loc = rng.uniform(-0.45, 0.45, size=(150, 3)) * d
        ca, sa = np.cos(a), np.sin(a)
        # mmdet3d 0.18.1 depth boxes: canonical = (p - c) rotated by -yaw ... about z
        rot = np.array([[ca, sa, 0], [-sa, ca, 0], [0, 0, 1.0]])
        xyz = np.concatenate([xyz, loc @ rot + c + [0, 0, d[2] / 2]], 0)
    raw = np.concatenate([xyz, rng.uniform(0, 1, size=(len(xyz), 3))], 1).astype(np.float32)   # + rgb
    path = tmp_path / f"{i:06d}.bin"
    raw.tofile(path)
    pts = data.load_points_bin(str(path))                       # (N,4): xyz + height
    boxes = np.concatenate([ctr, dims, yaw], 1).astype(np.float32)
    K = np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]])
    tilt = np.deg2rad(6.0 + i)
    Rt = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
    meta = data.resize_meta(dict(depth2img=data.depth2img_from_calib(K, Rt)), ORI, (1333, 600 - 40 * i))
    pts, _ = data.sample_points(pts, n_pts, rng)
    pts, boxes, meta = data.augment_3d(pts, boxes, meta, rng, flip_ratio=1.0 if i % 2 else 0.0,
                                       translation_std=(0.1, 0.1, 0.05))
    return pts, boxes, rng.integers(0, 10, size=3).astype(np.int64), meta


def _batch(seed, tmp_path, B=3):
    rng = np.random.default_rng(seed)
    s = [_dataset_sample(rng, tmp_path, i) for i in range(B)]
    bis = (max(m["batch_input_shape"][0] for *_, m in s), max(m["batch_input_shape"][1] for *_, m in s))
    metas = [dict(m, batch_input_shape=bis) for *_, m in s]
    pyramid = tuple((int(np.ceil(bis[0] / d)), int(np.ceil(bis[1] / d))) for d in (32, 64, 128, 256))
    return (np.stack([p for p, *_ in s]), [b for _, b, _, _ in s], [l for _, _, l, _ in s], metas, pyramid)


def test_data_path_output_through_the_hip_kernels_vs_oracle(tmp_path):
    from demf_amd import _ffi
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    points, gtb, gtl, metas, pyramid = _batch(7, tmp_path)
    assert any(m["pcd_horizontal_flip"] for m in metas) and not all(m["pcd_horizontal_flip"] for m in metas)
    assert all(m["transformation_3d_flow"] == ["HF", "R", "S", "T"] for m in metas)
    B = len(metas)
    rng = np.random.default_rng(0)
    feats = [rng.standard_normal((B, cfg.head.embed_dims, h, w)).astype(np.float32) for h, w in pyramid]

    ref = OracleDeMF(cfg)
    fixtures.seed_weights(ref, 3)
    ref.train().double()
    losses, T, tg = ref.forward_train(torch.from_numpy(points).double(),
                                      [torch.from_numpy(f).double() for f in feats], metas,
                                      [torch.from_numpy(b).double() for b in gtb],
                                      [torch.from_numpy(l) for l in gtl])
    assert int(tg["vote_target_masks"].sum()) > 100 and int(tg["objectness_targets"].sum()) >= 0

    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, 3)
    model.cuda().train()
    pts_d = torch.from_numpy(points).cuda()
    f_d = [torch.from_numpy(f).cuda() for f in feats]
    gb = [torch.from_numpy(b).cuda() for b in gtb]
    gl = [torch.from_numpy(l).cuda() for l in gtl]
    G = model.forward_head(pts_d, f_d, metas)
    head = model.pts_bbox_head

    # ---- (1) reference points of the aggregated points, straight through demf_msda_prep_fwd --------
    agg = T["aggregated_points"].float().cuda().contiguous()             # the ORACLE's query points
    Q = agg.shape[1]
    H, L, P = cfg.head.num_heads, cfg.head.num_levels, cfg.head.num_points
    spatial = [tuple(f.shape[-2:]) for f in f_d]
    mt = head._meta_tensors(metas, spatial, agg.device, agg.dtype)
    inp = ref.pts_bbox_head.decoder_inputs(T["aggregated_points"], [torch.from_numpy(f).double() for f in feats],
                                           metas)
    vr = mt["valid_ratios"].contiguous()                                  # the product's own, from the metas
    assert torch.allclose(vr.cpu().double(), inp["valid_ratios"].double(), atol=1e-6), "valid ratios"
    shapes = torch.tensor(spatial, dtype=torch.int64, device="cuda")
    R = B * Q
    raw = torch.zeros(R, 3 * H * L * P, device="cuda")                    # zero offsets / logits
    loc = torch.empty(R, H, L, P, 2, device="cuda")
    w = torch.empty(R, H, L, P, device="cuda")
    uvw = torch.empty(R, 4, device="cuda")
    _ffi.call("demf_msda_prep_fwd", R, Q, H, L, P, agg.data_ptr(), mt["M"].data_ptr(), mt["ab"].data_ptr(),
              vr.data_ptr(), shapes.data_ptr(), raw.data_ptr(), loc.data_ptr(), w.data_ptr(), uvw.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    want = (inp["reference_points"][:, :, None].double() * inp["valid_ratios"][:, None].double()).reshape(R, 1, L, 1, 2)
    got = loc.double().cpu()
    err = (got - want).abs().max().item()
    assert err <= 1e-4, f"reference points through msda_prep: {err:.2e}"
    inside = ((inp["reference_points"] > 0) & (inp["reference_points"] < 1)).all(-1).float().mean().item()
    assert 0.2 < inside < 1.0, inside              # both the projected and the clamped branch occur
    assert torch.allclose(w, torch.full_like(w, 1.0 / (L * P)))

    # ---- (2) training targets of the augmented boxes on the HIP target kernels ----------------------
    with torch.no_grad():
        preds_o = dict(aggregated_points=agg.view(B, Q, 3))
        got_t = head.get_targets(pts_d, gb, gl, preds_o)
    names = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets", "mask_targets",
             "objectness_targets", "objectness_weights", "box_loss_weights", "distance_targets",
             "dir_targets", "size_targets", "center_targets")
    for n, t in zip(names, got_t):
        wt = tg[n]
        if wt.dtype in (torch.int64, torch.int32, torch.bool):
            assert torch.equal(t.cpu().long(), wt.long()), "target " + n
        else:
            e = (t.double().cpu() - wt).abs().max().item()
            assert e <= 1e-4 * max(1.0, wt.abs().max().item()), f"target {n}: {e:.2e}"

    # ---- (3) the whole forward on this batch ---------------------------------------------------------
    for k in ("seed_indices", "aggregated_indices"):
        assert torch.equal(G[k].cpu().long(), T[k].long()), k
    for k in ("seed_points", "vote_points", "aggregated_points"):
        e = (G[k].double().cpu() - T[k]).abs().max().item()
        assert e <= 1e-4 * max(1.0, T[k].abs().max().item()), f"{k}: {e:.2e}"
    for i, d in enumerate(T["decode_res_all"]):
        for k, v in d.items():
            e = (G["decode_res_all"][i][k].double().cpu() - v).abs().max().item()
            assert e <= 1e-3 * max(1.0, v.abs().max().item()), f"decode{i}.{k}: {e:.2e}"
    lo = head.loss(G, pts_d, gb, gl, None, None, metas)
    lo.pop("_total")
    for k, v in losses.items():
        np.testing.assert_allclose(lo[k].item(), v.item(), rtol=2e-3, atol=1e-5, err_msg=k)
