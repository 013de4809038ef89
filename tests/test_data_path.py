"""CPU tests of demf_amd/data.py: formats and, above all, that the
augmentation metadata it writes is exactly what the head's reference-point projection
(class_agnostic_vote_head.py:524-547, pinned to the real reference in test_host_logic.py) undoes."""
import numpy as np
import torch

from demf_amd import data, synthetic
from demf_amd.modules.head import compose_projection


def test_points_bin_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((500, 6)).astype(np.float32)
    path = tmp_path / "000001.bin"
    raw.tofile(path)
    pts = data.load_points_bin(str(path))
    assert pts.shape == (500, 4) and pts.dtype == np.float32
    np.testing.assert_array_equal(pts[:, :3], raw[:, :3])
    np.testing.assert_allclose(pts[:, 3], raw[:, 2] - np.percentile(raw[:, 2], 0.99), rtol=1e-6)


def test_depth2img_from_calib_matches_axis_convention():
    K = np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]])
    d2i = data.depth2img_from_calib(K, np.eye(3))
    # a point 2 m in front of the camera (depth +y) on the optical axis lands on the principal point
    uvw = d2i @ np.array([0.0, 2.0, 0.0])
    np.testing.assert_allclose(uvw[:2] / uvw[2], [365.0, 265.0], atol=1e-4)
    # +z (up) moves the pixel up (smaller v)
    uvw = d2i @ np.array([0.0, 2.0, 0.5])
    assert uvw[1] / uvw[2] < 265.0


def _project(points, meta):
    M, au, bu, av, bv = compose_projection(meta)
    p = np.concatenate([points[:, :3], np.ones((len(points), 1))], 1) @ M.T
    return np.stack([p[:, 0] / p[:, 2] * au + bu, p[:, 1] / p[:, 2] * av + bv], 1)


def test_augmentation_metadata_is_undone_by_the_projection():
    """Round trip: for 200 random augmentations, projecting the AUGMENTED points with the
    metadata written by augment_3d gives the same normalised pixel as projecting the ORIGINAL points
    with the un-augmented metadata (mirrored horizontally when the image was flipped too)."""
    rng = np.random.default_rng(1)
    base = data.resize_meta(dict(depth2img=synthetic.depth2img()), (530, 730), (1333, 600))
    pts = rng.uniform([-2.0, 1.0, -1.0], [2.0, 5.0, 1.0], size=(64, 3)).astype(np.float32)
    boxes = np.array([[0.2, 3.0, -0.5, 1.0, 0.8, 0.9, 0.3]], np.float32)
    plain = dict(base, flip=False, transformation_3d_flow=[])
    uv0 = _project(pts, plain)
    flips = 0
    for _ in range(200):
        apts, abox, meta = data.augment_3d(pts, boxes, base, rng, translation_std=(0.1, 0.1, 0.05),
                                            sync_2d=True)
        uv = _project(apts, meta)
        want = uv0.copy()
        if meta["flip"]:                     # mirrored image: u -> (W - u) / (W - 1) in normalised form
            w = meta["img_shape"][1]
            want[:, 0] = (w - uv0[:, 0] * (w - 1)) / (w - 1)
            flips += 1
        np.testing.assert_allclose(uv, want, rtol=0, atol=2e-5)
        # the box follows its points: a point at the box centre stays at the box centre
        c0 = boxes[:, :3] + [0, 0, boxes[0, 5] / 2]
        assert abox.shape == boxes.shape
    assert 60 < flips < 140


def test_resize_meta_and_sampling():
    m = data.resize_meta({}, (530, 730), (1333, 600))
    assert m["img_shape"][:2] == (600, 826) and m["batch_input_shape"] == (608, 832)
    np.testing.assert_allclose(m["scale_factor"][:2], [826 / 730, 600 / 530], rtol=1e-6)
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((30000, 4)).astype(np.float32)
    s, idx = data.sample_points(pts, 20000, rng)
    assert s.shape == (20000, 4) and len(np.unique(idx)) == 20000
    s, idx = data.sample_points(pts[:5000], 20000, rng)
    assert s.shape == (20000, 4)


def test_remap_checkpoint_follows_the_reference_rename():
    """demf/modeling/detectors/demfnet.py:85-101 on a synthetic stage-1 key set: encoder /
    level_embeds keys of img_bbox_head.transformer move to img_encoder, every other img_bbox_head key
    is dropped, everything else passes through; the image-stream keys then load into ImageStream."""
    t = lambda: torch.zeros(1)
    sd = {"pts_backbone.SA_modules.0.mlps.0.layer0.conv.weight": t(),
          "pts_bbox_head.decoder.0.layer.norms.0.weight": t(),
          "img_backbone.conv1.weight": t(), "img_neck.convs.0.conv.weight": t(),
          "img_bbox_head.transformer.encoder.layers.0.attentions.0.sampling_offsets.weight": t(),
          "img_bbox_head.transformer.encoder.layers.5.ffns.0.layers.1.bias": t(),
          "img_bbox_head.transformer.level_embeds": t(),
          "img_bbox_head.transformer.decoder.layers.0.attentions.0.attn.in_proj_weight": t(),
          "img_bbox_head.transformer.reference_points.weight": t(),
          "img_bbox_head.cls_branches.0.weight": t(), "img_bbox_head.query_embedding.weight": t()}
    keys_before = set(sd)
    out = data.remap_checkpoint(sd)
    assert set(sd) == keys_before                                  # input untouched
    assert set(out) == {
        "pts_backbone.SA_modules.0.mlps.0.layer0.conv.weight",
        "pts_bbox_head.decoder.0.layer.norms.0.weight",
        "img_backbone.conv1.weight", "img_neck.convs.0.conv.weight",
        "img_encoder.encoder.layers.0.attentions.0.sampling_offsets.weight",
        "img_encoder.encoder.layers.5.ffns.0.layers.1.bias", "img_encoder.level_embeds"}
    hot, img = data.split_checkpoint(sd)
    assert len(hot) == 2 and len(img) == 5
    # a full stage-1-shaped state: ImageStream's own keys re-homed under img_bbox_head.transformer
    from demf_amd.modules import ImageStream
    stream = ImageStream(base=8, blocks=(1, 1, 1, 1), embed_dims=32, num_layers=2, num_heads=4,
                         feedforward_channels=64, gn_groups=4)
    stage1 = {}
    for k, v in stream.state_dict().items():
        if k.startswith("img_encoder."):
            k = k.replace("img_encoder", "img_bbox_head.transformer")
        stage1[k] = torch.full_like(v, 0.5) if v.is_floating_point() else v
    stage1["img_bbox_head.fc_cls.weight"] = torch.zeros(3)
    _, img = data.split_checkpoint(stage1)
    missing, unexpected = stream.load_state_dict(img, strict=True)
    assert not missing and not unexpected
    assert float(stream.img_encoder.level_embeds.mean()) == 0.5


def test_boxes_follow_their_points_under_augmentation():
    """Membership of points in (elongated, rotated) boxes is invariant under augment_3d - flip,
    rotation, scale and translation - with the box convention the target kernels use
    (geometry.DepthBoxes.points_in_boxes, mmdet3d 0.18.1 depth boxes: rotate => yaw -= angle)."""
    from demf_amd.geometry import DepthBoxes
    rng = np.random.default_rng(3)
    boxes = np.array([[0.2, 3.0, -0.5, 2.4, 0.5, 0.9, 0.3], [-1.0, 2.0, -1.0, 0.4, 1.9, 1.2, -1.1],
                      [1.2, 4.2, 0.0, 1.0, 1.0, 0.6, 2.5]], np.float32)
    pts = rng.uniform([-2.5, 0.8, -1.2], [2.5, 5.5, 1.0], size=(6000, 3)).astype(np.float32)
    inside0 = DepthBoxes(torch.from_numpy(boxes)).points_in_boxes(torch.from_numpy(pts)).numpy()
    assert inside0.sum(0).min() > 30
    base = dict(depth2img=synthetic.depth2img())
    flips = 0
    for _ in range(50):
        apts, abox, meta = data.augment_3d(pts, boxes, base, rng, translation_std=(0.1, 0.1, 0.05))
        flips += meta["pcd_horizontal_flip"]
        assert meta["flip"] is False          # sync_2d=False as demf_votenet.py:198-202
        inside = DepthBoxes(torch.from_numpy(abox)).points_in_boxes(torch.from_numpy(apts)).numpy()
        # fp32 round-off may move a point that sits on a face: allow 0.2 % of the memberships
        assert (inside != inside0).sum() <= 0.002 * inside0.sum(), (inside != inside0).sum()
    assert 10 < flips < 40
