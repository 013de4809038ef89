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
        apts, abox, meta = data.augment_3d(pts, boxes, base, rng, translation_std=(0.1, 0.1, 0.05))
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


def test_remap_checkpoint_splits_by_prefix():
    sd = {"pts_backbone.SA_modules.0.mlps.0.layer0.conv.weight": torch.zeros(1),
          "pts_bbox_head.decoder.0.layer.norms.0.weight": torch.zeros(1),
          "img_backbone.conv1.weight": torch.zeros(1), "img_encoder.level_embeds": torch.zeros(1),
          "img_bbox_head.fc_cls.weight": torch.zeros(1)}
    hot, img = data.remap_checkpoint(sd)
    assert len(hot) == 2 and len(img) == 2
