"""CPU tests of the product's host-side / pure-torch logic (no HIP kernels involved):
state-dict compatibility with the oracle (and therefore the reference's module names),
the composed reference-point projection, padding masks / valid ratios, and the batched,
loop-free target generation - all against golden vectors from the REAL reference."""
import os

import numpy as np
import pytest
import torch

from demf_amd.modules import DeMFHotPath
from oracle import fixtures
from oracle.model import OracleDeMF



@pytest.fixture(autouse=True)
def _torch_linear_on_cpu(monkeypatch):
    """The product's ops.linear is device-only (no CPU path); the glue exercised here contains
    one value projection, which this CPU test runs through torch's own linear instead."""
    import torch.nn.functional as F
    from demf_amd import ops
    def linear(x, w, b=None, row_mask=None):
        y = F.linear(x, w, b)
        return y if row_mask is None else y.masked_fill(row_mask.unsqueeze(-1), 0.0)
    monkeypatch.setattr(ops, "linear", linear)


NAMES = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets",
         "mask_targets", "objectness_targets", "objectness_weights", "box_loss_weights",
         "distance_targets", "dir_targets", "size_targets", "center_targets")


def _setup(name, seed, B, n_gt, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"ref_head_{name}.npz"))
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(B, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                      cfg.head.embed_dims, seed=seed, n_gt=n_gt)
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, seed)
    return gold, cfg, batch, model


def test_state_dict_keys_match_reference_naming():
    cfg = fixtures.tiny_cfg()
    a, b = DeMFHotPath(cfg).state_dict(), OracleDeMF(cfg).state_dict()
    assert list(sorted(a)) == list(sorted(b))
    for k in a:
        assert a[k].shape == b[k].shape, k
    from demf_amd.config import DeMFCfg
    full = DeMFHotPath(DeMFCfg())
    n = sum(p.numel() for p in full.parameters() if p.requires_grad)
    assert n == 2189975  # ~2.19 M trainable parameters
    groups = full.param_groups()
    assert abs(groups[1]["lr"] - 0.008 * 0.05) < 1e-12 and len(groups[1]["params"]) > 0


@pytest.mark.parametrize("name,seed,B,n_gt", [("tiny_a", 1, 2, 4), ("tiny_b", 2, 3, 2)])
def test_decoder_inputs_vs_reference(name, seed, B, n_gt, golden_dir):
    gold, cfg, batch, model = _setup(name, seed, B, n_gt, golden_dir)
    head = model.pts_bbox_head
    feats = [torch.from_numpy(f) for f in batch["img_features"]]
    agg = torch.from_numpy(gold["aggregated_points"])
    ff, mf, rp, ss, lsi, vr = head.prepare_decoder_inputs(agg, feats, batch["img_metas"])
    np.testing.assert_allclose(rp.numpy(), gold["reference_points"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(vr.numpy(), gold["valid_ratios"])
    np.testing.assert_array_equal(np.packbits(mf.numpy(), axis=1), gold["mask_flatten"])
    np.testing.assert_array_equal(ss.numpy(), gold["spatial_shapes"])
    np.testing.assert_array_equal(lsi.numpy(), gold["level_start_index"])
    np.testing.assert_array_equal(ff[::37, :, ::5].numpy(), gold["feat_flatten_probe"])


@pytest.mark.parametrize("name,seed,B,n_gt", [("tiny_a", 1, 2, 4), ("tiny_b", 2, 3, 2)])
def test_batched_targets_vs_reference_loops(name, seed, B, n_gt, golden_dir):
    gold, cfg, batch, model = _setup(name, seed, B, n_gt, golden_dir)
    head = model.pts_bbox_head
    gtb = [torch.from_numpy(gold[f"gt_boxes.{b}"]) for b in range(B)]
    gtl = [torch.from_numpy(gold[f"gt_labels.{b}"]) for b in range(B)]
    preds = dict(aggregated_points=torch.from_numpy(gold["aggregated_points"]))
    t = head.get_targets(torch.from_numpy(batch["points"]), gtb, gtl, preds)
    for n, v in zip(NAMES, t):
        g = gold["target." + n]
        if v.dtype == torch.long:
            np.testing.assert_array_equal(v.numpy(), g, err_msg=n)
        else:
            np.testing.assert_allclose(v.numpy(), g, rtol=1e-5, atol=1e-6, err_msg=n)


def test_targets_empty_scene_and_padding():
    cfg = fixtures.tiny_cfg()
    head = DeMFHotPath(cfg).pts_bbox_head
    pts = torch.rand(2, 500, 4) * 4
    box = torch.tensor([[2.0, 2.0, 1.0, 1.5, 1.5, 1.5, 0.3]])
    agg = torch.rand(2, 32, 3) * 4
    t = head.get_targets(pts, [box, torch.zeros(0, 7)], [torch.tensor([3]), torch.zeros(0, dtype=torch.long)],
                         dict(aggregated_points=agg))
    vt, vm = t[0], t[1]
    assert vm[1].sum() == 0 and (vt[1] == 0).all()      # empty scene: the zero fake box holds nothing
    assert (t[4][1] == 0).all()                          # labels of the fake box
    assert vm[0].sum() > 0
    # a point in exactly one box carries that vote in all three slots (reference :852-854)
    i = int(torch.nonzero(vm[0])[0])
    assert torch.equal(vt[0, i, 0:3], vt[0, i, 3:6]) and torch.equal(vt[0, i, 0:3], vt[0, i, 6:9])


def test_pad_gt_slots_and_lazy_dir_res():
    """pad_gt: per-scene GT lists -> (B,G,7) / (B,G) by one concatenation + one row gather each; an empty
    scene gets the reference's all-zero fake box with label 0 (class_agnostic_vote_head.py:766-773), padding
    slots are zero boxes with label 0 (or -1 in the slot form the device target kernels read).  split_pred:
    `dir_res` (coder.py:233) is computed on first access only, whichever dict access that is."""
    from demf_amd.modules.head import DeMFVoteHead
    from demf_amd.modules.coder import DeMFClassAgnosticBBoxCoder
    g = torch.Generator().manual_seed(0)
    boxes = [torch.randn(3, 7, generator=g), torch.zeros(0, 7), torch.randn(1, 7, generator=g)]
    labels = [torch.tensor([4, 0, 9]), torch.zeros(0, dtype=torch.long), torch.tensor([2])]
    gt, lab, valid = DeMFVoteHead.pad_gt(boxes, labels, torch.device("cpu"))
    assert gt.shape == (3, 3, 7) and lab.shape == (3, 3)
    assert torch.equal(gt[0], boxes[0]) and torch.equal(gt[2, 0], boxes[2][0])
    assert (gt[1] == 0).all() and (gt[2, 1:] == 0).all()
    assert lab.tolist() == [[4, 0, 9], [0, 0, 0], [2, 0, 0]]
    assert valid.tolist() == [[True, True, True], [True, False, False], [True, False, False]]
    _, slot, _ = DeMFVoteHead.pad_gt(boxes, labels, torch.device("cpu"), with_slot_labels=True)
    assert slot.tolist() == [[4, 0, 9], [0, -1, -1], [2, -1, -1]]
    assert torch.equal(slot >= 0, valid)

    coder = DeMFClassAgnosticBBoxCoder(num_dir_bins=12)
    cls, reg = torch.randn(2, 12, 5, generator=g), torch.randn(2, 30, 5, generator=g)
    res = coder.split_pred(cls, reg, torch.zeros(2, 5, 3))
    # not materialised by split_pred itself (no launch on the training path) ...
    assert not dict.__contains__(res, "dir_res")
    want = res["dir_res_norm"] * (np.pi / 12)
    # ... but every read of the dict contract sees it, as the reference's split_pred result (coder.py:233)
    res2 = coder.split_pred(cls, reg, torch.zeros(2, 5, 3))
    assert "dir_res" in res2 and torch.equal(res2.get("dir_res"), want)
    for view in (dict(coder.split_pred(cls, reg, torch.zeros(2, 5, 3))),
                 {**coder.split_pred(cls, reg, torch.zeros(2, 5, 3))},
                 coder.split_pred(cls, reg, torch.zeros(2, 5, 3)).copy()):
        assert torch.equal(view["dir_res"], want)
        assert torch.equal(view["center"], reg.transpose(2, 1)[..., 0:3])      # base_xyz = 0
    assert "dir_res" in list(coder.split_pred(cls, reg, torch.zeros(2, 5, 3)))
    assert torch.equal(res["dir_res"], want) and "dir_res" in res
    with pytest.raises(KeyError):
        res["no_such_key"]


def test_compute_dtype_names_and_native_override(monkeypatch):
    """ops.set_compute_dtype: 'f32' is the three-term split (mode 2) unless DEMF_F32_NATIVE=1 asks for
    the fp32 MFMA (mode 0); unknown names are rejected; the C entry refuses modes outside 0..2.
    (demf_set_compute_dtype only stores the mode: no GPU needed.)"""
    from demf_amd import _ffi, ops
    seen = []
    real = _ffi.call
    monkeypatch.setattr(_ffi, "call", lambda name, *a: seen.append((name, a)) or real(name, *a))
    try:
        monkeypatch.delenv("DEMF_F32_NATIVE", raising=False)
        monkeypatch.delenv("DEMF_F16_TERMS", raising=False)
        # (mode 2 is followed by demf_set_f16_terms: the two-fp16-term kernel forms on for "f32" / "f32h2", off for "f32x3")
        for name, mode, h2 in (("f32", 2, 1), ("f32x3", 2, 0), ("f32h2", 2, 1), ("f32_native", 0, None), ("bf16", 1, None)):
            ops.set_compute_dtype(name)
            calls = [c for c in seen if c[0] in ("demf_set_compute_dtype", "demf_set_f16_terms")]
            if h2 is None:
                assert calls[-1] == ("demf_set_compute_dtype", (mode,))
            else:
                assert calls[-2:] == [("demf_set_compute_dtype", (mode,)), ("demf_set_f16_terms", (h2,))]
            assert ops.get_compute_dtype() == name
        monkeypatch.setenv("DEMF_F16_TERMS", "0")
        ops.set_compute_dtype("f32")
        assert seen[-1] == ("demf_set_f16_terms", (0,))
        monkeypatch.delenv("DEMF_F16_TERMS", raising=False)
        monkeypatch.setenv("DEMF_F32_NATIVE", "1")
        ops.set_compute_dtype("f32")
        assert seen[-1] == ("demf_set_compute_dtype", (0,))
        with pytest.raises(ValueError):
            ops.set_compute_dtype("fp8")
        with pytest.raises(Exception):
            real("demf_set_compute_dtype", 3)
    finally:
        monkeypatch.delenv("DEMF_F32_NATIVE", raising=False)
        ops.set_compute_dtype("f32")


def test_image_stream_conv_path_host_logic_vs_oracle(monkeypatch):
    """The HOST side of the convolution kernel path (demf_amd/modules/image_stream.py: ImageStream._pyramid_tokens) with
    the four device operators replaced by torch restatements of their C-ABI contract: frozen-BatchNorm folding, the
    (Cout, KH, KW, Cin) reduction order of ops.conv_weight_planes, the stem's 7 x (8 pixels x 4 channels) packing, the
    bottleneck wiring (downsample branch, residual, strides), the neck incl. its 3x3 stride-2 level and GroupNorm
    written into token rows - against the CPU oracle's NCHW pyramid (oracle/model.py) at the reference's channel widths
    (configs/deformdetr/imvotenet_image.py:3-20).  The kernels themselves: tests/test_gpu_conv.py."""
    import torch
    import torch.nn.functional as F
    from oracle import fixtures
    from oracle.model import OracleImageStream
    from demf_amd import ops
    from demf_amd.modules import ImageStream

    def w_of(wp, shape):                                 # planes -> the fp32 weight they sum to
        return wp.float().sum(0).view(*shape)

    def conv_nhwc(x, wp, bias, KH, KW, stride=1, pad=0, resid=None, relu=False, out=None, ksplit=None):
        w = w_of(wp, (wp.shape[1], KH, KW, x.shape[3])).permute(0, 3, 1, 2)
        y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
        y = y if resid is None else y + resid
        return (y.relu() if relu else y).contiguous()

    def conv_stem7(img, wp, bias, relu=True):
        w = w_of(wp, (wp.shape[1], 7, 8, 4))
        assert (w[:, :, 7] == 0).all() and (w[..., 3] == 0).all()          # the padding taps of the packing
        y = F.conv2d(img, w[:, :, :7, :3].permute(0, 3, 1, 2), bias, stride=2, padding=3).permute(0, 2, 3, 1)
        return y.relu().contiguous()

    def maxpool(x):
        return F.max_pool2d(x.permute(0, 3, 1, 2), 3, stride=2, padding=1).permute(0, 2, 3, 1).contiguous()

    def groupnorm_into(x, groups, gamma, beta, eps, tokens, row0):
        B, h, w, C = x.shape
        tokens[:, row0:row0 + h * w] = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma, beta, eps) \
            .permute(0, 2, 3, 1).reshape(B, h * w, C)

    monkeypatch.setattr(ops, "conv_nhwc", conv_nhwc)
    monkeypatch.setattr(ops, "conv_stem7", conv_stem7)
    monkeypatch.setattr(ops, "maxpool3x3s2_nhwc", maxpool)
    monkeypatch.setattr(ops, "groupnorm_nhwc_into", groupnorm_into)
    cfg = dict(base=64, blocks=(2, 1, 2, 1), embed_dims=256, num_layers=1, num_heads=8, feedforward_channels=256,
               gn_groups=32, num_feats=128)
    img, _ = fixtures.make_images(12, B=2, H=96, W=160)
    ref = OracleImageStream(**cfg)
    fixtures.seed_weights(ref, 12)
    with torch.no_grad():
        want = ref.img_neck(ref.img_backbone(torch.from_numpy(img)))
    m = ImageStream(**cfg)
    fixtures.seed_weights(m, 12)
    with torch.no_grad():
        got = m._pyramid_tokens(torch.from_numpy(img))
    assert got["spatial"] == [tuple(p.shape[-2:]) for p in want]
    start = 0
    for p in want:
        h, w = p.shape[-2:]
        g = got["tokens"][:, start:start + h * w].view(2, h, w, 256).permute(0, 3, 1, 2)
        start += h * w
        assert (g - p).abs().max().item() <= 5e-5 * max(1.0, p.abs().max().item())
    # the pack is cached per (parameter versions, planes) and rebuilt by load_state_dict
    pk = m._conv_pack(3)
    assert m._conv_pack(3) is pk
    m.load_state_dict(m.state_dict())
    assert m._conv_pack(3) is not pk
