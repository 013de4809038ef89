"""The frozen image stream's convolutional half on csrc/conv.hip (reference configuration
configs/deformdetr/imvotenet_image.py:3-20, run at demf/modeling/detectors/demfnet.py:124-132): implicit-GEMM
convolutions on channels-last rows against fp64 torch convolutions, and the whole ResNet-50 + ChannelMapper at the
reference's channel widths against the CPU oracle (oracle/model.py: OracleImageStream)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures

pytestmark = pytest.mark.gpu


def _rb(t, planes):
    """What the kernel multiplies: fp32 values (3 planes: exact) or their bf16 roundings (1 plane)."""
    return t.double() if planes == 3 else t.bfloat16().double()


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("case", [
    dict(B=2, H=25, W=35, Cin=64, Cout=256, k=1, s=1, p=0),                       # 1 750 rows: partial 128-row tile
    dict(B=3, H=25, W=35, Cin=256, Cout=128, k=1, s=2, p=0),                      # the strided downsample form
    dict(B=2, H=25, W=35, Cin=64, Cout=64, k=3, s=1, p=1, resid=True, relu=True),  # 64-column tiles
    dict(B=2, H=25, W=35, Cin=128, Cout=128, k=3, s=2, p=1, relu=True),           # odd sizes, stride 2: 13 x 18
    dict(B=1, H=13, W=18, Cin=512, Cout=256, k=3, s=2, p=1),                      # the neck's extra level: 7 x 9
    dict(B=1, H=9, W=7, Cin=32, Cout=192, k=3, s=1, p=1, resid=True),             # Cout = 3 x 64
    dict(B=2, H=13, W=18, Cin=512, Cout=256, k=3, s=2, p=1, plain=True),          # split-K: 126 rows x K = 4 608
    dict(B=1, H=20, W=20, Cin=1024, Cout=256, k=1, s=1, p=0, plain=True),         # split-K on a 1x1 (8 tiles)
    # >= 16 384 pixels, 1 x 1 stride 1, 64 / 128 / 256 input channels: the weight-resident streaming kernels
    # (csrc/conv.hip conv1x1_c64_stream_kernel / conv1x1_stream_kernel); 129 x 131 = 16 899 pixels: a ragged last tile
    dict(B=1, H=129, W=131, Cin=64, Cout=256, k=1, s=1, p=0, resid=True, relu=True),
    dict(B=1, H=129, W=131, Cin=64, Cout=256, k=1, s=1, p=0),
    dict(B=1, H=129, W=131, Cin=64, Cout=64, k=1, s=1, p=0, relu=True),
    dict(B=1, H=129, W=131, Cin=128, Cout=512, k=1, s=1, p=0, resid=True, relu=True),
    dict(B=1, H=129, W=131, Cin=256, Cout=1024, k=1, s=1, p=0, resid=True, relu=True),
    dict(B=1, H=129, W=131, Cin=256, Cout=64, k=1, s=1, p=0, relu=True),
    dict(B=2, H=128, W=64, Cin=256, Cout=128, k=1, s=1, p=0),
])
def test_conv_nhwc_vs_fp64(case, planes):
    from demf_amd import ops
    g = torch.Generator().manual_seed(7)
    c = dict(resid=False, relu=False, plain=False)
    c.update(case)
    x = torch.randn(c["B"], c["Cin"], c["H"], c["W"], generator=g)
    w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"], generator=g) / (c["Cin"] * c["k"] ** 2) ** 0.5
    scale = torch.rand(c["Cout"], generator=g) + 0.5
    bias = torch.randn(c["Cout"], generator=g)
    ws = w * scale.view(-1, 1, 1, 1)
    if c["plain"]:
        bias = None
    want = F.conv2d(_rb(x, planes), _rb(ws, planes), None if bias is None else bias.double(), stride=c["s"],
                    padding=c["p"])
    res = None
    if c["resid"]:
        res = torch.randn(want.shape, generator=g)
        want = want + res.double()
    if c["relu"]:
        want = want.relu()
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    if c["plain"]:
        from demf_amd import _ffi
        seen = []
        orig = _ffi.call
        _ffi.call = lambda name, *a: (seen.append(a[15]) if name == "demf_conv_nhwc_f32" else None, orig(name, *a))[1]
    got = ops.conv_nhwc(xc, ops.conv_weight_planes(w.cuda(), planes, scale.cuda()),
                        None if bias is None else bias.cuda(), c["k"], c["k"],
                        c["s"], c["p"], resid=None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda(),
                        relu=c["relu"])
    if c["plain"]:
        _ffi.call = orig
        assert seen and seen[0] > 1, "the split-K form must be the one under test"
    assert tuple(got.shape) == (want.shape[0], want.shape[2], want.shape[3], want.shape[1])
    err = (got.cpu().double().permute(0, 3, 1, 2) - want).abs().max().item()
    tol = 3e-6 if planes == 3 else 2e-5            # fp32 accumulation of exact products
    assert err <= tol * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("planes", [3, 1])
def test_stem_maxpool_groupnorm_vs_torch(planes):
    from demf_amd import ops
    g = torch.Generator().manual_seed(9)
    img = torch.randn(2, 3, 50, 70, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    want = F.conv2d(_rb(img, planes), _rb(w * scale.view(-1, 1, 1, 1), planes), bias.double(), stride=2, padding=3).relu()
    got = ops.conv_stem7(img.cuda(), ops.stem_weight_planes(w.cuda(), planes, scale.cuda()), bias.cuda(), relu=True)
    err = (got.cpu().double().permute(0, 3, 1, 2) - want).abs().max().item()
    assert tuple(got.shape) == (2, 25, 35, 64) and err <= (3e-6 if planes == 3 else 2e-5) * want.abs().max().item(), err
    # max-pool 3x3 s2 p1 on odd sizes: bit-exact
    pooled = ops.maxpool3x3s2_nhwc(got)
    wantp = F.max_pool2d(got.permute(0, 3, 1, 2), 3, stride=2, padding=1)
    assert torch.equal(pooled.permute(0, 3, 1, 2), wantp)
    # GroupNorm(32, 256) into rows [5, 5 + h*w) of a token buffer
    x = torch.randn(3, 256, 13, 18, generator=g) * 3 + 1
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    wantg = F.group_norm(x.double(), 32, gam.double(), bet.double(), 1e-5)
    tok = torch.full((3, 5 + 13 * 18 + 2, 256), 7.0, device="cuda")
    ops.groupnorm_nhwc_into(x.permute(0, 2, 3, 1).contiguous().cuda(), 32, gam.cuda(), bet.cuda(), 1e-5, tok, 5)
    gotg = tok[:, 5:5 + 13 * 18].cpu().double().view(3, 13, 18, 256).permute(0, 3, 1, 2)
    assert (gotg - wantg).abs().max().item() <= 1e-5
    assert (tok[:, :5] == 7).all() and (tok[:, 5 + 13 * 18:] == 7).all()


REAL_WIDTHS = dict(base=64, blocks=(2, 1, 2, 1), embed_dims=256, num_layers=1, num_heads=8,
                   feedforward_channels=256, gn_groups=32, num_feats=128)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_backbone_and_neck_on_conv_kernels_vs_oracle(mode):
    """ResNet (bottlenecks at the reference's widths 64 .. 2048, stem, max-pool, strided blocks, downsample
    branches) + ChannelMapper (three 1x1 levels, the 3x3 s2 extra level, GroupNorm(32)) on csrc/conv.hip, padded
    128 x 224 images: the token pyramid against the CPU oracle's NCHW pyramid.  fp32-grade mode: 1e-4 of each
    level's scale (the library-convolution path is held to 2e-3 / 5e-3 in tests/test_gpu_image_stream.py); bf16
    mode: the bound of the bf16 GEMM tests.  And tokens() / forward() of the stream through the encoder."""
    from demf_amd import ops
    from demf_amd.modules import ImageStream
    from oracle.model import OracleImageStream
    img, metas = fixtures.make_images(12, B=2, H=128, W=224)
    ref = OracleImageStream(**REAL_WIDTHS)
    fixtures.seed_weights(ref, 12)
    with torch.no_grad():
        pyr = ref.img_neck(ref.img_backbone(torch.from_numpy(img)))
        want_enc = ref.img_encoder(pyr, metas)
    m = ImageStream(**REAL_WIDTHS)
    fixtures.seed_weights(m, 12)
    m.cuda()
    x = torch.from_numpy(img).cuda()
    ops.set_compute_dtype(mode)
    try:
        assert m._conv_ok(x), "the convolution kernel path must be the one under test"
        got = m.pyramid(x)
        enc = m(x, metas)
    finally:
        ops.set_compute_dtype("f32")
    assert isinstance(got, dict) and got["spatial"] == [tuple(p.shape[-2:]) for p in pyr]
    start = 0
    tol = 1e-4 if mode == "f32" else 4e-2
    for p in pyr:
        h, w = p.shape[-2:]
        g = got["tokens"][:, start:start + h * w].cpu().view(2, h, w, 256).permute(0, 3, 1, 2)
        start += h * w
        err = (g - p).abs().max().item()
        assert err <= tol * max(1.0, p.abs().max().item()), (tuple(p.shape), err)
    assert start == got["tokens"].shape[1]
    # through the encoder (kernel path at 256 dims): image positions, 2e-4 (one conv stack + one encoder layer)
    from test_gpu_image_stream import _assert_close_on_image
    for w_, g_ in zip(want_enc, enc):
        _assert_close_on_image(g_.cpu(), w_, metas, 2e-4 if mode == "f32" else 6e-2)


def test_conv_path_falls_back_and_rejects_cpu():
    from demf_amd import ops
    from demf_amd.modules import ImageStream
    tiny = ImageStream(**fixtures.TINY_IMAGE_STREAM).cuda()           # 8-channel base: library convolutions
    assert not tiny._conv_ok(torch.zeros(1, 3, 64, 96, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.conv_nhwc(torch.zeros(1, 4, 4, 32), torch.zeros(1, 64, 32, dtype=torch.bfloat16), None, 1, 1)
