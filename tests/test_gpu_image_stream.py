"""The frozen image stream on the GPU: against golden vectors of the REAL
reference DeformableDetrEncoder, against the CPU oracle at a larger size, and the channels-last
token hand-over to the head."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures
from oracle.model import OracleImageStream

pytestmark = pytest.mark.gpu


def _valid(metas, shape):
    """(B,1,h,w) bool: positions of a level that lie on the image, not on its padding.  On padded
    positions the reference's positional encoding evaluates sin/cos of arguments ~1e6
    ((0 - 0.5) / (0 + 1e-6) * 2pi): the last bit of the argument decides the value, so those
    outputs are not comparable between any two libm's - and nothing consumes them (every consumer
    masks the padding)."""
    from demf_amd.geometry import level_masks
    return torch.from_numpy(~level_masks(metas, [shape])[0])[:, None]


def _assert_close_on_image(got, want, metas, tol):
    v = _valid(metas, tuple(want.shape[-2:])).expand_as(want)
    err = ((got - want).abs() * v).max().item()
    assert err <= tol * max(1.0, (want.abs() * v).max().item()), err


def _product(cfg, seed):
    from demf_amd.modules import ImageStream
    m = ImageStream(**cfg)
    fixtures.seed_weights(m, seed)
    return m.cuda()


def test_image_stream_vs_real_reference_encoder(golden_dir):
    gold = np.load(os.path.join(golden_dir, "ref_encoder.npz"))
    m = _product(fixtures.TINY_IMAGE_STREAM, 4)
    img, metas = fixtures.make_images(4)
    x = torch.from_numpy(img).cuda()
    pyramid = m.img_neck(m.img_backbone(x))
    # the convolutional part is library code (MIOpen picks Winograd / implicit-GEMM kernels whose
    # fp32 round-off is ~1e-4 relative per layer): checked loosely ...
    for i, p in enumerate(pyramid):
        np.testing.assert_allclose(p.cpu().numpy(), gold[f"neck{i}"], rtol=2e-3, atol=2e-3)
    # ... and the encoder - this package's MSDA kernel + linears - on the reference's own inputs
    neck = [torch.from_numpy(gold[f"neck{i}"]).cuda() for i in range(4)]
    for i, o in enumerate(m.img_encoder(neck, metas)):
        _assert_close_on_image(o.cpu(), torch.from_numpy(gold[f"enc{i}"]), metas, 1e-4)


def test_image_stream_vs_oracle_mid_size():
    """128x192 images, 3 scenes, wider network: the encoder within 1e-4 (relative to the output
    scale) of the CPU oracle on identical inputs, the whole stream within the library-convolution
    tolerance; tokens() equals forward() re-laid out."""
    cfg = dict(base=16, blocks=(2, 2, 2, 2), embed_dims=64, num_layers=3, num_heads=8,
               feedforward_channels=128, gn_groups=16, num_feats=32)
    img, metas = fixtures.make_images(9, B=3, H=128, W=192)
    ref = OracleImageStream(**cfg)
    fixtures.seed_weights(ref, 9)
    want = ref(torch.from_numpy(img), metas)
    m = _product(cfg, 9)
    x = torch.from_numpy(img).cuda()
    got = m(x, metas)
    for w, g in zip(want, got):                    # end to end: library-convolution tolerance
        assert tuple(w.shape) == tuple(g.shape)
        _assert_close_on_image(g.cpu(), w, metas, 5e-3)
    with torch.no_grad():                          # encoder alone, on the oracle's pyramid: 1e-4
        pyr = ref.img_neck(ref.img_backbone(torch.from_numpy(img)))
        want_enc = ref.img_encoder(pyr, metas)
    for w, g in zip(want_enc, m.img_encoder([p.cuda() for p in pyr], metas)):
        _assert_close_on_image(g.cpu(), w, metas, 1e-4)
    tok = m.tokens(x, metas)
    flat = torch.cat([g.flatten(2).transpose(1, 2) for g in got], 1)
    assert torch.allclose(tok["tokens"], flat, rtol=1e-4, atol=1e-4)   # two runs of library convs
    assert tok["spatial"] == [tuple(g.shape[-2:]) for g in got]


def test_head_accepts_tokens():
    """prepare_image_inputs on the channels-last tokens == on the NCHW pyramid."""
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    head = DeMFHotPath(cfg).pts_bbox_head.cuda()
    raw = fixtures.make_scene_batch(2, 256, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                    cfg.head.embed_dims, seed=3, n_gt=2)
    feats = [torch.from_numpy(f).cuda() for f in raw["img_features"]]
    a = head.prepare_image_inputs(feats, raw["img_metas"])
    tokens = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1).contiguous()
    b = head.prepare_image_inputs(dict(tokens=tokens, spatial=[tuple(f.shape[-2:]) for f in feats]),
                                  raw["img_metas"])
    assert torch.equal(a["feat_flatten"], b["feat_flatten"])
    assert torch.equal(a["mask_flatten"], b["mask_flatten"])
    for x, y in zip(a["value_projected"], b["value_projected"]):
        assert torch.allclose(x, y, rtol=0, atol=1e-6)     # GEMM on differently strided inputs


@pytest.mark.parametrize("variant", ["default", "pre_add", "ln_epilogue"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("num_points", [4, 2])
def test_fused_encoder_layers_match_the_module_path(mode, num_points, variant):
    """embed_dims = 256, 8 heads, 4 levels (the reference's encoder shape, configs/demf/demf_votenet.py:28-47): the
    layers on demf_rows_gemm_f32 + demf_msda_fwd_raw_f32 (5 launches per layer) against the module path - the one
    the REAL-encoder goldens pin above - on padded images (masked value rows, partial 128-row tiles).  fp32-grade
    mode within 1e-4 of the output scale; the bf16 mode against the same bound as the decoder's bf16 GEMMs."""
    from demf_amd import ops
    from demf_amd.modules import image_stream as ims
    from demf_amd.modules.image_stream import DeformableDetrEncoder
    torch.manual_seed(3)
    enc = DeformableDetrEncoder(num_layers=2, embed_dims=256, num_heads=8, num_points=num_points,
                                feedforward_channels=384, num_feats=128).cuda().eval()
    enc.init_weights()
    with torch.no_grad():                       # offsets / logits that actually depend on the query
        for layer in enc.encoder.layers:
            a = layer.attentions[0]
            a.sampling_offsets.weight.normal_(0, 0.02)
            a.attention_weights.weight.normal_(0, 0.05)
            a.attention_weights.bias.normal_(0, 0.3)
            for n in layer.norms:
                n.weight.uniform_(0.5, 1.5)
                n.bias.normal_(0, 0.1)
    img, metas = fixtures.make_images(5, B=3, H=96, W=160)
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    feats = [torch.randn(3, 256, h, w, device="cuda") for h, w in shapes]
    ops.set_compute_dtype(mode)
    saved = (ims.PRE_ADD, ims.SPLIT_FFN_LN)
    try:
        ims.FUSED_LAYERS = False
        want = enc.forward_tokens(feats, metas)["tokens"]
        ims.FUSED_LAYERS = True
        # the opt-in forms: q = x + pos written by the LayerNorm pass and fed as a second operand
        # (demf_rows_ln_pos_f32, a2_op = 1); FFN-down with the LayerNorm epilogue instead of GEMM + pass
        ims.PRE_ADD = variant == "pre_add"
        ims.SPLIT_FFN_LN = variant != "ln_epilogue"
        assert enc._fused_ok(torch.empty(3, 10, 256, device="cuda"))
        got = enc.forward_tokens(feats, metas)["tokens"]
    finally:
        ims.FUSED_LAYERS = True
        ims.PRE_ADD, ims.SPLIT_FFN_LN = saved
        ops.set_compute_dtype("f32")
    keep = ~enc._static(metas, shapes, feats[0].device)["mask_flatten"]          # tokens on the image
    err = ((got - want).abs() * keep[..., None]).max().item()
    scale = max(1.0, (want.abs() * keep[..., None]).max().item())
    tol = 1e-4 if mode == "f32" else 3e-2
    assert err <= tol * scale, (err, scale)
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("planes", [3, 1])
def test_rows_gemm_epilogues_vs_fp64(planes):
    """demf_rows_gemm_f32 (csrc/rows_gemm.hip) alone against fp64: the positional addend on a column range, bias,
    padding-row zeroing, ReLU, residual + LayerNorm; R not a multiple of the 128-row tile, K = 256 and 1024.
    planes = 3: the error of an fp32 GEMM (<= 2e-6 of the output scale here); planes = 1: operands rounded to bf16
    once, checked against the same rounding in fp64."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    R = 1000
    rb = (lambda t: t.bfloat16().double()) if planes == 1 else (lambda t: t.double())
    tol = 2e-6 if planes == 3 else 2e-5          # (planes = 1: fp32 accumulation of exactly representable products)
    # mode 0: [x + pos | x] . W^T + b, rows zeroed from column 256 on
    x, pos, w, b = rnd(R, 256), rnd(R, 256), rnd(384, 256) / 16, rnd(384)
    mask = torch.rand(R, generator=g).cuda() < 0.2
    out = torch.empty(R, 384, device="cuda")
    ops.rows_gemm(x, ops.split_planes(w, planes), b, out, a2=pos, a2_cols=256, row_mask=mask, mask_col0=256)
    want = torch.cat([rb(x + pos) @ rb(w[:256]).t(), rb(x) @ rb(w[256:]).t()], 1) + b.double()
    want[:, 256:][mask] = 0
    err = (out.double() - want).abs().max().item()
    assert err <= tol * want.abs().max().item(), err
    # mode 1: ReLU, K = 1024 -> N = 128; a strided input (row pitch 1280)
    big = rnd(R, 1280)
    h, w1, b1 = big[:, :1024], rnd(128, 1024) / 32, rnd(128)
    out = torch.empty(R, 128, device="cuda")
    ops.rows_gemm(h, ops.split_planes(w1, planes), b1, out, relu=True)
    want = torch.relu(rb(h) @ rb(w1).t() + b1.double())
    err = (out.double() - want).abs().max().item()
    assert err <= tol * want.abs().max().item(), err
    # mode 2: LayerNorm(residual + x . W^T + b), K = 256 and 1024
    for K in (256, 1024):
        a, w2, b2, res = rnd(R, K), rnd(256, K) / K ** 0.5, rnd(256), rnd(R, 256)
        gam, bet = torch.rand(256, generator=g).cuda() + 0.5, rnd(256)
        out = torch.empty(R, 256, device="cuda")
        ops.rows_gemm(a, ops.split_planes(w2, planes), b2, out, ln=(res, gam, bet, 1e-5))
        s = rb(a) @ rb(w2).t() + b2.double() + res.double()
        want = torch.nn.functional.layer_norm(s, (256,), gam.double(), bet.double(), 1e-5)
        err = (out.double() - want).abs().max().item()
        assert err <= 5e-6 * max(1.0, want.abs().max().item()), (K, err)


def test_msda_raw_matches_the_composed_operator():
    """demf_msda_fwd_raw_f32 (softmax + sampling locations inside, value rows with a pitch) against
    MultiScaleDeformableAttnFunction on explicitly computed locations / weights - both its kernels (one wave per
    query, and 8 lanes per (query, head) via DEMF_MSDA_RAW_LANES in a fresh process is covered by the encoder test)."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, Dh, L = 2, 8, 32, 4
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    S = sum(h * w for h, w in shapes)
    shp = torch.tensor(shapes, dtype=torch.long, device="cuda")
    lsi = torch.tensor([0] + list(np.cumsum([h * w for h, w in shapes])[:-1]), dtype=torch.long, device="cuda")
    for P in (4, 2):
        n_off, n_lgt = H * L * P * 2, H * L * P
        v0 = (n_off + n_lgt + 127) // 128 * 128
        raw = torch.randn(B * S, v0 + H * Dh, generator=g).cuda()
        raw[:, :n_off] *= 3.0                                    # offsets of a few pixels, some off the image
        ref = torch.rand(B, S, L, 2, generator=g).cuda()
        out = torch.empty(B * S, H * Dh, device="cuda")
        ops.msda_fwd_raw(raw, v0, 0, n_off, ref, shp, lsi, B, S, H, Dh, P, out)
        off = raw[:, :n_off].view(B, S, H, L, P, 2)
        w = raw[:, n_off:n_off + n_lgt].view(B, S, H, L * P).softmax(-1).view(B, S, H, L, P)
        norm = torch.stack([shp[:, 1], shp[:, 0]], -1).float()
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        value = raw[:, v0:].reshape(B, S, H, Dh).contiguous()
        want = ops.MultiScaleDeformableAttnFunction.apply(value, shp, lsi, loc.contiguous(), w.contiguous(), 64)
        err = (out.view(B, S, -1) - want).abs().max().item()
        assert err <= 1e-5 * max(1.0, want.abs().max().item()), (P, err)


@pytest.mark.parametrize("shapes,first", [
    ([(40, 56), (20, 28), (10, 14), (5, 7)], 2),          # levels 2-3 (175 tokens) resident
    ([(120, 160), (60, 80), (30, 40), (15, 20)], 3),      # level 2 (1 200 tokens = 150 KB) does not fit: level 3 only
])
def test_msda_raw_head_form_matches_the_wave_form(shapes, first):
    """demf_msda_fwd_raw_head_f32 ((scene, head)-major, the coarse levels' value rows in LDS - what the encoder runs
    at >= 2 048 tokens) against demf_msda_fwd_raw_f32 on the same operands: equal up to the summation order."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, Dh, L = 2, 8, 32, 4
    sizes = [h * w for h, w in shapes]
    S = sum(sizes)
    shp = torch.tensor(shapes, dtype=torch.long, device="cuda")
    lsi = torch.tensor([0] + list(np.cumsum(sizes)[:-1]), dtype=torch.long, device="cuda")
    for P in (4, 2):
        n_off, n_lgt = H * L * P * 2, H * L * P
        v0 = (n_off + n_lgt + 127) // 128 * 128
        raw = torch.randn(B * S, v0 + H * Dh, generator=g).cuda()
        raw[:, :n_off] *= 3.0                                    # offsets of a few pixels, some off the image
        ref = torch.rand(B, S, L, 2, generator=g).cuda()
        want = torch.empty(B * S, H * Dh, device="cuda")
        got = torch.full((B * S, H * Dh), float("nan"), device="cuda")
        ops.msda_fwd_raw(raw, v0, 0, n_off, ref, shp, lsi, B, S, H, Dh, P, want)
        seen = []
        from demf_amd import _ffi
        orig = _ffi.call
        _ffi.call = lambda name, *a: (seen.append(a[14]) if name == "demf_msda_fwd_raw_head_f32" else None, orig(name, *a))[1]
        try:
            ops.msda_fwd_raw(raw, v0, 0, n_off, ref, shp, lsi, B, S, H, Dh, P, got, level_sizes=tuple(sizes))
        finally:
            _ffi.call = orig
        assert seen == [first], (seen, "the head form with this first resident level must be the one under test")
        err = (got - want).abs().max().item()
        assert err <= 1e-5 * max(1.0, want.abs().max().item()), (P, err)


def _enc256(seed, **over):
    from demf_amd.modules.image_stream import DeformableDetrEncoder
    kw = dict(fixtures.ENC256)
    kw.update(over)
    enc = DeformableDetrEncoder(**kw)
    fixtures.seed_weights(enc, seed)
    return enc.cuda().eval()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_fused_encoder_vs_real_reference_encoder_256(golden_dir, mode):
    """The KERNEL path (demf_rows_gemm_f32 + demf_msda_fwd_raw_f32 - the default at embed_dims = 256, i.e. what
    bench.py's image-stream leg and DeMFVoteNet run) against the REAL reference DeformableDetrEncoder at the
    reference's own encoder shape (256 dims, 8 heads, 4 levels, P = 4, FFN 1024, two layers;
    tests/golden/ref_encoder256.npz from oracle/pin_reference.py: encoder256_goldens): 1e-4 of the output scale on
    image positions in the fp32-grade mode, the bf16 bound of the decoder's GEMM tests in bf16 mode."""
    from demf_amd import ops
    gold = np.load(os.path.join(golden_dir, "ref_encoder256.npz"))
    enc = _enc256(6)
    assert sorted(enc.state_dict()) == list(gold["state_keys"])
    feats, metas = fixtures.make_encoder_pyramid(6)
    x = [torch.from_numpy(f).cuda() for f in feats]
    ops.set_compute_dtype(mode)
    try:
        assert enc._fused_ok(torch.empty(2, 10, 256, device="cuda")), "the kernel path must be the one under test"
        calls = _count_calls(("demf_rows_gemm_f32", "demf_msda_fwd_raw_f32"))
        with calls:
            outs = enc(x, metas)
        assert calls.n["demf_msda_fwd_raw_f32"] == 2 and calls.n["demf_rows_gemm_f32"] >= 8, calls.n
    finally:
        ops.set_compute_dtype("f32")
    for i, o in enumerate(outs):
        _assert_close_on_image(o.cpu(), torch.from_numpy(gold[f"enc{i}"]), metas, 1e-4 if mode == "f32" else 3e-2)


class _count_calls:
    """Counts _ffi.call launches by entry-point name while active (proof of which path ran)."""

    def __init__(self, names):
        self.n = {k: 0 for k in names}

    def __enter__(self):
        from demf_amd import _ffi
        self._ffi, self._orig = _ffi, _ffi.call

        def call(name, *a, **k):
            if name in self.n:
                self.n[name] += 1
            return self._orig(name, *a, **k)
        _ffi.call = call
        return self

    def __exit__(self, *exc):
        self._ffi.call = self._orig


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_fused_encoder_full_size_vs_oracle(mode):
    """forward_tokens on the kernel path at the PRODUCTION shape - B = 2, 100x140 ... 13x18 (S = 18 609; 37 218 rows =
    290 full 128-row tiles + a 98-row one), six layers, padded scene - against the CPU oracle's encoder
    (oracle/model.py: OracleEncoder, pinned bit-exact to the REAL class by tests/test_oracle_model.py) on the same
    neck pyramid and weights: 1e-4 of the output scale on image positions (f32 mode), the bf16 bound (bf16 mode)."""
    from demf_amd import ops
    from demf_amd.config import BATCH_INPUT_SHAPE, PYRAMID_SHAPES
    kw = dict(fixtures.ENC256, num_layers=6)
    ref = fixtures.oracle_encoder(**kw)
    fixtures.seed_weights(ref, 8)
    ref.eval()
    feats, metas = fixtures.make_encoder_pyramid(8, B=2, shapes=PYRAMID_SHAPES, input_hw=BATCH_INPUT_SHAPE)
    with torch.no_grad():
        want = ref([torch.from_numpy(f) for f in feats], metas)
    enc = _enc256(8, num_layers=6)
    x = [torch.from_numpy(f).cuda() for f in feats]
    ops.set_compute_dtype(mode)
    try:
        assert enc._fused_ok(torch.empty(2, 18609, 256, device="cuda"))
        tok = enc.forward_tokens(x, metas)
    finally:
        ops.set_compute_dtype("f32")
    assert tok["spatial"] == [tuple(s) for s in PYRAMID_SHAPES]
    got, start = tok["tokens"].permute(0, 2, 1), 0
    for w, (h, wd) in zip(want, PYRAMID_SHAPES):
        g = got[:, :, start:start + h * wd].reshape(2, 256, h, wd)
        start += h * wd
        _assert_close_on_image(g.cpu(), w, metas, 1e-4 if mode == "f32" else 3e-2)
    assert torch.isfinite(tok["tokens"]).all()
