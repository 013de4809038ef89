"""Round-4 additions to the device path, each against the code it replaces or the torch statement of
the reference behaviour:
  * demf_pad_gt (per-scene GT lists -> padded (B,G,7)/(B,G), class_agnostic_vote_head.py:766-773)
    against the host table path of DeMFVoteHead.pad_gt, for count signatures that never repeat;
  * the ``asum`` fallback of demf_gemm_f32 (bias gradient of a linear layer whose weight-gradient
    launch does not fit the small-tile path: a wider FFN than the reference's 1024, a lowered A/B knob).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gt_lists(counts, seed, box_dim=7):
    g = torch.Generator().manual_seed(seed)
    boxes = [torch.randn(n, box_dim, generator=g) for n in counts]
    labels = [torch.randint(0, 10, (n,), generator=g) for n in counts]
    return boxes, labels


@pytest.mark.parametrize("counts", [(3, 0, 5, 1), (0, 0), (8,) * 8, (1, 2, 3, 4, 5, 6, 7, 0, 9, 2, 4, 6, 1, 1, 0, 3)])
@pytest.mark.parametrize("G", [None, 12])
def test_pad_gt_device_lists_match_the_host_tables(counts, G):
    from demf_amd.modules.head import DeMFVoteHead
    boxes, labels = _gt_lists(counts, seed=sum(counts) + len(counts))
    for slot in (False, True):
        want = DeMFVoteHead.pad_gt(boxes, labels, torch.device("cpu"), with_slot_labels=slot, G=G)
        got = DeMFVoteHead.pad_gt([b.cuda() for b in boxes], [l.cuda() for l in labels], torch.device("cuda"),
                                  with_slot_labels=slot, G=G)
        for w, g_, name in zip(want, got, ("gt", "labels", "valid")):
            assert g_.is_cuda and g_.shape == w.shape and g_.dtype == w.dtype, name
            assert torch.equal(g_.cpu(), w), (name, slot)


def test_pad_gt_issues_no_host_to_device_copy():
    """The per-batch path must not upload anything: count signatures that were never seen before go
    through the same single launch (the table cache only hid the copies for repeats)."""
    from demf_amd.modules.head import DeMFVoteHead
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda")
    lists = []
    for s in range(6):
        counts = tuple(int(c) for c in np.random.default_rng(s).integers(0, 9, size=8))
        b, l = _gt_lists(counts, seed=100 + s)
        lists.append(([x.cuda() for x in b], [x.cuda() for x in l]))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for b, l in lists:
            DeMFVoteHead.pad_gt(b, l, dev, with_slot_labels=True, G=8)
        torch.cuda.synchronize()
    names = [e.name.lower() for e in prof.events()]
    assert not any("memcpy" in n and "htod" in n for n in names), [n for n in names if "memcpy" in n]


def test_pad_gt_rejects_overfull_scene():
    from demf_amd.modules.head import DeMFVoteHead
    boxes, labels = _gt_lists((3, 9), 0)
    with pytest.raises(ValueError, match="ground-truth boxes"):
        DeMFVoteHead.pad_gt([b.cuda() for b in boxes], [l.cuda() for l in labels], torch.device("cuda"), G=8)


@pytest.mark.parametrize("F", [1024, 2048])
def test_linear_bias_gradient_beyond_the_small_tile_path(F, monkeypatch):
    """ops.linear backward with a (F x 256) weight on 4 096 rows: F = 2048 is 2 048 tiles - past the
    small-tile limit the in-kernel row sums need; the library then runs the product without them and
    takes the bias gradient with demf_colsum_f32 instead of failing."""
    from demf_amd import ops
    g = torch.Generator().manual_seed(F)
    x = torch.randn(4096, 256, generator=g).cuda().requires_grad_()     # split-K 16: 1 024 / 2 048 tiles
    w = (torch.randn(F, 256, generator=g) * 0.05).cuda().requires_grad_()
    b = torch.randn(F, generator=g).cuda().requires_grad_()
    dy = torch.randn(4096, F, generator=g).cuda()
    y = ops.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
    x64, w64, b64 = (t.detach().double() for t in (x, w, b))
    torch.testing.assert_close(y.double(), x64 @ w64.t() + b64, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gb.double(), dy.double().sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(gw.double(), dy.double().t() @ x64, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(gx.double(), dy.double() @ w64, rtol=1e-4, atol=1e-3)


def test_gemm_asum_fallback_through_the_c_abi():
    """demf_gemm_f32 with ``asum`` on a launch shape that is NOT the small-tile reduction-strided form:
    (a) more tiles than the limit, (b) the grouped entry point's single-launch fallback."""
    from demf_amd import fused
    g = torch.Generator().manual_seed(7)
    R, N, K = 4096, 2304, 64          # dW (N,K) = dy^T x : 36 x 1 tiles of 64 x 64, x split-K 16 x ... > 1024
    dy = torch.randn(R, N, generator=g).cuda()
    x = torch.randn(R, K, generator=g).cuda()
    want_w = dy.double().t() @ x.double()
    want_b = dy.double().sum(0)
    for grouped in (False, True):
        dw = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        grp = [] if grouped else None
        fused.gemm(N, K, R, dy.data_ptr(), (1, N), x.data_ptr(), (1, K), dw.data_ptr(), K,
                   splitk=64, asum=db.data_ptr(), group=grp)
        if grouped:
            fused.gemm_group(grp)
        torch.testing.assert_close(dw.double(), want_w, rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(db.double(), want_b, rtol=1e-4, atol=2e-3)
