"""Test-time decode + NMS on the GPU against golden vectors of the REAL
reference DeMFVoteHead.get_bboxes (tests/golden/ref_bboxes.npz, oracle/pin_reference.py) and against
the oracle restatement on larger random cases."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures
from oracle.model import OracleDeMF

pytestmark = pytest.mark.gpu


def _head():
    from demf_amd.modules import DeMFHotPath
    return DeMFHotPath(fixtures.tiny_cfg()).pts_bbox_head.cuda().eval()


def _run(head, pts, dec, **kw):
    preds = dict(decode_res_all=[{k: torch.from_numpy(v).cuda() for k, v in d.items()} for d in dec])
    return head.get_bboxes(torch.from_numpy(pts).cuda(), preds, [dict() for _ in range(len(pts))], **kw)


def _sorted(boxes, scores, labels):
    """Canonical order: the product emits survivors in proposal order per class like the
    reference, but ties in device sorting are not part of the contract."""
    o = np.lexsort((boxes[:, 0], scores, labels))
    return boxes[o], scores[o], labels[o]


@pytest.mark.parametrize("seed", [0, 1])
def test_get_bboxes_vs_real_reference(seed, golden_dir):
    gold = np.load(os.path.join(golden_dir, "ref_bboxes.npz"))
    pts, dec = fixtures.make_decode_results(seed)
    head = _head()
    raw = _run(head, pts, dec, use_nms=False)
    np.testing.assert_allclose(raw.cpu().numpy(), gold[f"s{seed}.bbox3d"], rtol=1e-6, atol=1e-6)
    res = _run(head, pts, dec)
    for b, (bx, sc, lb) in enumerate(res):
        got = _sorted(bx.tensor.cpu().numpy(), sc.cpu().numpy(), lb.cpu().numpy())
        want = _sorted(gold[f"s{seed}.b{b}.boxes"], gold[f"s{seed}.b{b}.scores"],
                       gold[f"s{seed}.b{b}.labels"])
        assert got[0].shape == want[0].shape, "different survivor set"
        np.testing.assert_array_equal(got[2], want[2])
        np.testing.assert_allclose(got[1], want[1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("seed,B,K,N", [(5, 3, 256, 20000), (6, 8, 256, 20000), (7, 1, 512, 3000)])
def test_get_bboxes_vs_oracle(seed, B, K, N):
    """Full-size ensembles (2 x 256 proposals, 20000 points): same survivors as the oracle."""
    pts, dec = fixtures.make_decode_results(seed, B=B, K=K, N=N)
    oracle = OracleDeMF(fixtures.tiny_cfg()).pts_bbox_head
    want = oracle.get_bboxes(torch.from_numpy(pts), [{k: torch.from_numpy(v) for k, v in d.items()}
                                                     for d in dec])
    got = _run(_head(), pts, dec)
    for b in range(B):
        g = _sorted(got[b][0].tensor.cpu().numpy(), got[b][1].cpu().numpy(), got[b][2].cpu().numpy())
        w = _sorted(want[b][0].numpy(), want[b][1].numpy(), want[b][2].numpy())
        assert g[0].shape == w[0].shape, f"scene {b}: {g[0].shape} vs {w[0].shape}"
        np.testing.assert_array_equal(g[2], w[2])
        np.testing.assert_allclose(g[1], w[1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g[0], w[0], rtol=1e-5, atol=1e-5)


def test_nms_edge_cases():
    """No valid box, a single box, identical boxes of one class / of different classes."""
    from demf_amd import ops
    ext = torch.tensor([[[0, 0, 0, 1, 1, 1.0]] * 4], device="cuda")
    sc = torch.tensor([[0.9, 0.8, 0.7, 0.6]], device="cuda")
    same = torch.zeros((1, 4), dtype=torch.int64, device="cuda")
    diff = torch.arange(4, device="cuda")[None]
    none = torch.zeros((1, 4), dtype=torch.bool, device="cuda")
    allv = ~none
    assert ops.aligned_nms(ext, sc, same, none, 0.25).sum() == 0
    assert ops.aligned_nms(ext, sc, same, allv, 0.25)[0].tolist() == [True, False, False, False]
    assert ops.aligned_nms(ext, sc, diff, allv, 0.25)[0].tolist() == [True] * 4
    only3 = torch.tensor([[False, False, True, False]], device="cuda")
    assert ops.aligned_nms(ext, sc, same, only3, 0.25)[0].tolist() == [False, False, True, False]
