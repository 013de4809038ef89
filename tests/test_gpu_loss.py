"""Fused head-loss / vote-loss kernels (csrc/loss.hip) against the PyTorch composition of the
same losses (DeMFVoteHead._loss restating class_agnostic_vote_head.py:622-712), forward and
gradients, including negative predicted sizes (signed IoU areas) and ties in the IoU min/max."""
import numpy as np
import pytest
import torch

from oracle import fixtures

pytestmark = pytest.mark.gpu


def _inputs(seed, B=3, Q=64, S=128, N=500):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    cls = r(B, Q, 12).cuda().requires_grad_()
    reg = (r(B, Q, 30) * 0.7).cuda().requires_grad_()
    base = (r(B, Q, 3) * 2).cuda().requires_grad_()
    center_t = (base.detach() + 0.3 * r(B, Q, 3).cuda())
    size_t = (torch.rand(B, Q, 3, generator=g) + 0.4).cuda()
    obj_t = (torch.rand(B, Q, generator=g) < 0.3).long().cuda()
    obj_mask = (torch.rand(B, Q, generator=g) < 0.8).float().cuda()
    targets = (
        (r(B, N, 9)).cuda(), (torch.rand(B, N, generator=g) < 0.4).long().cuda(),   # vote targets, masks
        torch.randint(0, 12, (B, Q), generator=g).cuda(), (r(B, Q) * 0.3).cuda(),   # dir class / res
        torch.randint(0, 10, (B, Q), generator=g).cuda(), obj_t,
        obj_mask / (obj_mask.sum() + 1e-6), obj_t.float() / (obj_t.sum().float() + 1e-6),
        None, None, size_t, center_t)
    seed_points = r(B, S, 3).cuda()
    vote_points = (seed_points + 0.2 * r(B, S, 3).cuda()).requires_grad_()
    seed_idx = torch.randint(0, N, (B, S), generator=g).cuda()
    return cls, reg, base, targets, seed_points, vote_points, seed_idx


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fused_losses_match_torch_composition(seed):
    from demf_amd.modules import DeMFHotPath
    head = DeMFHotPath(fixtures.tiny_cfg()).pts_bbox_head.cuda()
    cls, reg, base, targets, seed_points, vote_points, seed_idx = _inputs(seed)
    if seed == 2:   # exact ties in the IoU bounds + a flat box
        with torch.no_grad():
            reg[0, 0, 0:3] = targets[11][0, 0] - base[0, 0]
            reg[0, 0, 3:6] = targets[10][0, 0]
            reg[0, 1, 3] = -0.4
            targets[5][0, :2] = 1
            targets = targets[:7] + (targets[5].float() / (targets[5].sum().float() + 1e-6),) + targets[8:]
    preds = dict(seed_points=seed_points, vote_points=vote_points, seed_indices=seed_idx,
                 **head.bbox_coder.split_pred(cls.transpose(1, 2), reg.transpose(1, 2), base))
    ref = head._loss(preds, targets)                       # torch composition (no '_rows')
    total_r = sum(v * (i + 1) for i, v in enumerate(ref.values()))   # distinct upstream grads
    gr = torch.autograd.grad(total_r, [cls, reg, base, vote_points])
    fused = head._loss({**preds, "_rows": (cls, reg, base)}, targets)
    assert list(fused) == list(ref)
    for k in ref:
        torch.testing.assert_close(fused[k], ref[k], rtol=2e-4, atol=1e-5, msg=k)
    total_f = sum(v * (i + 1) for i, v in enumerate(fused.values()))
    gf = torch.autograd.grad(total_f, [cls, reg, base, vote_points])
    for a, b, n in zip(gf, gr, ("cls", "reg", "base", "vote_points")):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-5 * max(1.0, b.abs().max().item()), msg=n)


@pytest.mark.parametrize("B,N,G,seed", [(2, 5000, 6, 0), (3, 20000, 17, 1), (1, 257, 1, 2), (2, 1000, 64, 3)])
def test_vote_targets_kernel_bit_exact(B, N, G, seed):
    """demf_vote_targets against its torch specification (the batched restatement of
    class_agnostic_vote_head.py:828-858 in DeMFVoteHead.vote_targets, run on the CPU): bit-exact,
    including points inside 0, 1, 2 and >= 3 boxes, padded (invalid) GT slots and an empty scene."""
    from demf_amd import ops
    from demf_amd.modules.head import DeMFVoteHead
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-3, -3, 0, 0], [3, 3, 3, 3], size=(B, N, 4)).astype(np.float32)
    gt = np.zeros((B, G, 7), np.float32)
    gt[..., :3] = rng.uniform([-2.5, -2.5, 0], [2.5, 2.5, 1.5], size=(B, G, 3))
    gt[..., 3:6] = rng.uniform(0.5, 3.0, size=(B, G, 3))         # big boxes: plenty of overlaps
    gt[..., 6] = rng.uniform(-np.pi, np.pi, size=(B, G))
    valid = np.ones((B, G), bool)
    if G > 2:
        valid[0, G // 2:] = False
        gt[0, G // 2:] = 0.0
    if B > 1:                                                        # the reference's fake box
        valid[1] = False
        valid[1, 0] = True
        gt[1] = 0.0
    head = DeMFVoteHead.__new__(DeMFVoteHead)                        # vote_targets uses no state
    # list-of-boxes call path, so that padded (invalid) slots exist
    boxes = [torch.from_numpy(gt[b][valid[b]]) if not (B > 1 and b == 1) else torch.zeros((0, 7))
             for b in range(B)]
    labels = [torch.zeros(len(bx), dtype=torch.long) for bx in boxes]
    spec = DeMFVoteHead.vote_targets(head, torch.from_numpy(pts), boxes, labels)
    g_pad, _, v_pad = DeMFVoteHead.pad_gt(boxes, labels, torch.device("cpu"))
    vt, mask = ops.vote_targets(torch.from_numpy(pts).cuda(), g_pad.cuda(), v_pad.cuda())
    np.testing.assert_array_equal(mask.cpu().numpy(), spec["vote_target_masks"].numpy())
    np.testing.assert_array_equal(vt.cpu().numpy(), spec["vote_targets"].numpy())
    inside = spec["vote_target_masks"].numpy()
    assert 0 < inside.sum() < inside.size


@pytest.mark.parametrize("B,Q,G,seed", [(2, 256, 5, 0), (8, 256, 12, 1), (3, 33, 1, 2)])
def test_proposal_targets_kernel_matches_spec(B, Q, G, seed):
    """demf_proposal_targets against the torch specification in DeMFVoteHead.get_targets (the
    batched restatement of class_agnostic_vote_head.py:877-934) evaluated on the CPU: integer
    targets and the assignment exact, gathered float targets bit-exact, rotated ones to 1e-6."""
    from oracle import fixtures
    from demf_amd.modules import DeMFHotPath
    rng = np.random.default_rng(seed)
    head = DeMFHotPath(fixtures.tiny_cfg()).pts_bbox_head
    N = 512
    pts = rng.uniform([-3, -3, 0, 0], [3, 3, 3, 3], size=(B, N, 4)).astype(np.float32)
    boxes, labels = [], []
    for b in range(B):
        n = G if b != 1 else max(G - 2, 0)                  # one scene with fewer (or zero) boxes
        g = np.zeros((n, 7), np.float32)
        g[:, :3] = rng.uniform([-2.5, -2.5, 0], [2.5, 2.5, 1.5], size=(n, 3))
        g[:, 3:6] = rng.uniform(0.4, 2.0, size=(n, 3))
        g[:, 6] = rng.uniform(-np.pi, np.pi, size=n)
        boxes.append(torch.from_numpy(g))
        labels.append(torch.from_numpy(rng.integers(0, 10, size=n)))
    agg = rng.uniform([-3, -3, 0], [3, 3, 3], size=(B, Q, 3)).astype(np.float32)
    for b in range(B):                                        # some proposals right at box centres
        if len(boxes[b]):
            c = boxes[b][0].numpy()
            agg[b, :8] = [c[0], c[1], c[2] + 0.5 * c[5]] + rng.normal(0, 0.05, size=(8, 3))
    names = ("vote_targets", "vote_target_masks", "dir_class_targets", "dir_res_targets",
             "mask_targets", "objectness_targets", "objectness_weights", "box_loss_weights",
             "distance_targets", "dir_targets", "size_targets", "center_targets")
    spec = head.get_targets(torch.from_numpy(pts), boxes, labels,
                            dict(aggregated_points=torch.from_numpy(agg)))
    got = head.get_targets(torch.from_numpy(pts).cuda(), [b.cuda() for b in boxes],
                           [l.cuda() for l in labels], dict(aggregated_points=torch.from_numpy(agg).cuda()))
    assert int(spec[5].sum()) > 0                             # positives exist
    for n, a, b in zip(names, spec, got):
        a, b = a.numpy(), b.cpu().numpy()
        if a.dtype.kind in "iu":
            np.testing.assert_array_equal(a, b, err_msg=n)
        elif n in ("objectness_weights", "box_loss_weights"):
            np.testing.assert_allclose(a, b, rtol=1e-6, err_msg=n)    # normalised by a float sum
        elif n == "distance_targets":
            # rotated by cos/sin(-yaw): device and host libm differ in the last ulp
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6, err_msg=n)
        else:
            np.testing.assert_array_equal(a, b, err_msg=n)


def test_gt_prep_matches_the_torch_specification():
    """demf_gt_prep against the host code it replaces in the captured step: cos / sin(-yaw),
    bbox_coder.angle2class(yaw) (class exact, residual bit-exact: same fp32 remainder /
    floor-divide arithmetic), valid mask, clamped labels, gravity centres."""
    from demf_amd import ops
    from demf_amd.modules.coder import DeMFClassAgnosticBBoxCoder
    g = torch.Generator().manual_seed(5)
    B, G, nb = 8, 11, 12
    gt = torch.randn(B, G, 7, generator=g)
    gt[..., 3:6] = gt[..., 3:6].abs() + 0.1
    gt[..., 6] = (torch.rand(B, G, generator=g) - 0.5) * 4 * np.pi       # beyond one period, both signs
    per = 2 * np.pi / nb
    gt[0, :6, 6] = torch.tensor([0.0, per / 2, -per / 2, np.pi, -np.pi, 2 * np.pi - 1e-7])  # bin edges
    lab = torch.randint(0, 10, (B, G), generator=g)
    lab[:, 7:] = -1
    lab[3] = -1
    gt, lab = gt.cuda(), lab.cuda()
    p = ops.gt_prep(gt, lab, nb)
    yaw = gt[..., 6]
    cls, res = DeMFClassAgnosticBBoxCoder(nb).angle2class(yaw)
    assert torch.equal(p["dir_class"], cls)
    assert torch.equal(p["dir_res"], res)
    assert torch.equal(p["cs"], torch.cos(-yaw)) and torch.equal(p["sn"], torch.sin(-yaw))
    assert torch.equal(p["valid"], lab >= 0) and p["valid"].dtype == torch.bool
    assert torch.equal(p["lab"], lab.clamp(min=0))
    center = torch.cat([gt[..., :2], gt[..., 2:3] + gt[..., 5:6] * 0.5], dim=-1)
    assert torch.equal(p["center"], center)


def test_target_weights_match_the_torch_specification():
    from demf_amd import ops
    g = torch.Generator().manual_seed(6)
    for R in (2048, 1000, 7):
        m = (torch.rand(R, generator=g) < 0.6).float().cuda()
        o = (torch.rand(R, generator=g) < 0.1).long().cuda()
        ow, bw = ops.target_weights(m, o)
        np.testing.assert_allclose(ow.cpu().numpy(), (m / (m.sum() + 1e-6)).cpu().numpy(), rtol=1e-6)
        np.testing.assert_allclose(bw.cpu().numpy(), (o.float() / (o.sum().float() + 1e-6)).cpu().numpy(), rtol=1e-6)
    z = torch.zeros(64, device="cuda")
    ow, bw = ops.target_weights(z, z.long())
    assert float(ow.abs().max()) == 0.0 and float(bw.abs().max()) == 0.0
