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
