"""oracle/emulate.py (the bf16-emulating mode of the CPU oracle, test infrastructure for BASELINE
configs[3]) on the CPU: with the rounding function replaced by the identity the emulating forward /
backward - the product's factored first SA layer, the kernels' attention operand order, sample-then-
project cross-attention - must reproduce the plain oracle exactly (forward and every gradient); with real
rounding the primitive gradient rules are the kernels' (dx = q(g) q(W), dW = q(g)^T q(x))."""
import numpy as np
import torch

import parity_tools as P
from oracle import emulate, fixtures

PYR, INP = ((32, 44), (16, 22), (8, 11), (4, 6)), (256, 352)      # more tokens than sampled corners


def _case(seed=1):
    cfg = fixtures.tiny_cfg()
    batch = fixtures.make_scene_batch(2, 1024, PYR, INP, cfg.head.embed_dims, seed=seed, n_gt=4)
    return cfg, batch


def test_emulation_structure_equals_plain_oracle_when_rounding_is_identity(monkeypatch):
    cfg, batch = _case()
    T = P.oracle_run(cfg, batch, batch["gt_boxes"], batch["gt_labels"], 1, torch.float64, tap=False)
    monkeypatch.setattr(emulate, "q", lambda x: x)
    E = P.oracle_run(cfg, batch, batch["gt_boxes"], batch["gt_labels"], 1, torch.float64, tap=False,
                     emulate_bf16=True)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
    for k in ("vote_points", "vote_features", "aggregated_points"):
        assert rel(E["preds"][k], T["preds"][k]) < 1e-10, k
    for i, d in enumerate(T["preds"]["decode_res_all"]):
        for k in d:
            assert rel(E["preds"]["decode_res_all"][i][k], d[k]) < 1e-9, (i, k)
    for k in T["losses"]:
        assert abs(E["losses"][k].item() - T["losses"][k].item()) <= 1e-9 * max(1.0, abs(T["losses"][k].item()))
    assert sorted(E["grads"]) == sorted(T["grads"])
    gmax = max(v.norm().item() for v in T["grads"].values())
    for n, g in T["grads"].items():
        assert (E["grads"][n] - g).norm().item() <= 1e-8 * gmax, n


def test_emulation_rounds_and_restores_the_patched_functions():
    import torch.nn.functional as F
    from oracle import deps
    before = (F.linear, F.conv1d, F.conv2d, deps.PointSAModule.forward, deps.MultiheadAttention.forward,
              deps.MultiScaleDeformableAttention.forward)
    cfg, batch = _case(2)
    T = P.oracle_run(cfg, batch, batch["gt_boxes"], batch["gt_labels"], 2, torch.float64, tap=False)
    E = P.oracle_run(cfg, batch, batch["gt_boxes"], batch["gt_labels"], 2, torch.float64, tap=False,
                     emulate_bf16=True)
    after = (F.linear, F.conv1d, F.conv2d, deps.PointSAModule.forward, deps.MultiheadAttention.forward,
             deps.MultiScaleDeformableAttention.forward)
    assert all(a is b for a, b in zip(before, after))
    # rounding is really applied (bf16 eps = 4e-3 per operand), and only to values - never to indices
    d = ((E["preds"]["vote_features"] - T["preds"]["vote_features"]).norm() / T["preds"]["vote_features"].norm()).item()
    assert 1e-4 < d < 0.5, d
    assert torch.equal(E["preds"]["seed_indices"], T["preds"]["seed_indices"])
    assert torch.equal(E["preds"]["aggregated_indices"], T["preds"]["aggregated_indices"])
    assert all(torch.isfinite(g).all() for g in E["grads"].values())


def test_primitive_gradient_rules():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 20, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(11, 20, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(37, 11, generator=g, dtype=torch.float64)
    q = emulate.q
    y = emulate.mm(x, w)
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    assert torch.equal(y, q(x) @ q(w).t())
    assert torch.equal(gx, q(dy) @ q(w)) and torch.equal(gw, q(dy).t() @ q(x))
    gw2, = torch.autograd.grad(emulate.mm(x, w, round_dw=False), [w], dy)
    assert torch.equal(gw2, dy.t() @ x)                     # SA1 layer 0: weight gradient from fp32 sums
    a = torch.randn(3, 5, 7, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, 7, 4, generator=g, dtype=torch.float64, requires_grad=True)
    do = torch.randn(3, 5, 4, generator=g, dtype=torch.float64)
    o = emulate._BMM.apply(a, b, 0.25)
    ga, gb = torch.autograd.grad(o, [a, b], do)
    assert torch.equal(o, (q(a) @ q(b)) * 0.25)
    assert torch.equal(ga, (q(do) @ q(b).transpose(1, 2)) * 0.25)
    assert torch.equal(gb, (q(a).transpose(1, 2) @ q(do)) * 0.25)
    # q is round-to-nearest-even onto 8 significand bits
    v = torch.tensor([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9, 1.0 + 2.0 ** -7], dtype=torch.float64)
    assert q(v).tolist() == [1.0, 1.0, 1.0 + 2.0 ** -7, 1.0 + 2.0 ** -7]


def test_pooled_last_layer_emulation_equals_autograd_when_rounding_is_identity(monkeypatch):
    """``_PooledLast`` (the no-store form of SA1's last layer: backward written around the layer's input) is
    algebraically the layer's backward: with identity rounding it must equal torch autograd of
    conv -> BN(train) -> ReLU -> max over 64 rows, incl. a negative and a zero BN scale."""
    import torch.nn.functional as F
    monkeypatch.setattr(emulate, "q", lambda x: x)
    g = torch.Generator().manual_seed(3)
    G, ns, K, N = 40, 64, 64, 128
    A = torch.relu(torch.randn(G * ns, K, generator=g, dtype=torch.float64)).requires_grad_()
    W = (torch.randn(N, K, generator=g, dtype=torch.float64) / 8).requires_grad_()
    gamma = (1.0 + 0.2 * torch.randn(N, generator=g, dtype=torch.float64))
    gamma[5], gamma[9] = -0.8, 0.0
    gamma.requires_grad_()
    beta = (0.1 * torch.randn(N, generator=g, dtype=torch.float64)).requires_grad_()
    dP = torch.randn(G, N, generator=g, dtype=torch.float64)
    out = emulate._PooledLast.apply(A, W, gamma, beta, ns, 1e-5)
    got = torch.autograd.grad(out, [A, W, gamma, beta], dP)
    A2, W2, g2, b2 = (t.detach().clone().requires_grad_() for t in (A, W, gamma, beta))
    ref = F.relu(F.batch_norm(A2 @ W2.t(), None, None, g2, b2, True, 0.1, 1e-5)).view(G, ns, N).max(1)[0]
    want = torch.autograd.grad(ref, [A2, W2, g2, b2], dP)
    assert torch.allclose(out, ref, rtol=0, atol=1e-12)
    for a, b, n in zip(got, want, ("dA", "dW", "dgamma", "dbeta")):
        assert ((a - b).norm() / b.norm()).item() < 1e-10, n
