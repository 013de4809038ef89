"""Whole-path parity tooling: oracle runs with activation taps, seed qualification by DISCRETE-EVENT
margins, and per-tensor gradient comparison.

Why seeds are screened.  The hot path is piecewise smooth: ReLUs, max-pools over neighbours, the
radius test of the vote-aggregation ball query and the positive / negative thresholds of the
target assignment are discontinuities (of the gradient or of the value).  Two correct fp32
implementations that differ only in summation order can land on different sides of one whenever a
pre-activation lies within their round-off - and when the element carries a large share of a
layer's gradient (box losses are carried by the ~10 positive proposals) one such flip moves whole
gradient tensors by per cent.  That is not an error of either implementation, so inputs on which it
CAN happen are not parity inputs.  A seed qualifies iff, in the fp64 oracle,

  * no neighbour lies within 1e-5 (relative, squared distance) of the vote-aggregation ball radius;
  * no proposal lies within 1e-4 m of the 0.3 / 0.6 m assignment thresholds, of a face of its
    assigned box, or has two GT centres within 1e-4 m of being equally near;
  * for every tapped ReLU / max-pool, the elements whose pre-activation (top-2 gap) lies within
    TAU_MULT x the layer's MEASURED fp32 noise (rms of z_cpu32 - z_fp64) carry at most RISK_MAX of
    the layer's gradient energy: risk = sqrt(sum_{near} g^2 / sum g^2) <= RISK_MAX.  (Measured on
    the tiny config: seeds with risk < 1e-2 show cpu32-vs-fp64 gradient errors <= 2e-4 on every
    tensor; a seed with risk 0.18 shows 13 % on one tensor - an actual flip.)  The residual risk
    of the chosen seed is added to the gradient tolerance as an allowance.

All of it is computed from the two oracle runs alone - the HIP path has no say in which seed is
used - and the first qualifying seed is THE test input: no retry on failure.
"""
import numpy as np
import torch
import torch.nn as nn

from oracle import deps, fixtures
from oracle.model import OracleDeMF

RISK_MAX = 1e-1
NUM_EXTRA_BOXES = 3     # GT boxes dropped on proposals per scene (so that positives exist)
TAU_MULT = 4.0          # near-tie window = TAU_MULT x the layer's measured fp32 rms noise
TAP_NUMEL_MAX = 40_000_000      # larger tensors (SA1/SA2 at B=8) are not tapped: >= 1M rows share the
#                                 gradient, a single element carries ~1e-6 of the layer's energy


class Taps:
    """Forward hooks on every ReLU and neighbour max-pool of an OracleDeMF.
    mode 'truth': keeps z (pre-activation) and the post-activation tensor with retain_grad;
    mode 'noise': compares z with the truth taps on the fly and keeps max |dz| per layer."""

    def __init__(self, model, truth=None):
        self.truth = truth
        self.z, self.out, self.pool, self.noise = {}, {}, {}, {}
        self.handles = []
        for name, m in model.named_modules():
            if isinstance(m, deps.ConvModule):
                self.handles.append(m.bn.register_forward_hook(self._z_hook(name)))
                self.handles.append(m.register_forward_hook(self._out_hook(name)))
            elif isinstance(m, nn.ReLU):
                self.handles.append(m.register_forward_pre_hook(self._pre_hook(name)))
                self.handles.append(m.register_forward_hook(self._out_hook(name)))
            elif isinstance(m, deps.PointSAModule):
                self.handles.append(m.mlps[0].register_forward_hook(self._prepool_hook(name)))
                self.handles.append(m.register_forward_hook(self._pooled_hook(name)))

    def close(self):
        for h in self.handles:
            h.remove()

    def _keep(self, name, z):
        if z.numel() > TAP_NUMEL_MAX:
            return
        if self.truth is None:
            self.z[name] = z.detach().clone()
        elif name in self.truth.z:
            self.noise[name] = (z.detach().double() - self.truth.z[name]).pow(2).mean().sqrt().item()

    def _z_hook(self, name):
        return lambda mod, inp, out: self._keep(name, out)

    def _pre_hook(self, name):
        return lambda mod, inp: self._keep(name, inp[0])

    def _out_hook(self, name):
        def hook(mod, inp, out):
            if self.truth is None and out.requires_grad and out.numel() <= TAP_NUMEL_MAX:
                out.retain_grad()
                self.out[name] = out
        return hook

    def _prepool_hook(self, name):
        def hook(mod, inp, out):                       # (B,C,M,ns), post-ReLU
            if out.numel() > TAP_NUMEL_MAX:
                return
            if self.truth is None:
                self.pool[name] = [out.detach().clone(), None]
            elif name in self.truth.pool:
                self.noise["pool:" + name] = (out.detach().double() - self.truth.pool[name][0]).pow(2).mean().sqrt().item()
        return hook

    def _pooled_hook(self, name):
        def hook(mod, inp, out):
            if self.truth is None and name in self.pool and out[1].requires_grad:
                out[1].retain_grad()
                self.pool[name][1] = out[1]
        return hook


def flip_risks(truth, noise):
    """-> {layer: (risk, tau, n_near)} from the fp64 taps (after backward) and the fp32 noise."""
    risks = {}
    for name, z in truth.z.items():
        out = truth.out.get(name)
        if out is None or out.grad is None or name not in noise:
            continue
        tau = TAU_MULT * noise[name]
        g2 = out.grad.double() ** 2
        tot = (g2 * (z > 0)).sum().item()
        near = z.abs() < tau
        risks[name] = (float(np.sqrt((g2 * near).sum().item() / max(tot, 1e-300))), tau, int(near.sum()))
    for name, (pre, pooled) in truth.pool.items():
        key = "pool:" + name
        if pooled is None or pooled.grad is None or key not in noise:
            continue
        tau = TAU_MULT * noise[key]
        top = pre.max(dim=-1, keepdim=True)[0]
        second = torch.where(pre < top, pre, torch.full_like(pre, -float("inf"))).max(dim=-1)[0]
        top = top.squeeze(-1)
        # exact ties are the ball query's duplicated neighbours (same source row, same arithmetic on
        # every path) or all-zero groups (no gradient through the ReLU either way): benign
        near = (top - second < tau) & (top > 0)
        g2 = pooled.grad.double() ** 2
        risks[key] = (float(np.sqrt((g2 * near).sum().item() / max(g2.sum().item(), 1e-300))), tau,
                      int(near.sum()))
    return risks


def target_margins(cfg, preds, targets, gtb):
    """Smallest distance (metres) of any proposal to a discontinuity of the target assignment."""
    agg = preds["aggregated_points"].detach().double()
    pos, neg = cfg.head.pos_distance_thr, cfg.head.neg_distance_thr
    m = float("inf")
    for b in range(agg.shape[0]):
        box = torch.as_tensor(gtb[b]).double()
        ctr = torch.cat([box[:, :2], box[:, 2:3] + box[:, 5:6] * 0.5], 1)
        d = torch.sqrt(((agg[b][:, None] - ctr[None]) ** 2).sum(-1) + 1e-6)
        s = torch.sort(d, dim=1)[0]
        m = min(m, (s[:, 0] - pos).abs().min().item(), (s[:, 0] - neg).abs().min().item())
        if s.shape[1] > 1:
            m = min(m, (s[:, 1] - s[:, 0]).min().item())
        close = s[:, 0] < pos
        if close.any():
            m = min(m, targets["distance_targets"][b][close].abs().min().item())
    return m


def ball_margin(xyz, center, r):
    d2 = ((xyz[:, None, :, :].astype(np.float64) - center[:, :, None, :].astype(np.float64)) ** 2).sum(-1)
    return np.abs(d2 - r * r).min() / (r * r)


def oracle_run(cfg, batch, gtb, gtl, seed, dtype, truth_taps=None, tap=True, emulate_bf16=False, state=None):
    """``emulate_bf16``: the oracle rounds where the product's bf16 compute mode rounds
    (oracle/emulate.py); float64 = the emulated truth, float32 = the accumulation noise around it.
    ``state``: a state dict to load instead of the seeded weights."""
    import contextlib
    from oracle import emulate
    ref = OracleDeMF(cfg)
    if state is None:
        fixtures.seed_weights(ref, seed)
    else:
        ref.load_state_dict(state)
    ref.train().to(dtype)
    taps = Taps(ref, truth_taps) if tap and not emulate_bf16 else None
    pts = torch.from_numpy(batch["points"]).to(dtype)
    feats = [torch.from_numpy(f).to(dtype) for f in batch["img_features"]]
    with (emulate.bf16_emulation() if emulate_bf16 else contextlib.nullcontext()):
        losses, preds, targets = ref.forward_train(pts, feats, batch["img_metas"],
                                                   [torch.from_numpy(b).to(dtype) for b in gtb],
                                                   [torch.from_numpy(l) for l in gtl])
        sum(losses.values()).backward()
    if taps is not None:
        taps.close()
    grads = {n: p.grad.detach().double() for n, p in ref.named_parameters() if p.grad is not None}
    return dict(preds=preds, losses=losses, targets=targets, grads=grads, taps=taps)


def make_case(cfg, B, N, pyramid, in_shape, img_shape, seed):
    """Seeded scene batch + GT boxes: in-room boxes + a few boxes dropped on proposals (so that
    positives exist).  -> (batch, gtb, gtl) or None when the vote-aggregation ball is too close."""
    batch = fixtures.make_scene_batch(B, N, pyramid, in_shape, cfg.head.embed_dims, seed=seed,
                                      n_gt=5, img_shape=img_shape)
    probe = OracleDeMF(cfg)
    fixtures.seed_weights(probe, seed)
    probe.train()
    with torch.no_grad():
        p0 = probe.forward_head(torch.from_numpy(batch["points"]),
                                [torch.from_numpy(f) for f in batch["img_features"]],
                                batch["img_metas"])
    if ball_margin(p0["vote_points"].numpy(), p0["aggregated_points"].numpy(),
                   cfg.head.agg_radius) < 1e-5:
        return None
    agg = p0["aggregated_points"].numpy()
    rng = np.random.default_rng(seed)
    gtb, gtl = [], []
    ne = NUM_EXTRA_BOXES
    for b in range(B):
        pick = rng.choice(agg.shape[1], ne, replace=False)
        dims = rng.uniform(0.6, 1.4, size=(ne, 3))
        ctr = agg[b, pick] + rng.normal(0, 0.04, size=(ne, 3))
        extra = np.concatenate([ctr - [0, 0, 1] * dims * 0.5, dims, rng.uniform(-3, 3, (ne, 1))], 1)
        gtb.append(np.concatenate([batch["gt_boxes"][b], extra.astype(np.float32)], 0))
        gtl.append(np.concatenate([batch["gt_labels"][b], rng.integers(0, 10, ne)]))
    return batch, gtb, gtl


_CASES = {}      # qualified cases of this pytest session: the B=8 oracle runs take minutes of CPU


def qualified_case(cfg, B, N, pyramid, in_shape, img_shape, seeds=range(1, 12), log=print, risk_max=RISK_MAX):
    """The first seed whose discrete events all keep their distance (module docstring); ``risk_max``
    tightens the flip-risk bound for a test whose gradient cap is below 2 x RISK_MAX.
    -> dict(seed, batch, gtb, gtl, truth, cpu32) ; raises if none of ``seeds`` qualifies.
    Cached per argument set for the session (the fp32 and the bf16 whole-path tests of the same
    configuration share the oracle runs; the choice still depends on the oracle alone)."""
    key = (repr(cfg), B, N, tuple(pyramid), tuple(in_shape), tuple(img_shape) if img_shape else None,
           tuple(seeds), risk_max)
    if key not in _CASES:
        _CASES[key] = _qualified_case(cfg, B, N, pyramid, in_shape, img_shape, seeds, log, risk_max)
    return _CASES[key]


def _qualified_case(cfg, B, N, pyramid, in_shape, img_shape, seeds, log, risk_max=RISK_MAX):
    why = []
    for seed in seeds:
        case = make_case(cfg, B, N, pyramid, in_shape, img_shape, seed)
        if case is None:
            why.append(f"seed {seed}: neighbour on the aggregation ball boundary")
            continue
        batch, gtb, gtl = case
        truth = oracle_run(cfg, batch, gtb, gtl, seed, torch.float64)
        tm = target_margins(cfg, truth["preds"], truth["targets"], gtb)
        if tm < 1e-4:
            why.append(f"seed {seed}: target-assignment margin {tm:.1e} m")
            continue
        cpu32 = oracle_run(cfg, batch, gtb, gtl, seed, torch.float32, truth_taps=truth["taps"])
        risks = flip_risks(truth["taps"], cpu32["taps"].noise)
        worst = max(risks.items(), key=lambda kv: kv[1][0])
        if worst[1][0] > risk_max:
            why.append(f"seed {seed}: flip risk {worst[1][0]:.1e} at {worst[0]} "
                       f"({worst[1][2]} elements within {worst[1][1]:.1e})")
            continue
        log(f"[parity] seed {seed} qualifies: target margin {tm:.1e} m, worst flip risk "
            f"{worst[1][0]:.1e} at {worst[0]} over {len(risks)} tapped layers; skipped: {why}")
        truth["taps"] = cpu32["taps"] = None            # free the activations
        return dict(seed=seed, batch=batch, gtb=gtb, gtl=gtl, truth=truth, cpu32=cpu32,
                    risk=worst[1][0])
    raise AssertionError("no qualifying seed: " + "; ".join(why))


def emulated_runs(cfg, case):
    """The two bf16-emulating oracle runs of a qualified case (cached on the case): emu64 = emulated
    truth, emu32 = the same rounding points with fp32 accumulation."""
    if "emu64" not in case:
        a = (cfg, case["batch"], case["gtb"], case["gtl"], case["seed"])
        case["emu64"] = oracle_run(*a, torch.float64, tap=False, emulate_bf16=True)
        case["emu32"] = oracle_run(*a, torch.float32, tap=False, emulate_bf16=True)
    return case["emu64"], case["emu32"]


def rel_l2(a, t):
    return ((a.double().cpu() - t).norm() / t.norm()).item()


def _group(name):
    """Parameters that receive the same output gradient: 'a.b.layer1.conv.weight' and
    'a.b.layer1.bn.bias' -> 'a.b.layer1'; 'x.conv_out.bias' -> 'x.conv_out'."""
    parts = name.split(".")[:-1]
    if parts and parts[-1] in ("conv", "bn"):
        parts = parts[:-1]
    return ".".join(parts)


def compare_grads(truth, cpu32, gpu_grads, rtol=1e-3, mult=4.0, allowance=0.0, cap=None):
    """EVERY gradient tensor: rel-L2 error vs the fp64 oracle <= max(rtol, mult x the CPU fp32
    oracle's own error on that tensor) + ``allowance`` (the qualified seed's residual flip risk:
    the share of a layer's gradient that elements inside the fp32 noise window still carry).
    A flipped element's gradient d enters its layer's weight gradient as d x (input row) and the
    1-D parameters of the same layer (bias, BN scale / shift: column sums, which cancel) as d itself,
    so for 1-D tensors the flip allowance is taken relative to the norm of the whole layer group
    (weight + bias + BN affine) instead of the cancelled sum's own norm; the no-flip part of the bound
    stays relative to the tensor itself.  Tensors that are mathematically zero (conv biases in front
    of a train-mode BN) must be at noise level relative to the largest gradient.
    ``cap``: hard ceiling on any tensor's error whatever the allowance says (2x the worst error
    MEASURED on MI355X for the case, tests/test_gpu_model.py) - the bound a regression has to beat.
    -> list of failure strings (empty = pass) and the table rows (name, |g|, gpu err, cpu32 err,
    error in units of the flip scale)."""
    gt, gc = truth["grads"], cpu32["grads"]
    assert sorted(gt) == sorted(gpu_grads), set(gt) ^ set(gpu_grads)
    gmax = max(v.norm().item() for v in gt.values())
    gnorm2 = {}
    for n, v in gt.items():
        gnorm2[_group(n)] = gnorm2.get(_group(n), 0.0) + v.double().norm().item() ** 2
    bad, rows = [], []
    for n in sorted(gt):
        nt = gt[n].norm().item()
        if nt < 1e-6 * gmax:
            e = gpu_grads[n].double().cpu().norm().item()
            rows.append((n, nt, e, float("nan"), float("nan")))
            if e > 1e-4 * gmax:
                bad.append(f"{n}: should be ~0, got norm {e:.2e} (largest gradient {gmax:.2e})")
            continue
        rg, rc = rel_l2(gpu_grads[n], gt[n]), rel_l2(gc[n], gt[n])
        flip_scale = max(1.0, gnorm2[_group(n)] ** 0.5 / nt) if gt[n].dim() == 1 else 1.0
        rows.append((n, nt, rg, rc, rg / flip_scale))
        bound = max(rtol, mult * rc) + allowance * flip_scale
        if cap is not None:
            bound = min(bound, max(rtol, cap * flip_scale))
        if rg > bound:
            bad.append(f"{n}: gpu {rg:.2e} vs cpu32 {rc:.2e} (norm {nt:.2e}, flip scale {flip_scale:.1f}, "
                       f"bound {bound:.2e})")
    return bad, rows
