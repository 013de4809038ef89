"""CPU ORACLE (test infrastructure, NOT a product path) - bf16-emulating mode of the oracle model.

BASELINE.json configs[3] is a variant the reference does not have (``fp16_enabled = False``,
demf/modeling/heads/class_agnostic_vote_head.py:384): the product's bf16 compute mode rounds the
operands of every dense contraction to bf16 (round-to-nearest-even) and accumulates in fp32, everything
else stays fp32.  With seeded random weights the 30 train-mode BatchNorm layers of the path amplify
operand-rounding noise by ~400x towards the heads, so the fp64 / fp32 oracle is no yard-stick for that
mode.  ``bf16_emulation()`` makes the SAME oracle model (oracle/model.py, oracle/deps.py - same
parameters, same state dict) round at exactly the places the kernels round, forward and backward:

  * every 1x1 convolution / linear layer:  y = q(x) q(W)^T (+ b in full precision);
    backward  dx = q(dy) q(W),  dW = q(dy)^T q(x),  db = sum dy  (csrc/mlp.hip CM = 1, csrc/mlp_bwd.hip,
    csrc/dense.hip BF16 instantiations; q = round to bf16);
  * the first layer of a set-abstraction level whose input carries features (SA2-4, the vote
    aggregation) in the product's factored form (csrc/group_first.hip): U = q(feat) q(Wf)^T per SOURCE
    point, y = U[idx] + rel_xyz Wx^T with the coordinate term in full precision - so the backward rounds
    the gradient summed per source point (dU), as the kernels do, not the per-neighbour rows;
  * SA1's first layer (4-float rows): forward rounded, but its weight gradient comes out of the fp32
    epilogue sums of the layer above (the FIRST epilogue): NOT rounded;
  * SA1's pooled last layer in the no-store form (``_PooledLast``): same forward values, backward re-associated
    around the layer's INPUT, A and W^T diag(a) W rounded where dY and W were before round 4;
  * self attention with the kernels' operand order: S = q(Q) q(K)^T * (1/sqrt(Dh)) (scale after the
    product), O = q(dropout(softmax S)) q(V);
  * the fusion cross-attention as sample-then-project (demf_amd/ops.py msda_sample_then_project):
    z = MSDA(keep * tokens) in full precision, out_h = q(z_h) q(Wv_h)^T + bv_h * MSDA(keep), whose
    backward rounds the operands of all four small products (dz, dWv, dbv, dksum).

Coordinates, indices, BatchNorm / LayerNorm arithmetic, interpolation, sampling, targets and losses
are untouched.  Works for float32 and float64 models: the float64 run is the "emulated truth" (same
rounding points, no accumulation noise), the float32 run measures the accumulation noise around it.
"""
import contextlib
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import deps
from . import torch_ops as O


def q(x):
    """round-to-nearest-even to bf16, kept in x's dtype"""
    return x.to(torch.bfloat16).to(x.dtype)


class _MM(Function):
    """y = q(x) q(w)^T ; dx = q(g) q(w) ; dw = q(g)^T q(x)  (round_dw False: dw = g^T x unrounded)."""

    @staticmethod
    def forward(ctx, x, w, round_dw):
        ctx.save_for_backward(x, w)
        ctx.round_dw = round_dw
        return q(x) @ q(w).t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = q(g) @ q(w) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = q(g).t() @ q(x) if ctx.round_dw else g.t() @ x
        return gx, gw, None


def mm(x, w, round_dw=True):
    """rows (R,K) x weight (N,K) -> (R,N) with the kernels' rounding points."""
    return _MM.apply(x, w, round_dw)


class _BMM(Function):
    """batched alpha * (q(a) (.., M, K) @ q(b) (.., K, N)): both operands rounded, forward and backward;
    ``alpha`` is applied to the fp32 accumulator (the GEMM epilogue), never to an operand."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return (q(a) @ q(b)) * alpha

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gq = q(g)
        return (gq @ q(b).transpose(-1, -2)) * ctx.alpha, (q(a).transpose(-1, -2) @ gq) * ctx.alpha, None


class _RowBias(Function):
    """out (R,H,Dh) = ks (R,H,1) * bias (H,Dh): exact forward (the GEMM epilogue's fp32 fma), backward
    through the two N = 1 bf16 GEMMs of demf_amd/fused.py (d_vp_b, dks4)."""

    @staticmethod
    def forward(ctx, ks, bias):
        ctx.save_for_backward(ks, bias)
        return ks.unsqueeze(-1) * bias.unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        ks, bias = ctx.saved_tensors
        gq = q(g)
        return (gq * q(bias).unsqueeze(0)).sum(-1), (gq * q(ks).unsqueeze(-1)).sum(0)


class _PooledLast(Function):
    """Conv1x1 -> train-mode BN -> ReLU -> max over ns rows of an SA1-shaped last layer (64 -> 128 channels,
    64-row groups, >= 16 384 rows) the way the product runs it since round 4 (csrc/mlp_bwd.hip
    mlp_bwd_pool_kernel): the raw output is never stored, and the backward is written in terms of the
    layer's input A:   dA = q(gi dZ) q(W) + q(A) q(M) + b^T W,   M = W^T diag(a) W  (formed in full precision),
    dW = (gi dZ)^T A + diag(a) W (q(A)^T q(A)) + b (x) colsum(A)   (sparse product and column sums in full
    precision).  Forward values equal the plain rounded layer's."""

    @staticmethod
    def forward(ctx, A, W, gamma, beta, ns, eps):
        R, K = A.shape
        y = q(A) @ q(W).t()
        mean = y.mean(0)
        invstd = torch.rsqrt(y.var(0, unbiased=False) + eps)
        sc = gamma * invstd
        sh = beta - mean * sc
        z = torch.relu(y * sc + sh).view(R // ns, ns, -1)
        out = z.max(1)[0]
        first = (z == out.unsqueeze(1)).to(torch.int64).argmax(1)       # the FIRST maximal row (upstream's max-pool)
        yraw = torch.gather(y.view(R // ns, ns, -1), 1, first.unsqueeze(1)).squeeze(1)
        ctx.save_for_backward(A, W, gamma, mean, invstd, sh, first, yraw)
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, dP):
        A, W, gamma, mean, invstd, sh, arg, yraw = ctx.saved_tensors
        ns = ctx.ns
        R, K = A.shape
        G, N = dP.shape
        gi = gamma * invstd
        dz = torch.where(yraw * gi + sh > 0, dP, torch.zeros_like(dP))
        g1, g2 = dz.sum(0), (dz * ((yraw - mean) * invstd)).sum(0)
        a = -gi * invstd * (g2 / R)
        b = -gi * (g1 / R) - a * mean
        dzs = torch.zeros(G, ns, N, dtype=A.dtype)
        dzs.scatter_(1, arg.unsqueeze(1), (gi * dz).unsqueeze(1))
        dzs = dzs.view(R, N)
        M = W.t() @ (a.unsqueeze(1) * W)
        dA = q(dzs) @ q(W) + q(A) @ q(M) + (b @ W)
        dW = dzs.t() @ A + a.unsqueeze(1) * (W @ (q(A).t() @ q(A))) + b.unsqueeze(1) * A.sum(0).unsqueeze(0)
        return dA, dW, g2, g1, None, None


def _no_store_last(R, ns, mlp):
    """demf_amd/ops.py _pool_noy_ok: the stacks whose last layer runs without its stored output."""
    if len(mlp) < 2 or ns != 64 or R < 16384 or R % 64:
        return False
    last, prev = mlp[len(mlp) - 1].conv, mlp[len(mlp) - 2].conv
    return last.out_channels == 128 and last.in_channels == 64 and prev.out_channels == 64


_STATE = {"on": False, "first_no_round_dw": False}
_ORIG = {}


def _linear(x, w, b=None):
    if not _STATE["on"] or not x.is_floating_point():
        return _ORIG["linear"](x, w, b)
    y = mm(x.reshape(-1, x.shape[-1]), w).reshape(*x.shape[:-1], w.shape[0])
    return y if b is None else y + b


def _conv(nd):
    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        one = all(int(k) == 1 for k in w.shape[2:])
        if not _STATE["on"] or not one or groups != 1 or not x.is_floating_point():
            return _ORIG["conv%dd" % nd](x, w, b, stride, padding, dilation, groups)
        C = x.shape[1]
        rows = x.movedim(1, -1).reshape(-1, C)
        y = mm(rows, w.reshape(w.shape[0], C), round_dw=not _STATE["first_no_round_dw"])
        if b is not None:
            y = y + b
        return y.reshape(*x.shape[:1], *x.shape[2:], w.shape[0]).movedim(-1, 1)
    return conv


# ---- PointSAModule: the product's two forms of the first layer ---------------------------------------------
def _sa_forward(self, points_xyz, features=None, indices=None, target_xyz=None):
    xyz_flipped = points_xyz.transpose(1, 2).contiguous()
    if indices is not None:
        new_xyz = O.gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
    elif target_xyz is not None:
        new_xyz = target_xyz.contiguous()
    else:
        indices = O.furthest_point_sample(points_xyz, self.num_point)
        new_xyz = O.gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
    idx = O.ball_query(0.0, self.radius, self.num_sample, points_xyz.contiguous(), new_xyz)
    grouped_xyz = O.grouping_operation(xyz_flipped, idx)
    grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
    if self.normalize_xyz:
        grouped_xyz = grouped_xyz / self.radius
    mlp = self.mlps[0]
    C = 0 if features is None else features.shape[1]
    layer0 = mlp[0]
    factored = self.use_xyz and features is not None and len(mlp) >= 2 and C % 4 == 0 and \
        layer0.conv.out_channels in (64, 128, 256)              # demf_amd/modules/pointnet2.py: group_first
    if factored:
        W = layer0.conv.weight.reshape(layer0.conv.out_channels, 3 + C)      # reference order [xyz | feat]
        B, _, N = features.shape
        U = mm(features.transpose(1, 2).reshape(B * N, C), W[:, 3:]).reshape(B, N, -1).transpose(1, 2)
        yu = O.grouping_operation(U.contiguous(), idx)                        # (B, C1, M, ns): gathered rows of U
        yx = torch.einsum("bcms,oc->boms", grouped_xyz, W[:, :3])            # coordinate term: full precision
        x = F.relu(layer0.bn(yu + yx))
        for blk in list(mlp)[1:]:
            x = blk(x)
    else:
        if features is not None:
            grouped_features = O.grouping_operation(features.contiguous(), idx)
            x = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            x = grouped_xyz
        # SA1-shaped stacks (4-float rows, >= 3 layers, <= 64 first channels, no input gradient): layer 0's
        # weight gradient is formed from fp32 sums (the FIRST epilogue), not by a bf16 GEMM
        first = x.shape[1] == 4 and len(mlp) >= 3 and layer0.conv.out_channels <= 64 and not x.requires_grad
        _STATE["first_no_round_dw"] = first
        try:
            x = layer0(x)
        finally:
            _STATE["first_no_round_dw"] = False
        Bx, _, Mx, nsx = x.shape
        if _no_store_last(Bx * Mx * nsx, nsx, mlp) and self.training:
            for blk in list(mlp)[1:-1]:
                x = blk(x)
            last = mlp[len(mlp) - 1]
            rows = x.permute(0, 2, 3, 1).reshape(Bx * Mx * nsx, x.shape[1])
            Wl = last.conv.weight.reshape(last.conv.out_channels, -1)
            pooled = _PooledLast.apply(rows, Wl, last.bn.weight, last.bn.bias, nsx, last.bn.eps)
            return new_xyz, pooled.view(Bx, Mx, -1).transpose(1, 2), indices
        for blk in list(mlp)[1:]:
            x = blk(x)
    new_features = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
    return new_xyz, new_features, indices


# ---- self attention (nn.MultiheadAttention restated with the kernels' operand order) --------------------
def _mha_forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                 attn_mask=None, key_padding_mask=None, **kw):
    assert attn_mask is None and key_padding_mask is None and (key is None or key is query)
    identity = query if identity is None else identity
    a = self.attn
    x = query                                          # (Q, B, E)
    xp = x if query_pos is None else x + query_pos
    Qn, B, E = x.shape
    H = a.num_heads
    Dh = E // H
    w, b = a.in_proj_weight, a.in_proj_bias
    qk = _linear(xp, w[:2 * E], b[:2 * E])             # q | k see x + pos (one GEMM with the A2 prologue)
    v = _linear(x, w[2 * E:], b[2 * E:])
    heads = lambda t: t.reshape(Qn, B * H, Dh).transpose(0, 1)                 # (B*H, Q, Dh)
    qh, kh, vh = heads(qk[..., :E]), heads(qk[..., E:]), heads(v)
    s = _BMM.apply(qh, kh.transpose(1, 2), 1.0 / math.sqrt(Dh))             # scale AFTER the product
    p = F.dropout(s.softmax(-1), a.dropout, self.training)
    o = _BMM.apply(p, vh, 1.0).transpose(0, 1).reshape(Qn, B, E)
    out = _linear(o, a.out_proj.weight, a.out_proj.bias)
    return identity + self.dropout_layer(self.proj_drop(out))


# ---- fusion cross-attention: sample, THEN project -----------------------------------------------------------
def _msda_forward(self, query, key=None, value=None, identity=None, query_pos=None,
                  key_padding_mask=None, reference_points=None, spatial_shapes=None,
                  level_start_index=None, **kw):
    value = query if value is None else value
    identity = query if identity is None else identity
    if query_pos is not None:
        query = query + query_pos
    if not self.batch_first:
        query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
    bs, nq, E = query.shape
    bs, nv, C = value.shape
    H, L, P = self.num_heads, self.num_levels, self.num_points
    Dh = E // H
    samples = nq * L * P * 4
    if not (samples < nv and key_padding_mask is not None):
        raise NotImplementedError("bf16 emulation covers the sample-then-project form only (decoder layers)")
    keep = (~key_padding_mask).to(value.dtype)                                    # (bs, nv)
    # (the product keeps the image tokens as bf16 rows in this mode - ops.pyramid_to_tokens(bf16=True): the
    # gather reads rounded tokens; no gradient flows to them)
    tokens = q(value * keep.unsqueeze(-1))
    off = _linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(bs, nq, H, L, P, 2)
    aw = _linear(query, self.attention_weights.weight, self.attention_weights.bias) \
        .view(bs, nq, H, L * P).softmax(-1).view(bs, nq, H, L, P)
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    loc1 = loc.reshape(bs, nq * H, 1, L, P, 2).contiguous()
    aw1 = aw.reshape(bs, nq * H, 1, L, P).contiguous()
    z = O.MultiScaleDeformableAttnFunction.apply(tokens.reshape(bs, nv, 1, C).contiguous(), spatial_shapes,
                                                 level_start_index, loc1, aw1, self.im2col_step)
    ks = O.MultiScaleDeformableAttnFunction.apply(keep.reshape(bs, nv, 1, 1).contiguous(), spatial_shapes,
                                                  level_start_index, loc1, aw1, self.im2col_step)
    z = z.reshape(bs * nq, H, C)
    ks = ks.reshape(bs * nq, H)
    Wv = self.value_proj.weight.reshape(H, Dh, C)
    bv = self.value_proj.bias.reshape(H, Dh)
    out = torch.stack([mm(z[:, h], Wv[h]) for h in range(H)], 1) + _RowBias.apply(ks, bv)   # (R, H, Dh)
    out = _linear(out.reshape(bs, nq, E), self.output_proj.weight, self.output_proj.bias)
    if not self.batch_first:
        out = out.permute(1, 0, 2)
    return self.dropout(out) + identity


@contextlib.contextmanager
def bf16_emulation():
    """Inside: every oracle model forward AND backward rounds where the bf16 compute mode rounds.
    (The backward must run inside the context too - autograd calls the patched functions' own
    backward, which does not depend on the patches, so it may also run after it.)"""
    assert not _STATE["on"], "bf16_emulation is not re-entrant"
    _ORIG.update(linear=F.linear, conv1d=F.conv1d, conv2d=F.conv2d, sa=deps.PointSAModule.forward,
                 mha=deps.MultiheadAttention.forward, msda=deps.MultiScaleDeformableAttention.forward)
    F.linear, F.conv1d, F.conv2d = _linear, _conv(1), _conv(2)
    deps.PointSAModule.forward = _sa_forward
    deps.MultiheadAttention.forward = _mha_forward
    deps.MultiScaleDeformableAttention.forward = _msda_forward
    _STATE["on"] = True
    try:
        yield
    finally:
        _STATE["on"] = False
        F.linear, F.conv1d, F.conv2d = _ORIG["linear"], _ORIG["conv1d"], _ORIG["conv2d"]
        deps.PointSAModule.forward = _ORIG["sa"]
        deps.MultiheadAttention.forward = _ORIG["mha"]
        deps.MultiScaleDeformableAttention.forward = _ORIG["msda"]
