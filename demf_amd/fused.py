"""The DeMF fusion decoder layer as ONE autograd node on the gfx950 dense kernels (csrc/dense.hip +
the MSDA kernels of csrc/msda.hip).

Reference: ``DeMFTransformerDecoderLayer.forward`` (demf/modeling/layers/transformer.py:55-80) ->
mmcv ``DetrTransformerDecoderLayer`` with operation_order ('self_attn','norm','cross_attn','norm',
'ffn','norm') (configs/demf/demf_votenet.py:71-91), fed by ``DeMFVoteHead.get_reference_points``
(class_agnostic_vote_head.py:524-547).  Upstream this is ~190 ATen / cuBLAS launches forward and as
many backward for 2 048 query rows; here it is 19 launches forward and ~40 backward, every one a
kernel of this package:

  qkv = [x+pos | x] . Win^T            1 GEMM (pos added in the A prologue for the q|k columns)
  S = q k^T / sqrt(Dh) ; P = softmax ; Pd = dropout(P) ; O = Pd v     2 batched GEMMs + 1 row kernel
  x1 = LN(x + dropout(O Wout^T))       1 GEMM + 1 row kernel
  raw = (x1+pos) . [Woff;Waw]^T        2 GEMMs ; locations / weights: 1 kernel (projects the query
                                        points into the image on the way: get_reference_points)
  z = MSDA(tokens; loc, w), ksum = MSDA(keep; loc, w)       sample-then-project, see ops.py
  x2 = LN(x1 + dropout((Wv_h z_h + bv_h ksum_h)_h Wop^T))   2 GEMMs + 1 row kernel
  x3 = LN(x2 + dropout(dropout(relu(x2 W0^T)) W1^T))         2 GEMMs + 1 row kernel

Dropout masks are counter-based (seed, step, op, element) and recomputed in the backward.  The step
counter lives on the device; every TRAINING forward of the node advances it and takes a snapshot
(``demf_rng_next``, one 1-thread launch inside the node, hence inside a captured hipGraph: every replay
draws fresh masks) and the backward re-derives its masks from that snapshot, not from the live counter -
so plain ``loss.backward()`` loops, gradient accumulation and a user's own optimizer all see fresh,
correctly paired masks without any help from the step engine.  Streams of different decoder layers are
salted with the layer index (``op = base + 8 * layer``).
"""
import ctypes
import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _ffi

RELU, DROPOUT, GATE, ACCUM, ROWBIAS, ACCUM2 = 1, 2, 4, 8, 16, 32

# dropout streams of the layer
OP_ATTN, OP_LN1, OP_LN2, OP_FFN, OP_LN3 = 1, 2, 3, 4, 5

_RNG = {}
ATTN_CORE = True  # the one-launch attention core (csrc/attn.hip); False: QK^T / softmax / PV as three launches
_DEBUG = None     # tools/debug_fused.py: a dict that FusedDecoderLayer.backward fills with its intermediates


def rng_state(device, seed=None):
    """(2,) int64 device tensor [seed, step] shared by every fused dropout on ``device``."""
    device = torch.device(device)
    if device.index is None:                      # "cuda" and "cuda:0" are the same state
        device = torch.device(device.type, torch.cuda.current_device())
    key = str(device)
    if key not in _RNG or seed is not None:
        s = torch.initial_seed() if seed is None else int(seed)
        t = torch.tensor([s & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        if key in _RNG:
            _RNG[key].copy_(t)        # in place: captured graphs keep reading the same buffer
        else:
            _RNG[key] = t
    return _RNG[key]


def advance_rng(device):
    """Skip one step of the counter (the fused node advances it by itself on every training forward)."""
    _ffi.call("demf_rng_advance", rng_state(device).data_ptr(), torch.cuda.current_stream().cuda_stream)


def next_rng(device):
    """Advance the device counter and return the (seed, step) snapshot the caller's masks are drawn
    from (what a training forward of ``FusedDecoderLayer`` does first)."""
    state = rng_state(device)
    snap = torch.empty_like(state)
    _ffi.call("demf_rng_next", state.data_ptr(), snap.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return snap


def peek_next_rng(device):
    """The snapshot the NEXT training forward will draw from, without advancing (test hook)."""
    snap = rng_state(device).clone()
    snap[1] += 1
    return snap


def get_rng_state(device):
    """[seed, step] as Python ints (checkpointing: Trainer.state_dict)."""
    return [int(v) for v in rng_state(device).tolist()]


def set_rng_state(device, seed_step):
    rng_state(device).copy_(torch.tensor([int(seed_step[0]), int(seed_step[1])], dtype=torch.int64))


def dropout_mask(numel, p, op_id, device, state=None):
    """keep / (1-p) per element, exactly as the fused kernels draw it for (``state``, op_id);
    ``state`` defaults to the live counter."""
    out = torch.empty(numel, dtype=torch.float32, device=device)
    st = rng_state(device) if state is None else state
    _ffi.call("demf_dropout_mask", numel, float(p), st.data_ptr(), int(op_id),
              out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out


def _st():
    return torch.cuda.current_stream().cuda_stream


def _p(t, off=0):
    return None if t is None else t.data_ptr() + 4 * off


def gemm(M, N, K, A, sa, B, sb, C, scm, *, batch=1, zdiv=1, sab=(0, 0), sbb=(0, 0), scb=(0, 0),
         A2=None, a2_cols=0, B2=None, b2_rows=0, C2=None, bias=None, sbias_b=0, rowscale=None,
         srs=(0, 0), alpha=1.0, flags=0, gate=None, sg=(0, 0), gate_scale=1.0, drop_p=0.0, rng=None,
         op_id=0, splitk=1, asum=None, group=None):
    """demf_gemm_f32 (include/demf_hip.h).  A, B, C, ... are device ADDRESSES (``_p(tensor, offset)``);
    sa = (sam, sak), sb = (sbn, sbk) element strides.  ``group``: a list that collects the descriptor
    instead of launching it (``gemm_group`` then issues the whole list, demf_gemm_group_f32); descriptors
    hold raw addresses, so the caller keeps every operand tensor alive until ``gemm_group`` has run."""
    d = _ffi.GemmDesc()
    d.M, d.N, d.K, d.batch, d.zdiv, d.splitk = M, N, K, batch, zdiv, splitk
    d.A, d.sam, d.sak, d.sab, d.sab2 = A, sa[0], sa[1], sab[0], sab[1]
    d.A2, d.a2_cols = A2, a2_cols
    d.B, d.sbn, d.sbk, d.sbb, d.sbb2 = B, sb[0], sb[1], sbb[0], sbb[1]
    d.B2, d.b2_rows = B2, b2_rows
    d.C, d.scm, d.scb, d.scb2 = C, scm, scb[0], scb[1]
    d.C2 = C2
    d.bias, d.sbias_b = bias, sbias_b
    d.rowscale, d.srs_m, d.srs_b = rowscale, srs[0], srs[1]
    d.alpha, d.flags = alpha, flags
    d.gate, d.sgm, d.sgb, d.gate_scale = gate, sg[0], sg[1], gate_scale
    d.drop_p, d.rng, d.op_id = drop_p, rng, op_id
    d.asum = asum
    if group is not None:
        group.append(d)
        return
    _ffi.call("demf_gemm_f32", ctypes.addressof(d), _st())


def gemm_group(descs):
    """Issue the collected descriptors: consecutive weight-gradient-shaped ones share one launch."""
    if not descs:
        return
    arr = (_ffi.GemmDesc * len(descs))(*descs)
    _ffi.call("demf_gemm_group_f32", ctypes.addressof(arr), len(descs), _st())


def _splitk(K):
    """Reduction over rows (weight gradients): enough K-slices to fill the chip."""
    return max(1, min(16, K // 256))


def linear_fwd(x, w, b, out=None, **kw):
    """rows (R,K) . w (N,K)^T + b -> (R,N)"""
    R, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
    gemm(R, N, K, _p(x), (K, 1), _p(w), (K, 1), _p(out), N, bias=_p(b), **kw)
    return out


def weight_grad(dy, x, dw, db=None, x2=None, dy_cols=None, dy_off=0, x2_rows=0, group=None):
    """dw (N,K) += dy[:, off:off+N]^T . (x + x2) ;  db (N) += column sums of dy (taken by the same
    launch from the rows it stages: ``asum``).  dw / db arrive ZEROED (split-K accumulates with atomics).
    ``group``: collect the descriptor for one grouped launch (``gemm_group``)."""
    R = dy.shape[0]
    N, K = dw.shape
    ldy = dy.shape[1]
    fused_bias = db is not None and N % 4 == 0
    gemm(N, K, R, _p(dy, dy_off), (1, ldy), _p(x), (1, x.shape[1]), _p(dw), K,
         B2=_p(x2), b2_rows=x2_rows if x2 is not None else 0, splitk=_splitk(R),
         asum=_p(db) if fused_bias else None, group=group)
    if db is not None and not fused_bias:
        _ffi.call("demf_colsum_f32", R, N, ldy, _p(dy, dy_off), _p(db), _st())


class FusedDecoderLayer(Function):
    """x (R,E), pos (R,E), pts (R,3) -> x3 (R,E), rows batch-major (R = B*Q)."""

    @staticmethod
    def forward(ctx, x, pos, pts, tokens, keep4, shapes, lsi, M, ab, vr, dims, training,
                in_w, in_b, out_w, out_b, g1, b1, off_w, off_b, aw_w, aw_b, vp_w, vp_b, op_w, op_b,
                g2, b2, f0_w, f0_b, f1_w, f1_b, g3, b3):
        B, Q, H, L, P, p_attn, p_ffn, eps = dims[:8]
        salt = 8 * (int(dims[8]) if len(dims) > 8 else 0)     # decoder layer index: independent streams
        if not training:
            p_attn = p_ffn = 0.0
        dev = x.device
        R, E = x.shape
        Dh, F, Ct, S = E // H, f0_w.shape[0], tokens.shape[2], tokens.shape[1]
        HLP = H * L * P
        assert R == B * Q and x.is_contiguous() and pos.is_contiguous() and pts.is_contiguous()
        # this forward's own (seed, step): drawn here, saved for the backward
        snap = next_rng(dev) if (p_attn > 0.0 or p_ffn > 0.0) else rng_state(dev)
        rng = snap.data_ptr()
        OP_ATTN, OP_LN1, OP_LN2, OP_FFN, OP_LN3 = (salt + o for o in (1, 2, 3, 4, 5))
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        st = _st()
        # ---- self attention (nn.MultiheadAttention: q = k = x + pos, v = x) ----
        qkv = new(R, 3 * E)
        gemm(R, 3 * E, E, _p(x), (E, 1), _p(in_w), (E, 1), _p(qkv), 3 * E, A2=_p(pos), a2_cols=2 * E,
             bias=_p(in_b))
        att = new(R, E)
        core = ATTN_CORE and Q == 256 and Dh == 32
        if core:
            # QK^T -> softmax -> dropout -> PV as ONE launch per direction (csrc/attn.hip); the backward
            # recomputes the probabilities from two floats per query
            prob = pd = new(0)
            ast = new(B * H * Q, 2)
            dbg = _DEBUG is not None
            if dbg:
                prob, pd = new(B * H, Q, Q), new(B * H, Q, Q)
            _ffi.call("demf_attn_core_fwd", B, H, Q, Dh, _p(qkv), 1.0 / math.sqrt(Dh), p_attn, rng, OP_ATTN,
                      _p(att), _p(ast), _p(prob) if dbg else None, _p(pd) if dbg else None, st)
        else:
            ast = new(0)
            sc = new(B * H, Q, Q)
            zb = (Q * 3 * E, Dh)                                     # (scene, head) -> offset into qkv
            gemm(Q, Q, Dh, _p(qkv), (3 * E, 1), _p(qkv, E), (3 * E, 1), _p(sc), Q, batch=B * H, zdiv=H,
                 sab=zb, sbb=zb, scb=(H * Q * Q, Q * Q), alpha=1.0 / math.sqrt(Dh))
            prob, pd = new(B * H, Q, Q), sc                          # dropout(prob) overwrites the scores
            _ffi.call("demf_softmax_dropout_fwd", B * H * Q, Q, _p(sc), p_attn, rng, OP_ATTN, _p(prob),
                      _p(pd), st)
            gemm(Q, Dh, Q, _p(pd), (Q, 1), _p(qkv, 2 * E), (1, 3 * E), _p(att), E, batch=B * H, zdiv=H,
                 sab=(H * Q * Q, Q * Q), sbb=zb, scb=(Q * E, Dh))
        s1 = linear_fwd(att, out_w, out_b)
        x1, st1 = new(R, E), new(R, 2)
        _ffi.call("demf_add_dropout_ln_fwd", R, E, _p(s1), _p(x), _p(g1), _p(b1), eps, p_attn, rng,
                  OP_LN1, _p(s1), _p(x1), _p(st1), st)
        # ---- cross attention into the image tokens (MultiScaleDeformableAttention) ----
        raw = new(R, 3 * HLP)
        gemm(R, 2 * HLP, E, _p(x1), (E, 1), _p(off_w), (E, 1), _p(raw), 3 * HLP, A2=_p(pos),
             a2_cols=2 * HLP, bias=_p(off_b))
        gemm(R, HLP, E, _p(x1), (E, 1), _p(aw_w), (E, 1), _p(raw, 2 * HLP), 3 * HLP, A2=_p(pos),
             a2_cols=HLP, bias=_p(aw_b))
        loc, w, uvw = new(R, H, L, P, 2), new(R, H, L, P), new(R, 4)
        _ffi.call("demf_msda_prep_fwd", R, Q, H, L, P, _p(pts), _p(M), _p(ab), _p(vr), shapes.data_ptr(),
                  _p(raw), _p(loc), _p(w), _p(uvw), st)
        z, ks4 = new(R * H, Ct), new(R * H, 4)
        # sample-then-project (ops.msda_sample_then_project): every (query, head) is one item of a
        # single-head MSDA over the unprojected tokens / the keep mask
        # (bf16 token rows in the bf16 compute mode: ops.pyramid_to_tokens(bf16=True), half the gather's bytes)
        _ffi.call("demf_msda_fwd_bf16" if tokens.dtype == torch.bfloat16 else "demf_msda_fwd_f32", B, S, 1, Ct, L,
                  Q * H, P, _p(tokens), shapes.data_ptr(), lsi.data_ptr(), _p(loc), _p(w), _p(z), st)
        _ffi.call("demf_msda_fwd_f32", B, S, 1, 4, L, Q * H, P, _p(keep4), shapes.data_ptr(),
                  lsi.data_ptr(), _p(loc), _p(w), _p(ks4), st)
        mo = new(R, E)
        gemm(R, Dh, Ct, _p(z), (H * Ct, 1), _p(vp_w), (Ct, 1), _p(mo), E, batch=H, sab=(Ct, 0),
             sbb=(Dh * Ct, 0), scb=(Dh, 0), bias=_p(vp_b), sbias_b=Dh, rowscale=_p(ks4), srs=(4 * H, 4),
             flags=ROWBIAS)
        s2 = linear_fwd(mo, op_w, op_b)
        x2, st2 = new(R, E), new(R, 2)
        _ffi.call("demf_add_dropout_ln_fwd", R, E, _p(s2), _p(x1), _p(g2), _p(b2), eps, p_attn, rng,
                  OP_LN2, _p(s2), _p(x2), _p(st2), st)
        # ---- FFN ----
        hid = new(R, F)
        gemm(R, F, E, _p(x2), (E, 1), _p(f0_w), (E, 1), _p(hid), F, bias=_p(f0_b),
             flags=RELU | (DROPOUT if p_ffn > 0 else 0), drop_p=p_ffn, rng=rng, op_id=OP_FFN)
        s3 = linear_fwd(hid, f1_w, f1_b)
        x3, st3 = new(R, E), new(R, 2)
        _ffi.call("demf_add_dropout_ln_fwd", R, E, _p(s3), _p(x2), _p(g3), _p(b3), eps, p_ffn, rng,
                  OP_LN3, _p(s3), _p(x3), _p(st3), st)
        ctx.dims = (B, Q, H, L, P, p_attn, p_ffn, eps, salt)
        ctx.core = core
        ctx.save_for_backward(snap, x, pos, pts, tokens, keep4, shapes, lsi, M, ab, vr, qkv, prob, pd, att, ast,
                              s1, st1, x1, w, loc, uvw, z, ks4, mo, s2, st2, x2, hid, s3, st3,
                              in_w, out_w, g1, off_w, aw_w, vp_w, vp_b, op_w, g2, f0_w, f1_w, g3)
        return x3

    @staticmethod
    @once_differentiable
    def backward(ctx, dx3):
        (snap, x, pos, pts, tokens, keep4, shapes, lsi, M, ab, vr, qkv, prob, pd, att, ast, s1, st1, x1, w, loc,
         uvw, z, ks4, mo, s2, st2, x2, hid, s3, st3, in_w, out_w, g1, off_w, aw_w, vp_w, vp_b, op_w,
         g2, f0_w, f1_w, g3) = ctx.saved_tensors
        B, Q, H, L, P, p_attn, p_ffn, eps, salt = ctx.dims
        OP_ATTN, OP_LN1, OP_LN2, OP_FFN, OP_LN3 = (salt + o for o in (1, 2, 3, 4, 5))
        dev = x.device
        R, E = x.shape
        Dh, F, Ct, S = E // H, f0_w.shape[0], tokens.shape[2], tokens.shape[1]
        HLP = H * L * P
        rng = snap.data_ptr()                        # the forward's own (seed, step)
        st = _st()
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        dx3 = dx3.contiguous()
        wg = []     # the weight / bias gradients depend on saved tensors only: ONE grouped launch at the end
        # every parameter gradient lives in ONE zero-filled workspace (split-K GEMMs, column sums and
        # the LayerNorm reductions accumulate into it)
        shp = [(3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (2 * HLP, E), (2 * HLP,), (HLP, E), (HLP,),
               (E, Ct), (E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,), (E,), (E,)]
        sizes = [math.prod(s) for s in shp]
        from . import ops
        ws = ops.zeros(sum(sizes), dev)
        gr, o = [], 0
        for s_, n in zip(shp, sizes):
            gr.append(ws[o:o + n].view(s_))
            o += n
        (d_in_w, d_in_b, d_out_w, d_out_b, d_g1, d_b1, d_off_w, d_off_b, d_aw_w, d_aw_b, d_vp_w, d_vp_b,
         d_op_w, d_op_b, d_g2, d_b2, d_f0_w, d_f0_b, d_f1_w, d_f1_b, d_g3, d_b3) = gr
        # ---- FFN ----
        dx2, df = new(R, E), new(R, E)               # gradient of x2 (accumulated), of fc1's output
        _ffi.call("demf_add_dropout_ln_bwd", R, E, _p(dx3), None, _p(s3), _p(st3), _p(g3), p_ffn, rng,
                  OP_LN3, _p(dx2), 0, _p(df), _p(d_g3), _p(d_b3), st)
        weight_grad(df, hid, d_f1_w, d_f1_b, group=wg)
        dh = new(R, F)                               # through dropout + ReLU: gate on hid != 0
        gemm(R, F, E, _p(df), (E, 1), _p(f1_w), (1, F), _p(dh), F, flags=GATE, gate=_p(hid), sg=(F, 0),
             gate_scale=1.0 / (1.0 - p_ffn))
        weight_grad(dh, x2, d_f0_w, d_f0_b, group=wg)
        gemm(R, E, F, _p(dh), (F, 1), _p(f0_w), (1, E), _p(dx2), E, flags=ACCUM)
        # ---- cross attention ----
        dx1, dco = new(R, E), new(R, E)
        _ffi.call("demf_add_dropout_ln_bwd", R, E, _p(dx2), None, _p(s2), _p(st2), _p(g2), p_attn, rng,
                  OP_LN2, _p(dx1), 0, _p(dco), _p(d_g2), _p(d_b2), st)
        weight_grad(dco, mo, d_op_w, d_op_b, group=wg)
        dmo = new(R, E)
        gemm(R, E, E, _p(dco), (E, 1), _p(op_w), (1, E), _p(dmo), E)
        # per-head value projection applied after sampling: mo[:, h] = z_h Wv_h^T + bv_h * ksum_h
        dz, dks4 = new(R * H, Ct), ops.zeros((R * H, 4), dev)
        gemm(R, Ct, Dh, _p(dmo), (E, 1), _p(vp_w), (1, Ct), _p(dz), H * Ct, batch=H, sab=(Dh, 0),
             sbb=(Dh * Ct, 0), scb=(Ct, 0))
        gemm(Dh, Ct, R, _p(dmo), (1, E), _p(z), (1, H * Ct), _p(d_vp_w), Ct, batch=H, sab=(Dh, 0),
             sbb=(Ct, 0), scb=(Dh * Ct, 0), splitk=_splitk(R), group=wg)
        gemm(Dh, 1, R, _p(dmo), (1, E), _p(ks4), (0, 4 * H), _p(d_vp_b), 1, batch=H, sab=(Dh, 0),
             sbb=(4, 0), scb=(Dh, 0), splitk=_splitk(R))
        gemm(R, 1, Dh, _p(dmo), (E, 1), _p(vp_b), (0, 1), _p(dks4), 4 * H, batch=H, sab=(Dh, 0),
             sbb=(Dh, 0), scb=(4, 0))
        dloc, dw = new(R, H, L, P, 2), new(R, H, L, P)
        dloc2, dw2 = new(R, H, L, P, 2), new(R, H, L, P)
        _ffi.call("demf_msda_bwd_bf16" if tokens.dtype == torch.bfloat16 else "demf_msda_bwd_f32", B, S, 1, Ct, L,
                  Q * H, P, _p(tokens), shapes.data_ptr(), lsi.data_ptr(), _p(loc), _p(w), _p(dz), None, _p(dloc),
                  _p(dw), st)
        _ffi.call("demf_msda_bwd_f32", B, S, 1, 4, L, Q * H, P, _p(keep4), shapes.data_ptr(),
                  lsi.data_ptr(), _p(loc), _p(w), _p(dks4), None, _p(dloc2), _p(dw2), st)
        draw, dpts = new(R, 3 * HLP), new(R, 3)
        _ffi.call("demf_msda_prep_bwd", R, Q, H, L, P, _p(pts), _p(M), _p(ab), _p(vr), shapes.data_ptr(),
                  _p(w), _p(uvw), _p(dloc), _p(dloc2), _p(dw), _p(dw2), _p(draw), _p(dpts), st)
        weight_grad(draw, x1, d_off_w, d_off_b, x2=pos, x2_rows=2 * HLP, group=wg)
        weight_grad(draw, x1, d_aw_w, d_aw_b, x2=pos, x2_rows=HLP, dy_off=2 * HLP, group=wg)
        dpos = new(R, E)
        gemm(R, E, 2 * HLP, _p(draw), (3 * HLP, 1), _p(off_w), (1, E), _p(dx1), E, C2=_p(dpos), flags=ACCUM)
        gemm(R, E, HLP, _p(draw, 2 * HLP), (3 * HLP, 1), _p(aw_w), (1, E), _p(dx1), E, C2=_p(dpos),
             flags=ACCUM | ACCUM2)
        # ---- self attention ----
        dx, dao = new(R, E), new(R, E)
        _ffi.call("demf_add_dropout_ln_bwd", R, E, _p(dx1), None, _p(s1), _p(st1), _p(g1), p_attn, rng,
                  OP_LN1, _p(dx), 0, _p(dao), _p(d_g1), _p(d_b1), st)
        weight_grad(dao, att, d_out_w, d_out_b, group=wg)
        datt = new(R, E)
        gemm(R, E, E, _p(dao), (E, 1), _p(out_w), (1, E), _p(datt), E)
        dqkv = new(R, 3 * E)
        ds = None
        if ctx.core:
            _ffi.call("demf_attn_core_bwd", B, H, Q, Dh, _p(qkv), _p(att), _p(datt), _p(ast), 1.0 / math.sqrt(Dh),
                      p_attn, rng, OP_ATTN, _p(dqkv), st)
        else:
            ds = new(B * H, Q, Q)
            zb, zs, za = (Q * 3 * E, Dh), (H * Q * Q, Q * Q), (Q * E, Dh)
            # dPd = dO v^T ; dv = Pd^T dO
            gemm(Q, Q, Dh, _p(datt), (E, 1), _p(qkv, 2 * E), (3 * E, 1), _p(ds), Q, batch=B * H, zdiv=H,
                 sab=za, sbb=zb, scb=zs)
            gemm(Q, Dh, Q, _p(pd), (1, Q), _p(datt), (1, E), _p(dqkv, 2 * E), 3 * E, batch=B * H, zdiv=H,
                 sab=zs, sbb=za, scb=zb)
            _ffi.call("demf_softmax_dropout_bwd", B * H * Q, Q, _p(prob), p_attn, rng, OP_ATTN, _p(ds), st)
            a = 1.0 / math.sqrt(Dh)
            gemm(Q, Dh, Q, _p(ds), (Q, 1), _p(qkv, E), (1, 3 * E), _p(dqkv), 3 * E, batch=B * H, zdiv=H,
                 sab=zs, sbb=zb, scb=zb, alpha=a)                                    # dq = a dS k
            gemm(Q, Dh, Q, _p(ds), (1, Q), _p(qkv), (1, 3 * E), _p(dqkv, E), 3 * E, batch=B * H, zdiv=H,
                 sab=zs, sbb=zb, scb=zb, alpha=a)                                    # dk = a dS^T q
        weight_grad(dqkv, x, d_in_w, d_in_b, x2=pos, x2_rows=2 * E, group=wg)
        gemm(R, E, 2 * E, _p(dqkv), (3 * E, 1), _p(in_w), (1, E), _p(dx), E, C2=_p(dpos),
             flags=ACCUM | ACCUM2)                                               # q, k see x + pos
        gemm(R, E, E, _p(dqkv, 2 * E), (3 * E, 1), _p(in_w, 2 * E * E), (1, E), _p(dx), E, flags=ACCUM)
        gemm_group(wg)
        if _DEBUG is not None:
            _DEBUG.update(df=df, dh=dh, dx2=dx2, dco=dco, dmo=dmo, dz=dz, dks4=dks4, dloc=dloc, dw=dw,
                          dloc2=dloc2, dw2=dw2, draw=draw, dx1=dx1, dao=dao, datt=datt, ds=ds, dqkv=dqkv,
                          dx=dx, dpos=dpos)
        return (dx, dpos, dpts, None, None, None, None, None, None, None, None, None,
                d_in_w, d_in_b, d_out_w, d_out_b, d_g1, d_b1, d_off_w, d_off_b, d_aw_w, d_aw_b,
                d_vp_w, d_vp_b, d_op_w, d_op_b, d_g2, d_b2, d_f0_w, d_f0_b, d_f1_w, d_f1_b, d_g3, d_b3)
