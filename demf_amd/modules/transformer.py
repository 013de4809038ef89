"""The DeMF fusion decoder layer.

``DeMFTransformerDecoderLayer`` / ``PositionEmbeddingLearned`` mirror
demf/modeling/layers/transformer.py:18-80.  The mmcv layer it wraps
(DetrTransformerDecoderLayer = BaseTransformerLayer with operation_order
('self_attn','norm','cross_attn','norm','ffn','norm'), configs/demf/demf_votenet.py:71-91)
is restated here with identical sub-module names so reference checkpoints load:
``layer.attentions.0.attn.*`` (nn.MultiheadAttention), ``layer.attentions.1.{sampling_offsets,
attention_weights,value_proj,output_proj}``, ``layer.ffns.0.layers.{0.0,1}``, ``layer.norms.{0,1,2}``.
The deformable sampling itself runs on the gfx950 kernel (ops.MultiScaleDeformableAttnFunction).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import MultiScaleDeformableAttnFunction


class PositionEmbeddingLearned(nn.Module):
    """transformer.py:18-36 - Conv1d(6->256) BN ReLU Conv1d(256->256) on (B,N,6)."""

    def __init__(self, cfg):
        super().__init__()
        cin, cpos = cfg["input_channel"], cfg["num_pos_feats"]
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(cin, cpos, kernel_size=1), nn.BatchNorm1d(cpos), nn.ReLU(inplace=True),
            nn.Conv1d(cpos, cpos, kernel_size=1))

    def forward(self, xyz):
        """(B,N,cin) -> (B,cpos,N).  The two k=1 Conv1d run as row GEMMs (MIOpen's conv
        path picks naive kernels for these shapes); parameters stay in the Conv1d/BN1d
        holders so checkpoints load unchanged."""
        c0, bn, _, c1 = self.position_embedding_head
        B, N, cin = xyz.shape
        x = ops.linear(xyz.reshape(B * N, cin), c0.weight.view(c0.out_channels, cin), c0.bias)
        if bn.training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        x = F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                                bn.training, bn.momentum, bn.eps), inplace=True)
        x = ops.linear(x, c1.weight.view(c1.out_channels, -1), c1.bias)
        return x.view(B, N, -1).transpose(1, 2)

    def forward_rows(self, rows):
        """(R,cin) -> (R,cpos) on the fused kernels: Conv1d+BN1d(train stats)+ReLU through the
        shared-MLP GEMM (input rows zero-padded to a multiple of 4 columns), the second Conv1d through
        the strided GEMM of csrc/dense.hip."""
        c0, bn, _, c1 = self.position_embedding_head
        cin = c0.in_channels
        pad = (-cin) % 4
        # (rows that already carry the zero columns - the head concatenates them in - skip the pad launch)
        x = rows.contiguous() if rows.shape[1] == cin + pad else F.pad(rows, (0, pad))
        R = x.shape[0]
        w0 = F.pad(c0.weight.view(c0.out_channels, cin), (0, pad)) if pad else c0.weight.view(c0.out_channels, cin)
        h = ops.shared_mlp_pool(x, 1, [(w0.contiguous(), bn.weight, bn.bias, bn.running_mean,
                                        bn.running_var, c0.bias, bn.num_batches_tracked)],
                                training=bn.training, eps=bn.eps, momentum=bn.momentum)
        return ops.linear(h, c1.weight.view(c1.out_channels, -1), c1.bias)


class MultiheadAttention(nn.Module):
    """mmcv.cnn.bricks.transformer.MultiheadAttention (seq-first).  The deprecated
    ``dropout`` kwarg of the reference config sets BOTH the attention-weight dropout and
    the output dropout layer, as upstream does."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout=None, **unused):
        super().__init__()
        out_drop = 0.0
        if dropout is not None:
            attn_drop, out_drop = dropout, dropout
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(out_drop) if out_drop > 0 else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        shared_qk = key is query and key_pos is query_pos      # self-attention: one q|k projection
        if query_pos is not None:
            query = query + query_pos
        if shared_qk:
            key = query
        elif key_pos is not None:
            key = key + key_pos
        out = self._attend(query, key, value, attn_mask, key_padding_mask)
        return identity + self.dropout_layer(self.proj_drop(out))


    def _attend(self, query, key, value, attn_mask, key_padding_mask):
        """torch.nn.MultiheadAttention's math (seq-first, packed in_proj) spelled out on
        ops.linear; ``self.attn`` stays the parameter holder so state-dict keys are upstream's."""
        a = self.attn
        L, B, E = query.shape
        S, H = key.shape[0], a.num_heads
        Dh = E // H
        W, b = a.in_proj_weight, a.in_proj_bias
        if key is query:
            qk = ops.linear(query, W[:2 * E], None if b is None else b[:2 * E])
            q, k = qk[..., :E], qk[..., E:]
        else:
            q = ops.linear(query, W[:E], None if b is None else b[:E])
            k = ops.linear(key, W[E:2 * E], None if b is None else b[E:2 * E])
        v = ops.linear(value, W[2 * E:], None if b is None else b[2 * E:])
        q = q.reshape(L, B * H, Dh).transpose(0, 1)
        k = k.reshape(S, B * H, Dh).transpose(0, 1)
        v = v.reshape(S, B * H, Dh).transpose(0, 1)
        if q.is_cuda:
            # (the device path of the head runs the layer as one fused node - demf_amd/fused.py - and never gets here)
            ops.library_fallback("MultiheadAttention module path (torch.bmm for QK^T and PV)")
        scores = torch.bmm(q * (1.0 / math.sqrt(Dh)), k.transpose(1, 2))        # (B*H, L, S)
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask, float("-inf")) if attn_mask.dtype == torch.bool \
                else scores + attn_mask
        if key_padding_mask is not None:
            scores = scores.view(B, H, L, S).masked_fill(key_padding_mask[:, None, None, :],
                                                         float("-inf")).view(B * H, L, S)
        attn = F.dropout(scores.softmax(dim=-1), a.dropout, self.training)
        out = torch.bmm(attn, v).transpose(0, 1).reshape(L, B, E)
        return ops.linear(out, a.out_proj.weight, a.out_proj.bias)


class MultiScaleDeformableAttention(nn.Module):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention (transformer.py:8-15
    imports it; built from demf_votenet.py:79-85).  forward signature as upstream."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, **unused):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.im2col_step, self.embed_dims = im2col_step, embed_dims
        self.num_levels, self.num_heads, self.num_points = num_levels, num_heads, num_points
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid.view(-1)
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def project_value(self, value, key_padding_mask=None):
        """value (S,B,C) [batch_first: (B,S,C)] -> masked, projected (B,S,H,Dh).  Depends only on
        the image tokens, so the head can run it ahead of time on a side stream."""
        if not self.batch_first:
            value = value.permute(1, 0, 2)
        bs, num_value, _ = value.shape
        value = ops.linear(value, self.value_proj.weight, self.value_proj.bias,
                           row_mask=key_padding_mask)
        return value.view(bs, num_value, self.num_heads, -1)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, value_projected=None, value_tokens=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        if value_tokens is None:
            value = value_projected if value_projected is not None else \
                self.project_value(value, key_padding_mask)
        offsets = ops.linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(bs, num_query, self.num_heads, self.num_levels,
                                                    self.num_points, 2)
        weights = ops.linear(query, self.attention_weights.weight, self.attention_weights.bias).view(bs, num_query, self.num_heads,
                                                     self.num_levels * self.num_points)
        weights = weights.softmax(-1).view(bs, num_query, self.num_heads, self.num_levels,
                                           self.num_points)
        assert reference_points.shape[-1] == 2, "the DeMF path passes 2-d reference points"
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        locations = reference_points[:, :, None, :, None, :] + \
            offsets / normalizer[None, None, None, :, None, :]
        if value_tokens is not None:
            # few queries against many tokens: sample the (masked) tokens first, project the
            # sampled rows afterwards (ops.msda_sample_then_project) - same result, no (B,S,C) GEMM
            tokens, keep4 = value_tokens
            output = ops.msda_sample_then_project(
                tokens, keep4, spatial_shapes, level_start_index, locations.contiguous(),
                weights.contiguous(), self.value_proj.weight, self.value_proj.bias)
        else:
            output = MultiScaleDeformableAttnFunction.apply(
                value.contiguous(), spatial_shapes, level_start_index, locations.contiguous(),
                weights.contiguous(), self.im2col_step)
        output = ops.linear(output, self.output_proj.weight, self.output_proj.bias)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return self.dropout(output) + identity


class FFN(nn.Module):
    """mmcv FFN(embed_dims, feedforward_channels, num_fcs=2, ReLU, ffn_drop, add_identity)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, ffn_drop=0.0, **unused):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True),
                          nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        (fc0, _, drop0), fc1, drop1 = self.layers
        h = drop0(F.relu(ops.linear(x, fc0.weight, fc0.bias)))
        out = drop1(ops.linear(h, fc1.weight, fc1.bias))
        return (x if identity is None else identity) + out


class DetrTransformerDecoderLayer(nn.Module):
    """BaseTransformerLayer for operation_order ('self_attn','norm','cross_attn','norm',
    'ffn','norm') - the only order the reference config uses (demf_votenet.py:89-90)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=2, attn_dropout=0.4,
                 feedforward_channels=1024, ffn_dropout=0.1):
        super().__init__()
        self.attentions = nn.ModuleList([
            MultiheadAttention(embed_dims, num_heads, dropout=attn_dropout),
            MultiScaleDeformableAttention(embed_dims, num_heads, num_levels, num_points,
                                          dropout=attn_dropout)])
        self.ffns = nn.ModuleList([FFN(embed_dims, feedforward_channels, ffn_dropout)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims) for _ in range(3)])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                key_padding_mask=None, **kwargs):
        query = self.attentions[0](query, query, query, None, query_pos=query_pos,
                                   key_pos=query_pos, **kwargs)
        query = self.norms[0](query)
        query = self.attentions[1](query, key, value, None, query_pos=query_pos,
                                   key_pos=key_pos, key_padding_mask=key_padding_mask, **kwargs)
        query = self.norms[1](query)
        query = self.ffns[0](query, None)
        return self.norms[2](query)


class DeMFTransformerDecoderLayer(nn.Module):
    """transformer.py:39-80.  ``transformerlayers``/``posembed`` are the reference's cfg
    dicts (only the keys the DeMF config sets are honoured)."""

    def __init__(self, *args, transformerlayers=None, posembed=None, **kwargs):
        super().__init__()
        t = dict(transformerlayers or {})
        attn = t.get("attn_cfgs", [{}, {}])
        self.layer = DetrTransformerDecoderLayer(
            embed_dims=attn[1].get("embed_dims", 256), num_heads=attn[1].get("num_heads", 8),
            num_levels=attn[1].get("num_levels", 4), num_points=attn[1].get("num_points", 4),
            attn_dropout=attn[1].get("dropout", 0.1),
            feedforward_channels=t.get("feedforward_channels", 1024),
            ffn_dropout=t.get("ffn_dropout", 0.0))
        self.posembed = PositionEmbeddingLearned(posembed)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()

    def forward_rows(self, x, query_pos, points, value_tokens, spatial_shapes, level_start_index,
                     proj, valid_ratios, batch, layer_index=0):
        """The layer on batch-major rows through ONE autograd node (demf_amd/fused.py):
        x (B*Q, E) query rows, query_pos (B*Q, 6), points (B*Q, 3) the 3-D query points (projected
        into the image inside the sampling-location kernel: get_reference_points,
        class_agnostic_vote_head.py:524-547, with ``proj`` = (M (B,4,4), ab (B,4)) composed on the
        host), value_tokens = (tokens (B,S,C) padding-zeroed, keep4 (B,S,4)).  -> (B*Q, E).
        Same mathematics as ``forward`` (self-attn, norm, deformable cross-attn with the value
        projection applied after sampling, norm, FFN, norm).  ``layer_index`` salts the dropout
        streams so that stacked decoder layers draw independent masks (as nn.Dropout does)."""
        from ..fused import FusedDecoderLayer
        lyr = self.layer
        mha, msda, ffn = lyr.attentions[0], lyr.attentions[1], lyr.ffns[0]
        a = mha.attn
        (fc0, _, drop0), fc1, _ = ffn.layers
        n1, n2, n3 = lyr.norms
        assert isinstance(mha.proj_drop, nn.Dropout) and mha.proj_drop.p == 0.0
        p_attn = float(a.dropout)
        assert (not isinstance(mha.dropout_layer, nn.Dropout) and p_attn == 0.0) or \
            mha.dropout_layer.p == p_attn == msda.dropout.p, "one attention dropout rate (cfg:78,84)"
        pos = self.posembed.forward_rows(query_pos)
        tokens, keep4 = value_tokens
        M, ab = proj
        Q = x.shape[0] // batch
        dims = (batch, Q, msda.num_heads, msda.num_levels, msda.num_points, p_attn, float(drop0.p),
                float(n1.eps), int(layer_index))
        return FusedDecoderLayer.apply(
            x.contiguous(), pos, points.contiguous(), tokens, keep4, spatial_shapes, level_start_index,
            M, ab, valid_ratios, dims, self.training,
            a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, n1.weight, n1.bias,
            msda.sampling_offsets.weight, msda.sampling_offsets.bias, msda.attention_weights.weight,
            msda.attention_weights.bias, msda.value_proj.weight, msda.value_proj.bias,
            msda.output_proj.weight, msda.output_proj.bias, n2.weight, n2.bias,
            fc0.weight, fc0.bias, fc1.weight, fc1.bias, n3.weight, n3.bias)

    def forward(self, query, query_pos, *args, reference_points=None, valid_ratios=None, **kwargs):
        if reference_points.shape[-1] == 4:
            reference_points_input = reference_points[:, :, None] * \
                torch.cat([valid_ratios, valid_ratios], -1)[:, None]
        else:
            assert reference_points.shape[-1] == 2
            reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
        query_pos_embed = self.posembed(query_pos).permute(2, 0, 1)
        return self.layer(query, *args, query_pos=query_pos_embed,
                          reference_points=reference_points_input, **kwargs)
