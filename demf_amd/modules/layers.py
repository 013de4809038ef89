"""Pointwise conv + BatchNorm + ReLU blocks in point-major ("channels-last") form.

The reference stacks mmcv ConvModule(Conv2d/Conv1d k=1 -> BN -> ReLU) on channel-major
tensors.  Here the same parameters (identical state-dict names: ``conv.weight``,
``bn.weight`` ...) are applied to (rows, C) matrices so a 1x1 conv is a plain GEMM over
rows = B*M*ns and BatchNorm statistics run over dim 0 - numerically the same batch
statistics as BN2d/BN1d over (B, *, spatial).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBNReLU(nn.Module):
    """Parameter-compatible stand-in for mmcv ConvModule(conv -> bn -> relu), k=1.

    ``dim`` picks the holder type (Conv2d+BN2d for SA/FP shared MLPs, Conv1d+BN1d for
    vote / prediction heads) so state-dict shapes match the reference checkpoints.
    """

    def __init__(self, cin, cout, dim=2, bias=False, act=True):
        super().__init__()
        if dim == 2:
            self.conv = nn.Conv2d(cin, cout, kernel_size=1, bias=bias)
            self.bn = nn.BatchNorm2d(cout)
        else:
            self.conv = nn.Conv1d(cin, cout, kernel_size=1, bias=bias)
            self.bn = nn.BatchNorm1d(cout)
        self.act = act
        self.cin, self.cout = cin, cout

    def weight2d(self):
        return self.conv.weight.view(self.cout, self.cin)

    def forward_rows(self, x, weight=None):
        """x (rows, cin[+pad]) -> (rows, cout).  ``weight`` overrides the (cout, K) matrix
        when the caller has permuted / padded the input columns."""
        w = self.weight2d() if weight is None else weight
        y = F.linear(x, w, self.conv.bias)
        bn = self.bn
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                         bn.training or not bn.track_running_stats, bn.momentum, bn.eps)
        return F.relu(y, inplace=True) if self.act else y


class RowsMLP(nn.Sequential):
    """nn.Sequential of ConvBNReLU named layer0, layer1, ... (mmdet3d naming)."""

    def __init__(self, channels, dim=2, bias=False):
        super().__init__()
        for i in range(len(channels) - 1):
            self.add_module(f"layer{i}", ConvBNReLU(channels[i], channels[i + 1], dim, bias))

    def forward_rows(self, x, first_weight=None):
        for i, layer in enumerate(self):
            x = layer.forward_rows(x, first_weight if i == 0 else None)
        return x
