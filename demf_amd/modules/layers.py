"""Pointwise conv + BatchNorm + ReLU blocks in point-major ("channels-last") form.

The reference stacks mmcv ConvModule(Conv2d/Conv1d k=1 -> BN -> ReLU) on channel-major
tensors.  Here the same parameters (identical state-dict names: ``conv.weight``,
``bn.weight`` ...) are applied to (rows, C) matrices by the fused gfx950 kernels of
csrc/mlp.hip: a 1x1 conv is an fp32-MFMA GEMM over rows = B*M*ns whose epilogue reduces the
BatchNorm batch statistics (over dim 0 - the same statistics as BN2d/BN1d over
(B, *, spatial)) and whose prologue applies the previous layer's BN + ReLU.
"""
import torch.nn as nn


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


def fused_rows(blocks, x, ns=1, first_weight=None, geo=None):
    """Run a stack of ConvBNReLU blocks on rows x (R, K) through the fused gfx950 kernels
    (ops.shared_mlp_pool): GEMM + BN statistics + BN/ReLU, optionally max over ``ns`` rows."""
    from .. import ops
    layers = []
    training = blocks[0].bn.training
    for i, blk in enumerate(blocks):
        bn = blk.bn
        w = first_weight if (i == 0 and first_weight is not None) else blk.weight2d()
        layers.append((w.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                       blk.conv.bias, bn.num_batches_tracked))
    bn0 = blocks[0].bn
    return ops.shared_mlp_pool(x, ns, layers, training=training, eps=bn0.eps, momentum=bn0.momentum,
                               geo=geo)


class RowsMLP(nn.Sequential):
    """nn.Sequential of ConvBNReLU named layer0, layer1, ... (mmdet3d naming)."""

    def __init__(self, channels, dim=2, bias=False):
        super().__init__()
        for i in range(len(channels) - 1):
            self.add_module(f"layer{i}", ConvBNReLU(channels[i], channels[i + 1], dim, bias))

    def forward_rows(self, x, first_weight=None, ns=1, geo=None):
        """x (R, K) -> (R/ns, C_out): the whole stack (+ max over ns neighbours) fused.
        ``geo``: see ops.shared_mlp_pool (first layer applied per source point)."""
        return fused_rows(list(self), x.contiguous(), ns, first_weight, geo)
