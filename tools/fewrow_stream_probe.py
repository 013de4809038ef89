"""Probe: a few-row linear layer (M rows, 256 -> 256) through the weight-resident streaming 1x1 kernel of
csrc/conv.hip (DEMF_CONV_STREAM_MIN=1) vs the tile kernel (DEMF_CONV_STREAMK=0) - is the barrier-free form faster
at 2 k - 32 k rows, where csrc/mlp.hip's mlp_gemm_kernel takes 12-60 us per layer?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops
ops.set_compute_dtype("f32")
def timed(fn, n=20):
    # (a launch through ctypes costs the host 10-20 us: 20 launches captured in one hipGraph, replayed)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3
for M in (2048, 4096, 8192, 16384, 32768):
    for K, N in ((256, 256), (128, 128)):
        x = torch.randn(1, 32, M // 32, K, device="cuda")
        w = ops.conv_weight_planes(torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5, 3)
        b = torch.randn(N, device="cuda")
        y = torch.empty(1, 32, M // 32, N, device="cuda")
        us = timed(lambda: ops.conv_nhwc(x, w, b, 1, 1, 1, 0, relu=True, out=y))
        print("M %6d  %d -> %d: %6.1f us  %6.1f TF/s" % (M, K, N, us, 2.0 * M * K * N / us * 1e-6), flush=True)
