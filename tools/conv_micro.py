"""Per-convolution timing of ResNet-50 + ChannelMapper on csrc/conv.hip at the bench shape (B x 3 x 800 x 1120):
every launch alone, HIP events, TF/s of the algorithmic (fp32) work.  usage: conv_micro.py [B] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops
from demf_amd.config import BATCH_INPUT_SHAPE
from demf_amd.modules import ImageStream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
ops.set_compute_dtype(mode)
ist = ImageStream().cuda()
img = torch.randn(B, 3, *BATCH_INPUT_SHAPE, device="cuda")
planes = 1 if mode == "bf16" else 3
pk = ist._conv_pack(planes)
rows = []
real = ops.conv_nhwc


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def conv(x, wp, bias, KH, KW, stride=1, pad=0, resid=None, relu=False, out=None, ksplit=None):
    y = real(x, wp, bias, KH, KW, stride, pad, resid=resid, relu=relu)
    ms = timed(lambda: real(x, wp, bias, KH, KW, stride, pad, resid=resid, relu=relu, out=y))
    M = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * M * wp.shape[1] * wp.shape[2]
    rows.append((ms, f"{KH}x{KW} s{stride} {tuple(x.shape[1:])} -> {wp.shape[1]:5d}  rows {M:7d} K {wp.shape[2]:5d}  {ms:7.3f} ms {fl / ms * 1e-9:7.1f} TF/s" +
                 ("  +resid" if resid is not None else "")))
    return y


ops.conv_nhwc = conv
with torch.no_grad():
    t_stem = timed(lambda: ops.conv_stem7(img, pk["stem"]["w"], pk["stem"]["b"]))
    x = ops.conv_stem7(img, pk["stem"]["w"], pk["stem"]["b"])
    t_pool = timed(lambda: ops.maxpool3x3s2_nhwc(x))
    ist._pyramid_tokens(img)
ops.conv_nhwc = real
print(f"stem (nchw->nhwc4 + 7x7 s2): {t_stem:.3f} ms   max-pool: {t_pool:.3f} ms")
tot = 0.0
for ms, line in rows:
    print(line)
    tot += ms
print(f"sum of {len(rows)} convolutions: {tot:.2f} ms")
with torch.no_grad():
    print(f"whole pyramid(): {timed(lambda: ist.pyramid(img)):.2f} ms")

# the encoder's linear layers as 1x1 "convolutions" over (1, 1, R, K) rows: the same kernel, coalesced epilogue
R = B * 18609
print(f"--- encoder linears through demf_conv_nhwc_f32, R = {R}")
for K, N, relu in ((256, 1024, True), (1024, 256, False), (256, 640, False), (256, 256, False)):
    x = torch.randn(1, 24, R // 24, K, device="cuda")
    w = ops.split_planes(torch.randn(N, K, device="cuda") / K ** 0.5, planes)
    b = torch.randn(N, device="cuda")
    y = torch.empty(1, 24, R // 24, N, device="cuda")
    ms = timed(lambda: real(x, w, b, 1, 1, 1, 0, relu=relu, out=y))
    y2 = torch.empty(R, N, device="cuda")
    ms2 = timed(lambda: ops.rows_gemm(x.view(R, K), w, b, y2, relu=relu))
    print(f"K {K:5d} -> N {N:5d}: conv kernel {ms:.3f} ms {2.0 * R * K * N / ms * 1e-9:6.1f} TF/s   rows_gemm {ms2:.3f} ms {2.0 * R * K * N / ms2 * 1e-9:6.1f} TF/s"
          f"   max diff {(y.view(R, N) - y2).abs().max().item():.2e}")
