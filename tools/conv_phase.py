"""Phase-skip timing of demf_conv_nhwc_f32 (DEMF_CV_DBG bits: 1 no MFMA, 2 no B loads, 4 no A loads, 8 no commit,
16 no fragment reads, 32 no stores) on an encoder-linear shape and a 3x3 convolution.  One process per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

out = []
for name, shp, K, N, k in (("lin 148872x256->1024", (1, 24, 6203, 256), 256, 1024, 1),
                           ("3x3 8x100x140x128->128", (8, 100, 140, 128), 128 * 9, 128, 3)):
    x = torch.randn(*shp, device="cuda")
    w = ops.split_planes(torch.randn(N, K, device="cuda") / K ** 0.5, 3)
    b = torch.randn(N, device="cuda")
    y = ops.conv_nhwc(x, w, b, k, k, 1, k // 2, relu=True)
    ms = timed(lambda: ops.conv_nhwc(x, w, b, k, k, 1, k // 2, relu=True, out=y))
    fl = 2.0 * y.numel() * K
    out.append(f"{name}: {ms * 1e3:7.1f} us {fl / ms * 1e-9:6.1f} TF/s")
print("DBG=%s  " % os.environ.get("DEMF_CV_DBG", "0") + "   ".join(out))
