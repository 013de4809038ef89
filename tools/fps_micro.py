"""Furthest point sampling alone (demf_fps_f32) at the backbone's shapes: average launch time and
cycles per dependent round at 2.4 GHz."""
import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
from demf_amd import ops
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for (B, N, M) in [(8, 20000, 2048), (1, 20000, 2048), (8, 16384, 2048), (8, 4096, 1024), (8, 1024, 256)]:
    pts = (torch.rand(B, N, 3, device="cuda") * torch.tensor([6.0, 6.0, 2.5], device="cuda")).contiguous()
    us = t(lambda: ops.furthest_point_sample(pts, M))
    print("B=%d N=%d M=%d: %.1f us = %.0f cycles/round" % (B, N, M, us, us * 1e-6 / (M - 1) * 2.4e9))
