"""Isolated timing of the strided GEMM (csrc/dense.hip) against hipBLASLt through torch, for the
shapes of the decoder layer; graph-replayed so that host launch cost does not count."""
import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
from demf_amd import fused
from demf_amd.fused import _p
def bench(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for (M, N, K) in [(2048, 256, 256), (2048, 768, 256), (2048, 1024, 256), (2048, 256, 1024), (8192, 256, 256), (16384, 128, 128)]:
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    t_own = bench(lambda: fused.gemm(M, N, K, _p(x), (K, 1), _p(w), (K, 1), _p(y), N, bias=_p(b)))
    t_lib = bench(lambda: torch.addmm(b, x, w.t(), out=y))
    g = torch.randn(M, N, device="cuda"); dw = torch.zeros(N, K, device="cuda")
    t_dw = bench(lambda: fused.gemm(N, K, M, _p(g), (1, N), _p(x), (1, K), _p(dw), K, splitk=fused._splitk(M)))
    t_dwl = bench(lambda: torch.mm(g.t(), x, out=dw))
    print(f"{M}x{N}x{K}: fwd own {t_own:6.1f} us  hipBLASLt {t_lib:6.1f} us | dW own {t_dw:6.1f} us  hipBLASLt {t_dwl:6.1f} us")
