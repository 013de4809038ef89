"""Times demf_gemm_f32 on the decoder layer's projection shapes (rows x out x in), split-K variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import fused, ops
from demf_amd.fused import _p
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (R, N, K) in [(2048, 256, 256), (2048, 768, 256), (2048, 1024, 256), (2048, 256, 1024), (2048, 128, 256), (2048, 64, 256)]:
    x, w, b = torch.randn(R, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    y = torch.zeros(R, N, device="cuda")
    row = []
    for sk in (1, 2, 4):
        f = lambda: fused.gemm(R, N, K, _p(x), (K, 1), _p(w), (K, 1), _p(y), N, bias=_p(b), splitk=sk)
        try:
            row.append("splitk %d: %.1f us" % (sk, timeit(f)))
        except Exception as e:
            row.append("splitk %d: %s" % (sk, str(e)[:40]))
    # dx form: dy (R,N) . W (N,K) -> (R,K)
    dx = torch.empty(R, K, device="cuda")
    g = lambda: fused.gemm(R, K, N, _p(y), (N, 1), _p(w), (1, K), _p(dx), K)
    row.append("dx %.1f us" % timeit(g))
    print((R, N, K), "; ".join(row), flush=True)
