import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for R, K, N in ((1048576, 64, 64), (1048576, 64, 128), (262144, 128, 128), (262144, 128, 256)):
    X = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Y = torch.empty(R, N, device="cuda")
    ss = torch.rand(2 * K, device="cuda"); stats = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
    res = {}
    for name, pro, stt in (("none", None, None), ("none+stats", None, stats), ("bnrelu", ss, None), ("bnrelu+stats", ss, stats)):
        res[name] = t(lambda: _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, X.data_ptr(), pro.data_ptr() if pro is not None else None, W.data_ptr(), Y.data_ptr(), stt.data_ptr() if stt is not None else None, st))
    cp = t(lambda: Y.copy_(X[:, :N]) if N <= K else Y[:, :K].copy_(X))
    mm = t(lambda: torch.mm(X, W.t(), out=Y))
    byt = R * (K + N) * 4; fl = 2 * R * K * N
    print(f"R={R} K={K} N={N}: " + "  ".join(f"{k} {v:6.1f}us" for k, v in res.items()) + f" | torch.mm {mm:6.1f}us | ideal hbm {byt/5e12*1e6:5.1f}us mfma {fl/157e12*1e6:5.1f}us")
