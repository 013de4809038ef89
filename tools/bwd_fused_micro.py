"""Times demf_mlp_bwd_fused alone at the SA shapes (synthetic operands), next to the two launches it
replaces.  DEMF_BWDF_DBG phase-skip bits: 1 transform+LDS writes, 2 prefetch, 4 dX MFMAs, 8 ds_add,
16 dW MFMAs, 32 epilogue, 64 dW flush."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
p = lambda t: None if t is None else t.data_ptr()
SHAPES = [("SA2.L3", 262144, 256, 128, 32), ("SA1.L3", 1048576, 128, 64, 64), ("SA1.L2", 1048576, 64, 64, 0), ("SA2.L2", 262144, 128, 128, 0),
          ("AGG.L3", 32768, 128, 128, 16)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for name, R, N, K, ns in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    Y = torch.randn(R, N, device="cuda"); Yp = torch.randn(R, K, device="cuda")
    G = torch.randn(R, N, device="cuda") if ns == 0 else None
    dP = torch.randn(R // ns, N, device="cuda") if ns else None
    arg = torch.randint(0, ns, (R // ns, N), device="cuda", dtype=torch.int32) if ns else None
    vec = torch.randn(5 * N, device="cuda"); W = torch.randn(N, K, device="cuda") / 8
    pss = torch.randn(2 * K, device="cuda"); pmi = torch.rand(2 * K, device="cuda") + 0.5
    dX = torch.empty(R, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
    g12 = torch.zeros(2 * K, dtype=torch.float64, device="cuda")
    first = name == "SA1.L2"
    X0 = torch.randn(R, 4, device="cuda") if first else None
    fs = torch.zeros(10 * K + 4, dtype=torch.float64, device="cuda") if first else None
    def fused():
        _ffi.call("demf_mlp_bwd_fused", R, N, K, p(G), p(dP), p(arg), max(ns, 1), p(Y), p(vec), p(W), p(Yp), p(pss), p(pmi),
                  None if first else p(dX), p(dW), None if first else p(g12), p(X0), p(fs), None, None, None, None, 0, st)
    def two():
        _ffi.call("demf_mlp_gemm_bwd_dw", R, N, K, K, p(G), p(dP), p(arg), max(ns, 1), p(Y), p(vec), p(Yp), p(pss), p(dW), st)
        if first:
            _ffi.call("demf_mlp_gemm_bwd_dx_first", R, N, K, p(G), p(Y), p(vec), p(W), p(X0), p(Yp), p(pss), p(pmi), p(fs), st)
        else:
            _ffi.call("demf_mlp_gemm_bwd_dx_red", R, N, K, K, p(G), p(dP), p(arg), max(ns, 1), p(Y), p(vec), p(W), p(dX), p(Yp), p(pss), p(pmi), p(g12), st)
    tf = timeit(fused)
    t2 = timeit(two) if not os.environ.get("DEMF_BWDF_DBG") else float("nan")
    byt = 4 * R * (N + (N if ns == 0 else 0) + K + (0 if first else K))
    print(f"{name}: fused {tf:7.1f} us ({byt/tf/1e6:5.0f} GB/s algo)   two launches {t2:7.1f} us", flush=True)
