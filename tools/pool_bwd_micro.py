"""Times demf_mlp_bwd_pool alone (SA1's last layer, synthetic operands) next to demf_mlp_bwd_fused on the
stored output.  DEMF_PB_DBG phase-skip bits: 1 transform, 2 MFMAs, 4 fold rounds, 8 dX store, 16 sparse S,
32 prologue M."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
p = lambda t: None if t is None else t.data_ptr()
R, N, K, ns = 1048576, 128, 64, 64
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
Y = torch.randn(R, N, device="cuda"); Yp = torch.randn(R, K, device="cuda")
dP = torch.randn(R // ns, N, device="cuda"); arg = torch.randint(0, ns, (R // ns, N), device="cuda", dtype=torch.int32)
yraw = torch.randn(R // ns, N, device="cuda")
vec = torch.randn(5 * N, device="cuda"); W = torch.randn(N, K, device="cuda") / 8
pss = torch.randn(2 * K, device="cuda"); pmi = torch.rand(2 * K, device="cuda") + 0.5
dX = torch.empty(R, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
g12 = torch.zeros(2 * K, dtype=torch.float64, device="cuda")
nws = ctypes.c_longlong(); _ffi.call("demf_mlp_bwd_pool_ws", R, ctypes.addressof(nws))
ws = torch.empty(nws.value, device="cuda"); cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
def pool():
    _ffi.call("demf_mlp_bwd_pool", R, N, K, ns, p(dP), p(arg), p(yraw), p(vec), p(W), p(Yp), p(pss), p(pmi), p(dX), p(dW), p(g12),
              None, None, None, None, p(ws), st)
def fused():
    _ffi.call("demf_mlp_bwd_fused", R, N, K, None, p(dP), p(arg), ns, p(Y), p(vec), p(W), p(Yp), p(pss), p(pmi),
              p(dX), p(dW), p(g12), None, None, None, None, None, None, 0, st)
print("dbg %s: pool %.1f us   fused (stored Y) %.1f us" % (os.environ.get("DEMF_PB_DBG", "0"), timeit(pool), timeit(fused)), flush=True)
