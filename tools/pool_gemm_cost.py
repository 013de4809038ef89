"""SA1 last layer (R = 1 M rows, 64 -> 128, ns = 64): the pooled GEMM against the same GEMM without the
pooling epilogue and against the separate BN+ReLU+max pass it replaces."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from demf_amd import _ffi
dev = torch.device("cuda:0")
R, K, N, ns = 1 << 20, 64, 128, 64
x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / 8
ss = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1])
Y = torch.empty(R, N, device=dev); stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
pm = torch.empty(2, R // ns, N, device=dev); am = torch.empty(2, R // ns, N, dtype=torch.int32, device=dev)
out = torch.empty(R // ns, N, device=dev); arg = torch.empty(R // ns, N, dtype=torch.int32, device=dev)
ss2 = torch.cat([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1])
def bench(f, n=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (3 * n) * 1e6
st = lambda: torch.cuda.current_stream().cuda_stream
t_pool = bench(lambda: _ffi.call("demf_mlp_gemm_fwd_pool", R, K, N, K, x.data_ptr(), ss.data_ptr(), W.data_ptr(), Y.data_ptr(), stats.data_ptr(), ns, pm[0].data_ptr(), pm[1].data_ptr(), am[0].data_ptr(), am[1].data_ptr(), st()))
t_plain = bench(lambda: _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, x.data_ptr(), ss.data_ptr(), W.data_ptr(), Y.data_ptr(), stats.data_ptr(), st()))
t_nostat = bench(lambda: _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, x.data_ptr(), ss.data_ptr(), W.data_ptr(), Y.data_ptr(), None, st()))
t_mp = bench(lambda: _ffi.call("demf_bnrelu_maxpool_fwd", R // ns, ns, N, Y.data_ptr(), ss2.data_ptr(), out.data_ptr(), arg.data_ptr(), st()))
print(f"pooled GEMM {t_pool:.1f} us | plain GEMM + stats {t_plain:.1f} us | plain GEMM {t_nostat:.1f} us | separate BN+ReLU+max pass {t_mp:.1f} us")
print(f"algorithmic bytes 839 MB: pooled {839e6 / t_pool / 1e6:.2f} TB/s; plain (805 MB) {805e6 / t_plain / 1e6:.2f} TB/s")
