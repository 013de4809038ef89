"""What do the per-block fp64 column-sum atomics at the end of a persistent GEMM cost?
Same launch with and without the statistics epilogue."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from demf_amd import _ffi
dev = torch.device("cuda:0")
for R, K, N in ((1 << 20, 64, 64), (1 << 20, 4, 64), (1 << 18, 128, 128), (1 << 16, 128, 256), (8192, 256, 256)):
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    Y = torch.empty(R, N, device=dev); stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    res = []
    for st_ptr in (None, stats.data_ptr()):
        def f():
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(20):
                _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, x.data_ptr(), None, W.data_ptr(), Y.data_ptr(), st_ptr, s)
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f()
        g.replay(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / 100 * 1e6)
    print(f"R={R:8d} K={K:3d} N={N:3d}: no stats {res[0]:7.1f} us, with stats {res[1]:7.1f} us  (+{res[1]-res[0]:.1f})")
