"""Plain GEMMs of the decoder's sizes: demf_gemm_f32 (gemm_ks_kernel: 32 x 32 tiles, K split over the waves) against the
64 x 64-tile form of the shared-MLP forward (demf_mlp_gemm_fwd_bn -> mlp_fwd_tile_kernel, which also takes the column
statistics), replayed from a hipGraph: microseconds per launch.  Decides whether dense.hip wants a tile-form kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demf_amd import _ffi, ops, fused
torch.manual_seed(0)
dev = "cuda"
SHAPES = [("FFN1", 2048, 256, 1024), ("in-proj", 2048, 256, 768), ("proj", 2048, 256, 256), ("FFN2", 2048, 1024, 256),
          ("value", 8192, 256, 256), ("wide", 8192, 256, 512), ("k512", 2048, 512, 256)]
P = lambda t: None if t is None else t.data_ptr()

def timeit(go):
    go(); torch.cuda.synchronize()
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        for _ in range(3): go()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            for _ in range(20): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gph.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): gph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 100

for name, R, K, N in SHAPES:
    x = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(R, N, device=dev)
    y2 = torch.empty(R, N, device=dev)
    stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    ss, mi = torch.empty(2 * N, device=dev), torch.empty(2 * N, device=dev)
    t_ks = timeit(lambda: fused.gemm(R, N, K, P(x), (K, 1), P(w), (K, 1), P(y), N))
    try:
        t_tile = timeit(lambda: _ffi.call("demf_mlp_gemm_fwd_bn", R, K, N, K, P(x), None, P(w), P(y2), P(stats), P(g), P(b), 1e-5,
                                          0.1, P(rm), P(rv), None, P(ss), P(mi), None, torch.cuda.current_stream().cuda_stream))
        err = (y - y2).abs().max().item()
    except Exception as e:
        t_tile, err = float("nan"), str(e)[:60]
    print("%-8s M=%5d K=%4d N=%4d: gemm_f32 %6.1f us (%5.1f TF/s)   mlp tile form %6.1f us   |diff| %s"
          % (name, R, K, N, t_ks, 2.0 * R * K * N / t_ks * 1e-6, t_tile, err))
