"""Forward of the few-row shared-MLP layers (demf_mlp_gemm_fwd_bn: GEMM + BN statistics + bookkeeping) at the hot path's
shapes, replayed from a hipGraph: microseconds per launch.  DEMF_FWD_TILE=0 / 1 selects mlp_gemm_kernel / mlp_fwd_tile_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demf_amd import _ffi, ops
torch.manual_seed(0)
dev = "cuda"
SHAPES = [("FP1 L1", 4096, 512, 256, False), ("FP1 L2", 4096, 256, 256, True), ("FP2 L1", 8192, 512, 256, False),
          ("FP2 L2", 8192, 256, 256, True), ("vote L1", 8192, 256, 256, False), ("agg L2", 32768, 256, 256, True),
          ("head L1", 2048, 256, 128, False), ("head L2", 2048, 128, 128, True)]
for name, R, K, N, pro in SHAPES:
    x = torch.randn(R, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(R, N, device=dev)
    vec = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1]) if pro else None
    stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    ss, mi = torch.empty(2 * N, device=dev), torch.empty(2 * N, device=dev)
    P = lambda t: None if t is None else t.data_ptr()
    def go():
        _ffi.call("demf_mlp_gemm_fwd_bn", R, K, N, K, P(x), P(vec), P(w), P(y), P(stats), P(g), P(b), 1e-5, 0.1, P(rm), P(rv),
                  None, P(ss), P(mi), None, torch.cuda.current_stream().cuda_stream)
    go(); torch.cuda.synchronize()
    # reference (fp64)
    a = x.double()
    if pro:
        a = torch.relu(a * vec[:K].double() + vec[K:].double())
    want = a @ w.double().t()
    err = (y.double() - want).abs().max().item() / want.abs().max().item()
    mean_err = (mi[:N].double() - want.mean(0)).abs().max().item()
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
    us = e0.elapsed_time(e1) * 1e3 / 100
    print("%-8s R=%5d K=%3d N=%3d: %6.1f us  (%5.1f TF/s fp32-equivalent)  max err %.1e of scale, mean err %.1e"
          % (name, R, K, N, us, 2.0 * R * K * N / us * 1e-6, err, mean_err))
