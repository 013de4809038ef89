"""Input-gradient launches of the few-row shared-MLP layers (demf_mlp_gemm_bwd_dx_red_v / _dx_w) at the hot path's shapes,
replayed from a hipGraph: microseconds per launch.  DEMF_DX_TILE=0 / 1 selects mlp_gemm_kernel / mlp_dx_tile_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demf_amd import _ffi
torch.manual_seed(0)
dev = "cuda"
# (name, rows, N = this layer's channels (reduction), K = input channels (output columns), sparse upstream, RED)
SHAPES = [("FP2 L2", 8192, 256, 256, True, True), ("FP2 L1", 8192, 256, 512, False, False), ("vote L2", 8192, 256, 256, True, True),
          ("FP1 L2", 4096, 256, 256, True, True), ("head L2", 2048, 128, 128, True, True), ("head L1", 2048, 128, 256, False, False)]
P = lambda t: None if t is None else t.data_ptr()
for name, R, N, K, sparse, red in SHAPES:
    Y = torch.randn(R, N, device=dev)
    G = torch.randn(R, N, device=dev)
    arg = torch.zeros(R, N, dtype=torch.int32, device=dev)
    vec = torch.cat([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5,
                     torch.randn(N, device=dev) * 0.01, torch.randn(N, device=dev) * 0.01])
    W = torch.randn(N, K, device=dev) / N ** 0.5
    dX = torch.empty(R, K, device=dev)
    Yp = torch.randn(R, K, device=dev)
    ssp = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1])
    mip = torch.cat([torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5])
    g12 = torch.zeros(2 * K, dtype=torch.float64, device=dev)
    gam = torch.ones(K, device=dev)
    v6, dg, db = torch.empty(5 * K, device=dev), torch.empty(K, device=dev), torch.empty(K, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    def go():
        a = (R, N, K, K, None if sparse else P(G), P(G) if sparse else None, P(arg) if sparse else None, 1, P(Y), P(vec), P(W), P(dX))
        if red:
            _ffi.call("demf_mlp_gemm_bwd_dx_red_v", *a, P(Yp), P(ssp), P(mip), P(g12), P(gam), P(v6), P(dg), P(db), st())
        else:
            _ffi.call("demf_mlp_gemm_bwd_dx_w", *a, st())
    go(); torch.cuda.synchronize()
    y = Y.double(); on = (y * vec[:N].double() + vec[N:2 * N].double()) > 0
    dy = vec[2 * N:3 * N].double() * torch.where(on, G.double(), torch.zeros_like(y)) + vec[3 * N:4 * N].double() * y + vec[4 * N:].double()
    want = dy @ W.double()
    err = (dX.double() - want).abs().max().item() / want.abs().max().item()
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
    print("%-8s R=%5d N=%3d K=%3d %s%s: %6.1f us, max err %.1e of scale" % (name, R, N, K, "sparse " if sparse else "dense  ", "RED" if red else "   ",
          e0.elapsed_time(e1) * 1e3 / 100, err))
