"""dW and dX launches of one small-row layer: back to back on one stream vs side by side on two streams
(how much an "both in one launch" kernel could gain).  Graph-replayed so launch overheads are the graph's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
p = lambda t: None if t is None else t.data_ptr()
SHAPES = [("FP 8192x256x256", 8192, 256, 256), ("AGG 32768x256x256", 32768, 256, 256), ("FP1 4096x256x256", 4096, 256, 256),
          ("FP 8192x256x512", 8192, 256, 512), ("vote 8192x256x256", 8192, 256, 256)]
def graph_time(fn, n=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n / 4 * 1e3
def eager_time(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
side = torch.cuda.Stream()
for name, R, N, K in SHAPES:
    Y = torch.randn(R, N, device="cuda"); Yp = torch.randn(R, K, device="cuda"); G = torch.randn(R, N, device="cuda")
    vec = torch.randn(5 * N, device="cuda"); W = torch.randn(N, K, device="cuda") / 8
    pss = torch.randn(2 * K, device="cuda"); pmi = torch.rand(2 * K, device="cuda") + 0.5
    dX = torch.empty(R, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
    g12 = torch.zeros(2 * K, dtype=torch.float64, device="cuda")
    def dw(st):
        _ffi.call("demf_mlp_gemm_bwd_dw", R, N, K, K, p(G), None, None, 1, p(Y), p(vec), p(Yp), p(pss), p(dW), st)
    def dx(st):
        _ffi.call("demf_mlp_gemm_bwd_dx_red", R, N, K, K, p(G), None, None, 1, p(Y), p(vec), p(W), p(dX), p(Yp), p(pss), p(pmi), p(g12), st)
    cur = lambda: torch.cuda.current_stream().cuda_stream
    t_dw = graph_time(lambda: dw(cur())); t_dx = graph_time(lambda: dx(cur()))
    t_seq = graph_time(lambda: (dw(cur()), dx(cur())))
    def both():
        side.wait_stream(torch.cuda.current_stream())
        dw(side.cuda_stream); dx(cur())
        torch.cuda.current_stream().wait_stream(side)
    t_par = eager_time(both); t_seq_e = eager_time(lambda: (dw(cur()), dx(cur())))
    print(f"{name}: dW {t_dw:6.1f}  dX {t_dx:6.1f}  seq(graph) {t_seq:6.1f}  seq(eager) {t_seq_e:6.1f}  two streams(eager) {t_par:6.1f} us", flush=True)
