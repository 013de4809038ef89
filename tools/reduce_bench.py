import sys, torch
sys.path.insert(0, "/root/repo")
from demf_amd import _ffi
dev = torch.device("cuda:0")
def p(t): return t.data_ptr()
for R, N in ((1048576, 64), (1048576, 128), (262144, 128), (262144, 256)):
    G = torch.randn(R, N, device=dev); Y = torch.randn(R, N, device=dev)
    ss = torch.randn(2 * N, device=dev); mi = torch.rand(2 * N, device=dev) + 0.5
    g12 = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _ffi.call("demf_bn_bwd_reduce", R, N, 1, p(G), None, None, p(Y), None, p(ss), p(mi), p(g12), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _ffi.call("demf_bn_bwd_reduce", R, N, 1, p(G), None, None, p(Y), None, p(ss), p(mi), p(g12), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"R={R} N={N}: {us:.1f} us  {2*R*N*4/us/1e3:.0f} GB/s")
    x = torch.empty(R * N * 2, device=dev)
    e0.record()
    for _ in range(20): x.copy_(torch.cat([G.view(-1), Y.view(-1)]) if False else x)
    e1.record(); torch.cuda.synchronize()
