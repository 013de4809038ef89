"""Times the step's weight-gradient products (the 13 few-row jobs + the vote-aggregation pair traced with
DEMF_DW_TRACE=1) through demf_mlp_gemm_bwd_dw_group: each job alone and all of them as the deferred group.
Phase-skip bits DEMF_DW_DBG need a library built with -DDEMF_DW_PROFILE.  Knobs: DEMF_DW_GRID (row chunks x sub-blocks per job), DEMF_DW_SMALL_R (64 x 64 sub-blocks below it)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
p = lambda t: None if t is None else t.data_ptr()
# (R, N, K, sparse, ns, bnrelu_in, count per step)
JOBS = [(16384, 128, 128, 0, 1, 0, 1), (2048, 128, 128, 1, 1, 1, 2), (2048, 128, 256, 0, 1, 0, 2), (2048, 256, 8, 1, 1, 0, 1),
        (32768, 256, 256, 0, 16, 1, 1), (32768, 256, 256, 1, 16, 1, 1), (4096, 128, 256, 0, 1, 0, 1),
        (4096, 256, 256, 1, 1, 1, 1), (4096, 256, 512, 0, 1, 0, 1), (8192, 128, 256, 0, 1, 0, 1),
        (8192, 256, 256, 0, 1, 0, 2), (8192, 256, 256, 1, 1, 1, 2), (8192, 256, 512, 0, 1, 0, 1)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
keep, jobs, flops = [], [], []
for R, N, K, sparse, ns, bnin, cnt in JOBS:
    for _ in range(cnt):
        Y = torch.randn(R, N, device="cuda"); X = torch.randn(R, K, device="cuda")
        vec = torch.randn(6 * N, device="cuda"); pss = torch.randn(2 * K, device="cuda") if bnin else None
        dW = torch.zeros(N, K, device="cuda")
        if sparse:
            G = None; dP = torch.randn(R // ns, N, device="cuda")
            arg = torch.randint(0, ns, (R // ns, N), device="cuda", dtype=torch.int32)
        else:
            G = torch.randn(R, N, device="cuda"); dP = arg = None
        keep += [Y, X, vec, pss, dW, G, dP, arg]
        jobs.append(_ffi.DwJob(R, N, K, K, p(G), p(dP), p(arg), ns, p(Y), p(vec), p(X), p(pss), p(dW), K))
        flops.append(2.0 * R * N * K)
tot = 0.0
for j, f in zip(jobs, flops):
    us = timeit(lambda: _ffi.call("demf_mlp_gemm_bwd_dw_group", 1, ctypes.addressof(j), st))
    tot += us
    print("R %6d N %4d K %4d sparse %d: %6.1f us  %6.1f TF/s" % (j.R, j.N, j.K, j.G is None, us, f / us * 1e-6), flush=True)
arr = (_ffi.DwJob * len(jobs))(*jobs)
g = timeit(lambda: _ffi.call("demf_mlp_gemm_bwd_dw_group", len(jobs), ctypes.addressof(arr), st))
print("grid %s small_r %s: sum of single launches %.1f us; one grouped call %.1f us (%.1f TF/s)" % (
    os.environ.get("DEMF_DW_GRID", "512"), os.environ.get("DEMF_DW_SMALL_R", "16384"), tot, g, sum(flops) / g * 1e-6))
