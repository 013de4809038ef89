"""Times demf_rows_gemm_f32 on the encoder's four linear shapes (R = 8 x 18 609 rows) and checks it against fp64.
DEMF_RG_BIG selects the big-tile forms (csrc/rows_gemm.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops
mode = os.environ.get("MODE", "f32")
ops.set_compute_dtype(mode)
planes = 3 if mode == "f32" else 1
R = 8 * 18609
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(0)
tot = 0.0
for K, N, relu in ((256, 1024, True), (1024, 256, False), (256, 640, False), (256, 256, False)):
    x = torch.randn(R, K, device="cuda")
    wf = torch.randn(N, K, device="cuda") / K ** 0.5
    w = ops.split_planes(wf, planes)
    b = torch.randn(N, device="cuda")
    y = torch.empty(R, N, device="cuda")
    ms = timed(lambda: ops.rows_gemm(x, w, b, y, relu=relu))
    tot += ms
    rows = torch.cat([torch.arange(0, 300, device="cuda"), torch.arange(R - 300, R, device="cuda")])
    ref = x[rows].double() @ wf.double().t() + b.double()
    if relu: ref = ref.clamp_min(0)
    err = (y[rows].double() - ref).abs().max().item()
    print(f"big {os.environ.get('DEMF_RG_BIG', '0')}  K {K:5d} -> N {N:5d}: {ms:.3f} ms {2.0 * R * K * N / ms * 1e-9:6.1f} TF/s   max err vs fp64 {err:.2e}", flush=True)
print(f"big {os.environ.get('DEMF_RG_BIG', '0')}  sum {tot:.3f} ms")
