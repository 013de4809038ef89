"""Accuracy and speed of the three compute modes on one shared-MLP launch (SA1's last layer by
default): max / rms error of Y against an fp64 product, and the average launch time.
usage: python tools/split_micro.py [R K N ns]"""
import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
from demf_amd import _ffi, ops
a = [int(v) for v in sys.argv[1:5]]
R, K, N, ns = a if len(a) == 4 else (8 * 2048 * 64, 64, 128, 64)
torch.manual_seed(0)
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
pro = torch.cat([sc, sh]).contiguous()
y = torch.empty(R, N, device="cuda"); stats = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
pm = torch.empty(2, R // ns, N, device="cuda"); am = torch.empty(2, R // ns, N, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    _ffi.call("demf_mlp_gemm_fwd_pool", R, K, N, K, x.data_ptr(), pro.data_ptr(), w.data_ptr(), y.data_ptr(),
              stats.data_ptr(), ns, pm[0].data_ptr(), pm[1].data_ptr(), am[0].data_ptr(), am[1].data_ptr(), st)
rows = min(R, 1 << 16)
a64 = torch.relu(x[:rows] * sc + sh).double()          # the fp32 prologue, then exact products
ref = a64 @ w.double().t()
for mode in ("f32", "f32x3", "bf16"):
    ops.set_compute_dtype(mode)
    for _ in range(3): run()
    torch.cuda.synchronize()
    err = (y[:rows].double() - ref)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print("%-6s max err %.3e  rms err %.3e  (rms ref %.3f)   avg %.1f us" % (
        mode, err.abs().max().item(), err.pow(2).mean().sqrt().item(), ref.pow(2).mean().sqrt().item(),
        s.elapsed_time(e) * 1e3 / 20))
ops.set_compute_dtype("f32")
