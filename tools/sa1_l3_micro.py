"""The dominant kernel alone: demf_mlp_gemm_fwd_pool at SA1's last layer (R = 8*2048*64 rows, 64 -> 128,
ns = 64), N launches back to back; DEMF_FWD_RES=0/1 selects the register-staged / weight-resident kernel,
DEMF_MODE=f32|f32_native|f32x3|bf16 the compute mode.
Used under rocprofv3 --pmc for pipe-utilisation counters."""
import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("DEMF_MODE", "f32"))
R, K, N, ns = 8 * 2048 * 64, 64, 128, 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 8
pro = torch.cat([torch.ones(K), torch.zeros(K)]).cuda()
y = torch.empty(R, N, device="cuda"); stats = torch.zeros(2 * N, dtype=torch.float64, device="cuda")
pm = torch.empty(2, R // ns, N, device="cuda"); am = torch.empty(2, R // ns, N, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
gamma = torch.ones(N, device="cuda"); gamma[::3] = -1.0
beta = torch.zeros(N, device="cuda"); rm = torch.zeros(N, device="cuda"); rv = torch.ones(N, device="cuda")
ss = torch.empty(2 * N, device="cuda"); mi = torch.empty(2 * N, device="cuda")
VARIANT = os.environ.get("DEMF_VARIANT", "pool")        # pool | pool_bn (4 outputs + finalize) | sel
def run():
    if VARIANT == "pool":
        _ffi.call("demf_mlp_gemm_fwd_pool", R, K, N, K, x.data_ptr(), pro.data_ptr(), w.data_ptr(), y.data_ptr(),
                  stats.data_ptr(), ns, pm[0].data_ptr(), pm[1].data_ptr(), am[0].data_ptr(), am[1].data_ptr(), st)
    else:
        sel = VARIANT == "sel"
        _ffi.call("demf_mlp_gemm_fwd_pool_bn", R, K, N, K, x.data_ptr(), pro.data_ptr(), w.data_ptr(), y.data_ptr(),
                  stats.data_ptr(), ns, pm[0].data_ptr(), 0 if sel else pm[1].data_ptr(), am[0].data_ptr(),
                  0 if sel else am[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(),
                  rv.data_ptr(), 0, ss.data_ptr(), mi.data_ptr(), 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n): run()
e.record(); torch.cuda.synchronize()
print("DEMF_MODE=%s DEMF_VARIANT=%s DEMF_FWD_RES=%s  avg %.1f us" % (os.environ.get("DEMF_MODE", "f32"), VARIANT, os.environ.get("DEMF_FWD_RES", "1"), s.elapsed_time(e) * 1e3 / n))
