"""Does a row chunk that fits the 256 MiB Infinity Cache run the backward kernels faster per row?
SA1 last layer (N=128, K=64, ns=64): weight-gradient and input-gradient launches at the full
R = 8*2048*64 rows (HBM-streamed) vs at R/16 rows repeated on the same buffers (L3-resident)."""
import sys, os, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
from demf_amd import _ffi, ops
N, K, ns = 128, 64, 64
RF = 8 * 2048 * 64
st = torch.cuda.current_stream().cuda_stream
y = torch.randn(RF, N, device="cuda"); xp = torch.randn(RF, K, device="cuda")
dP = torch.randn(RF // ns, N, device="cuda"); arg = torch.randint(0, ns, (RF // ns, N), dtype=torch.int32, device="cuda")
vec = torch.stack([torch.rand(N) + 0.5, torch.randn(N) * 0.3, torch.rand(N) + 0.5, torch.randn(N) * 0.1, torch.randn(N) * 0.1, torch.zeros(N)]).cuda().contiguous()
pss = torch.cat([torch.rand(K) + 0.5, torch.randn(K) * 0.3]).cuda(); mi = torch.cat([torch.randn(K) * 0.1, torch.rand(K) + 0.5]).cuda()
W = torch.randn(N, K, device="cuda") / N ** 0.5
dW = torch.zeros(N, K, device="cuda"); dX = torch.empty(RF, K, device="cuda"); g12 = torch.zeros(2 * K, dtype=torch.float64, device="cuda")
def dw(R, off=0):
    _ffi.call("demf_mlp_gemm_bwd_dw", R, N, K, K, 0, dP[off // ns:].data_ptr(), arg[off // ns:].data_ptr(), ns, y[off:].data_ptr(), vec.data_ptr(), xp[off:].data_ptr(), pss.data_ptr(), dW.data_ptr(), st)
def dx(R, off=0):
    _ffi.call("demf_mlp_gemm_bwd_dx_red", R, N, K, K, 0, dP[off // ns:].data_ptr(), arg[off // ns:].data_ptr(), ns, y[off:].data_ptr(), vec.data_ptr(), W.data_ptr(), dX[off:].data_ptr(), xp[off:].data_ptr(), pss.data_ptr(), mi.data_ptr(), g12.data_ptr(), st)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for mode in ("f32", "bf16"):
    ops.set_compute_dtype(mode)
    print(mode)
    print("  full      dw %.1f us   dx %.1f us   dx+dw %.1f us" % (timeit(lambda: dw(RF)), timeit(lambda: dx(RF)), timeit(lambda: (dx(RF), dw(RF)))))
    for parts in (4, 8, 16, 32):
        Rc = RF // parts
        resident = (timeit(lambda: dw(Rc)), timeit(lambda: dx(Rc)))
        def chunked():
            for c in range(parts):
                dx(Rc, c * Rc); dw(Rc, c * Rc)
        print("  R/%-2d  resident x parts: dw %.1f us  dx %.1f us | chunked dx,dw pairs over all rows: %.1f us" % (parts, resident[0] * parts, resident[1] * parts, timeit(chunked, 5)))
ops.set_compute_dtype("f32")
