"""What makes the main queue slower while the furthest-point chain is resident?  A graph of 4000
tiny kernels (and one of 60 large persistent GEMMs) is replayed while a second stream holds
 (a) nothing, (b) the real FPS launch, (c) one spinning workgroup with a small footprint,
 (d) one spinning workgroup that claims its CU's whole register file (as the FPS kernel does),
 (e) 8 of (d)."""
import ctypes, os, subprocess, sys, time, torch
sys.path.insert(0, "/root/repo")
from demf_amd import ops, _ffi
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libspin.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so,
                os.path.join(here, "src", "spin_kernel.hip")], check=True)
lib = ctypes.CDLL(so)
lib.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
pts = (torch.rand(8, 20000, 3, device=dev) * 6).contiguous()
side = torch.cuda.Stream()
TICKS = int(2.7e-3 * 100e6)     # wall_clock64 runs at 100 MHz

def spin(blocks, threads, fat, ms=2.7):
    def f():
        lib.spin_launch(blocks, threads, fat, int(ms * 1e-3 * 100e6), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return f

SIDES = [("nothing", None), ("FPS 1 scene", lambda: ops.furthest_point_sample(pts[:1], 2048)),
         ("spin 1 WG x 64 thr, small", spin(1, 64, 0)), ("spin 1 WG x 1024 thr, small", spin(1, 1024, 0)),
         ("spin 1 WG x 1024 thr, whole register file", spin(1, 1024, 1)),
         ("spin 8 WG x 1024 thr, whole register file", spin(8, 1024, 1)),
         ("spin 1 WG x 64 thr, small, 1 ms", spin(1, 64, 0, 1.0)),
         ("spin 1 WG x 64 thr, small, 5 ms", spin(1, 64, 0, 5.0))]

a = torch.randn(2048, 256, device=dev); b = torch.randn(2048, 256, device=dev)
def tiny():
    c = a
    for _ in range(4000):
        c = c + b
R, K, N = 1 << 20, 64, 64
x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / 8
Y = torch.empty(R, N, device=dev); stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
def big():
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(60):
        _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, x.data_ptr(), None, W.data_ptr(), Y.data_ptr(), stats.data_ptr(), st)

ss = torch.zeros(512, device=dev); mi = torch.zeros(512, device=dev); st64 = torch.zeros(512, dtype=torch.float64, device=dev)
gam = torch.ones(256, device=dev); bet = torch.zeros(256, device=dev)
def tiny_own():
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4000):
        _ffi.call("demf_bn_finalize", 256, 1000, st64.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1, None, None, None, ss.data_ptr(), mi.data_ptr(), st)
def medium():
    for _ in range(300):
        torch.mm(x[:8192], W.t())
for name, f in (("4000 tiny kernels", tiny), ("4000 tiny own kernels", tiny_own), ("300 mm 8192x64x64", medium), ("60 persistent GEMMs", big)):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    for sname, sf in SIDES:
        def it():
            if sf is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    sf()
            g.replay()
            if sf is not None:
                torch.cuda.current_stream().wait_stream(side)
        for _ in range(3): it()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): it()
        torch.cuda.synchronize()
        print(f"{name:22s} | side: {sname:44s} {(time.perf_counter() - t) / 10 * 1e3:6.2f} ms")
