"""Which kind of kernel pays for the concurrent FPS chain?  A fixed sequence of ONE kind of
kernel is captured in a hipGraph and replayed alone / with k scenes of 20000->2048 FPS running on a
second stream (the FPS chain takes ~2.7 ms on k CUs).  Reports the slowdown of the sequence."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from demf_amd import ops, _ffi
dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = (torch.rand(8, 20000, 3, device=dev) * 6).contiguous()
side = torch.cuda.Stream()

def seq_mlp(R, K, N, reps):
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    Y = torch.empty(R, N, device=dev); stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    def f():
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(reps):
            _ffi.call("demf_mlp_gemm_fwd", R, K, N, K, x.data_ptr(), None, W.data_ptr(), Y.data_ptr(), stats.data_ptr(), st)
    return f

def seq_tiny(n):
    a = torch.randn(2048, 256, device=dev); b = torch.randn(2048, 256, device=dev)
    def f():
        c = a
        for _ in range(n):
            c = c + b
        return c
    return f

def seq_mm(M, K, N, reps):
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    def f():
        for _ in range(reps):
            torch.mm(a, b)
    return f

def seq_copy(mb, reps):
    a = torch.empty(mb * 1024 * 256, device=dev); b = torch.empty_like(a)
    def f():
        for _ in range(reps):
            b.copy_(a)
    return f

CASES = [("mlp_gemm 1M x 64 -> 64 (persistent 512 blocks) x60", seq_mlp(1 << 20, 64, 64, 60)),
         ("mlp_gemm 262144 x 128 -> 128 x84", seq_mlp(1 << 18, 128, 128, 84)),
         ("mlp_gemm 8192 x 256 -> 256 (column split) x4000", seq_mlp(8192, 256, 256, 400)),
         ("tiny elementwise add (2 MB) x4000", seq_tiny(4000)),
         ("hipBLASLt mm 2048x256x256 x1200", seq_mm(2048, 256, 256, 1200)),
         ("copy 256 MB x72", seq_copy(256, 72))]

for name, f in CASES:
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    res = []
    for k in (0, 1, 8):
        def it():
            if k:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ops.furthest_point_sample(pts[:k], 2048)
            g.replay()
            if k:
                torch.cuda.current_stream().wait_stream(side)
        for _ in range(3): it()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): it()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / 10 * 1e3)
    print(f"{name:55s} alone {res[0]:6.2f} ms | +1 scene FPS {res[1]:6.2f} | +8 scenes {res[2]:6.2f}   (FPS alone ~2.7 ms)")
