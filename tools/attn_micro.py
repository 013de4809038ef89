"""Times demf_attn_core_{fwd,bwd} alone at the reference shape (8 scenes x 8 heads x 256 queries x 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from demf_amd import _ffi, fused, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
p_ = lambda t: None if t is None else t.data_ptr()
dev = torch.device("cuda")
fused.rng_state(dev, seed=5)
B, H, Q, Dh = 8, 8, 256, 32
E, R = H * Dh, B * Q
qkv, dout = torch.randn(R, 3 * E, device="cuda"), torch.randn(R, E, device="cuda")
rng, st = fused.rng_state(dev).data_ptr(), torch.cuda.current_stream().cuda_stream
out, stats, dqkv = torch.empty(R, E, device="cuda"), torch.empty(B * H * Q, 2, device="cuda"), torch.empty(R, 3 * E, device="cuda")
a = 1.0 / np.sqrt(Dh)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for p in (0.0, 0.1):
    f = lambda: _ffi.call("demf_attn_core_fwd", B, H, Q, Dh, p_(qkv), a, p, rng, 7, p_(out), p_(stats), None, None, st)
    b = lambda: _ffi.call("demf_attn_core_bwd", B, H, Q, Dh, p_(qkv), p_(out), p_(dout), p_(stats), a, p, rng, 7, p_(dqkv), st)
    print("p=%.1f: fwd %.1f us  bwd %.1f us" % (p, timeit(f), timeit(b)), flush=True)
