"""SA1's shared-MLP stack alone (R = 8 x 2048 x 64 grouped rows, 4 -> 64 -> 64 -> 128, max over 64): forward
and forward + backward time with and without the no-store last layer (DEMF_POOL_NOY) / the input-row
recompute of layer 1 (DEMF_SA1_X4).  usage: python tools/sa1_micro.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops
ops.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32")
R, ns, ld, ch = 1048576, 64, 4, (64, 64, 128)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
x = torch.randn(R, ld, device="cuda")
layers, k = [], ld
for n in ch:
    layers.append((torch.randn(n, k, device="cuda").div_(k ** 0.5).requires_grad_(), torch.ones(n, device="cuda", requires_grad=True),
                   torch.zeros(n, device="cuda", requires_grad=True), torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")))
    k = n
go = torch.randn(R // ns, ch[-1], device="cuda")
for flags in ((False, False), (True, False), (True, True)):
    ops._POOL_NOY = flags[0]
    if hasattr(ops, "_SA1_X4"):
        ops._SA1_X4 = flags[1]
    elif flags[1]:
        continue
    f = lambda: ops.shared_mlp_pool(x, ns, layers, True)
    def fb():
        out = ops.shared_mlp_pool(x, ns, layers, True); out.backward(go)
    tf, tfb = timeit(f), timeit(fb)
    print(f"noy={flags[0]} x4={flags[1]}: fwd {tf:7.1f} us  fwd+bwd {tfb:7.1f} us  bwd {tfb - tf:7.1f} us", flush=True)
