"""Times ops.shared_mlp_pool (fused shared-MLP chain) forward / backward at the SA shapes of the
B=8 bench and reports algorithmic GB/s and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops
SHAPES = [("SA1", 1048576, 64, 4, (64, 64, 128), False), ("SA2", 262144, 32, 132, (128, 128, 256), True),
          ("SA3", 65536, 16, 260, (128, 128, 256), True), ("SA4", 32768, 16, 260, (128, 128, 256), True),
          ("AGG", 32768, 16, 260, (256, 256, 256), True), ("FP1", 4096, 1, 512, (256, 256), True)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tot_f = tot_b = 0
for name, R, ns, ld, ch, xg in SHAPES:
    x = torch.randn(R, ld, device="cuda", requires_grad=xg)
    layers, k = [], ld
    for n in ch:
        layers.append((torch.randn(n, k, device="cuda").div_(k ** 0.5).requires_grad_(), torch.ones(n, device="cuda", requires_grad=True),
                       torch.zeros(n, device="cuda", requires_grad=True), torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")))
        k = n
    go = torch.randn(R // ns, ch[-1], device="cuda")
    f = lambda: ops.shared_mlp_pool(x, ns, layers, True)
    tf = timeit(f)
    def fb():
        out = ops.shared_mlp_pool(x, ns, layers, True); out.backward(go)
    tfb = timeit(fb)
    dims = [ld] + list(ch)
    fbytes = sum(R * (dims[i] + dims[i + 1]) * 4 for i in range(len(ch)))
    flops = sum(2 * R * dims[i] * dims[i + 1] for i in range(len(ch)))
    print(f"{name}: fwd {tf*1e3:7.1f} us ({fbytes/tf/1e6:6.0f} GB/s algo, {flops/tf/1e9:5.1f} TF/s)   bwd {(tfb-tf)*1e3:7.1f} us ({2*flops/(tfb-tf)/1e9:5.1f} TF/s)")
    tot_f += tf; tot_b += tfb - tf
print(f"total fwd {tot_f:.2f} ms, bwd {tot_b:.2f} ms")
