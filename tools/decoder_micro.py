"""Wall time of the fused decoder layer alone (forward, forward + backward) at the bench size, replayed
from a hipGraph: B = 8 scenes x 256 queries, E = 256, F = 1024, 18 609 image tokens."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
import test_gpu_dense as T
from demf_amd import fused, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
B, Q, H, L, P, E, Fd = 8, 256, 8, 4, 2, 256, 1024
shapes = ((100, 140), (50, 70), (25, 35), (13, 18))
c = T._make_case(B, Q, H, L, P, E, Fd, shapes, seed=1)
dims = (B, Q, H, L, P, 0.4, 0.1, 1e-5)
ins = [c["x"].clone().requires_grad_(), c["pos"].clone().requires_grad_(), c["pts"].clone().requires_grad_()]
prm = [c["prm"][k].clone().requires_grad_() for k in T.PARAM_ORDER]
gout = torch.randn(B * Q, E, device="cuda")
def fwd():
    return fused.FusedDecoderLayer.apply(ins[0], ins[1], ins[2], c["tokens"], c["keep4"], c["shapes"], c["lsi"], c["M"],
                                         c["ab"], c["vr"], dims, True, *prm)
def fwdbwd():
    ops.ARENA.begin(torch.device("cuda:0"))
    out = fwd()
    g = torch.autograd.grad(out, ins + prm, gout)
    ops.ARENA.end()
    return g
def graph_time(fn, n=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    tf = graph_time(lambda: fwd())
tfb = graph_time(fwdbwd)
print(f"decoder layer (graph replay): forward {tf:.1f} us, forward+backward {tfb:.1f} us")
