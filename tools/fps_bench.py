"""FPS launch times: the 20000 -> 2048 level and the already-ordered levels behind it."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from demf_amd import ops
dev = torch.device("cuda:0")
def t(name, x, m, n=10):
    for _ in range(3): ops.furthest_point_sample(x, m)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): idx = ops.furthest_point_sample(x, m)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1)/n*1e3:8.1f} us   arange: {bool((idx[0] == torch.arange(m, device=dev)).all())}")
for B in (1, 8):
    pts = torch.rand(B, 20000, 3, device=dev) * torch.tensor([6.0, 6.0, 3.0], device=dev)
    i1 = ops.furthest_point_sample(pts, 2048)
    l1 = torch.gather(pts, 1, i1.long()[..., None].expand(-1, -1, 3)).contiguous()
    t(f"B={B} 20000->2048", pts, 2048)
    t(f"B={B} 2048->1024 (ordered input)", l1, 1024)
    t(f"B={B} 2048->1024 (shuffled input)", l1[:, torch.randperm(2048, device=dev)].contiguous(), 1024)
    l2 = l1[:, :1024].contiguous()
    t(f"B={B} 1024->512 (ordered)", l2, 512)
    t(f"B={B} 1024->256 (ordered)", l2, 256)
    t(f"B={B} 512->256 (ordered)", l2[:, :512].contiguous(), 256)
