"""How does a long, narrow kernel on a second HIP stream affect a train of tiny kernels?"""
import sys, torch
sys.path.insert(0, "/root/repo")
from demf_amd import ops
dev = torch.device("cuda:0")
pts = torch.rand(8, 20000, 3, device=dev)
x = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def train(n=300):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): x.add_(1.0)
    e1.record()
    return e0, e1

# as graphs (like the product) to remove host launch effects
g_main = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(main)
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1.0)
main.wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g_main):
    for _ in range(300): x.add_(1.0)
def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if mode != "alone":
        with torch.cuda.stream(side):
            ops.furthest_point_sample(pts, 2048)
            if mode == "fps+tail":
                for _ in range(5): pts.mul_(1.0)
    e0.record()
    g_main.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for mode in ("alone", "fps", "fps+tail", "alone", "fps", "fps+tail"):
    print(mode, "300 tiny kernels: %.3f ms" % run(mode))
