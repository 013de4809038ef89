"""Host-side cost of one replay() call vs the GPU step time."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
step = tr.capture(batch)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host loop {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
g = tr._graph
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a = time.perf_counter(); g.replay(); b = time.perf_counter(); torch.cuda.synchronize(); ts.append(1e3*(b-a))
print("graph.replay() host ms:", ["%.2f" % t for t in ts])
