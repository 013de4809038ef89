"""Host time of the per-step input path (replay.load) and of the replay call, and the loop's pace with and
without the load - to see whether the load's host work overlaps the previous step on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
a, _ = bench.make_batch(8, seed=1000, device=dev)
b, _ = bench.make_batch(8, seed=2000, device=dev)
step = tr.capture(a, prefetch_geometry=True)
pair = [a, b]
for i in range(6):
    step.load(pair[i & 1]); step(next_points=pair[(i + 1) & 1]["points"])
torch.cuda.synchronize()
n = 40
tl = tsr = 0.0
t0 = time.perf_counter()
for i in range(n):
    cur, nxt = pair[i & 1], pair[(i + 1) & 1]
    t1 = time.perf_counter(); step.load(cur); t2 = time.perf_counter(); step(next_points=nxt["points"]); t3 = time.perf_counter()
    tl += t2 - t1; tsr += t3 - t2
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"with load: {1e3 * tot / n:.2f} ms/step; host time in load() {1e3 * tl / n:.2f} ms, in the replay call {1e3 * tsr / n:.2f} ms")
t0 = time.perf_counter()
for i in range(n):
    step(next_points=pair[(i + 1) & 1]["points"])
torch.cuda.synchronize()
print(f"no load (next_points alternating): {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step")
t0 = time.perf_counter()
for i in range(n):
    step()
torch.cuda.synchronize()
print(f"plain replay: {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step")
# pieces of load()
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    step.load(pair[i & 1]); step(next_points=pair[(i + 1) & 1]["points"])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
