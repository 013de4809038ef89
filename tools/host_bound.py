"""Is the replayed step bound by the GPU or by the host enqueueing it?  Times N resident replays twice: the
host's own time to ENQUEUE them (before any synchronisation) and the time until the GPU has finished."""
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
batch, _ = bench.make_batch(8, seed=1000, device=dev)
rep = tr.capture(batch, prefetch_geometry=True)
for _ in range(5): rep()
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for _ in range(N): rep()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, until GPU idle %.3f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
# the graph alone, back to back (no pre-pass, no optimizer step): GPU time of the captured fwd+bwd
g = tr._graph
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph only: host %.3f ms, GPU %.3f ms per replay" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
