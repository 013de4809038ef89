"""Library (non-demf) launches of one eager training step, by op, input shapes and the innermost
demf_amd call site (where do the remaining glue launches come from)."""
import sys, os, collections, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
import bench
from torch.profiler import profile, ProfilerActivity
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
geo = model.index_geometry(batch["points"])
for _ in range(2):
    tr._fwd_bwd(batch, geo); tr._update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr._fwd_bwd(batch, geo); tr._update()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels: continue
    if any(c.kernels for c in e.cpu_children): continue
    if all("demf::" in k.name for k in e.kernels): continue
    site = next((f.split("/root/repo/")[-1] for f in (e.stack or []) if "demf_amd/" in f), "?")
    cnt[(e.name, str(e.input_shapes)[:70], site[:60])] += len(e.kernels)
for (n, sh, site), c in cnt.most_common(80):
    print(f"{c:3d} {n:24s} {site:60s} {sh}")
