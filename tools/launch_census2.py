"""Which source lines still launch library (ATen / rocclr) kernels in one eager training step:
torch.profiler with stacks; every CPU op that owns device kernels is attributed to the innermost
frame inside this repository."""
import sys, os, collections, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr._fwd_bwd(batch, geo); tr._update()
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    # leaf ops only: skip an op if one of its children also owns the kernels
    if any(c.kernels for c in e.cpu_children):
        continue
    kn = [k.name for k in e.kernels]
    if all("demf::" in k for k in kn):
        continue
    site = "(no repo frame: autograd engine)"
    for fr in e.stack:
        if "/demf_amd/" in fr or "bench.py" in fr:
            site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-90:]
            break
    key = (site, e.name)
    cnt[key] += len(kn); tim[key] += sum(k.duration for k in e.kernels)
tot = sum(cnt.values())
print("library launches in one eager step:", tot, " kernel time %.0f us" % sum(tim.values()))
for (site, name), n in cnt.most_common(70):
    print(f"{n:4d} {tim[(site, name)]:7.0f} us  {name:38s} {site}")
