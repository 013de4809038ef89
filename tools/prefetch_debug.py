"""Checks that graph replay with the pipelined geometry pre-pass trains exactly like replay without it."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath

dev = torch.device("cuda:0")
out = {}
for mode in ("nopf", "pf"):
    torch.manual_seed(0)
    model = DeMFHotPath(DeMFCfg()).to(dev).train()
    tr = engine.Trainer(model)
    batch, _ = bench.make_batch(8, seed=1000, device=dev)
    step = tr.capture(batch, prefetch_geometry=(mode == "pf"))
    ls = []
    for i in range(6):
        ls.append(float(step()))
    torch.cuda.synchronize()
    out[mode] = ls
    print(mode, ["%.5f" % v for v in ls])
    if mode == "pf":
        geo = model.index_geometry(batch["points"])
        print("sample_indices[0,:8]", geo["sample_indices"][0, :8].tolist())
