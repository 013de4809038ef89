import sys, torch
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

def flat(g):
    out = []
    for lvl in g["sa"]: out += list(lvl)
    for lvl in g["fp"]: out += list(lvl)
    out.append(g["sample_indices"])
    return out

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): tr.step(batch)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
ref = [t.clone() for t in flat(model.index_geometry(batch["points"]))]
static_geo = model.index_geometry(batch["points"])
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss = tr._fwd_bwd(batch, static_geo)
static_pts = batch["points"].clone()
torch.cuda.synchronize()
geo_graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(geo_graph, stream=side):
    fresh = flat(model.index_geometry(static_pts))
torch.cuda.synchronize()

def cmp(tag):
    torch.cuda.synchronize()
    bad = [(i, tuple(a.shape), str(a.dtype), int((a != b).sum())) for i, (a, b) in enumerate(zip(fresh, ref)) if not torch.equal(a, b)]
    print(tag, "mismatching tensors:", bad)

import os
main = torch.cuda.current_stream()
if os.environ.get("PRE") == "geo":
    with torch.cuda.stream(side):
        geo_graph.replay()
    torch.cuda.synchronize()
names = [n for n, p in model.named_parameters() if p.requires_grad]
def report(tag):
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print(tag, "loss", float(loss), "non-finite grads:", len(bad), bad[:12])
if os.environ.get("PRE") == "main2":
    graph.replay(); report("main replay 1")
    graph.replay(); report("main replay 2")
    graph.replay(); report("main replay 3")
    sys.exit(0)
static_flat = flat(static_geo)
MODE = os.environ.get("MODE", "conc")
for it in range(8):
    side.wait_stream(main)
    if MODE == "seq_before":
        with torch.cuda.stream(side):
            geo_graph.replay()
        main.wait_stream(side)
    graph.replay()
    if MODE == "conc":
        with torch.cuda.stream(side):
            geo_graph.replay()
    torch.cuda.synchronize()
    g_ok = bool(torch.isfinite(tr.flat.flat).all())
    nbad = int((~torch.isfinite(tr.flat.flat)).sum())
    if not g_ok and not globals().get('_rep'):
        _rep = True
        off = 0
        _nm = {id(p): n for n, p in model.named_parameters()}
        for p in tr.flat.params:
            n = _nm[id(p)]
            k = p.numel(); seg = tr.flat.flat[off:off + k]; off += k
            if not bool(torch.isfinite(seg).all()): print("   bad:", n, int((~torch.isfinite(seg)).sum()), k, (~torch.isfinite(seg)).nonzero().flatten()[:6].tolist())
    tr._update()
    main.wait_stream(side)
    torch._foreach_copy_(static_flat, fresh)
    torch.cuda.synchronize()
    print(MODE, it, "loss", float(loss), "grads finite after graph:", g_ok, nbad,
          "params finite:", all(bool(torch.isfinite(p).all()) for p in model.parameters()))
