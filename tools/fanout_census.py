"""Which tensors of the training step have more than one consumer in the autograd graph (each extra consumer is
one gradient-accumulation `add` launch in the backward), and which pooled-MLP nodes receive a non-contiguous
gradient (one `.contiguous()` copy each)."""
import sys, os, collections, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
import bench
from demf_amd import engine, ops
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
geo = model.index_geometry(batch["points"])
tr._arena(True)
total = tr._fwd(batch, geo)
indeg = collections.Counter()
seen, stack = set(), [total.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    for nxt, nr in fn.next_functions:
        if nxt is not None:
            indeg[(nxt, nr)] += 1
            stack.append(nxt)
print("autograd nodes:", len(seen))
for (fn, nr), c in sorted(indeg.items(), key=lambda kv: -kv[1]):
    if c > 1 and "AccumulateGrad" not in type(fn).__name__:
        users = [type(u).__name__ for u in seen if any(n is fn and k == nr for n, k in u.next_functions)]
        meta = getattr(fn, "_input_metadata", None)
        print("%d consumers of %s output %d  <- %s" % (c, type(fn).__name__, nr, users))
orig = ops._SharedMLPPool.backward
def spy(ctx, g):
    if not g.is_contiguous():
        print("non-contiguous grad into _SharedMLPPool: shape", tuple(g.shape), "strides", g.stride())
    return orig(ctx, g)
ops._SharedMLPPool.backward = staticmethod(spy)
tr.flat.backward_into(total)
tr._arena(False)
torch.cuda.synchronize()
