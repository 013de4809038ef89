"""ATen ops (library launches) issued by one eager training step, with the demf_amd call site of each
(TorchDispatchMode + Python stack; backward ops run on the autograd thread: attributed to 'backward')."""
import sys, os, collections, traceback, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
import bench
from torch.utils._python_dispatch import TorchDispatchMode
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
cnt = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.expand",
        "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.detach", "aten.alias", "aten.as_strided", "aten.empty",
        "aten.reshape", "aten.split", "aten.unbind", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar",
        "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.result_type", "aten.unfold", "prim.", "aten.narrow",
        "aten.chunk", "aten.flatten", "aten.contiguous", "aten.numel", "aten.dim", "aten.item")
class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "backward/autograd"
            f = sys._getframe(0)
            while f is not None:                      # no source-line lookups: frames only
                fn = f.f_code.co_filename
                if "/demf_amd/" in fn:
                    site = f"{fn.split('/demf_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
                    break
                f = f.f_back
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            cnt[(name, site, str(shp))] += 1
        return func(*args, **(kwargs or {}))
with Census():
    tr._fwd_bwd(batch, geo); tr._update()
torch.cuda.synchronize()
for (n, site, shp), c in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{c:3d} {n:34s} {site:58s} {shp}")
print("total", sum(cnt.values()))
