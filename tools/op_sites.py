"""Every ATen op of one eager training step (forward, backward, update, load + pre-pass) that launches a library
kernel, with the innermost demf_amd source line that issued it (TorchDispatchMode + the Python stack)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath

VIEWS = ("view", "reshape", "transpose", "permute", "slice", "select", "unsqueeze", "squeeze", "expand", "as_strided",
         "detach", "alias", "t.default", "unbind", "split", "_unsafe_view", "empty", "is_", "size", "stride", "numel",
         "record_stream", "lift_fresh", "_local_scalar", "item", "unflatten", "narrow", "set_", "resize_", "_to_copy_same")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()
        self.phase = "?"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/demf_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    site = "%s:%d" % (fr.filename.split("/root/repo/")[-1].split("/repo/")[-1], fr.lineno)
                    break
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            self.cnt[(self.phase, name.replace("aten.", ""), site, str(shp))] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
geo = model.index_geometry(batch["points"])
for _ in range(2):
    tr._fwd_bwd(batch, geo); tr._update()
torch.cuda.synchronize()
s = Sites()
with s:
    s.phase = "prepass"
    geo = model.index_geometry(batch["points"])
    s.phase = "fwd"
    tr._arena(True)
    total = tr._fwd(batch, geo)
    s.phase = "bwd"
    tr.flat.backward_into(total)
    tr._arena(False)
    s.phase = "update"
    tr._update()
torch.cuda.synchronize()
for (ph, n, site, shp), c in sorted(s.cnt.items()):
    print("%-8s %3d  %-28s %-52s %s" % (ph, c, n, site, shp))
