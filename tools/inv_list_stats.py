"""Length distribution of the inverse neighbour lists (rows per source point) of every SA level of the bench
configuration: the work of one group of demf::group_first_bwd_k is one such list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
batch, _ = bench.make_batch(8, seed=1000, device=dev)
geo = model.index_geometry(batch["points"])
g = geo["backbone"] if "backbone" in geo else geo
for i, lvl in enumerate(g["sa"]):
    idx = lvl[2]                      # (B, M, ns) int32 into the level's source points
    B, M, ns = idx.shape
    N = (g["xyz"] if i == 0 else g["sa"][i - 1][1]).shape[1]
    cnt = torch.stack([torch.bincount(idx[b].reshape(-1).long(), minlength=N) for b in range(B)]).float()
    q = torch.quantile(cnt.reshape(-1), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
    print("SA%d N=%d M=%d ns=%d rows/pt mean %.1f  p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f  zero %.1f%%" % (
        i + 1, N, M, ns, cnt.mean().item(), *q.tolist(), cnt.max().item(), 100 * (cnt == 0).float().mean().item()))
