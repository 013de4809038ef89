"""Length distribution of the inverse neighbour lists (demf_invert_index) per SA level on the
bench's synthetic batch: the heavy tail (ball-query padding repeats a group's first neighbour)
is what the per-point backward kernels have to balance."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
batch = bench.make_batch(8, 0, dev)
pts = batch[0]["points"]
geo = model.pts_backbone.index_geometry(pts)
for i, lvl in enumerate(geo["sa"]):
    if len(lvl) < 5:
        continue
    off = lvl[3].cpu().numpy()
    ln = np.diff(off, axis=1)
    print(f"SA{i+1}: points/scene {ln.shape[1]}, rows/scene {off[0,-1]}, mean {ln.mean():.1f}, median {np.median(ln):.0f}, "
          f"p99 {np.percentile(ln,99):.0f}, max {ln.max()}, lists > 256: {(ln>256).sum()}, rows in lists > 256: {ln[ln>256].sum()} of {ln.sum()}")
