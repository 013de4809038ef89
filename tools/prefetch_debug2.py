import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).eval()
batch, _ = bench.make_batch(8, seed=1000, device=dev)
pts = batch["points"]
geo = model.index_geometry(pts)
bb = model.pts_backbone
with torch.no_grad():
    a = bb(pts)
    b = bb(pts, geo)
for k in ("sa_xyz", "sa_features", "sa_indices", "fp_features"):
    for i, (x, y) in enumerate(zip(a[k], b[k])):
        if x is None: continue
        print(k, i, tuple(x.shape), float((x.float() - y.float()).abs().max()))
for i, (gi, gx, gg) in enumerate(geo["sa"]):
    print("level", i, gi.dtype, gi.shape, gg.shape, "idx==arange:", bool((gi[0] == torch.arange(gi.shape[1], device=dev)).all()))
la = model.forward_train(pts, batch["img_features"], batch["img_metas"], batch["gt_bboxes_3d"], batch["gt_labels_3d"])
lb = model.forward_train(pts, batch["img_features"], batch["img_metas"], batch["gt_bboxes_3d"], batch["gt_labels_3d"], geometry=geo)
for k in la: print(k, float(la[k]), float(lb[k]))
