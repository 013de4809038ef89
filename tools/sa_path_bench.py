"""BASELINE.json configs[1]: the PointNet++ SA path only (FPS / ball_query / group + shared MLPs of the
4 SA + 2 FP levels), fp32, one MI355X - forward and forward+backward, B = 1 and 8, uniform and
clustered clouds.  A parity-case configuration, reported in DESIGN.md; not the bench.py line."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
bb = model.pts_backbone

def cloud(B, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        xyz = rng.uniform([-3, -3, 0], [3, 3, 3], size=(B, 20000, 3))
    else:
        c = rng.uniform([-3, -3, 0], [3, 3, 3], size=(B, 20, 1, 3))
        xyz = (c + 0.3 * rng.standard_normal((B, 20, 1000, 3))).reshape(B, 20000, 3)
    h = xyz[..., 2:3] - xyz[..., 2:3].min(axis=1, keepdims=True)
    return torch.from_numpy(np.concatenate([xyz, h], -1).astype(np.float32)).to(dev)

def timed(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for kind in ("uniform", "clustered"):
    for B in (1, 8):
        pts = cloud(B, kind)
        def fwd():
            with torch.no_grad(): return bb(pts)
        def fwdbwd():
            out = bb(pts); out["fp_features"][-1].sum().backward()
        def geo():
            return bb.index_geometry(pts)
        a, b, c = timed(fwd), timed(fwdbwd), timed(geo)
        print(f"{kind:9s} B={B}: forward {a:6.2f} ms ({B/a*1e3:7.1f} scenes/s)  fwd+bwd {b:6.2f} ms ({B/b*1e3:7.1f} scenes/s)"
              f"  index pre-pass alone (FPS/ball/3-NN) {c:5.2f} ms")
