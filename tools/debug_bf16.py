"""bf16-vs-fp32 error budget: shared MLP stack and the tiny / mid hot path, per tensor."""
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch, numpy as np
import test_gpu_bf16 as T
from demf_amd import ops
from oracle import fixtures
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
outs = {}
for mode in ("f32", "bf16"):
    ops.set_compute_dtype(mode)
    mlp, x = T._mlp_case(3, 4096 * 16, 16, [64, 64, 128, 256])
    y = mlp.forward_rows(x, ns=16)
    (y * T._r(*y.shape, seed=9)).sum().backward()
    outs[mode] = [("y", y.detach()), ("dx", x.grad)] + [(n, p.grad) for n, p in mlp.named_parameters()]
ops.set_compute_dtype("f32")
for (n, a), (_, b) in zip(outs["bf16"], outs["f32"]):
    print(f"mlp {n:24s} rel {rel(a, b):.3e}")
import parity_tools as P
from demf_amd.modules import DeMFHotPath
cfg = fixtures.tiny_cfg()
batch, gtb, gtl = P.make_case(cfg, 2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT, None, 4)
res = {}
for mode in ("f32", "bf16"):
    ops.set_compute_dtype(mode)
    model = DeMFHotPath(cfg); fixtures.seed_weights(model, 4); model.cuda().train()
    pts = torch.from_numpy(batch["points"]).cuda(); feats = [torch.from_numpy(f).cuda() for f in batch["img_features"]]
    with torch.no_grad():
        bb = model.pts_backbone(pts)
        preds = model.forward_head(pts, feats, batch["img_metas"])
    res[mode] = dict(sa=[f for f in bb["sa_features"]] if "sa_features" in bb else [], fp=bb["fp_features"], preds=preds)
ops.set_compute_dtype("f32")
a, b = res["bf16"], res["f32"]
for i, (u, v) in enumerate(zip(a["sa"], b["sa"])):
    if u is not None and v is not None: print(f"sa_features[{i}] rel {rel(u, v):.3e}")
for i, (u, v) in enumerate(zip(a["fp"], b["fp"])): print(f"fp_features[{i}] rel {rel(u, v):.3e}")
for k in ("vote_points", "vote_features", "aggregated_points"):
    print(k, f"rel {rel(a['preds'][k], b['preds'][k]):.3e}  max {float((a['preds'][k]-b['preds'][k]).abs().max()):.3e}")
for i in range(2):
    for k, v in b["preds"]["decode_res_all"][i].items():
        if not k.startswith("_"): print(f"decode{i}.{k:14s} rel {rel(a['preds']['decode_res_all'][i][k], v):.3e}")
