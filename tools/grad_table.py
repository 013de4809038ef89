import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
from demf_amd.config import PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE
full = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0)) if full else DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
              head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
ARGS = (2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2]) if full else (2, 6000, ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551))
r = None
for seed in range(1, 8):
    r = T._run_triple(cfg, *ARGS, seed)
    if r is not None: break
pt, pc, pg = (dict(r[k]["model"].named_parameters()) for k in ("truth", "cpu32", "gpu"))
rows = []
for n in pt:
    if pt[n].grad is None: continue
    t = pt[n].grad.double(); nt = t.norm().item()
    rows.append((((pg[n].grad.double().cpu() - t).norm().item()) / max(nt, 1e-30), ((pc[n].grad.double() - t).norm().item()) / max(nt, 1e-30), nt, n))
rows = [x for x in rows if x[2] > 1e-6]
rows.sort(reverse=True)
for rg, rc, nt, n in rows[:25]:
    print(f"{rg:9.2e} {rc:9.2e} {nt:10.3e}  {n}")
for k in r["truth"]["losses"]:
    print(k, r["truth"]["losses"][k].item(), r["cpu32"]["losses"][k].item(), r["gpu"]["losses"][k].item())

G, C, Tr = r["gpu"]["preds"], r["cpu32"]["preds"], r["truth"]["preds"]
for i in (0, 1):
    for k in ("obj_scores", "center", "size"):
        t = Tr["decode_res_all"][i][k].detach().double()
        print(i, k, "gpu", (G["decode_res_all"][i][k].detach().double().cpu() - t).abs().max().item(), "cpu", (C["decode_res_all"][i][k].detach().double() - t).abs().max().item())
