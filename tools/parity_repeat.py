"""The parity step of tests/test_gpu_parity_full.py repeated N times on ONE (conditioned weights, held-out batch) pair:
the HIP path is deterministic per pair to ~9 digits; what varies between runs of the test is the network.
Usage: python tools/parity_repeat.py [N]"""
import sys, os
ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT_); sys.path.insert(0, os.path.join(ROOT_, "tests"))
import numpy as np, torch
import parity_tools as P
import test_gpu_parity_full as T
from demf_amd.modules import DeMFHotPath
cfg = T._cfg(); B = 8
state = T.conditioned_state()
raw, gtb, gtl = T.held_out_case(B, state)
truth = P.oracle_run(cfg, raw, gtb, gtl, 0, torch.float64, tap=False, state=state)
model = DeMFHotPath(cfg); model.load_state_dict(state); model.cuda().train()
dev = T._dev_batch(raw, gtb, gtl)
head = model.pts_bbox_head
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = [p for p in model.parameters() if p.requires_grad]
res = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    preds = model.forward_head(dev["points"], dev["img_features"], raw["img_metas"])
    losses = head.loss(preds, dev["points"], dev["gt_bboxes_3d"], dev["gt_labels_3d"], None, None, raw["img_metas"])
    grads = torch.autograd.grad(losses["_total"], params, allow_unused=True)
    worst = (0.0, None)
    for n, g in zip(names, grads):
        want = truth["grads"][n]
        if want.dim() < 2: continue
        e = (g.double().cpu() - want).norm().item() / max(want.norm().item(), 1e-30)
        worst = max(worst, (e, n))
    res.append(worst)
vals = sorted(r[0] for r in res)
print("TILE=%s: %d repeats on ONE state: worst weight-gradient rel-L2 min %.2e median %.2e max %.2e" % (os.environ.get("DEMF_FWD_TILE","1"), len(vals), vals[0], vals[len(vals)//2], vals[-1]))
print(sorted(res, key=lambda r: -r[0])[:4])
