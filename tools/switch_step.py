"""One training step (forward + loss + backward + update) of a mid-size configuration on the HIP path, printed as one
JSON line {loss, grad_norm, param_delta}: what tests/test_gpu_switches.py runs in a subprocess under every non-default
value of the most-used DEMF_* A/B switches (they are read once per process)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import engine
from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg
from demf_amd.modules import DeMFHotPath
from oracle import fixtures

MID = ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551)
cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)),
              head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
raw = fixtures.make_scene_batch(2, 6000, MID[0], MID[1], cfg.head.embed_dims, seed=5, n_gt=4, img_shape=MID[2])
model = DeMFHotPath(cfg)
fixtures.seed_weights(model, 5)
model.cuda().train()
batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
             img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]], img_metas=raw["img_metas"],
             gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
             gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
tr = engine.Trainer(model, lr=1e-4)
p0 = tr.opt.flat.clone()
graph = "--graph" in sys.argv
if graph:
    replay = tr.capture(batch, warmup=1, max_gt=8, dry=True)
    loss = replay()
else:
    loss = tr.step(batch)
torch.cuda.synchronize()
print(json.dumps(dict(loss=float(loss), grad_norm=float(tr.flat.flat.double().norm()),
                      param_delta=float((tr.opt.flat - p0).double().norm()),
                      finite=bool(torch.isfinite(tr.opt.flat).all()))))
