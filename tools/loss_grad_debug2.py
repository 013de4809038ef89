
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
from oracle import deps, fixtures
from demf_amd.config import DeMFCfg, HeadCfg, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE
import demf_amd.modules.vote as V
cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
ARGS = (2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2])
cap = {}
orig = V.BaseConvBboxHead.forward
def fwd(self, feats):
    c, r = orig(self, feats)
    tag = "p%d" % len([k for k in cap if k.startswith("x")])
    cap["x" + tag] = feats.detach().clone()
    if c.requires_grad:
        c.register_hook(lambda g, t=tag: cap.__setitem__("gc" + t, g.detach().clone()))
        r.register_hook(lambda g, t=tag: cap.__setitem__("gr" + t, g.detach().clone()))
    return c, r
V.BaseConvBboxHead.forward = fwd
for seed in range(1, 8):
    cap.clear()
    r = T._run_triple(cfg, *ARGS, seed)
    if r is not None: break
V.BaseConvBboxHead.forward = orig
gm, tm = r["gpu"]["model"], r["truth"]["model"]
for i in (0, 1):
    tag = "p%d" % i
    x, gc, gr = cap["x" + tag], cap["gc" + tag], cap["gr" + tag]
    kw = dict(in_channels=256, shared_conv_channels=(128, 128), num_cls_out_channels=12, num_reg_out_channels=30, bias=True)
    ref = deps.BaseConvBboxHead(**kw).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in getattr(gm.pts_bbox_head, "conv_pred%d" % i).state_dict().items()})
    ref.train()
    c, rr = ref(x.double().cpu())
    ((c * gc.double().cpu()).sum() + (rr * gr.double().cpu()).sum()).backward()
    print("conv_pred%d: gpu grads vs fp64 recomputation from the GPU's own captured x / upstream grads" % i)
    pg = dict(getattr(gm.pts_bbox_head, "conv_pred%d" % i).named_parameters())
    pt = dict(getattr(tm.pts_bbox_head, "conv_pred%d" % i).named_parameters())
    for n, p in ref.named_parameters():
        if p.grad.norm() < 1e-6: continue
        print("   %-34s vs-recomputed %.2e   vs-truth-pipeline %.2e" % (n, (pg[n].grad.double().cpu() - p.grad).norm().item() / p.grad.norm().item(),
              (pg[n].grad.double().cpu() - pt[n].grad).norm().item() / pt[n].grad.norm().item()))
