"""Runs N training steps in a given launch mode and reports the first step whose loss / gradients are non-finite."""
import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
mode, steps = sys.argv[1], int(sys.argv[2])
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.008
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model, lr=lr)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
nm = {id(p): n for n, p in model.named_parameters()}
if os.environ.get("GEMV"):
    import torch.nn.functional as F
    class LinGemv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return F.linear(x, w, b)
        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            g2 = g.reshape(-1, g.shape[-1]); x2 = x.reshape(-1, x.shape[-1])
            return (g2 @ w).view_as(x), g2.t() @ x2, torch.mv(g2.t(), torch.ones(g2.shape[0], device=g.device))
    ffn = model.pts_bbox_head.decoder[0].layer.ffns[0]
    for lin in (ffn.layers[0][0], ffn.layers[1]):
        lin.forward = (lambda l: (lambda x: LinGemv.apply(x.reshape(-1, x.shape[-1]), l.weight, l.bias).view(*x.shape[:-1], -1)))(lin)
if mode == "eager":
    fb = lambda: tr._fwd_bwd(batch)
else:
    geo = model.index_geometry(batch["points"]) if mode == "graph_geo" else None
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): tr.step(batch)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    dbg = {}
    if os.environ.get("HOOK"):
        lin = model.pts_bbox_head.decoder[0].layer.ffns[0].layers[0][0]
        lin.bias.register_hook(lambda gr: dbg.__setitem__("bias_grad", gr.clone()) or None)
        lin.weight.register_hook(lambda gr: dbg.__setitem__("weight_grad", gr.clone()) or None)
        _orig = lin.forward
        def _fwd(x):
            h = _orig(x)
            if h.requires_grad: h.register_hook(lambda gr: dbg.__setitem__("dY", gr.clone()) or None)
            return h
        lin.forward = _fwd
    g = torch.cuda.CUDAGraph()
    if os.environ.get("DOT"): g.enable_debug_mode()
    with torch.cuda.graph(g):
        loss_t = tr._fwd_bwd(batch, geo)
    if os.environ.get("DOT"):
        g.debug_dump(os.environ["DOT"]); print("dumped"); sys.exit(0)
    def fb():
        g.replay(); return loss_t
for it in range(steps):
    loss = fb()
    torch.cuda.synchronize()
    flat = tr.flat.flat
    if not bool(torch.isfinite(flat).all()) or not bool(torch.isfinite(loss)):
        print(mode, "step", it, "loss", float(loss), "non-finite grads:", int((~torch.isfinite(flat)).sum()))
        off = 0
        for p in tr.flat.params:
            k = p.numel(); seg = flat[off:off + k]; off += k
            bad = (~torch.isfinite(seg)).nonzero().flatten()
            if len(bad): print("   ", nm[id(p)], len(bad), "of", k, bad[:4].tolist(), seg[bad[:4]].tolist())
        for k, v in globals().get("dbg", {}).items():
            bad = (~torch.isfinite(v)).nonzero()
            print("   dbg", k, tuple(v.shape), "non-finite:", len(bad), bad[:4].tolist())
            if k == "dY":
                cs = v.reshape(-1, v.shape[-1]).sum(0)
                print("   dY eager colsum finite:", bool(torch.isfinite(cs).all()), "max|colsum - bias_grad| over finite:",
                      float((cs - dbg["bias_grad"])[torch.isfinite(dbg["bias_grad"])].abs().max()), "absmax dY", float(v.abs().max()))
        break
    tr._update()
else:
    print(mode, "all", steps, "steps finite; last loss", float(loss))
