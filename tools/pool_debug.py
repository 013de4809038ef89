import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from demf_amd import ops
torch.manual_seed(0)
Rp, ns, ld, chans = 700, 64, 4, (64, 64, 128)
R = Rp * ns
x = (torch.randn(R, ld) * 0.7 + 0.1).cuda()
layers, k = [], ld
for n in chans:
    layers.append((torch.randn(n, k).cuda() / np.sqrt(k), (1.0 + 0.2 * torch.randn(n)).cuda(), (0.1 * torch.randn(n)).cuda(),
                   torch.zeros(n).cuda(), torch.ones(n).cuda()))
    k = n
layers[-1][1][1] = -0.5; layers[-1][1][2] = 0.0
outs = {}
for flag in (True, False):
    ops._NO_FUSED_POOL = flag
    outs[flag] = ops.shared_mlp_pool(x, ns, [tuple(t.clone() for t in l) for l in layers], training=True)
d = (outs[True] - outs[False]).abs()
print("max diff fused vs unfused:", float(d.max()), "bad cols:", torch.nonzero(d.max(0).values > 1e-5).flatten().tolist()[:20], "bad rows:", torch.nonzero(d.max(1).values > 1e-5).flatten().tolist()[:20])
import torch.nn.functional as F
h = x.double().cpu()
for W, g, b, _, _ in layers:
    h = F.relu(F.batch_norm(F.linear(h, W.double().cpu()), None, None, g.double().cpu(), b.double().cpu(), True, 0.1, 1e-5))
ref = h.view(Rp, ns, -1).max(1)[0]
for flag in (True, False):
    d = (outs[flag].double().cpu() - ref).abs()
    print("no_fused" if flag else "fused", "max err vs fp64 ref", float(d.max()), "cols", torch.nonzero(d.max(0).values > 1e-3).flatten().tolist()[:10])
print("last gamma[:4]", layers[-1][1][:4].tolist())
