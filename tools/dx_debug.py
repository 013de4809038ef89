import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from demf_amd import ops
def run(Rp, ns, ld, chans, seed=0):
    torch.manual_seed(seed)
    R = Rp * ns
    x = torch.randn(R, ld, dtype=torch.float64) * 0.7 + 0.1
    layers, k = [], ld
    for n in chans:
        layers.append((torch.randn(n, k, dtype=torch.float64) / np.sqrt(k), 1.0 + 0.2 * torch.randn(n, dtype=torch.float64), 0.1 * torch.randn(n, dtype=torch.float64)))
        k = n
    go = torch.randn(Rp, chans[-1], dtype=torch.float64)
    xr = x.clone().requires_grad_()
    h = xr
    for W, g, b in layers:
        h = F.relu(F.batch_norm(F.linear(h, W), None, None, g, b, True, 0.1, 1e-5))
    h.view(Rp, ns, -1).max(1)[0].backward(go)
    xg = x.float().cuda().requires_grad_()
    lg = [(W.float().cuda().requires_grad_(), g.float().cuda().requires_grad_(), b.float().cuda().requires_grad_(), torch.zeros(W.shape[0]).cuda(), torch.ones(W.shape[0]).cuda()) for W, g, b in layers]
    ops.shared_mlp_pool(xg, ns, lg, training=True).backward(go.float().cuda())
    d = (xg.grad.double().cpu() - xr.grad).abs()
    scale = xr.grad.abs().max().item()
    rows = torch.nonzero(d.max(1).values > 1e-3 * scale).flatten()
    print(f"Rp={Rp} ns={ns} ld={ld} chans={chans}: dx max err {d.max().item():.3e} (scale {scale:.3e}); bad rows {len(rows)} of {Rp*ns}", rows[:6].tolist(), "row%256:", sorted(set((rows % 256).tolist()))[:12], " bad cols", torch.nonzero(d.max(0).values > 1e-3 * scale).flatten().tolist()[:10])
run(1500, 32, 132, (128, 128, 256))
run(1500, 32, 128, (128, 128, 256))
run(1500, 32, 132, (128, 256))
run(1500, 32, 132, (128,))
run(760, 32, 132, (128, 128, 256))
run(800, 32, 132, (128, 128, 256))
