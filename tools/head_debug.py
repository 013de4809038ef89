import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from demf_amd.modules.vote import BaseConvBboxHead
from oracle import deps, fixtures
torch.manual_seed(0)
for B, N in ((2, 256), (2, 128), (8, 256)):
    x = torch.randn(B, 256, N)
    gc, gr = torch.randn(B, 12, N), torch.randn(B, 30, N)
    kw = dict(in_channels=256, shared_conv_channels=(128, 128), num_cls_out_channels=12, num_reg_out_channels=30, bias=True)
    ref = deps.BaseConvBboxHead(**kw); fixtures.seed_weights(ref, 5); ref.train().double()
    xr = x.double().requires_grad_(); c, r = ref(xr); (c * gc.double()).sum().backward(retain_graph=True); (r * gr.double()).sum().backward()
    m = BaseConvBboxHead(**kw); fixtures.seed_weights(m, 5); m.cuda().train()
    # point-major storage view, as in the pipeline
    xg = x.cuda().transpose(1, 2).contiguous().transpose(1, 2).requires_grad_()
    c2, r2 = m(xg); ((c2 * gc.cuda()).sum() + (r2 * gr.cuda()).sum()).backward()
    print(B, N, "fwd", (c2.detach().cpu().double() - c.detach()).abs().max().item())
    pr = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        t = pr[n].grad
        print("   %-40s rel %.2e" % (n, (p.grad.double().cpu() - t).norm().item() / max(t.norm().item(), 1e-30)), "norm %.2e" % t.norm().item())
    print("   x.grad rel %.2e" % ((xg.grad.double().cpu() - xr.grad).norm().item() / xr.grad.norm().item()))
