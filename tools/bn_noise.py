import torch, torch.nn.functional as F
torch.manual_seed(0)
P, C = 1048576, 64
x = torch.randn(P, C) * 2 + 1; g = torch.randn(P, C); w = torch.rand(C) + 0.5; b = torch.randn(C)
def run(x, g, w, b, dev, dt):
    x = x.to(dev, dt).detach().requires_grad_(); w = w.to(dev, dt).detach().requires_grad_(); b = b.to(dev, dt).detach().requires_grad_()
    y = F.batch_norm(x, None, None, w, b, True, 0.1, 1e-5)
    y.backward(g.to(dev, dt))
    return [t.detach().double().cpu() for t in (y, x.grad, w.grad, b.grad)]
T = run(x, g, w, b, "cpu", torch.float64)
for name, dev in (("cpu32", "cpu"), ("gpu32", "cuda")):
    R = run(x, g, w, b, dev, torch.float32)
    print(name, ["%.2e" % ((r - t).norm() / t.norm()).item() for r, t in zip(R, T)])
# linear
W = torch.randn(128, 64) / 8
def lin(dev, dt):
    xx = x.to(dev, dt).detach().requires_grad_(); WW = W.to(dev, dt).detach().requires_grad_()
    y = F.linear(xx, WW); y.backward(torch.ones_like(y) * g.to(dev, dt)[:, :1])
    return [t.detach().double().cpu() for t in (y, xx.grad, WW.grad)]
T = lin("cpu", torch.float64)
for name, dev in (("cpu32", "cpu"), ("gpu32", "cuda")):
    R = lin(dev, torch.float32)
    print("linear", name, ["%.2e" % ((r - t).norm() / t.norm()).item() for r, t in zip(R, T)])
print(torch.backends.cudnn.enabled, torch.backends.cuda.matmul.allow_tf32)
