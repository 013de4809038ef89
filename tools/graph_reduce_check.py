"""Does a torch column-sum (global two-stage reduce with semaphores) inside a hipGraph replay reliably?"""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(2048, 1024, device=dev)
w = torch.randn(1024, 256, device=dev)
bad = torch.zeros((), dtype=torch.int64, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y = torch.relu(x @ w @ w.t()); r = y.sum(0)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
ref = torch.relu(x @ w @ w.t()).double().sum(0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = []
    for _ in range(8):
        y = torch.relu(x @ w @ w.t())
        r = y.sum(0)
        outs.append(r)
        bad += ((r.double() - ref).abs() > 1e-3 * ref.abs() + 1.0).sum() + (~torch.isfinite(r)).sum()
for it in range(500):
    g.replay()
torch.cuda.synchronize()
print("mismatching column sums over 4000 graph-replayed reductions:", int(bad))
bad.zero_()
for it in range(500):
    for _ in range(8):
        y = torch.relu(x @ w @ w.t()); r = y.sum(0)
        bad += ((r.double() - ref).abs() > 1e-3 * ref.abs() + 1.0).sum() + (~torch.isfinite(r)).sum()
torch.cuda.synchronize()
print("eager:", int(bad))
