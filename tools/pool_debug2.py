import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from demf_amd import ops
torch.manual_seed(0)
for Rp, ns, ld, chans in ((700, 64, 4, (64, 64, 128)), (760, 32, 132, (128, 128, 256)), (300, 16, 260, (128, 128, 256))):
    R = Rp * ns
    x = (torch.randn(R, ld) * 0.7 + 0.1).cuda().requires_grad_()
    layers, k = [], ld
    for n in chans:
        layers.append((torch.randn(n, k).cuda() / np.sqrt(k), (1.0 + 0.2 * torch.randn(n)).cuda(), (0.1 * torch.randn(n)).cuda(), torch.zeros(n).cuda(), torch.ones(n).cuda()))
        k = n
    args = {}
    for flag in (True, False):
        ops._NO_FUSED_POOL = flag
        o = ops.shared_mlp_pool(x, ns, [tuple(t.clone() for t in l) for l in layers], training=True)
        args[flag] = o.grad_fn.saved_tensors[1]
    a, b = args[True], args[False]
    neq = (a != b)
    print(f"ns={ns}: arg mismatches {int(neq.sum())} of {a.numel()}; unfused sample {a[0,:6].tolist()} fused {b[0,:6].tolist()}; fused range {int(b.min())}..{int(b.max())}")
