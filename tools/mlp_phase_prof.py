"""Per-phase shader cycles of mlp_gemm_kernel's K loop (block (0,0), thread 0) for few-row launches.
Needs a library built with -DDEMF_MLP_PROFILE (csrc/mlp.hip).  Phases: 0 wait at the first barrier (= the
previous step's stragglers + the prefetch still in flight), 1 transform + LDS writes, 2 second barrier,
3 prefetch issue, 4 LDS reads + split + MFMAs, 5 epilogue."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import _ffi, ops
ops.set_compute_dtype(os.environ.get("MODE", "f32"))
lib = _ffi.load()
buf = (ctypes.c_longlong * 16)()
def read(reset=1):
    assert lib.demf_mlp_prof_read(buf, reset) == 0
    return list(buf)
NAMES = ["wait@barrier1", "xform+LDS write", "barrier2", "prefetch issue", "LDS read+split+MFMA", "epilogue"]
def layer(R, K, N):
    g = torch.Generator().manual_seed(0)
    return (torch.randn(R, K, generator=g).cuda(),
            [((torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_(), torch.ones(N, device="cuda", requires_grad=True),
              torch.zeros(N, device="cuda", requires_grad=True), torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")),
             ((torch.randn(N, N, generator=g) / N ** 0.5).cuda().requires_grad_(), torch.ones(N, device="cuda", requires_grad=True),
              torch.zeros(N, device="cuda", requires_grad=True), torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"))])
CASES = [(128, 256, 256, 1), (8192, 256, 256, 1), (8192, 512, 256, 1), (262144, 128, 128, 1),
         (262144, 128, 256, 32)]          # last: SA2-like pooled stack (the second launch is the pooled GEMM)
for R, K, N, ns in CASES:
    x, ls = layer(R, K, N)
    x.requires_grad_()
    for _ in range(2):
        out = ops.shared_mlp_pool(x, ns, ls, True)
    torch.cuda.synchronize(); read()
    out = ops.shared_mlp_pool(x, ns, ls, True)
    torch.cuda.synchronize(); f = read()
    out.backward(torch.randn_like(out))
    torch.cuda.synchronize(); b = read()
    for tag, v in (("forward (2 launches)", f), ("backward dx launches", b)):
        tot = sum(v[:6]) or 1
        print(f"R={R} K={K} N={N} ns={ns} {tag}: launches {v[15]}, cycles of block 0: {tot}  " +
              "  ".join(f"{n} {100 * c / tot:.0f}%" for n, c in zip(NAMES, v[:6])), flush=True)
