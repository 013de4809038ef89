"""The frozen encoder alone (6 layers over 8 x 18 609 tokens, fp32, no_grad): wall time per call; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: python tools/enc_micro.py [f32|bf16] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from demf_amd import ops
from demf_amd.modules import ImageStream
ops.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
stream = ImageStream().to(dev)
batch, _ = bench.make_batch(B, seed=1000, device=dev)
metas = batch["img_metas"]
pyr = [torch.randn(B, 256, h, w, device=dev) for h, w in ((100, 140), (50, 70), (25, 35), (13, 18))]
f = lambda: stream.img_encoder.forward_tokens(pyr, metas)
for _ in range(3): f()
torch.cuda.synchronize(); t = time.perf_counter()
n = 5
for _ in range(n): f()
torch.cuda.synchronize()
print(f"encoder (6 layers, B={B}): {(time.perf_counter() - t) / n * 1e3:.2f} ms")
