"""Secondary, end-to-end figure: the frozen image stream forward (ResNet-50 ->
ChannelMapper -> 6-layer deformable encoder, fp32, no_grad) + the trainable hot-path step, 8 scenes
of (20000 points, 800x1120 image) on one MI355X.  Not the bench.py line."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd import engine
from demf_amd.config import DeMFCfg, BATCH_INPUT_SHAPE
from demf_amd.modules import DeMFHotPath, ImageStream

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
stream = ImageStream().to(dev)
tr = engine.Trainer(model)
batch, _ = bench.make_batch(B, seed=1000, device=dev)
img = torch.randn(B, 3, *BATCH_INPUT_SHAPE, device=dev)
metas = batch["img_metas"]

def sync_time(f, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

t0 = time.perf_counter()
tok = stream.tokens(img, metas); torch.cuda.synchronize()
print(f"first image-stream call (MIOpen find etc.): {time.perf_counter() - t0:.1f} s; tokens {tuple(tok['tokens'].shape)}")
for _ in range(2): stream.tokens(img, metas)
from demf_amd import ops as _ops
_ops.LIBRARY_FALLBACK = True       # the library convolutions are this line's A/B reference
print(f"backbone+neck      : {sync_time(lambda: stream.pyramid(img), 5):7.2f} ms  (library convolutions: {sync_time(lambda: stream._pyramid(img), 5):7.2f} ms)")
pyr = stream.pyramid(img)
print(f"encoder (6 layers) : {sync_time(lambda: stream.img_encoder.forward_tokens(pyr, metas), 5):7.2f} ms")
ims = sync_time(lambda: stream.tokens(img, metas), 5)
print(f"image stream total : {ims:7.2f} ms")
# hot path fed with channels-last tokens (static buffer), graph replay
static_tok = dict(tokens=tok["tokens"].clone(), spatial=tok["spatial"])
batch["img_features"] = static_tok
step = tr.capture(batch)
for _ in range(3): step()
hp = sync_time(step, 20)
def e2e():
    static_tok["tokens"].copy_(stream.tokens(img, metas)["tokens"])
    step()
for _ in range(2): e2e()
ee = sync_time(e2e, 10)
print(f"hot-path step (tokens in): {hp:6.2f} ms = {B / hp * 1e3:6.1f} scenes/s")
print(f"end-to-end step          : {ee:6.2f} ms = {B / ee * 1e3:6.1f} scenes/s")

# pipelined: the frozen stream of batch k+1 on a side stream while the hot-path step of batch k runs (they share
# nothing: the stream is no_grad and frozen); the step's launch-latency gaps are filled by the stream's kernels
side = engine.concurrent_stream()
nxt = {}
def e2e_pipe():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nxt["tok"] = stream.tokens(img, metas)["tokens"]
    step()
    torch.cuda.current_stream().wait_stream(side)
    static_tok["tokens"].copy_(nxt["tok"])
for _ in range(2): e2e_pipe()
ep = sync_time(e2e_pipe, 10)
print(f"end-to-end, pipelined    : {ep:6.2f} ms = {B / ep * 1e3:6.1f} scenes/s")
