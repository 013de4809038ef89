"""How much does the concurrent furthest-point chain cost the step?  The captured step (no pre-pass)
is replayed while k scenes of 20000 -> 2048 FPS run on a second stream."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from demf_amd import engine, ops
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
tr = engine.Trainer(model)
batch, _ = bench.make_batch(8, seed=1000, device=dev)
step = tr.capture(batch, prefetch_geometry=False)
side = torch.cuda.Stream()
pts = batch["points"][..., :3].contiguous()
for k in (0, 1, 2, 4, 8):
    def it():
        if k:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.furthest_point_sample(pts[:k], 2048)
        step()
        if k:
            torch.cuda.current_stream().wait_stream(side)
    for _ in range(5): it()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): it()
    torch.cuda.synchronize()
    print(f"{k} scenes of FPS alongside: {(time.perf_counter() - t) / 30 * 1e3:.2f} ms/step")
