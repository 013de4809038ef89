"""Eager forward+loss launch census: number of device kernels per region of the step
(torch.profiler, CUDA activity attributed to the enclosing record_function on the CPU side)."""
import sys, collections, torch
sys.path.insert(0, "/root/repo")
import bench
from torch.profiler import profile, ProfilerActivity, record_function
from demf_amd.config import DeMFCfg
from demf_amd.modules import DeMFHotPath
import demf_amd.modules.head as H

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DeMFHotPath(DeMFCfg()).to(dev).train()
batch, _ = bench.make_batch(8, seed=1000, device=dev)
head = model.pts_bbox_head

def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function("R:" + (label or name)):
            return f(*a, **k)
    setattr(obj, name, g)

wrap(model, "extract_pts_feat")
for n in ("prepare_image_inputs", "prepare_decoder_inputs", "get_targets", "vote_targets", "_loss_fused", "_loss", "transformer_decoder", "get_reference_points"):
    if hasattr(head, n): wrap(head, n)
wrap(head.vote_module, "forward", "vote_module")
wrap(head.vote_aggregation, "forward", "vote_aggregation")
for i, l in enumerate(head.decoder): wrap(l, "forward", f"decoder_layer{i}")
wrap(head.conv_pred0, "forward", "conv_pred0"); wrap(head.conv_pred1, "forward", "conv_pred1")

def run():
    losses = model.forward_train(batch["points"], batch["img_features"], batch["img_metas"], batch["gt_bboxes_3d"], batch["gt_labels_3d"])
    return torch.stack(list(losses.values())).sum()
for _ in range(2): run().backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("R:forward_total"):
        loss = run()
    torch.cuda.synchronize()
    with record_function("R:backward_total"):
        loss.backward()
    torch.cuda.synchronize()
ev = prof.events()
regions = [e for e in ev if e.name.startswith("R:")]
launches = [e for e in ev if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipExtLaunchKernel")]
print("launch events:", len(launches))
cnt = collections.Counter()
for l in launches:
    inner = None
    for r in regions:
        if r.time_range.start <= l.time_range.start <= r.time_range.end and r.thread == l.thread:
            if inner is None or r.time_range.start >= inner.time_range.start: inner = r
    cnt[inner.name if inner else "(backward thread / none)"] += 1
for k, v in cnt.most_common(): print(f"{v:6d}  {k}")
print("---- aten ops inside selected regions (count)")
for target in ("R:prepare_image_inputs", "R:vote_aggregation", "R:decoder_layer0"):
    rs = [r for r in regions if r.name == target]
    ops_c = collections.Counter()
    for e in ev:
        if e.name.startswith("aten::") or e.name.startswith("_") or "Function" in e.name:
            if any(r.time_range.start <= e.time_range.start <= r.time_range.end and r.thread == e.thread for r in rs):
                ops_c[e.name] += 1
    print(target, dict(ops_c.most_common(22)))
