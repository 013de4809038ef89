"""HIP bf16 compute mode against the bf16-EMULATING oracle (oracle/emulate.py) and against the plain fp64
oracle, per stage: how much of the bf16 deviation the emulation explains.  usage: python tools/bf16_emulation_check.py [tiny|mid|full2|full8]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_tools as P
import test_gpu_model as M
from oracle import fixtures
from demf_amd import ops
from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, BackboneCfg, DeMFCfg, HeadCfg

which = sys.argv[1] if len(sys.argv) > 1 else "mid"
if which == "tiny":
    cfg = fixtures.tiny_cfg()
    args = (cfg, 2, 1024, ((32, 44), (16, 22), (8, 11), (4, 6)), (256, 352), None)
elif which == "mid":
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)), head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    args = (cfg, 2, 6000, *M.MID)
else:
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    args = (cfg, 2 if which == "full2" else 8, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2])
t0 = time.time()
case = P.qualified_case(*args, seeds=tuple(range(1, 12)))
t1 = time.time()
E64, E32 = P.emulated_runs(cfg, case)
t2 = time.time()
print("oracle %.1f s, emulated oracle runs %.1f s" % (t1 - t0, t2 - t1))
T = case["truth"]
ops.set_compute_dtype("bf16")
G = M._gpu_run(cfg, case)
ops.set_compute_dtype("f32")
G32 = M._gpu_run(cfg, case)
rel = lambda a, t: ((a.detach().double().cpu() - t.double()).norm() / t.double().norm()).item()
for k in ("seed_indices", "aggregated_indices"):
    print(k, "HIP-bf16 == emu64:", bool((G["preds"][k].cpu() == E64["preds"][k]).all()), " == truth:", bool((G["preds"][k].cpu() == T["preds"][k]).all()))
for k in ("vote_points", "vote_features", "aggregated_points"):
    print("%-18s HIPbf16-vs-emu64 %.2e   emu32-vs-emu64 %.2e   HIPbf16-vs-fp64 %.2e   HIPf32-vs-fp64 %.2e" %
          (k, rel(G["preds"][k], E64["preds"][k]), rel(E32["preds"][k], E64["preds"][k]), rel(G["preds"][k], T["preds"][k]), rel(G32["preds"][k], T["preds"][k])))
for i, d in enumerate(E64["preds"]["decode_res_all"]):
    for k in ("center", "size", "obj_scores", "sem_scores"):
        print("decode%d.%-10s HIPbf16-vs-emu64 %.2e   emu32-vs-emu64 %.2e   HIPbf16-vs-fp64 %.2e" %
              (i, k, rel(G["preds"]["decode_res_all"][i][k], d[k]), rel(E32["preds"]["decode_res_all"][i][k], d[k]),
               rel(G["preds"]["decode_res_all"][i][k], T["preds"]["decode_res_all"][i][k])))
for k in E64["losses"]:
    print("loss %-16s HIPbf16 %.6f emu64 %.6f emu32 %.6f fp64 %.6f" % (k, G["losses"][k].item(), E64["losses"][k].item(), E32["losses"][k].item(), T["losses"][k].item()))
rows = []
for n, g in E64["grads"].items():
    if g.norm().item() < 1e-6 * max(v.norm().item() for v in E64["grads"].values()):
        continue
    rows.append((n, rel(G["grads"][n], g), rel(E32["grads"][n], g), rel(G["grads"][n], T["grads"][n])))
rows.sort(key=lambda r: -r[1])
print("gradients: median HIPbf16-vs-emu64 %.2e, emu32-vs-emu64 %.2e, HIPbf16-vs-fp64 %.2e" % tuple(np.median([r[i] for r in rows]) for i in (1, 2, 3)))
for r in rows[:12]:
    print("  %-70s %.2e %.2e %.2e" % r)
def cos(A, B):
    a = torch.cat([A[n].double().cpu().reshape(-1) for n in B]); b = torch.cat([B[n].double().reshape(-1) for n in B])
    return (a @ b / (a.norm() * b.norm())).item()
print("whole-gradient cosine: HIPbf16.emu64 %.4f  emu32.emu64 %.4f  HIPbf16.fp64 %.4f" % (cos(G["grads"], E64["grads"]), cos(E32["grads"], E64["grads"]), cos(G["grads"], T["grads"])))
