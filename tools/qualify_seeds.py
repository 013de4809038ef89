"""Oracle-only seed qualification for tests/test_gpu_model.py (see tests/parity_tools.py): prints, per
seed, the target-assignment margin, the residual flip risk and the CPU-fp32-vs-fp64 gradient errors.
usage: python tools/qualify_seeds.py {mid|full2|full2p4|full8}   (CPU only)"""
import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(R,'tests')); sys.path.insert(0,R)
import torch, numpy as np, dataclasses
import parity_tools as P
from demf_amd.config import BackboneCfg, DeMFCfg, HeadCfg, BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES
which=sys.argv[1]
if which=='mid':
    cfg = DeMFCfg(backbone=BackboneCfg(num_points=(1024, 512, 256, 128)), head=HeadCfg(num_proposal=128, attn_dropout=0.0, ffn_dropout=0.0))
    args=(cfg, 2, 6000, ((50, 70), (25, 35), (13, 18), (7, 9)), (400, 560), (400, 551))
elif which=='full2':
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    args=(cfg, 2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2])
elif which=='full2p4':
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0, num_points=4))
    args=(cfg, 2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2])
elif which=='full8':
    cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0))
    args=(cfg, 8, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, IMG_SHAPE[:2])
cfg=args[0]
for seed in range(1,12):
    t=time.time()
    case=P.make_case(*args,seed)
    if case is None: print(seed,'ball',flush=True); continue
    batch,gtb,gtl=case
    truth=P.oracle_run(cfg,batch,gtb,gtl,seed,torch.float64)
    t1=time.time()-t
    tm=P.target_margins(cfg,truth['preds'],truth['targets'],gtb)
    cpu32=P.oracle_run(cfg,batch,gtb,gtl,seed,torch.float32,truth_taps=truth['taps'])
    risks=P.flip_risks(truth['taps'],cpu32['taps'].noise)
    top=sorted(risks.items(), key=lambda kv:-kv[1][0])[:3]
    R=top[0][1][0]
    errs={n:P.rel_l2(cpu32['grads'][n],truth['grads'][n]) for n in truth['grads'] if truth['grads'][n].norm()>1e-6*max(v.norm() for v in truth['grads'].values())}
    w=sorted(errs.items(), key=lambda kv:-kv[1])[:3]
    npos=int(truth['targets']['objectness_targets'].sum())
    print(f"seed {seed} ({t1:.0f}s+{time.time()-t-t1:.0f}s) tm {tm:.1e} npos {npos} R {R:.1e} [{top[0][0][-40:]} n={top[0][1][2]}]  cpu32 worst {w[0][1]:.1e} {w[0][0][-40:]}, median {np.median(list(errs.values())):.1e}",flush=True)
    del truth,cpu32
