"""Measures the fp32 noise floor of the hot path: fp64 CPU oracle (truth) vs fp32 CPU oracle
vs the HIP path on identical inputs/weights.  Prints max-abs errors per output tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fixtures
from oracle.model import OracleDeMF

def run(cfg, B, N, pyr, inp, seed, dtype):
    batch = fixtures.make_scene_batch(B, N, pyr, inp, cfg.head.embed_dims, seed=seed, n_gt=4)
    m = OracleDeMF(cfg); fixtures.seed_weights(m, seed); m.train().to(dtype)
    for meta in batch["img_metas"]:
        pass
    pts = torch.from_numpy(batch["points"]).to(dtype)
    feats = [torch.from_numpy(f).to(dtype) for f in batch["img_features"]]
    with torch.no_grad():
        return batch, m.forward_head(pts, feats, batch["img_metas"])

def main():
    from demf_amd.modules import DeMFHotPath
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if which == "tiny":
        cfg = fixtures.tiny_cfg(); args = (2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT)
    else:
        from demf_amd.config import DeMFCfg, HeadCfg, PYRAMID_SHAPES, BATCH_INPUT_SHAPE
        cfg = DeMFCfg(head=HeadCfg(attn_dropout=0.0, ffn_dropout=0.0)); args = (2, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE)
    for seed in (1, 2, 3):
        batch, p64 = run(cfg, *args, seed, torch.float64)
        _, p32 = run(cfg, *args, seed, torch.float32)
        m = DeMFHotPath(cfg); fixtures.seed_weights(m, seed); m.cuda().train()
        with torch.no_grad():
            pg = m.forward_head(torch.from_numpy(batch["points"]).cuda(),
                                [torch.from_numpy(f).cuda() for f in batch["img_features"]], batch["img_metas"])
        print(f"seed {seed}: idx equal cpu32/gpu:", bool((p32['aggregated_indices'] == pg['aggregated_indices'].cpu()).all()),
              " cpu64/gpu:", bool((p64['aggregated_indices'] == pg['aggregated_indices'].cpu()).all()))
        for i in range(len(p64["decode_res_all"])):
            for k in ("center", "size", "dir_res_norm", "obj_scores", "sem_scores"):
                t = p64["decode_res_all"][i][k]
                e32 = (p32["decode_res_all"][i][k].double() - t).abs().max().item()
                eg = (pg["decode_res_all"][i][k].cpu().double() - t).abs().max().item()
                print(f"  decode{i}.{k:13s} scale {t.abs().max().item():8.3f}  |cpu32-f64| {e32:.2e}  |gpu32-f64| {eg:.2e}")
        for k in ("vote_points", "vote_features"):
            t = p64[k]
            print(f"  {k:21s} scale {t.abs().max().item():8.3f}  |cpu32-f64| {(p32[k].double()-t).abs().max().item():.2e}  |gpu32-f64| {(pg[k].cpu().double()-t).abs().max().item():.2e}")
main()
