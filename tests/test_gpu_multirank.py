"""The multi-rank DEVICE path on one GPU: two ranks share cuda:0
(DEMF_SHARE_DEVICE=1) and talk through gloo (DEMF_DIST_BACKEND=gloo), so that everything a real
8-GPU run does above the transport - Trainer's broadcast, the SUM all-reduce of the flat
gradient buffer, the norm of the SUM, 1/world folded into demf_adamw_f32 (engine.py `_update`) and
bench.py's barrier / max-over-ranks / single JSON line - runs on hardware in the round-end test
tier.  Reference shape: tools/dist_train.sh:8-9, train.py:56-63."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fixtures

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_and_batch(rank):
    from demf_amd import synthetic
    from demf_amd.modules import DeMFHotPath
    cfg = fixtures.tiny_cfg()
    model = DeMFHotPath(cfg)
    fixtures.seed_weights(model, 7)
    model.cuda().train()
    raw = synthetic.make_scene_batch(2, 1024, fixtures.TINY_PYRAMID, fixtures.TINY_INPUT,
                                     cfg.head.embed_dims, seed=50 + rank, n_gt=4)
    batch = dict(points=torch.from_numpy(raw["points"]).cuda(),
                 img_features=[torch.from_numpy(f).cuda() for f in raw["img_features"]],
                 img_metas=raw["img_metas"],
                 gt_bboxes_3d=[torch.from_numpy(b).cuda() for b in raw["gt_boxes"]],
                 gt_labels_3d=[torch.from_numpy(l).cuda() for l in raw["gt_labels"]])
    return model, batch


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), DEMF_SHARE_DEVICE="1",
                      DEMF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from demf_amd import engine
    engine.init_distributed()
    assert dist.get_backend() == "gloo" and torch.cuda.current_device() == 0
    model, batch = _model_and_batch(rank)
    if rank == 1:                       # replicas start different: Trainer must broadcast rank 0
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    tr = engine.Trainer(model, lr=1e-3)
    assert tr.fused, "the device path must be the flat fused AdamW"
    grads = []
    for _ in range(STEPS):
        tr._fwd_bwd(batch)
        grads.append(tr.flat.flat.clone().cpu())     # this rank's local gradient, pre-reduce
        tr._update()
    torch.cuda.synchronize()
    torch.save(dict(params=tr.opt.flat.clone().cpu(), grads=grads), os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_hand_averaged_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["params"], b["params"]), "replicas diverged"       # (i) bit-identical
    # (ii) single process: feed the two ranks' recorded local gradients, averaged by hand, through
    # the same fused optimizer (world 1).  The forward of step k depends on the parameters after
    # step k-1, which (i)+(ii) pin inductively, so the recorded gradients ARE the ones a
    # single-process run would see.
    from demf_amd import engine
    model, _ = _model_and_batch(0)
    tr = engine.Trainer(model, lr=1e-3)
    for k in range(STEPS):
        tr.flat.flat.copy_(((a["grads"][k].double() + b["grads"][k].double()) / world).float().cuda())
        tr._update()
    torch.cuda.synchronize()
    got, want = a["params"].double(), tr.opt.flat.cpu().double()
    # SUM-then-scale-in-kernel vs mean-by-hand differ by fp32 rounding of the gradient only
    assert ((got - want).abs().max() / want.abs().max()).item() < 1e-5
    # and the ranks really saw different data
    assert not torch.equal(a["grads"][0], b["grads"][0])


def test_bench_two_ranks_prints_one_json_line():
    """`torchrun --nproc-per-node 2 bench.py --gpus 2` (the driver's launch line) on one GPU under
    the share-device hook: rank 0 prints exactly one JSON line with n_gpus 2 and the whole-job
    aggregate; small batch so both ranks fit one GPU comfortably."""
    env = dict(os.environ, DEMF_SHARE_DEVICE="1", DEMF_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["scenes_per_gpu"] == 2
    # whole-job aggregate: 2 ranks x 2 scenes x 3 steps over the max-over-ranks time
    assert abs(out["value"] - 2 * 2 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out            # rank 0 at N=1 only


def test_plain_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun and without RANK in the environment: bench.py
    re-executes itself under `python -m torch.distributed.run --nproc-per-node 2` (the reference's
    launcher is the same one line: tools/dist_train.sh:8-9) and rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(DEMF_SHARE_DEVICE="1", DEMF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT, MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"


def test_bench_eight_ranks_on_one_gpu():
    """The driver's 8-GPU launch line, eight ranks sharing one GPU through gloo (one scene per rank):
    JSON shape, n_gpus, the weak-scaling arithmetic (value = all ranks' scenes over the max-over-ranks
    time), the all-reduce's own time, and that the eight replicas differ where they must - data seeds and
    dropout streams - while the reference's launcher gives every rank its own shard
    (tools/dist_train.sh:8-9, train.py:56-63)."""
    env = dict(os.environ, DEMF_SHARE_DEVICE="1", DEMF_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "2",
           "--batch", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["scenes_per_gpu"] == 1 and out["config"]["parallelism"] == "dp8"
    assert abs(out["value"] - 8 * 1 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert out["allreduce_us"] > 0.0
    assert "cpu_baseline" not in out and "secondary" not in out
    ranks = out["ranks"]
    assert [r_["rank"] for r_ in ranks] == list(range(8))
    assert len({r_["dropout_seed"] for r_ in ranks}) == 8
    seeds = [sd for r_ in ranks for sd in r_["batch_seeds"]]
    assert len(set(seeds)) == len(seeds)
