#!/usr/bin/env python
"""DeMF fusion hot-path benchmark (BASELINE.json metric: fwd+bwd scenes/s at 20 k points +
530x730 RGB -> 800x1120 pyramid), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the trainable hot path over one synthetic batch of 8 scenes per GPU
(BASELINE.json configs[2]): PointNet++ backbone -> vote -> aggregation -> DeMF fusion layer ->
heads -> loss -> backward -> one RCCL gradient all-reduce -> clip -> AdamW.  The frozen,
no_grad image stream is outside the path; its output pyramid is a resident input.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from demf_amd import engine, synthetic  # noqa: E402
from demf_amd.config import BATCH_INPUT_SHAPE, IMG_SHAPE, PYRAMID_SHAPES, DeMFCfg  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
CLOCK_GHZ = 2.4
# The kernel the headline `roofline` object prices: the single longest launch on the step's main stream -
# since round 4 the BACKWARD of SA1's last shared-MLP layer (R = B*2048*64 grouped rows, 128 -> 64
# channels; csrc/mlp_bwd.hip mlp_bwd_pool_kernel: the pooled layer's input gradient, weight gradient and
# the previous layer's BN sums without the layer's (R,128) output ever stored).  `prefix`: how rocprofv3
# prints the instantiation - matched BY PREFIX against the committed PMC / stats files.
DOMINANT_KERNEL = {"f32_native": "mlp_bwd_pool_kernel<0>", "f32x3": "mlp_bwd_pool_kernel<2>",
                   "f32h2": "mlp_bwd_pool_kernel<3>", "bf16": "mlp_bwd_pool_kernel<1>"}
# what "f32" means in this process: the fp32 MFMA (DEMF_F32_NATIVE=1), three bf16 terms everywhere (DEMF_F16_TERMS=0),
# or - the default - three bf16 terms with the SA stacks' kernels on two fp16 terms (ops.set_compute_dtype)
F32_MEANS = "f32_native" if int(os.environ.get("DEMF_F32_NATIVE", "0") or 0) else \
    ("f32h2" if int(os.environ.get("DEMF_F16_TERMS", "1") or 0) and int(os.environ.get("DEMF_F16_TERMS_BWD", "1") or 0)
     else "f32x3")
DOMINANT_KERNEL["f32"] = DOMINANT_KERNEL[F32_MEANS]
# its forward twin (the round-3 headline): SA1 layer 3, 64 -> 128 + BN statistics + max-pool, no output store
SA1_FWD_KERNEL = {"f32_native": "mlp_fwd_res_kernel<4, 2, true, 0", "f32x3": "mlp_fwd_res_kernel<4, 2, true, 2",
                  "f32h2": "mlp_fwd_res_kernel<4, 2, true, 3", "bf16": "mlp_fwd_res_kernel<4, 2, true, 1"}
SA1_FWD_KERNEL["f32"] = SA1_FWD_KERNEL[
    "f32_native" if F32_MEANS == "f32_native" else
    ("f32h2" if int(os.environ.get("DEMF_F16_TERMS", "1") or 0) else "f32x3")]
MFMA_PATH = {"f32_native": "v_mfma_f32_32x32x2_f32",
             "f32x3": "fp32 operands split exactly into 3 bf16 terms, 6 products on "
                      "v_mfma_f32_32x32x16_bf16, fp32 accumulate (error vs fp64 = the fp32 MFMA's, "
                      "tests/test_gpu_split.py)",
             "f32h2": "fp32 operands as 2 fp16 terms, 3 products on v_mfma_f32_32x32x16_f16 in the SA stacks' forward, "
                      "(256|128,128) backward and pooled-last-layer backward kernels (2^-22 per operand; gradient "
                      "operands scaled by a power of two per slab: csrc/common.h); every other kernel: 3 bf16 terms, "
                      "6 products on v_mfma_f32_32x32x16_bf16; fp32 accumulate either way",
             "bf16": "operands rounded to bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulate"}
MFMA_PATH["f32"] = MFMA_PATH[F32_MEANS]
# algorithmic (compulsory) HBM bytes and FLOPs of ONE scene, fwd + bwd
ALGO_BYTES_PER_SCENE = 190e6
ALGO_FLOP_PER_SCENE = 46e9


def newest_profile(pattern):
    """Newest committed profiles/<pattern> (round tags sort lexicographically: r04_b > r04_a > r03_final)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return hits[-1] if hits else None


LIVE_PMC = {}       # --pmc-live: {"FETCH_SIZE": summary csv, "WRITE_SIZE": summary csv} measured by THIS run


def pmc_live(args):
    """--pmc-live: the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, no tracing
    domains) of this same command at 3 + 2 steps, summarised per kernel by tools/pmc_summary.py into a scratch
    directory - so that `roofline.traffic` and `roofline_step.traffic_step` of this line are observed by this run
    instead of read from the committed profiles/.  Best effort: any failure leaves the committed files in charge."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 not found"
    scratch = tempfile.mkdtemp(prefix="demf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(scratch, c)
            cmd = ["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-secondary",
                   "--dtype", args.dtype, "--batch", str(args.batch), "--msda-points", str(args.msda_points),
                   "--cloud", args.cloud, "--no-pmc-live"]
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=150, capture_output=True, check=True)
            hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not hits:
                return "no counter_collection.csv for " + c
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), hits[0], c],
                                 capture_output=True, text=True, timeout=120, check=True).stdout
            path = os.path.join(scratch, "pmc_%s.csv" % c)
            with open(path, "w") as fh:
                fh.write(out)
            LIVE_PMC[c] = path
    except Exception as exc:            # noqa: BLE001 - the committed passes remain the source
        LIVE_PMC.clear()
        return repr(exc)[:200]
    return None


def pmc_per_launch(kernel_prefix, which="max", required=False):
    """HBM bytes per launch of the kernel whose name STARTS WITH ``kernel_prefix`` (after the
    ``void demf::`` decoration), from the NEWEST committed PMC summaries
    profiles/r*_pmc_{FETCH,WRITE}_SIZE.csv (separate --pmc passes of this same command, summarised by
    tools/pmc_summary.py: columns kernel, calls, mean KiB, max KiB).
    FETCH_SIZE is doubled: gfx950 counts 64 B per 128-B request for wide coalesced reads
    (MI355X_MICROARCH.md, HBM section; calibrated on the SA1 dx GEMM).
    ``which``: 'max' = the largest launch of that instantiation (the SA1-sized one), 'mean'.
    Only the newest pair of files counts: an older round's row for a kernel that has since changed is
    not evidence.  ``required``: raise instead of returning (None, source) when that pair has no row.
    -> (bytes | None, source file | None)"""
    import csv
    live = len(LIVE_PMC) == 2
    f = LIVE_PMC["FETCH_SIZE"] if live else newest_profile("r*_pmc_FETCH_SIZE.csv")
    if f is None or not os.path.exists(f.replace("FETCH_SIZE", "WRITE_SIZE")):
        if required:
            raise RuntimeError("no committed PMC passes under profiles/")
        return None, None
    vals = []
    for path in (f, f.replace("FETCH_SIZE", "WRITE_SIZE")):
        hit = None
        with open(path) as fh:
            for row in csv.reader(fh):
                if len(row) < 4:
                    continue
                name = row[0]
                for deco in ("void ", "demf::"):
                    if name.startswith(deco):
                        name = name[len(deco):]
                if name.startswith(kernel_prefix):
                    hit = float(row[3] if which == "max" else row[2])
                    break
        vals.append(hit)
    src = "live: two rocprofv3 --pmc passes of this run (bench.py --pmc-live)" if live else os.path.relpath(f, ROOT)
    if None in vals:
        if required:
            raise RuntimeError("%s has no row for kernel prefix %r" % (src, kernel_prefix))
        return None, src
    return (2.0 * vals[0] + vals[1]) * 1024.0, src


def pmc_per_step():
    """HBM bytes of one whole training step (every kernel of it) from the newest committed PMC passes:
    the __TOTAL_PER_STEP__ row tools/pmc_summary.py writes (FETCH doubled as above).
    -> (bytes | None, source | None)"""
    return pmc_per_launch("__TOTAL_PER_STEP__", which="mean")


def make_batch(B, seed, device, cloud="uniform", gt_counts=None):
    raw = synthetic.make_scene_batch(B, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, 256, seed=seed,
                                     n_gt=8, img_shape=IMG_SHAPE[:2], scale_factor=1.5094, cloud=cloud,
                                     gt_counts=gt_counts)
    return dict(points=torch.from_numpy(raw["points"]).to(device),
                img_features=[torch.from_numpy(f).to(device) for f in raw["img_features"]],
                img_metas=raw["img_metas"],
                gt_bboxes_3d=[torch.from_numpy(b).to(device) for b in raw["gt_boxes"]],
                gt_labels_3d=[torch.from_numpy(l).to(device) for l in raw["gt_labels"]]), raw


def top_kernels():
    """The newest committed per-kernel digest (tools/top_kernels.py, written by tools/prof.sh from a
    rocprofv3 kernel trace + the PMC passes of this same command) -> (dict | None, source | None)."""
    f = newest_profile("r*_top_kernels.json")
    if f is None:
        return None, None
    with open(f) as fh:
        return json.load(fh), os.path.relpath(f, ROOT)


class KernelTimer:
    """HIP-event timing of one kernel family, live inside the timed region, on the stream the
    kernel is launched on (demf_amd.ops launches on torch's current stream)."""

    def __init__(self, ops_module, fn_name):
        self.ops, self.fn_name = ops_module, fn_name
        self.orig = getattr(ops_module, fn_name)
        self.events, self.enabled = [], False
        timer = self

        def wrapped(*a, **k):
            if not timer.enabled:
                return timer.orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = timer.orig(*a, **k)
            e.record()
            timer.events.append((s, e, a[0].shape if hasattr(a[0], "shape") else None))
            return out
        setattr(ops_module, fn_name, wrapped)

    def results(self):
        return [(s.elapsed_time(e), shp) for s, e, shp in self.events]


class FfiTimer:
    """HIP-event timing of one C-ABI entry point (demf_amd._ffi.call) for calls whose leading
    integer arguments match ``match`` - used for the largest kernel on the step's critical path."""

    def __init__(self, ffi_module, symbol, match):
        self.ffi, self.symbol, self.match = ffi_module, symbol, tuple(match)
        self.orig = ffi_module.call
        self.events, self.enabled = [], False
        timer = self

        def call(name, *args):
            if not (timer.enabled and name == timer.symbol and tuple(args[:len(timer.match)]) == timer.match):
                return timer.orig(name, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = timer.orig(name, *args)
            e.record()
            timer.events.append((s, e))
            return out
        ffi_module.call = call

    def mean_ms(self):
        v = [s.elapsed_time(e) for s, e in self.events]
        return float(np.mean(v)) if v else float("nan")


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (oracle/model.py: a port of the reference path, checker-only code) timed
    on this host, ONE full-size scene per pass: leg (a) SA backbone forward,
    leg (b) full hot-path forward, leg (c) fwd + loss + bwd = the figure `value` reports.
    Threads: min(32, host cores) - on the 256-thread GPU-box host 32 threads measured faster than
    all of them (round 2: the probe that found that out cost ~200 s per run and is gone);
    `host_cores` in the result states what the machine has."""
    from oracle import fixtures
    from oracle.model import OracleDeMF
    host = os.cpu_count() or 1
    cfg = DeMFCfg()
    raw = synthetic.make_scene_batch(1, 20000, PYRAMID_SHAPES, BATCH_INPUT_SHAPE, 256, seed=0,
                                     n_gt=8, img_shape=IMG_SHAPE[:2], scale_factor=1.5094)
    m = OracleDeMF(cfg)
    fixtures.seed_weights(m, 0)
    m.train()
    pts = torch.from_numpy(raw["points"])
    feats = [torch.from_numpy(f) for f in raw["img_features"]]
    gtb = [torch.from_numpy(b) for b in raw["gt_boxes"]]
    gtl = [torch.from_numpy(l) for l in raw["gt_labels"]]

    def leg_c():
        m.zero_grad()
        losses, _, _ = m.forward_train(pts, feats, raw["img_metas"], gtb, gtl)
        sum(losses.values()).backward()

    def leg_a():
        with torch.no_grad():
            m.pts_backbone(pts)

    def leg_b():
        with torch.no_grad():
            m.forward_head(pts, feats, raw["img_metas"])

    def set_threads(n):
        torch.set_num_threads(n)
        os.environ["OMP_NUM_THREADS"] = str(n)

    set_threads(min(32, host))

    def timed(fn, budget, max_n):
        fn()
        ts = []
        t_end = time.perf_counter() + budget
        while len(ts) < max_n and (time.perf_counter() < t_end or len(ts) < 3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    ta = timed(leg_a, 0.15 * seconds_budget, 20)
    tb = timed(leg_b, 0.15 * seconds_budget, 20)
    tc = timed(leg_c, 0.7 * seconds_budget, 50)
    n, dt = len(tc), float(sum(tc))
    return dict(value=n / dt, unit="scenes/s", cores=torch.get_num_threads(), host_cores=host,
                kind="port",
                sample=f"{n} x (1 scene, 20000 pts + 800x1120 pyramid, fwd+loss+bwd) in {dt:.1f} s",
                legs={"a_sa_backbone_fwd": dict(scenes_per_s=1.0 / float(np.median(ta)), n=len(ta),
                                                min_ms=1e3 * min(ta), median_ms=1e3 * float(np.median(ta))),
                      "b_hot_path_fwd": dict(scenes_per_s=1.0 / float(np.median(tb)), n=len(tb),
                                             min_ms=1e3 * min(tb), median_ms=1e3 * float(np.median(tb))),
                      "c_fwd_loss_bwd": dict(scenes_per_s=1.0 / float(np.median(tc)), n=len(tc),
                                             min_ms=1e3 * min(tc), median_ms=1e3 * float(np.median(tc)))})


def sa_path_ms(model, device, B, seed=0):
    """BASELINE configs[1]: the PointNet++ SA path alone (FPS / ball query / grouping + the shared MLPs
    of the 4 SA + 2 FP levels), fp32, pre-pass NOT pipelined (every pass pays its own FPS chain):
    -> (forward ms, forward+backward ms, coordinate pre-pass alone ms)."""
    pts, _ = make_batch(B, seed, device)
    pts = pts["points"]
    bb = model.pts_backbone

    def fwd():
        with torch.no_grad():
            return bb(pts)

    def fwdbwd():
        out = bb(pts)
        torch.autograd.grad(out["fp_features"][-1].sum(), [p for p in bb.parameters() if p.requires_grad])

    def geo():
        return bb.index_geometry(pts)

    def timed(f, n=10):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    return timed(fwd), timed(fwdbwd), timed(geo)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU (BASELINE configs[2])")
    ap.add_argument("--cloud", choices=("uniform", "clustered"), default="uniform",
                    help="synthetic cloud of the headline loop (BASELINE.md section 3: uniform volume / 20 "
                         "Gaussian blobs); the other one is reported under `secondary`")
    ap.add_argument("--msda-points", type=int, default=2,
                    help="sampling points per level of the fusion attention: 2 = reference config "
                         "(demf_votenet.py:83), 4 = BASELINE.json's wording; secondary figure only")
    ap.add_argument("--dtype", choices=("f32", "f32_native", "f32x3", "bf16"), default="f32",
                    help="compute mode of the dense MFMA kernels.  f32 = the reference's precision "
                         "(headline, BASELINE configs[2]): fp32 results; the shared-MLP GEMMs split each "
                         "fp32 operand exactly into three bf16 terms and run the six significant "
                         "products on the bf16 MFMA; the SA stacks' forward kernels and their (256,128) backward "
                         "take TWO fp16 terms and three products instead (2^-22 per operand, gradient operand "
                         "scaled per slab; f32x3 or DEMF_F16_TERMS=0: three bf16 terms everywhere; "
                         "DEMF_F32_NATIVE=1 or f32_native: the fp32 MFMA itself).  bf16 = configs[3] (operands rounded to bf16, fp32 "
                         "accumulate, fp32 storage / statistics / indices / losses)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--resident", action="store_true",
                    help="time the replay of ONE resident batch (rounds 1-3's headline) instead of the "
                         "training loop that loads a different batch every step")
    ap.add_argument("--stream-priority", action="store_true",
                    help="run the step graph on a high-priority HIP stream (experiment)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute the FPS/ball-query pre-pass inside the step instead of pipelining it")
    ap.add_argument("--cu-mask", action="store_true",
                    help="pin the FPS pre-pass to its own CUs (hipExtStreamCreateWithCUMask); measured: "
                         "no effect under hipGraph replay, off by default")
    ap.add_argument("--allreduce-stub-us", type=int, default=0,
                    help="replace the gradient all-reduce by a spin kernel of this many microseconds "
                         "(measures the engine's communication / compute overlap on one GPU)")
    ap.add_argument("--allreduce-overlap", action="store_true",
                    help="issue the collective on a communication stream and defer norm + AdamW to the start of the "
                         "next step (default: on the step's own stream, the update right behind it)")
    ap.add_argument("--double-buffer", action="store_true",
                    help="load each batch into the idle one of TWO sets of static input buffers on an input stream "
                         "while the previous step runs (engine.DoubleBufferedStep; measured SLOWER on MI355X / ROCm "
                         "7.2: 5.15 vs 5.01 ms - a third active hardware queue costs more than the 0.1 ms it hides)")
    ap.add_argument("--no-pmc-live", action="store_true",
                    help="read the HBM traffic fields from the committed profiles/ instead of re-measuring them with "
                         "two rocprofv3 --pmc passes of this command (default at one GPU, ~10 s; falls back to the "
                         "committed files on any failure)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (resident / bf16 / P=4 / clustered / B=16 step times, SA path)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: launch the ranks ourselves, the way the reference's
        # tools/dist_train.sh:8-9 does (python -m torch.distributed.launch --nproc_per_node=$GPUS),
        # one process per GPU over RCCL; rank 0 of the children prints the JSON line
        import socket
        import subprocess
        port = os.environ.get("MASTER_PORT")
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank, local, world = engine.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    device = torch.device("cuda", local)

    from demf_amd import ops
    from demf_amd.modules import DeMFHotPath
    ops.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    cfg = DeMFCfg()
    if args.msda_points != cfg.head.num_points:
        import dataclasses
        cfg = dataclasses.replace(cfg, head=dataclasses.replace(cfg.head, num_points=args.msda_points))
    model = DeMFHotPath(cfg).to(device).train()
    trainer = engine.Trainer(model)
    trainer.allreduce_stub_us = args.allreduce_stub_us
    if args.allreduce_overlap:
        trainer.allreduce_overlap = True
    # The training loop's data: NB distinct batches per rank (weak scaling: B scenes per GPU, distinct
    # seeds per rank), every one with its own per-scene GT counts - a real loader never repeats a
    # count signature, and the per-batch input path (target padding, meta refresh, static-buffer
    # copies, the next cloud's pre-pass) is part of the step a training loop pays.  All of it is resident
    # in HBM before the timed region.
    NB, MAX_GT = 4, 8
    cnt_rng = np.random.default_rng(4242 + rank)
    batches = [make_batch(args.batch, seed=1000 + rank + 7919 * i, device=device, cloud=args.cloud,
                          gt_counts=None if i == 0 else cnt_rng.integers(0, MAX_GT + 1, size=args.batch))[0]
               for i in range(NB)]
    batch = batches[0]

    fps_timer = KernelTimer(ops, "furthest_point_sample")
    # the longest launch on the critical path (the FPS chain runs underneath the step on a side stream):
    # SA1's last shared-MLP layer backward, and its forward twin
    from demf_amd import _ffi
    sa1_rows = args.batch * 2048 * 64
    mlp_timer = FfiTimer(_ffi, "demf_mlp_bwd_pool", (sa1_rows, 128, 64, 64))
    fwd_timer = FfiTimer(_ffi, "demf_mlp_gemm_fwd_pool_bn_st", (sa1_rows, 64, 128))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if not args.no_graph and args.cu_mask:
        # FPS pre-pass of the next batch on B dedicated CUs, the step on the other 256-B
        main_s, trainer.side_stream = engine.cu_masked_streams(list(range(args.batch)))
        torch.cuda.set_stream(main_s)
    if not args.no_graph and args.stream_priority:
        # experiment: step graph on a high-priority stream, pre-pass on a normal one
        main_s = torch.cuda.Stream(priority=-1)
        trainer.side_stream = torch.cuda.Stream(priority=0)
        main_s.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_s)
    replay = None if args.no_graph else \
        trainer.capture(batch, prefetch_geometry=not args.no_prefetch, max_gt=MAX_GT)
    # --double-buffer: the loading loop on double-buffered static inputs (engine.DoubleBufferedStep): batch k+1 is
    # loaded on an input stream while step k runs.  The pre-pass pipeline is shared with the single-set replay above.
    replay2 = None
    if replay is not None and not args.resident and args.double_buffer and not args.no_prefetch:
        replay2 = trainer.capture_double(batches[1], batches[2], prefetch_geometry=True, max_gt=MAX_GT,
                                         geo_pipe=replay.geo, dry=True)
    k = [0]
    loop_replay = [None]        # (the all-reduce stub legs swap in a capture whose update stays eager)

    def step_resident():
        return trainer.step(batch) if replay is None else replay()

    def step_loop():
        """One step of the training loop: batch k is loaded into the captured step's static buffers
        (its pre-pass was launched underneath step k-1), the step runs, batch k+1's cloud is handed to
        the pipelined pre-pass."""
        k[0] += 1
        cur, nxt = batches[k[0] % NB], batches[(k[0] + 1) % NB]
        if replay is None:
            return trainer.step(cur)
        r = replay2 if replay2 is not None else (loop_replay[0] or replay)
        r.load(cur)
        return r(next_points=nxt["points"])

    step = step_resident if args.resident else step_loop
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    repeats = []
    for _ in range(2):
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        repeats.append(1000.0 * (time.perf_counter() - t1) / args.steps)
    trainer.flush()               # (overlapped collective: the last step's norm + AdamW are still owed)
    # a throughput figure over non-finite arithmetic would be meaningless: refuse to report it
    if not bool(torch.isfinite(trainer.flat.flat).all()) or \
            not all(bool(torch.isfinite(p).all()) for p in model.parameters()):
        raise RuntimeError("non-finite gradients/parameters after the timed steps")
    # the collective's own time (HIP events around the all-reduce of the flat 8.76 MB buffer)
    allreduce_us = None
    if world > 1 or args.allreduce_stub_us > 0:
        trainer.allreduce_events = []
        for _ in range(min(args.steps, 10)):
            step()
        torch.cuda.synchronize()
        ev = trainer.allreduce_events
        trainer.allreduce_events = None
        if ev:
            allreduce_us = 1e3 * float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # dominant-kernel duration: HIP events around the same launches, same inputs, same stream,
    # in an eager pass right after the timed region (a graph replay cannot host per-kernel
    # events); profiles/ holds the rocprofv3 figure for the same kernel inside the replays
    fps_timer.enabled = mlp_timer.enabled = fwd_timer.enabled = True
    for _ in range(min(args.steps, 5)):
        trainer.step(batch)
    torch.cuda.synchronize()
    fps_timer.enabled = mlp_timer.enabled = fwd_timer.enabled = False
    if not mlp_timer.events:
        raise RuntimeError("bench: demf_mlp_bwd_pool was not launched with R=%d - the roofline object would "
                           "describe a kernel this step does not run" % sa1_rows)

    def time_steps(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t1) / n

    secondary = {}
    if world == 1 and not args.no_secondary and not args.no_graph:
        # (a) the other way of running the same captured step: ONE resident batch replayed (the headline
        # of rounds 1-3) if `value` is the loading loop, and vice versa
        if args.resident:
            secondary["ms_per_step_with_load"] = time_steps(step_loop, args.steps)
        else:
            secondary["ms_per_step_resident"] = time_steps(step_resident, args.steps)
        # (a2) the collective's overlap, measured on ONE GPU: a 100 us spin kernel (demf_spin_us) stands in for the
        # all-reduce (the size class RCCL needs for 8.76 MB over xGMI) - on the step's own stream
        # with the update right behind it (serial), and on the communication stream with the update deferred behind
        # the next batch's input path (the engine's default for world > 1)
        if not args.resident and args.allreduce_stub_us == 0 and not os.environ.get("DEMF_BENCH_SKIP_AR_STUB"):
            # (the headline capture holds the optimizer update inside its graph; a collective sits between
            # backward and update, so these legs replay a second capture whose update stays eager)
            loop_replay[0] = trainer.capture(batch, prefetch_geometry=not args.no_prefetch, max_gt=MAX_GT,
                                             dry=True, geo_pipe=replay.geo, update_in_graph=False)
            secondary["ms_per_step_eager_update"] = time_steps(step_loop, args.steps)
            trainer.allreduce_stub_us = 100
            trainer.allreduce_overlap = False
            secondary["allreduce_stub100us_serial_ms_per_step"] = time_steps(step_loop, args.steps)
            trainer.flush()
            trainer.allreduce_overlap = True
            secondary["allreduce_stub100us_overlapped_ms_per_step"] = time_steps(step_loop, args.steps)
            trainer.flush()
            trainer.allreduce_stub_us, trainer.allreduce_overlap = 0, None
            loop_replay[0] = None
            torch.cuda.synchronize()
        # (b) the other BASELINE configurations on the same process / box, resident replay each: configs[3]
        # per GPU (bf16 compute mode), BASELINE's "8 heads x 4 points" wording (P = 4; the reference config
        # is 2), the other cloud distribution and the reference's own per-GPU batch
        # (samples_per_gpu=16, configs/_base_/datasets/sunrgbd-3d-10class.py:75)
        import dataclasses

        def other(dtype, points, cloud=args.cloud, B=args.batch):
            ops.set_compute_dtype(dtype)
            try:
                c2 = cfg if points == cfg.head.num_points else \
                    dataclasses.replace(cfg, head=dataclasses.replace(cfg.head, num_points=points))
                torch.manual_seed(0)
                m2 = DeMFHotPath(c2).to(device).train()
                t2 = engine.Trainer(m2)
                b2 = batch if (cloud == args.cloud and B == args.batch) else \
                    make_batch(B, seed=3000 + rank, device=device, cloud=cloud)[0]
                r2 = t2.capture(b2, prefetch_geometry=not args.no_prefetch)
                ms = time_steps(r2, args.steps)
                ok = bool(torch.isfinite(t2.flat.flat).all())
                return ms if ok else float("nan")
            finally:
                ops.set_compute_dtype(args.dtype)
        if args.dtype != "bf16":
            secondary["bf16_ms_per_step"] = other("bf16", args.msda_points)
        if args.dtype == "f32":
            # the same step with the two-fp16-term kernel forms off (three bf16 terms everywhere, round 5's arithmetic)
            secondary["f32x3_ms_per_step"] = other("f32x3", args.msda_points)
        if args.msda_points != 4:
            secondary["p4_ms_per_step"] = other(args.dtype, 4)
        oc = "clustered" if args.cloud == "uniform" else "uniform"
        secondary[oc + "_ms_per_step"] = other(args.dtype, args.msda_points, cloud=oc)
        if args.batch != 16:
            # its own process, as a user would run it (measured inside this one - after the other models,
            # graphs and static buffers - the same replay came out 40 % slower: 11.2 vs 7.7 ms)
            import subprocess
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", "16", "--steps", str(args.steps),
                                  "--warmup", str(args.warmup), "--dtype", args.dtype, "--msda-points",
                                  str(args.msda_points), "--no-secondary", "--no-cpu-baseline", "--no-pmc-live"],
                                 capture_output=True, text=True, timeout=600)
            try:
                d16 = json.loads(out.stdout.strip().splitlines()[-1])
                secondary["b16_ms_per_step"] = d16["ms_per_step"]
                secondary["b16_scenes_per_s"] = d16["value"]
            except Exception:
                secondary["b16_ms_per_step"] = float("nan")
        # (d) SURVEY 8(f) rank 1 / 8(d) "secondary = e2e": the frozen image stream (ResNet-50 + ChannelMapper on
        # csrc/conv.hip, the six encoder layers on csrc/rows_gemm.hip + the MSDA kernel) in front of the step
        def image_stream_secondary():
            from demf_amd.modules import ImageStream
            from demf_amd.config import BATCH_INPUT_SHAPE
            ist = ImageStream().to(device)
            img = torch.randn(args.batch, 3, *BATCH_INPUT_SHAPE, device=device)
            for _ in range(2):
                ist.tokens(img, batch["img_metas"])
            pyr = ist.pyramid(img)
            secondary["image_backbone_neck_path"] = "csrc/conv.hip (implicit-GEMM NHWC convolutions)" \
                if isinstance(pyr, dict) else "library convolutions (MIOpen)"
            secondary["image_backbone_neck_ms"] = time_steps(lambda: ist.pyramid(img), 5)
            if isinstance(pyr, dict):
                # the library path at its best: MIOpen's search on (first call per shape: seconds)
                from demf_amd.modules import image_stream as _ims
                _ims.MIOPEN_SEARCH = True
                prev_fb, ops.LIBRARY_FALLBACK = ops.LIBRARY_FALLBACK, True     # (the A/B reference: explicitly allowed)
                try:
                    secondary["image_backbone_neck_library_ms"] = time_steps(lambda: ist._pyramid(img), 5)
                finally:
                    _ims.MIOPEN_SEARCH = False
                    ops.LIBRARY_FALLBACK = prev_fb
                # ResNet-50 + ChannelMapper at 800 x 1120: 77.0 GMAC per image
                secondary["image_backbone_neck_tflops"] = 2 * 77.0e9 * args.batch / secondary["image_backbone_neck_ms"] * 1e-9
            secondary["image_encoder_ms"] = time_steps(lambda: ist.img_encoder.forward_tokens(pyr, batch["img_metas"]), 5)
            # its linear layers: 6 layers x 2 x rows x (256 x (640 + 256) + 2 x 256 x 1024) flop (csrc/rows_gemm.hip)
            enc_flop = 6 * 2.0 * args.batch * sum(h * w for h, w in PYRAMID_SHAPES) * (256 * 896 + 2 * 256 * 1024)
            secondary["image_encoder_gemm_tflops_incl_msda_time"] = enc_flop / secondary["image_encoder_ms"] * 1e-9

            def e2e():
                ist.tokens(img, batch["img_metas"])
                step_resident()
            secondary["e2e_ms_per_step"] = time_steps(e2e, 10)
            # pipelined: the (frozen, no_grad) stream of batch k+1 on a side stream under the step of batch k
            side = engine.concurrent_stream()      # (a pool stream may share the main stream's hardware queue)

            def e2e_pipe():
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ist.tokens(img, batch["img_metas"])
                step_resident()
                torch.cuda.current_stream().wait_stream(side)
            secondary["e2e_pipelined_ms_per_step"] = time_steps(e2e_pipe, 10)
            if args.dtype != "bf16":
                # the frozen, no_grad stream alone in the bf16 compute mode (operands rounded to bf16, fp32
                # accumulate; tests/test_gpu_conv.py + test_gpu_image_stream.py hold that mode to its own bars)
                # in front of the SAME fp32-grade step: what mixed precision buys end to end
                def stream_bf16():
                    ops.set_compute_dtype("bf16")
                    try:
                        ist.tokens(img, batch["img_metas"])
                    finally:
                        ops.set_compute_dtype(args.dtype)
                secondary["image_stream_bf16_ms"] = time_steps(stream_bf16, 5)

                def e2e_mixed():
                    stream_bf16()
                    step_resident()
                secondary["e2e_bf16_stream_ms_per_step"] = time_steps(e2e_mixed, 10)
            del ist, img, pyr
        if args.batch <= 8:
            try:
                image_stream_secondary()
            except Exception as exc:      # (a secondary figure must not take the headline line down with it)
                secondary["e2e_error"] = repr(exc)[:200]
        # (c) BASELINE configs[1]: the SA path alone (forward / forward+backward / the index pre-pass)
        for B1 in (1, 8):
            f, fb, g = sa_path_ms(model, device, B1)
            secondary["sa_path_b%d_fwd_ms" % B1] = f
            secondary["sa_path_b%d_fwdbwd_ms" % B1] = fb
            secondary["sa_path_b%d_prepass_ms" % B1] = g
        # (e) the multi-rank path in software: no 8-GPU node has run this code (SCALE_r0x: skipped), so at least the
        # launcher, the rank bookkeeping and the collective's call path are exercised by every bench run - two ranks
        # sharing THIS GPU through gloo (DEMF_SHARE_DEVICE; RCCL refuses two ranks on one device), four scenes each.
        # Its allreduce_us is gloo through host memory: a software-path number, not an xGMI figure.
        if not os.environ.get("DEMF_BENCH_SKIP_WORLD2") and not os.environ.get("DEMF_SHARE_DEVICE"):
            import socket
            import subprocess
            try:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = str(sk.getsockname()[1])
                env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
                env2.update(DEMF_SHARE_DEVICE="1", DEMF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
                            MASTER_PORT=port, OMP_NUM_THREADS="4")
                r2 = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "2", "--steps", "5", "--warmup",
                                     "2", "--batch", "4", "--dtype", args.dtype], env=env2, capture_output=True,
                                    text=True, timeout=300)
                line = [l for l in r2.stdout.splitlines() if l.startswith("{")]
                if r2.returncode == 0 and line:
                    o2 = json.loads(line[-1])
                    secondary["world2_shared_gpu"] = {
                        "backend": "gloo, two ranks on one GPU (software path only)", "scenes_per_rank": 4,
                        "ms_per_step": o2["ms_per_step"], "value": o2["value"], "allreduce_us": o2.get("allreduce_us"),
                        "allreduce": o2.get("allreduce")}
                else:
                    secondary["world2_shared_gpu"] = {"error": (r2.stderr or r2.stdout)[-300:]}
            except Exception as exc:      # (a secondary figure must not take the headline line down with it)
                secondary["world2_shared_gpu"] = {"error": repr(exc)[:200]}
    rank_info = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # what makes the ranks different replicas of one job: their data seeds and dropout streams
        from demf_amd import fused
        rank_info = [None] * world
        dist.all_gather_object(rank_info, dict(rank=rank, batch_seeds=[1000 + rank + 7919 * i for i in range(NB)],
                                               dropout_seed=fused.get_rng_state(device)[0]))

    if rank == 0:
        scenes = args.batch * world * args.steps
        ms_step = 1000.0 * elapsed / args.steps
        two_graphs = bool(os.environ.get("DEMF_GEO_AT_BWD"))
        if args.no_graph:
            launch = "eager"
        else:
            launch = ("hipGraphs(fwd+loss | bwd), pre-pass of the next batch launched in between" if two_graphs
                      else "one hipGraph(fwd+loss+bwd), pre-pass of the next batch launched in front of it") + \
                " on a side stream; " + ("norm + clip + AdamW inside the graph"
                                         if getattr(replay, "update_in_graph", False)
                                         else "eager allreduce / clip / AdamW behind it")
        out = {
            "metric": "DeMF fusion fwd+bwd scenes/sec at 20k pts + 530x730 RGB",
            "value": scenes / elapsed, "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            # fp32 RESULTS; in the default mode the shared-MLP products are issued as 3-term bf16
            # splits on the bf16 MFMA (emulated fp32, error vs fp64 = the fp32 MFMA's)
            "emulated_fp32": args.dtype in ("f32", "f32x3") and MFMA_PATH[args.dtype] in (MFMA_PATH["f32x3"], MFMA_PATH["f32h2"]),
            "config": {"workload": "BASELINE configs[2]: full DeMF fusion hot path fwd+loss+bwd+"
                                   "allreduce+AdamW, %d scenes/GPU x (20000 pts, 800x1120 -> "
                                   "4-level 256-ch pyramid), 256 queries, H=8 L=4 P=%d, %s"
                                   % (args.batch, args.msda_points,
                                      "fp32" if args.dtype != "bf16" else
                                      "bf16 MFMA / fp32 accumulate+storage (BASELINE configs[3] per GPU)"),
                       "compute_mode": args.dtype, "mfma_path": MFMA_PATH[args.dtype],
                       "scenes_per_gpu": args.batch, "parallelism": f"dp{world}",
                       "cloud": args.cloud,
                       "timed_loop": ("one resident batch replayed" if args.resident else
                                      "%d distinct HBM-resident batches cycled through replay.load(): target "
                                      "padding, meta refresh, static-buffer copies and the next cloud's "
                                      "pre-pass are inside the timed step" % NB) +
                                     ("; double-buffered static inputs: batch k+1 is loaded on an input stream "
                                      "while step k runs" if (not args.resident and replay2 is not None) else ""),
                       "launch": launch},
            # two more passes of the same K steps right after the timed region (stability check;
            # `value` is the first, contract-shaped region only)
            "repeat_ms_per_step": repeats,
        }
        if allreduce_us is not None:
            # HIP events around the one all-reduce of the flat 8.76 MB gradient buffer (rank 0's view)
            out["allreduce_us"] = allreduce_us
        w_, stub_, ov_ = trainer.allreduce_config()
        out["allreduce"] = {
            "buckets_bytes": [int(trainer.flat.flat.numel()) * 4], "collective": "one flat all-reduce (SUM), "
            "1/world folded into AdamW", "world": w_, "stub_us": stub_, "overlap": bool(ov_),
            "overlap_how": "issued on a communication stream behind the step's graph; norm + AdamW are enqueued at "
                           "the start of the next step, so the collective runs underneath the next batch's input "
                           "path (replay.load) and pre-pass launch" if ov_ else None,
            "us": allreduce_us}
        if rank_info is not None:
            out["ranks"] = [dict(rank=r["rank"], batch_seeds=r["batch_seeds"], dropout_seed=r["dropout_seed"])
                            for r in rank_info]
        # ---- headline roofline: the longest launch ON the step's main stream - SA1's last layer backward.
        # Algorithmic bytes of that layer's backward = read the (R,64) input rows once, write their (R,64)
        # gradient once, read the pooled gradient, its argmax rows and the raw pooled values (3 x (R/64,128)
        # words); weights and the (128,64) weight gradient < 1 %.  Algorithmic FLOPs: dX = dZ W and
        # dW = dZ^T A, 2 x 2 R 128 64.
        if not args.no_pmc_live and world == 1 and args.batch == 8 and not args.no_graph:
            err = pmc_live(args)
            out["pmc_live"] = "ok" if err is None else "failed (%s): committed profiles/ used" % err
        x3 = args.dtype in ("f32", "f32x3") and MFMA_PATH[args.dtype] == MFMA_PATH["f32x3"]
        h2 = args.dtype == "f32" and MFMA_PATH[args.dtype] == MFMA_PATH["f32h2"]      # (the dominant kernel: 3 fp16 products)
        # MFMA budget of the mode: native fp32 -> the fp32 MFMA peak; bf16 -> the bf16 peak; the
        # three-term split issues 6 bf16 MFMAs per algorithmic product -> bf16 peak / 6
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else \
            (MFMA_BF16_PEAK_TFLOPS / 6.0 if x3 else (MFMA_BF16_PEAK_TFLOPS / 3.0 if h2 else MFMA_F32_PEAK_TFLOPS))
        mlp_ms = mlp_timer.mean_ms()
        mlp_bytes = 2 * sa1_rows * 64 * 4 + 3 * (sa1_rows // 64) * 128 * 4
        mlp_flop = 2 * 2.0 * sa1_rows * 64 * 128
        dom = DOMINANT_KERNEL[args.dtype]
        # (the committed PMC passes are B = 8 runs of the default mode; no row for this kernel in the
        # newest pair is an error there, not a silent fall-back to an older round's file)
        strict = args.batch == 8 and args.dtype == "f32" and not os.environ.get("DEMF_F32_NATIVE")
        try:
            traffic, src = pmc_per_launch(dom, required=strict) if args.batch == 8 else (None, None)
        except RuntimeError as e:      # (a stale committed pair must cost the line its `traffic`, not the line itself)
            traffic, src = None, "unavailable: %s" % e
        tk, tk_src = top_kernels()
        out["roofline"] = {
            "kernel": "%s (SA1 layer 3 backward: dX, dW and layer 2's BN sums from the pooled gradient; the "
                      "layer's (R,128) output is never stored; R=%d)" % (dom, sa1_rows),
            # priced against the HBM roof (its algorithmic bytes dominate its FLOPs at either MFMA
            # rate); what limits it today is on-chip: MFMA issue of the three-term products (44 % of its
            # time, tools/pool_bwd_micro.py phase skips) + latency at 2 waves/SIMD
            "bound": "hbm", "limited_by": "mfma issue of the split products + latency (2 waves per SIMD), "
                                          "not HBM bytes",
            "achieved": mlp_bytes / (mlp_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": mlp_bytes / (mlp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "algorithmic_bytes": mlp_bytes,
            "avg_launch_ms": mlp_ms,
            # this ONE kernel's share of the step: `frac` above is a statement about it, not about the step
            "share_of_step": mlp_ms / ms_step,
            "mfma_frac": mlp_flop / (mlp_ms * 1e-3) / 1e12 / mfma_peak,
            "note": "also %.1f algorithmic GFLOP per launch; mfma_frac = of the %.1f TF/s the mode can "
                    "issue (%s); the whole step: roofline_step, its five longest kernels: roofline_top5"
                    % (mlp_flop / 1e9, mfma_peak,
                       "dense bf16 MFMA peak" if args.dtype == "bf16" else
                       ("dense bf16 MFMA peak / 6: three-term split" if x3 else
                        ("dense fp16 MFMA peak / 3: two-term split" if h2 else "dense fp32 MFMA peak")))}
        # the forward twin (round 3's headline kernel): reads the (R,64) rows, writes only the pooled
        # (R/64,128) extremum + its row; since round 4 the (R,128) output is not stored
        fwd_ms = fwd_timer.mean_ms()
        if fwd_ms == fwd_ms:
            fwd_bytes = sa1_rows * 64 * 4 + 2 * (sa1_rows // 64) * 128 * 4
            ftraffic, fsrc = pmc_per_launch(SA1_FWD_KERNEL[args.dtype]) if args.batch == 8 else (None, None)
            out["roofline_sa1_fwd"] = {
                "kernel": "%s...> (SA1 layer 3 forward: 64->128 + BN stats + max-pool, no output store, R=%d)"
                          % (SA1_FWD_KERNEL[args.dtype], sa1_rows),
                "bound": "hbm", "achieved": fwd_bytes / (fwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": ftraffic, "traffic_source": fsrc, "algorithmic_bytes": fwd_bytes,
                "avg_launch_ms": fwd_ms, "share_of_step": fwd_ms / ms_step,
                "mfma_frac": 2.0 * sa1_rows * 64 * 128 / (fwd_ms * 1e-3) / 1e12 / mfma_peak}
        if tk is not None:
            # the five kernels with the most time per step in the newest committed profile of this command
            # (rocprofv3 kernel trace + PMC passes; builder-run, committed - not measured by this run):
            # time, HBM bytes moved, achieved GB/s and fraction of the 8 TB/s roof, each
            out["roofline_top5"] = {
                "source": tk_src, "profiled_kernel_us_per_step": tk.get("kernel_us_per_step"),
                "profiled_launches_per_step": tk.get("launches_per_step"),
                "kernels": [dict(kernel=e["kernel"][:100], us_per_step=e["us_per_step"],
                                 launches_per_step=e["launches_per_step"],
                                 mb_per_step=(e["hbm_bytes_per_step"] / 1e6 if e.get("hbm_bytes_per_step") else None),
                                 gbps=e.get("gbps"), frac=e.get("frac_of_hbm_peak"),
                                 side_stream=e.get("side_stream", False))
                            for e in tk["kernels"][:5]]}
        # ---- step level: what the metric asks for ("as achieved fraction of HBM roofline")
        step_bytes = ALGO_BYTES_PER_SCENE * args.batch
        step_flop = ALGO_FLOP_PER_SCENE * args.batch
        traffic_step, src_step = pmc_per_step() if args.batch == 8 else (None, None)
        out["roofline_step"] = {
            "bound": "mfma" if args.dtype != "bf16" else "hbm", "algorithmic_bytes": step_bytes,
            # HBM bytes the whole step actually moved (PMC, every kernel) and their ratio to the
            # algorithmic bytes: > 1 = re-reads of materialised intermediates (Y_l read by the next
            # layer, the dx and the dW pass).  Builder-run PMC passes of this command, committed under
            # profiles/ (`traffic_source`) - not observed by this run.
            "traffic_step": traffic_step, "traffic_source": src_step,
            "traffic_over_algorithmic": (traffic_step / step_bytes) if traffic_step else None,
            "algorithmic_flop": step_flop,
            "hbm": {"achieved": step_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "mfma": {"achieved": step_flop / (ms_step * 1e-3) / 1e12, "peak": mfma_peak,
                     "unit": "TFLOP/s", "frac": step_flop / (ms_step * 1e-3) / 1e12 / mfma_peak},
            "note": "SURVEY 8(d) per-scene algorithmic figures (190 MB, 46 GFLOP fwd+bwd) x scenes/GPU "
                    "over the measured step time, per GPU; the path is fp32-MFMA / latency bound, "
                    "not HBM bound"}
        # ---- the FPS chain (underneath the step): latency-bound, neither HBM nor MFMA.  Reported as
        # what it is - milliseconds and shader cycles per dependent round - not as a fraction of a floor.
        big = [ms for ms, shp in fps_timer.results() if shp is not None and shp[1] == 20000]
        if big:
            fps_ms = float(np.mean(big))
            rounds = 2047
            out["roofline_fps"] = {
                "kernel": "FPS 20000->2048 (one workgroup per scene, %d scenes)" % args.batch,
                "bound": "latency", "avg_launch_ms": fps_ms, "dependent_rounds": rounds,
                "cycles_per_round": fps_ms * 1e-3 / rounds * CLOCK_GHZ * 1e9,
                "note": "cycles at the %.1f GHz peak clock; HBM bytes are 2 MB per launch by construction "
                        "(points read once, indices written once); DESIGN.md section 3.1" % CLOCK_GHZ}
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
