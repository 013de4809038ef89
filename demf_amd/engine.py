"""Training-step engine for the DeMF hot path on MI355X.

The reference trains with mmcv's EpochBasedRunner + MMDistributedDataParallel over NCCL
(train.py:56-63,140-147; tools/dist_train.sh:8-9): replicas only, gradients all-reduced,
AdamW with 'decoder' lr_mult 0.05 and grad-clip 10 (configs/demf/demf_votenet.py:16-24,
configs/_base_/schedules/schedule_3x.py:6).  MI355X-first restatement:

  * one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI);
  * the 2.19 M trainable parameters' gradients live in ONE flat fp32 buffer (8.76 MB):
    every p.grad is a view into it, so a step issues exactly one RCCL all-reduce - at this
    size the collective is latency-bound on xGMI, so one call beats DDP's bucket stream -
    and one fused clip + AdamW over the flat buffers;
  * no model sharding (the reference has none).
"""
import contextlib
import os

import torch
import torch.distributed as dist


def init_distributed():
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DEMF_SHARE_DEVICE"):
        # test hook: several ranks on ONE GPU (with DEMF_DIST_BACKEND=gloo), to exercise the
        # multi-rank code path of bench.py on a single-GPU box
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("DEMF_DIST_BACKEND") or \
            ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def cu_masked_streams(side_cus):
    """(main, side) torch streams on disjoint CU sets: ``side`` runs only on ``side_cus``, ``main``
    on every other CU (hipExtStreamCreateWithCUMask through libdemf_hip.so)."""
    import ctypes
    from . import _ffi
    arr = (ctypes.c_int * len(side_cus))(*side_cus)
    out = []
    for invert in (1, 0):
        h = ctypes.c_void_p()
        _ffi.call("demf_stream_create_cu_masked", ctypes.cast(arr, ctypes.c_void_p), len(side_cus),
                  invert, ctypes.cast(ctypes.byref(h), ctypes.c_void_p))
        out.append(torch.cuda.ExternalStream(h.value))
    return out[0], out[1]


_REJECTED_STREAMS = []      # streams found to share the main stream's hardware queue (kept: the pool must move on)


def concurrent_stream(main=None, tries=8, spin_us=300):
    """A new torch stream that really runs CONCURRENTLY with ``main`` (default: the current stream).
    HIP multiplexes streams onto a few hardware queues (4 by default); a side stream that lands on the main
    stream's queue is serialised with it, and the pipelined FPS pre-pass - 2.1 ms meant to run underneath the
    step - then runs in front of it (measured: 4.84 -> 7.45 ms per step for one unlucky stream of torch's pool,
    bench.py's clustered leg after another stream had been created).  Each candidate is probed with two spin
    kernels of known length (demf_spin_us), one per stream: together they take one length if the streams are
    concurrent, two if they share a queue."""
    from . import _ffi
    main = main if main is not None else torch.cuda.current_stream()
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream()
    last = None
    for _ in range(tries):
        cand = torch.cuda.Stream()
        last = cand
        cand.wait_stream(main)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        _ffi.call("demf_spin_us", spin_us, main.cuda_stream)
        with torch.cuda.stream(cand):
            _ffi.call("demf_spin_us", spin_us, cand.cuda_stream)
        main.wait_stream(cand)
        e1.record(main)
        e1.synchronize()
        if e0.elapsed_time(e1) * 1e3 < 1.5 * spin_us:
            return cand
        _REJECTED_STREAMS.append(cand)
    return last


class FlatGrads:
    """All trainable gradients as views of one contiguous buffer; one all-reduce per step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev, dt = self.params[0].device, self.params[0].dtype
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.views = []
        self._pending_tables, self._tables = [], []     # address tables of captured pack launches
        self.captured_pack = False      # set by Trainer.capture around its captures (it calls finish_capture)
        self.sumsq_state = None         # FlatAdamW.state while a capture wants the pack to take the clip norm
        self._reserved_table = None
        off = 0
        for p in self.params:
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            self.views.append(p.grad)
            off += k

    def backward_into(self, loss):
        """d(loss)/d(params) straight into the flat buffer: one autograd.grad + multi-tensor
        copies, instead of 119 per-parameter accumulate kernels (and no zero-fill dependence)."""
        if self.flat.is_cuda:
            from . import ops
            with ops.deferred_weight_grads():       # the few-row stacks' dW products as one grouped launch at the end
                grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        else:
            grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        dst = [v for v, g in zip(self.views, grads) if g is not None]
        src = [g for g in grads if g is not None]
        if len(src) != len(grads):
            self.flat.zero_()
        if self.captured_pack and self.flat.is_cuda and torch.cuda.is_current_stream_capturing() and src:
            # Under capture the gradients live at fixed addresses of the graph's pool: one
            # demf_multi_copy launch over an address table (two multi-tensor launches of 37 us each
            # otherwise).  The table cannot be uploaded while capturing; the recorded launch only
            # holds its device address, ``finish_capture`` fills it before the first replay.
            from . import _ffi
            src = [g if g.is_contiguous() else g.contiguous() for g in src]
            tab = [[g.data_ptr() for g in src], [d.data_ptr() for d in dst],
                   [d.numel() for d in dst]]
            # (allocated by Trainer.capture BEFORE capturing: memory of the graphs' shared pool is
            # scratch that the forward graph rewrites on every replay)
            table = self._reserved_table.view(-1)[:3 * len(src)].view(3, len(src))
            blocks = max(1, min(64, max(tab[2]) // 4096))
            if self.sumsq_state is not None:
                # + the squared norm of everything packed, for the clip of the in-graph update
                _ffi.call("demf_multi_copy_sumsq", len(src), table.data_ptr(), blocks,
                          self.sumsq_state.data_ptr(), torch.cuda.current_stream().cuda_stream)
            else:
                _ffi.call("demf_multi_copy", len(src), table.data_ptr(), blocks,
                          torch.cuda.current_stream().cuda_stream)
            self._pending_tables.append((table, tab))
            return
        torch._foreach_copy_(dst, src)

    def reserve_table(self):
        """Room for the address table of one captured pack launch, from the ordinary allocator."""
        self._reserved_table = torch.empty((3, len(self.params)), dtype=torch.int64,
                                           device=self.flat.device)

    def finish_capture(self):
        """Upload the address tables of the pack launches recorded while capturing (call once the
        capture has ended, before the first replay)."""
        for table, tab in self._pending_tables:
            table.copy_(torch.tensor(tab, dtype=torch.int64))
            self._tables.append(table)                 # the graphs hold its address
        self._reserved_table = None
        self._pending_tables = []

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def clip_(self, max_norm):
        """clip_grad_norm_(max_norm) on the flat buffer (one norm, one scale)."""
        norm = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm


class FlatAdamW:
    """torch.optim.AdamW over parameter groups whose parameters are re-homed into ONE flat fp32
    buffer (each p.data becomes a view of it, in FlatGrads order), with flat moment buffers: a
    step is ONE demf_adamw_state_f32 launch over all groups (csrc/optim.hip) with the clip coefficient
    and the 1/world_size of the gradient mean folded in, instead of 22 multi-tensor launches + a
    scaling pass.  Step count, learning-rate factor and the squared gradient norm live in a 64-byte
    DEVICE state block, so the launch takes no host argument that changes from step to step and the
    whole update can be a node of the step's hipGraph (Trainer.capture).  Group semantics as the
    reference's config (demf_votenet.py:16-24)."""

    def __init__(self, groups, flat_grads, betas=(0.9, 0.999), eps=1e-8):
        import ctypes
        params = flat_grads.params
        assert [id(p) for g in groups for p in g["params"] if p.requires_grad] == [id(p) for p in params]
        self.grads = flat_grads.flat
        self.flat = torch.empty_like(self.grads)
        off, self.segments = 0, []
        for g in groups:
            start = off
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                off += k
            if off > start:
                self.segments.append((start, off - start, float(g["lr"]), float(g["weight_decay"])))
        if not 1 <= len(self.segments) <= 4:
            raise ValueError("FlatAdamW: %d parameter groups (demf_adamw_state_f32 takes 1..4)" % len(self.segments))
        n = len(self.segments)
        self._seg = ((ctypes.c_longlong * n)(*[s[0] for s in self.segments]),
                     (ctypes.c_longlong * n)(*[s[1] for s in self.segments]),
                     (ctypes.c_float * n)(*[s[2] for s in self.segments]),
                     (ctypes.c_float * n)(*[s[3] for s in self.segments]))
        self.params = params
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.betas, self.eps = betas, eps
        # { double sumsq; int64 t; uint32 ticket; float lr_factor; pad } - include/demf_hip.h
        self.state = torch.zeros(64, dtype=torch.uint8, device=self.flat.device)
        self._lr_factor = 1.0
        self.state.view(torch.float32)[5] = 1.0

    @property
    def t(self):
        """Completed optimizer steps (read from the device: synchronises)."""
        return int(self.state.view(torch.int64)[1].item())

    @t.setter
    def t(self, value):
        self.state.view(torch.int64)[1] = int(value)

    @property
    def lr_factor(self):
        return self._lr_factor

    def set_lr_factor(self, factor):
        """Multiplies every group's base learning rate (the reference's step schedule:
        lr_config step=[24, 32], x0.1 each - configs/_base_/schedules/schedule_3x.py:7-9).
        Written into the device state: captured steps pick it up at their next replay."""
        self._lr_factor = float(factor)
        self.state.view(torch.float32)[5] = float(factor)

    def check_aliasing(self):
        """Every parameter must still be a view of the flat buffer: ``model.to()`` / ``.float()`` /
        ``.half()`` after the optimizer was built re-allocates ``p.data`` and would silently detach
        the parameter from the update."""
        lo = self.flat.data_ptr()
        hi = lo + 4 * self.flat.numel()
        for p in self.params:
            if not (lo <= p.data_ptr() < hi):
                raise RuntimeError("a parameter no longer lives in FlatAdamW's flat buffer (was the "
                                   "model moved or cast after the Trainer was built?)")

    def state_dict(self):
        """Moments + step count + lr factor (resume: mmcv CheckpointHook saves the optimizer too)."""
        return dict(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), t=self.t,
                    lr_factor=self.lr_factor, numel=self.flat.numel())

    def load_state_dict(self, sd):
        if int(sd["numel"]) != self.flat.numel():
            raise ValueError("optimizer state is for %d parameters, this model has %d"
                             % (int(sd["numel"]), self.flat.numel()))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.t = int(sd["t"])
        self.set_lr_factor(float(sd.get("lr_factor", 1.0)))

    def step(self, max_norm=0.0, grad_scale=1.0, norm_taken=False):
        """One update of every group from the flat gradient buffer.  ``norm_taken``: the squared norm
        of the gradients is already in the device state (the captured pack took it,
        demf_multi_copy_sumsq); otherwise one reduction launch over the flat buffer comes first."""
        import ctypes
        from . import _ffi
        self.check_aliasing()
        stream = torch.cuda.current_stream().cuda_stream
        if max_norm > 0.0 and not norm_taken:
            _ffi.call("demf_sumsq_f32", self.grads.numel(), self.grads.data_ptr(), self.state.data_ptr(), stream)
        a = [ctypes.cast(x, ctypes.c_void_p) for x in self._seg]
        _ffi.call("demf_adamw_state_f32", len(self.segments), a[0], a[1], a[2], a[3], self.flat.data_ptr(),
                  self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                  self.state.data_ptr(), float(max_norm), float(grad_scale), self.betas[0], self.betas[1],
                  self.eps, stream)


@contextlib.contextmanager
def _gc_paused():
    """No garbage collection inside a stream capture.  A collection that reaches an unreachable CUDAGraph (the trainer of
    an earlier phase, kept alive by a reference cycle until now) releases that graph's memory pool, and a hipFree under
    capture aborts the process (seen in the test suite: `Fatal Python error: Aborted`, "Garbage-collecting" on the
    stack of a capture).  Collect once before the capture begins, then keep the collector off until it has ended."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class Trainer:
    """fwd -> loss -> bwd -> one all-reduce -> clip -> AdamW, as one callable step."""

    def __init__(self, model, lr=0.008, weight_decay=0.01, max_grad_norm=10.0):
        self.model = model
        groups = model.param_groups(lr=lr, weight_decay=weight_decay)
        self.flat = FlatGrads([p for g in groups for p in g["params"]])
        self.max_grad_norm = max_grad_norm
        if dist.is_initialized() and dist.get_world_size() > 1:
            for p in model.parameters():           # identical replicas at step 0
                dist.broadcast(p.data, src=0)
            for b in model.buffers():
                dist.broadcast(b.data, src=0)
        # device path: flat fused AdamW (HIP); the torch optimizer serves the CPU/gloo host-logic
        # tests only
        self.fused = self.flat.flat.is_cuda
        if self.fused:
            from . import fused
            rank = dist.get_rank() if dist.is_initialized() else 0
            fused.rng_state(self.flat.flat.device, seed=torch.initial_seed() + 7919 * rank)
        self.opt = FlatAdamW(groups, self.flat) if self.fused else \
            torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, foreach=True)
        self._base_lrs = [float(g["lr"]) for g in groups]

    def set_epoch(self, epoch, steps=(24, 32), gamma=0.1):
        """The reference's step schedule (configs/_base_/schedules/schedule_3x.py:7-9:
        lr_config = dict(policy='step', step=[24, 32]), 36 epochs): lr x gamma at each milestone."""
        factor = gamma ** sum(1 for s in steps if epoch >= s)
        if self.fused:
            self.flush()                 # (an overlapped step's owed update still uses the old rate)
            self.opt.set_lr_factor(factor)
        else:
            for g, base in zip(self.opt.param_groups, self._base_lrs):
                g["lr"] = base * factor
        return factor

    def state_dict(self):
        """Model + optimizer state for resume (mmcv CheckpointHook, cfg:280), plus the dropout
        counter [seed, step] of the fused decoder layer so that a resumed run continues the mask
        sequence instead of replaying it from step 0."""
        if self.fused:
            self.flush()
        sd = dict(model=self.model.state_dict(), optimizer=self.opt.state_dict())
        if self.fused:
            from . import fused
            sd["dropout_rng"] = fused.get_rng_state(self.flat.flat.device)
        return sd

    def load_state_dict(self, sd):
        # copies INTO the existing (flat-buffer-resident) parameters: the views stay intact
        if self.fused:
            self.flush()                 # (never apply an owed update of the OLD gradients to the loaded state)
        self.model.load_state_dict(sd["model"])
        self.opt.load_state_dict(sd["optimizer"])
        if self.fused and sd.get("dropout_rng") is not None:
            from . import fused
            fused.set_rng_state(self.flat.flat.device, sd["dropout_rng"])

    def _fwd(self, batch, geometry=None):
        kw = {} if geometry is None else dict(geometry=geometry)
        # (the fused decoder layer advances its counter-based dropout state by itself, inside the
        # forward and therefore inside the captured graph: demf_amd/fused.py)
        losses = self.model.forward_train(batch["points"], batch["img_features"],
                                          batch["img_metas"], batch["gt_bboxes_3d"],
                                          batch["gt_labels_3d"], **kw)
        # the head hands out the sum of its losses directly (one node instead of 8 selects)
        return losses["_total"] if "_total" in losses else torch.stack(list(losses.values())).sum()

    def _arena(self, begin):
        """Bracket forward + backward with the step's zero arena (ops._ZeroArena) on the device path."""
        if self.fused:
            from . import ops
            if begin:
                ops.ARENA.begin(self.flat.flat.device)
            else:
                ops.ARENA.end()

    def _fwd_bwd(self, batch, geometry=None):
        self._arena(True)
        try:
            total = self._fwd(batch, geometry)
            self.flat.backward_into(total)
        except BaseException:
            if self.fused and not torch.cuda.is_current_stream_capturing():
                from . import ops
                ops.reset_accumulators()          # (a step that raised half-way leaves BatchNorm sums behind)
            raise
        finally:
            self._arena(False)
        return total.detach()

    # ---- gradient all-reduce + clip + AdamW ------------------------------------------------------------
    # The collective is latency-bound at 8.76 MB and nothing after it in the step can start
    # before it ends (the clip needs the norm of the reduced gradients) - but the NEXT batch's input path can:
    # ``replay.load`` (token conversion of the image pyramid, target padding, constants) and the launch of the
    # coordinate pre-pass touch neither the gradients nor the parameters.  With ``overlap`` the collective is issued
    # on a communication stream when the step's graph has been enqueued, ``replay()`` returns, and norm + AdamW
    # are enqueued at the start of the next ``replay()`` (or by ``flush()``), behind the collective: it runs
    # underneath whatever the caller enqueues in between.  The RCCL call itself stays an ordinary eager call on a
    # stream - nothing is captured, so the multi-GPU path is the one the gloo / shared-GPU tests exercise.
    def _comm(self):
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = concurrent_stream()
        return self._comm_stream

    def allreduce_config(self):
        """(world, stub_us, overlap): ``stub_us`` > 0 replaces the collective by a spin kernel of that length
        (bench.py --allreduce-stub-us: the overlap measured on one GPU); overlap is opt-in (DEMF_AR_OVERLAP=1 or
        ``trainer.allreduce_overlap = True``)."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        stub = int(getattr(self, "allreduce_stub_us", 0) or 0)
        ov = getattr(self, "allreduce_overlap", None)
        if ov is None:
            # default OFF: measured on MI355X / ROCm 7.2 with a spin kernel in the collective's place
            # (bench.py secondary.allreduce_stub100us_*; profiles/r05_allreduce_overlap_probe.log) the extra
            # cross-queue dependencies of the deferred order cost 0.23 ms per step - more than a 100 us collective
            # takes on the step's own stream
            ov = bool(int(os.environ.get("DEMF_AR_OVERLAP", "0")))
        return world, stub, bool(ov) and (world > 1 or stub > 0)

    def _collective(self, world, stub):
        ev = getattr(self, "allreduce_events", None)   # bench.py: HIP events around the collective
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if stub > 0:
            from . import _ffi
            _ffi.call("demf_spin_us", stub, torch.cuda.current_stream().cuda_stream)
        if world > 1:
            dist.all_reduce(self.flat.flat, op=dist.ReduceOp.SUM)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))

    def _finish_update(self, norm_taken=False):
        world = dist.get_world_size() if dist.is_initialized() else 1
        # (the norm is that of the SUM over ranks; 1/world is applied in-kernel)
        self.opt.step(self.max_grad_norm, 1.0 / world, norm_taken=norm_taken)

    def _captured_update(self, world):
        """The update as nodes of the step's graph (called while capturing, behind the gradient pack).  One
        rank: the pack has taken the squared norm.  More ranks (DEMF_GRAPH_ALLREDUCE=1): the flat SUM
        all-reduce is captured too (torch's NCCL / RCCL process group records the collective into the
        capturing stream), then one reduction launch for the norm of the reduced gradients."""
        if world > 1:
            dist.all_reduce(self.flat.flat, op=dist.ReduceOp.SUM)
        self._finish_update(norm_taken=(world == 1))

    def flush(self):
        """Enqueue the update a deferred (overlapped) step still owes: call before reading parameters,
        gradients or optimizer state.  ``replay()``, ``step()`` and ``state_dict()`` do it themselves."""
        if getattr(self, "_pending_update", False):
            self._pending_update = False
            torch.cuda.current_stream().wait_stream(self._comm())
            self._finish_update()

    def _update(self, defer=False):
        if self.fused:
            self.flush()
            world, stub, overlap = self.allreduce_config()
            if (world > 1 or stub > 0) and overlap and defer:
                comm = self._comm()
                comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm):
                    self._collective(world, stub)
                self._pending_update = True
                return
            if world > 1 or stub > 0:
                self._collective(world, stub)
            self._finish_update()
            return
        self.flat.all_reduce_mean()
        self.flat.clip_(self.max_grad_norm)
        self.opt.step()

    def step(self, batch):
        if self.fused:
            self.flush()
        total = self._fwd_bwd(batch)
        self._update()
        return total

    @staticmethod
    def pad_targets(gt_boxes, gt_labels, G, device):
        """list of (n_i,7) / (n_i,) -> static-shape (B,G,7) float32, (B,G) int64 with label -1 on
        padding slots; an empty scene gets the reference's single all-zero fake box with label 0
        (class_agnostic_vote_head.py:766-773)."""
        B = len(gt_boxes)
        first = gt_boxes[0].tensor if hasattr(gt_boxes[0], "tensor") else gt_boxes[0]
        if first.is_cuda:
            # device-resident lists: padded on the device (one concatenation + one row gather per tensor,
            # DeMFVoteHead.pad_gt) - a `.cpu()` here would make the host wait for the step in flight and
            # serialise the input path with the GPU (measured: 6.66 instead of 5.6 ms per step)
            from .modules.head import DeMFVoteHead
            gt, lab, _ = DeMFVoteHead.pad_gt(gt_boxes, gt_labels, device, with_slot_labels=True, G=G)
            return gt, lab
        gt = torch.zeros((B, G, 7), dtype=torch.float32)
        lab = torch.full((B, G), -1, dtype=torch.int64)
        for i, (b, l) in enumerate(zip(gt_boxes, gt_labels)):
            b = b.tensor if hasattr(b, "tensor") else b
            n = int(b.shape[0])
            if n > G:
                raise ValueError(f"scene {i} has {n} ground-truth boxes, the captured step holds {G}")
            if n:
                gt[i, :n] = b.detach().float().cpu()
                lab[i, :n] = l.detach().cpu()
            else:
                lab[i, 0] = 0
        return gt.to(device), lab.to(device)

    def _snapshot_state(self):
        """What a forward + backward WITHOUT an optimizer update still changes: BatchNorm running
        statistics / counters and the dropout counter.  (-> restore closure)"""
        bufs = [(b, b.clone()) for b in self.model.buffers()]
        rng = None
        if self.fused:
            from . import fused
            rng = fused.get_rng_state(self.flat.flat.device)

        def restore():
            for b, c in bufs:
                b.copy_(c)
            if rng is not None:
                from . import fused
                fused.set_rng_state(self.flat.flat.device, rng)
        return restore

    def capture(self, batch, warmup=3, prefetch_geometry=True, max_gt=None, dry=False, geo_pipe=None,
                update_in_graph=None):
        """Capture forward + loss + backward of ``batch`` (static shapes, device-resident
        inputs) into one hipGraph; returns ``replay(next_points=None)`` = graph launch + eager
        all-reduce / clip / AdamW.  The path issues no host sync or host->device copy after
        warm-up (targets are batched, metas are cached), which is what makes it capturable; the
        collective and the optimizer stay outside the graph.

        The graph reads STATIC input buffers owned by the returned object: ``replay.load(batch)``
        copies another batch (same shapes; at most ``max_gt`` boxes per scene, default the
        largest count in the captured batch) into them and refreshes the cached per-image
        constants in place, so a training loop is

            replay = trainer.capture(batch_0)
            for k in range(steps):
                if k: replay.load(batch_k)
                loss = replay(next_points=batch_{k+1}["points"])

        Batches whose SHAPES differ (padded image size, point count, batch size) need their own
        graph: ``Trainer.bucketed()`` keeps one per shape.

        ``prefetch_geometry``: the coordinate-only pre-pass of the NEXT batch (every FPS level,
        the backbone ball queries, 3-NN - ``DeMFHotPath.index_geometry``) is issued on a side
        HIP stream in front of the current batch's fwd+bwd graph, so the latency-bound FPS chain
        (B workgroups on 256 CUs) runs underneath the current step's forward instead of in front
        of the next step (``DEMF_GEO_AT_BWD=1``: between a forward and a backward graph).  Every step still computes one full
        pre-pass; the graph reads it from static buffers that are refreshed by ONE ~6 MB multi-copy
        launch at the step boundary.

        ``dry``: the warm-up passes run forward + backward only (no optimizer update) and the state
        they touch (BatchNorm running statistics, dropout counter) is restored afterwards - a capture
        in the middle of training leaves the model exactly as it found it.
        ``geo_pipe``: the pre-pass pipeline (``replay.geo``) of another capture with the same cloud
        shape (B, N, channels): both graphs then read the same index buffers and share one
        pipelined pre-pass.
        ``update_in_graph`` (default: on one rank, ``DEMF_GRAPH_UPDATE=0`` turns it off): gradient norm
        (taken by the gradient pack on its way), clip and AdamW are nodes of the graph too - the
        optimizer's step count / learning-rate factor / norm live on the device (FlatAdamW.state), so a
        replay is the WHOLE step and nothing eager follows it.  With more than one rank the collective
        sits between backward and update and stays an eager RCCL call (``DEMF_GRAPH_ALLREDUCE=1``:
        captured as well, where the runtime allows it)."""
        dev = batch["points"].device
        world_c, stub_c, _ = self.allreduce_config()
        if update_in_graph is None:
            update_in_graph = bool(int(os.environ.get("DEMF_GRAPH_UPDATE", "1")))
        update_in_graph = bool(update_in_graph) and self.fused and stub_c == 0 and \
            (world_c == 1 or bool(int(os.environ.get("DEMF_GRAPH_ALLREDUCE", "0"))))
        gt_list = isinstance(batch["gt_bboxes_3d"], (list, tuple))
        if gt_list and dev.type == "cuda":
            G = max_gt or max(1, max(int((b.tensor if hasattr(b, "tensor") else b).shape[0])
                                     for b in batch["gt_bboxes_3d"]))
            gt, lab = self.pad_targets(batch["gt_bboxes_3d"], batch["gt_labels_3d"], G, dev)
        else:
            G, gt, lab = None, batch["gt_bboxes_3d"], batch["gt_labels_3d"]
        feats = batch["img_features"]
        head = getattr(self.model, "pts_bbox_head", None)
        # A pyramid handed over as the reference's list of (B,C,H_l,W_l) maps: the captured step reads TOKENS
        # (channels-last rows, padding zeroed) from a static buffer, and ``load`` converts every new batch
        # straight out of the caller's maps into that buffer (one tiled-transpose launch) - instead of copying
        # the 152 MB pyramid into static maps and transposing those inside the graph.
        tok_static = None
        if (not isinstance(feats, dict) and dev.type == "cuda" and hasattr(head, "pyramid_tokens")
                and int(os.environ.get("DEMF_ZERO_COPY_TOKENS", "1"))):
            tok_static = head.pyramid_tokens(feats, batch["img_metas"])
        static = dict(points=batch["points"].clone(),
                      img_features=(tok_static if tok_static is not None else
                                    dict(feats, tokens=feats["tokens"].clone())
                                    if isinstance(feats, dict) else [f.clone() for f in feats]),
                      img_metas=batch["img_metas"], gt_bboxes_3d=gt, gt_labels_3d=lab)
        batch = static
        if head is not None and hasattr(head, "pin_metas"):
            head.pin_metas(static["img_metas"])     # the graph holds raw pointers into its cache entry
        side = self.side_stream if getattr(self, "side_stream", None) is not None else \
            (concurrent_stream() if dev.type == "cuda" else torch.cuda.Stream())
        if geo_pipe is not None:
            side = geo_pipe.side
        restore = self._snapshot_state() if dry else None
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if dry:
                    self._fwd_bwd(batch)
                else:
                    self.step(batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        can_prefetch = prefetch_geometry and hasattr(self.model, "index_geometry")
        geo = None
        if can_prefetch:
            if geo_pipe is not None:
                if tuple(geo_pipe.static_pts.shape) != tuple(batch["points"].shape):
                    raise ValueError("geo_pipe was built for clouds of shape %s, this batch has %s"
                                     % (tuple(geo_pipe.static_pts.shape), tuple(batch["points"].shape)))
                geo = geo_pipe
                # the shared index buffers must describe THIS batch's cloud while it is being captured
                # (values do not matter to a capture, but the eager bookkeeping below assumes them)
                geo.ensure(torch.cuda.current_stream(), batch["points"], static["points"])
            else:
                geo = _GeoPipe(self.model, batch["points"], side)
        static_geo = geo.static_geo if geo is not None else None
        torch.cuda.synchronize()
        self.flat.reserve_table()
        graph = torch.cuda.CUDAGraph()
        graph_bwd = None
        if can_prefetch and os.environ.get("DEMF_GEO_AT_BWD"):
            # DEMF_GEO_AT_BWD=1: forward and backward as two graphs (one memory pool) with the pre-pass
            # started in between, underneath the backward (round 2's default: 9.26 -> 9.11 ms/step then).
            # With the one-pass backward kernels of round 3 the backward is ~230 mostly small launches and
            # every launch pays ~2 us more while a second hardware queue is active, so the pre-pass now
            # goes underneath the FORWARD (~110 launches) in front of a single fwd+bwd graph:
            # 6.49 -> 6.39 ms/step.
            self.flat.captured_pack = True
            self.flat.sumsq_state = self.opt.state if (update_in_graph and world_c == 1) else None
            try:
                with _gc_paused(), torch.cuda.graph(graph):
                    self._arena(True)            # the arena's single fill is the graph's first node
                    total = self._fwd(batch, static_geo)
                graph_bwd = torch.cuda.CUDAGraph()
                with _gc_paused(), torch.cuda.graph(graph_bwd, pool=graph.pool()):
                    self.flat.backward_into(total)
                    if update_in_graph:
                        self._captured_update(world_c)
            finally:
                self._arena(False)
                self.flat.captured_pack = False
                self.flat.sumsq_state = None
            loss = total.detach()
        else:
            self.flat.captured_pack = True
            self.flat.sumsq_state = self.opt.state if (update_in_graph and world_c == 1) else None
            try:
                with _gc_paused(), torch.cuda.graph(graph):
                    loss = self._fwd_bwd(batch, static_geo)
                    if update_in_graph:
                        self._captured_update(world_c)
            except Exception:
                if update_in_graph and world_c > 1:
                    # the runtime refused to capture the collective: fall back to the eager update
                    # (a second capture: the failed one left nothing behind)
                    self.flat.captured_pack = False
                    self.flat.sumsq_state = None
                    return self.capture(static, warmup=0, prefetch_geometry=prefetch_geometry, max_gt=max_gt,
                                        dry=dry, geo_pipe=geo if geo is not None else geo_pipe,
                                        update_in_graph=False)
                raise
            finally:
                self.flat.captured_pack = False
                self.flat.sumsq_state = None
        self.flat.finish_capture()
        if restore is not None:
            restore()
            torch.cuda.synchronize()
        one_deep = not os.environ.get("DEMF_GEO_TWO_DEEP")         # A/B: see replay()

        def replay(next_points=None):
            """One training step.  Default (one-deep): ``next_points`` is the cloud of the NEXT batch;
            its coordinate pre-pass is launched on the side stream in front of the step's graph and
            runs underneath the forward (with ``DEMF_GEO_AT_BWD=1``: between the forward and the
            backward graph); the result is moved into the static buffers at the end of the call.
            ``DEMF_GEO_TWO_DEEP=1``: ``next_points`` is the cloud of the batch AFTER the next one;
            the pre-pass is launched after the backward and runs underneath the optimizer update and
            the first layers of the next forward (measured: 7.80 vs 7.67 ms/step - the forward's
            statically striped pooled GEMMs lose more to the occupied CUs than the backward does;
            the step alone is 7.03 ms).  Any other usage stays correct: ``load`` checks whether the
            static / in-flight geometry belongs to the very tensor object it is given (same object,
            same version) and otherwise recomputes it.  ``next_points=None`` recomputes the
            pre-pass of the cloud last given (every step still pays for one full pre-pass)."""
            main = torch.cuda.current_stream()
            if update_in_graph:
                w_now, stub_now, _ = self.allreduce_config()
                if w_now != world_c or stub_now:
                    raise RuntimeError("this step was captured with its optimizer update inside the graph "
                                       "(world %d, no all-reduce stub); capture again with update_in_graph=False"
                                       % world_c)
            update = (lambda defer=False: None) if update_in_graph else self._update
            if can_prefetch and os.environ.get("DEMF_SKIP_GEO"):     # measurement only: the step alone
                self.flush()
                graph.replay()
                if graph_bwd is not None:
                    graph_bwd.replay()
                update()
                return loss
            if graph_bwd is not None:
                self.flush()
            if graph_bwd is not None and one_deep:
                graph.replay()
                geo.launch_prepass(main, next_points)
                graph_bwd.replay()
                update()
                geo.take_fresh(main)
                return loss
            if graph_bwd is not None:
                graph.replay()
                graph_bwd.replay()
                if geo.state["fresh"] is not None:
                    geo.take_fresh(main)              # no stall: that pre-pass had a whole step
                geo.launch_prepass(main, next_points)
                update()
                return loss
            if can_prefetch:
                # single-graph step (the default): the pre-pass goes first - enqueueing the
                # ~900-node step graph takes the host about a millisecond
                # (the owed update - norm + AdamW behind an overlapped collective - goes in front of the pre-pass
                # launch: behind it the step measured 0.17 ms slower, profiles/r05_allreduce_overlap_probe.log)
                self.flush()
                geo.launch_prepass(main, next_points)
            self.flush()
            graph.replay()
            update(defer=True)
            if can_prefetch:
                geo.take_fresh(main)
            return loss

        def load(new, skip_geo=False):
            """Copy another batch into the static input buffers of the captured step.  The captured
            forward reads its FPS / ball-query / 3-NN indices from the static geometry buffers; if
            they do not hold THIS cloud's geometry, it is fetched from the finished / in-flight
            pre-pass (waiting for it) or recomputed here, so geometry and targets can never belong
            to different batches.  ``skip_geo``: the caller has done that already (DoubleBufferedStep:
            on the main stream, while this call runs on its input stream)."""
            if can_prefetch and not skip_geo:
                geo.ensure(torch.cuda.current_stream(), new["points"], static["points"])
            static["points"].copy_(new["points"])
            nf = new["img_features"]
            metas_done = False
            if tok_static is not None and not isinstance(nf, dict):
                # the padding mask the conversion zeroes by is the NEW batch's: constants first
                head.refresh_metas(static["img_metas"], new["img_metas"])
                metas_done = True
                if head.pyramid_tokens(nf, static["img_metas"], out=tok_static) is None:
                    # (non-contiguous / non-fp32 maps or maps that require grad: the conversion did not run and
                    # the static token buffer would silently keep the previous batch's image features)
                    raise ValueError("load(): the new pyramid cannot be converted in place (it must be contiguous "
                                     "fp32 NCHW maps without requires_grad, as the captured batch's)")
            elif isinstance(nf, dict):
                if tok_static is not None and not nf.get("padding_zeroed"):
                    raise ValueError("this step was captured on a pyramid of feature maps: load() takes maps "
                                     "(or tokens built by DeMFVoteHead.pyramid_tokens)")
                static["img_features"]["tokens"].copy_(nf["tokens"])
            else:
                for d, f in zip(static["img_features"], nf):
                    d.copy_(f)
            if G is not None:
                nb = [b.tensor if hasattr(b, "tensor") else b for b in new["gt_bboxes_3d"]]
                if len(nb) <= 32 and all(b.is_cuda for b in nb) and all(l.is_cuda for l in new["gt_labels_3d"]):
                    # device lists: padded straight into the static buffers (one launch, no copies)
                    if max(int(b.shape[0]) for b in nb) > G:
                        raise ValueError("a scene has more ground-truth boxes than the captured step's %d slots" % G)
                    from . import ops
                    ops.pad_gt_lists(nb, list(new["gt_labels_3d"]), G,
                                     out=(static["gt_bboxes_3d"], static["gt_labels_3d"]))
                else:
                    g2, l2 = self.pad_targets(new["gt_bboxes_3d"], new["gt_labels_3d"], G, dev)
                    static["gt_bboxes_3d"].copy_(g2)
                    static["gt_labels_3d"].copy_(l2)
            else:
                static["gt_bboxes_3d"].copy_(new["gt_bboxes_3d"])
                static["gt_labels_3d"].copy_(new["gt_labels_3d"])
            if not metas_done and head is not None and hasattr(head, "refresh_metas"):
                head.refresh_metas(static["img_metas"], new["img_metas"])

        replay.load = load
        replay.update_in_graph = update_in_graph
        replay.static = static
        replay.geo = geo
        replay.max_gt = G
        self._graph = graph
        return replay

    def capture_double(self, batch_a, batch_b, **kw):
        """Two captured steps over two sets of static input buffers (``DoubleBufferedStep``): the next batch is
        loaded into the idle set on an input stream WHILE the current step runs."""
        return DoubleBufferedStep(self, batch_a, batch_b, **kw)

    def bucketed(self, **kw):
        """A step that keeps one captured graph per input SHAPE (``StepCache``)."""
        return StepCache(self, **kw)


def _flat_tensors(g):
    """Every tensor of the (nested) geometry structure, in a deterministic order."""
    if torch.is_tensor(g):
        return [g]
    if isinstance(g, dict):
        return [t for k in sorted(g) for t in _flat_tensors(g[k])]
    return [t for v in g for t in _flat_tensors(v)]


def _batch_tensors(batch):
    """Every tensor of a batch dict (points, pyramid maps / token dict, GT lists or padded GT)."""
    out = []

    def walk(v):
        if torch.is_tensor(v):
            out.append(v)
        elif hasattr(v, "tensor") and torch.is_tensor(v.tensor):
            out.append(v.tensor)
        elif isinstance(v, dict):
            for k, x in v.items():
                if k != "img_metas":
                    walk(x)
        elif isinstance(v, (list, tuple)):
            for x in v:
                walk(x)
    walk({k: v for k, v in batch.items() if k != "img_metas"})
    return out


class _Cloud:
    """Which cloud a set of index buffers describes.  Identity is the tensor OBJECT handed in (held
    here, so the caching allocator cannot hand its address to the next batch) plus its version
    counter; an address / shape tag would match a recycled buffer and silently pair new points with
    old indices."""
    __slots__ = ("t", "v")

    def __init__(self, t):
        self.t, self.v = t, t._version

    def holds(self, t):
        return self.t is t and self.v == t._version


class _GeoPipe:
    """The pipelined coordinate pre-pass of ONE cloud shape (B, N, channels): static index buffers that
    captured step graphs read, a small hipGraph that recomputes them for another cloud on a side stream
    (``DeMFHotPath.index_geometry``: every FPS level, ball queries + inverse lists, 3-NN, the head's seed
    FPS), and the one-launch refresh that moves a finished pre-pass into the static buffers.  Shared by
    every step graph captured for that cloud shape (graphs for different padded image sizes read the
    same index buffers)."""

    def __init__(self, model, points, side):
        from . import ops
        self.side = side
        self.static_geo = model.index_geometry(points)
        torch.cuda.synchronize()
        # the pre-pass is its own (small) hipGraph, replayed on the side stream: one launch
        # call instead of ~30, so the main graph is not held up behind host launch latency
        self.static_pts = points.clone()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with _gc_paused(), torch.cuda.graph(self.graph, stream=side):
            fresh = _flat_tensors(model.index_geometry(self.static_pts))
        torch.cuda.synchronize()
        static_flat = _flat_tensors(self.static_geo)
        # one launch instead of one copy per index tensor (~30 of them)
        pairs = [(d, s) for d, s in zip(static_flat, fresh) if d.numel()]
        self.refresh = ops.MultiCopy([d for d, _ in pairs], [s for _, s in pairs])
        # Which cloud the static geometry buffers hold and which cloud's pre-pass sits in `fresh`
        # (in flight or finished).
        captured = _Cloud(points)
        self.state = dict(static=captured, fresh=None, pts=captured)   # static_pts holds the captured cloud

    def take_fresh(self, main):
        """fresh -> static (one launch), after the pre-pass that filled `fresh` has finished."""
        main.wait_stream(self.side)
        self.refresh()
        self.state["static"], self.state["fresh"] = self.state["fresh"], None

    def launch_prepass(self, main, next_points):
        if next_points is not None:
            self.static_pts.copy_(next_points)
            self.state["pts"] = _Cloud(next_points)
        self.side.wait_stream(main)                # the cloud is in place, `fresh` has been consumed
        with torch.cuda.stream(self.side):
            self.graph.replay()
        self.state["fresh"] = self.state["pts"]

    def ensure(self, main, points, static_points=None):
        """Make the static index buffers describe ``points``: already there, taken from the finished /
        in-flight pre-pass (waiting for it), or recomputed here."""
        st = self.state
        if st["static"].holds(points) or (static_points is not None and st["static"].holds(static_points)
                                          and points is static_points):
            return
        if st["fresh"] is not None and st["fresh"].holds(points):
            self.take_fresh(main)
        else:
            # (an unrelated pre-pass in flight is left to finish; its result is dropped)
            self.launch_prepass(main, points)
            self.take_fresh(main)


class DoubleBufferedStep:
    """The captured step with DOUBLE-BUFFERED static inputs: two hipGraphs of the same step (same model, optimizer,
    pre-pass pipeline) reading two sets of static input buffers.  ``load(batch)`` fills the set the NEXT call will
    read, on an input stream, so the per-batch input path - the pyramid's conversion to tokens (63 us of HBM traffic),
    the copies of points / padded targets / constants - runs underneath the step in flight instead of between two
    steps (the host runs about one step ahead of the GPU, so ``load(k+1)`` is enqueued while step k executes).
    Same calling convention as ``Trainer.capture``'s replay:

        step = trainer.capture_double(batch_0, batch_1)
        for k in range(steps):
            if k: step.load(batch_k)            # k = 0: batch_0 is in place
            loss = step(next_points=batch_{k+1}["points"])

    Ordering: a set is overwritten only after the last step that read it has finished (event), and a step waits for
    its set's load (event).  The reference's loader does the same thing with pinned-memory prefetch threads
    (mmcv's DataLoader workers feeding MMDistributedDataParallel, /root/reference/train.py:140-147)."""

    def __init__(self, trainer, batch_a, batch_b, **kw):
        if batch_a["img_metas"] is batch_b["img_metas"]:
            raise ValueError("capture_double: the two batches must carry their own img_metas objects (each set of "
                             "static buffers owns the device constants cached for its metas)")
        self.trainer = trainer
        first = trainer.capture(batch_a, **kw)
        kw2 = dict(kw)
        kw2.update(dry=True, geo_pipe=first.geo)
        second = trainer.capture(batch_b, **kw2)
        self.slots = [first, second]
        self.geo, self.max_gt = first.geo, first.max_gt
        self.input_stream = concurrent_stream()
        self.loaded = [None, None]           # event: the set's load has finished
        self.done = [None, None]             # event: the last step that read the set has finished
        self.nxt = 0                         # the set the next call replays (set 0 holds batch_a)
        # the shared pre-pass pipeline must describe batch_a's cloud for the first call
        if self.geo is not None:
            self.geo.ensure(torch.cuda.current_stream(), batch_a["points"])

    @property
    def static(self):
        return self.slots[self.nxt].static

    def load(self, new):
        s = self.nxt
        main = torch.cuda.current_stream()
        if self.slots[s].geo is not None:
            # geometry bookkeeping (normally a no-op: the pre-pass of this cloud was launched under the previous step
            # and moved into the static index buffers at its end) belongs to the main stream
            self.slots[s].geo.ensure(main, new["points"])
        ins = self.input_stream
        if self.done[s] is not None:
            ins.wait_event(self.done[s])
        else:
            ins.wait_stream(main)            # first use of the set: behind its capture
        # The tensors of ``new`` (H2D copies, image-stream outputs, GT lists) are normally produced on the
        # caller's current stream: the input stream must not read them before they are written, and the
        # caching allocator must not hand their memory out again while the input stream still reads it.
        ready = torch.cuda.Event()
        ready.record(main)
        ins.wait_event(ready)
        for t in _batch_tensors(new):
            if t.is_cuda:
                t.record_stream(ins)
        with torch.cuda.stream(ins):
            self.slots[s].load(new, skip_geo=True)
            ev = torch.cuda.Event()
            ev.record(ins)
        self.loaded[s] = ev

    def __call__(self, next_points=None):
        s = self.nxt
        main = torch.cuda.current_stream()
        if self.loaded[s] is not None:
            main.wait_event(self.loaded[s])
            self.loaded[s] = None
        out = self.slots[s](next_points=next_points)
        ev = torch.cuda.Event()
        ev.record(main)
        self.done[s] = ev
        self.nxt = 1 - s
        return out


class StepCache:
    """Training step over batches of VARYING shape: one captured hipGraph per shape key

        (scenes, points, point channels, image-pyramid shapes, GT slots)

    in an LRU of ``max_graphs``, an eager step for shapes seen fewer than ``capture_on`` times.  The
    reference rebuilds its masks and shapes per batch from ``img_metas[0]['batch_input_shape']``
    (demf/modeling/heads/class_agnostic_vote_head.py:556-568) and its pipeline pads every batch to a
    multiple of 32 after ``Resize((1333, 800), keep_ratio)`` (configs/demf/demf_votenet.py:194-197): SUN
    RGB-D's four sensors give four padded image sizes, so a single static-shape graph would only ever
    serve one of them.  GT slots are bucketed (8 / 16 / 32 / 64) so that box counts do not multiply the
    graphs.  Graphs of the same cloud shape share one pipelined coordinate pre-pass (``_GeoPipe``).
    Captures are ``dry``: they leave parameters, optimizer state, BatchNorm statistics and the dropout
    counter untouched, so the sequence of updates equals that of eager steps."""

    GT_BUCKETS = (8, 16, 32, 64)

    def __init__(self, trainer, max_graphs=4, capture_on=2, warmup=2, prefetch_geometry=True):
        import collections
        self.trainer, self.max_graphs, self.capture_on = trainer, int(max_graphs), int(capture_on)
        self.warmup, self.prefetch = int(warmup), prefetch_geometry
        self.graphs = collections.OrderedDict()       # key -> replay
        self.pipes = {}                               # cloud shape -> _GeoPipe
        self.seen = collections.Counter()
        self.stats = dict(eager=0, replayed=0, captured=0, evicted=0)

    @classmethod
    def gt_slots(cls, gt_boxes):
        if not isinstance(gt_boxes, (list, tuple)):
            return int(gt_boxes.shape[1])
        n = max(1, max(int((b.tensor if hasattr(b, "tensor") else b).shape[0]) for b in gt_boxes))
        for g in cls.GT_BUCKETS:
            if n <= g:
                return g
        raise ValueError("a scene has %d ground-truth boxes; the target kernels hold %d" % (n, cls.GT_BUCKETS[-1]))

    @classmethod
    def key(cls, batch):
        f = batch["img_features"]
        if isinstance(f, dict):
            img = (tuple(f["tokens"].shape), tuple(tuple(s) for s in f["spatial"]))
        else:
            img = tuple(tuple(t.shape) for t in f)
        return (tuple(batch["points"].shape), img, cls.gt_slots(batch["gt_bboxes_3d"]))

    def step(self, batch, next_points=None):
        """One optimizer step on ``batch``.  ``next_points``: the cloud of the next batch (its
        coordinate pre-pass then runs underneath this step if the next batch replays a graph of the
        same cloud shape)."""
        key = self.key(batch)
        r = self.graphs.get(key)
        if r is None:
            self.seen[key] += 1
            if self.seen[key] < self.capture_on:
                self.stats["eager"] += 1
                return self.trainer.step(batch)
            while len(self.graphs) >= self.max_graphs:
                old, dead = self.graphs.popitem(last=False)
                self.stats["evicted"] += 1
                # (the pipe - static index buffers + the pre-pass hipGraph - of a cloud shape survives while any
                # graph of that shape does, INCLUDING the one about to be captured)
                if old[0] != key[0] and not any(k[0] == old[0] for k in self.graphs):
                    self.pipes.pop(old[0], None)
                head = getattr(self.trainer.model, "pts_bbox_head", None)
                if head is not None and hasattr(head, "unpin_metas"):
                    head.unpin_metas(dead.static["img_metas"])
                del dead
            pipe = self.pipes.get(key[0]) if self.prefetch else None
            r = self.trainer.capture(batch, warmup=self.warmup, prefetch_geometry=self.prefetch,
                                     max_gt=key[2] if isinstance(batch["gt_bboxes_3d"], (list, tuple)) else None,
                                     dry=True, geo_pipe=pipe)
            if self.prefetch and r.geo is not None:
                self.pipes[key[0]] = r.geo
            self.graphs[key] = r
            self.stats["captured"] += 1
        else:
            self.graphs.move_to_end(key)
        r.load(batch)
        self.stats["replayed"] += 1
        if next_points is not None and tuple(next_points.shape) != key[0]:
            next_points = None                      # another cloud shape: its own pipeline fetches it
        return r(next_points=next_points)

    __call__ = step
