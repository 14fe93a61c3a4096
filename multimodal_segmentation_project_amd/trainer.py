"""TrainStep / DannStep — the reference's per-step hot loops as native, hipGraph-capturable kernel sequences with
data-parallel gradient all-reduce over RCCL.

    TrainStep   train_unet.py:220-252 (== finetune_ct.py:159-191), distill_unet.py:107-133
        zero_grad -> model(images) -> loss_fn -> backward(loss/accum) -> [all-reduce grads / world] -> AdamW.step
        -> calculate_{iou,dice,accuracy} -> gather(loss, iou, dice, acc).mean()
    DannStep    train_dann.py:233-300
        source fwd -> target fwd (full net) -> GAP x2 -> grad_reverse x2 -> discriminator -> CE
        -> total = task + lambda*domain (lambda applied twice, Q3) -> ONE backward over both graphs -> two AdamW steps

Design (MI355X-first, not a DDP translation):
  * parameters, gradients and AdamW moments live in flat fp32 arenas (22.6 MB each); nn.Parameters are views, so
    state_dict()/load_state_dict() keep working, while the optimizer is ONE kernel and gradient all-reduce operates
    on contiguous arena ranges (no bucket copy-in/copy-out).
  * backward runs as runs of segments with TWO exchanges by default (dp.bucket_ranges): after encoder.L-1's segment the
    arena range [encoder.L-1 .. final_conv] (97 % of the gradient bytes) is all-reduced on a side stream (RCCL over xGMI)
    while the bandwidth-heavy encoder.L-2..0 backward runs; [encoder.0..L-2] follows at the end (MI3D_FINE_BUCKETS=1: four
    readiness-ordered buckets, measured slower).  The four scalar gathers of the reference (C4) are one 4-float all-reduce
    that rides on the first exchange.
  * a step is captured into hipGraphs: ONE graph at world 1; at world > 1 the step is launched eagerly (or, on request, as
    one graph per comm-free run of kernels with the all-reduces between them: RCCL calls stay outside the graphs).
  * BatchNorm statistics are per-GPU local (DDP + BatchNorm3d semantics); the per-forward buffer broadcast of DDP
     is replaced by `sync_buffers()` before eval/checkpoint.
  * no host synchronisation inside step(); results are device tensors.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib, engine
from ._lib import Mi3dError, call, ptr, ptr_table, stream_ptr
from . import metrics as M
from .dp import DataParallelComm, ParamArena

_LOSS_TABLE = {      # name -> (w_ce, region_kind, w_reg, alpha, beta, eps); train_unet.py:178-205
    "combined": (1.0, 1, 1.0, 0.0, 0.0, 1e-5), "dice": (0.0, 1, 1.0, 0.0, 0.0, 1e-5),
    "tversky": (0.0, 2, 1.0, 0.5, 0.5, 1e-6), "ce_tversky": (0.3, 2, 0.7, 0.5, 0.5, 1e-6),
    "ce": (1.0, 0, 0.0, 0.0, 0.0, 1e-6),
}


def _loss_cfg(kind, kd_alpha=None, temperature=2.0):
    if kd_alpha is not None:      # distillation_loss, utils/metrics.py:169-190
        return M._cfg(0.3 * kd_alpha, 2, 0.7 * kd_alpha, 0.7, 0.3, 1e-6, 1.0 - kd_alpha, temperature)
    if kind not in _LOSS_TABLE:
        raise Mi3dError(f"unknown loss {kind!r}: one of {sorted(_LOSS_TABLE)} (train_unet.py:538 --loss choices)")
    return M._cfg(*_LOSS_TABLE[kind])


def _priority_stream(device, cls):
    """torch stream object around a hipStream_t of priority class cls (-1 highest, 0 middle, +1 lowest; mi3d_stream_create).
    The handle lives as long as the process: streams are few and are shared through the per-device cache below."""
    h = C.c_void_p()
    call("mi3d_stream_create", int(cls), C.byref(h))
    st = torch.cuda.ExternalStream(h.value, device=device)
    st.mi3d_handle = h
    return st


_STREAM_CACHE = {}      # (device index, priority, role) -> stream chosen by concurrent_stream
_CAPTURE_STREAMS = {}   # device index -> the stream hipGraph captures run on (nothing else ever does)


def _capture_stream(device):
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[key] = _priority_stream(dev, 0)
    return _CAPTURE_STREAMS[key]


def _masked_stream(device, cus_per_xcd, from_top=1):
    """torch stream object around a hipStream_t confined to cus_per_xcd CUs of every XCD (mi3d_stream_create_masked)."""
    h = C.c_void_p()
    call("mi3d_stream_create_masked", int(cus_per_xcd), int(from_top), C.byref(h))
    st = torch.cuda.ExternalStream(h.value, device=device)
    st.mi3d_handle = h
    st.mi3d_cus_per_xcd = int(cus_per_xcd)
    return st


def concurrent_stream(device, candidates=6, hold_us=300, priority="high", role="aux", cus_per_xcd=0):
    """A stream whose kernels really run BESIDE those of the current (compute) stream.  HIP multiplexes streams onto a few
    hardware queues, and two streams that share a queue execute strictly one after the other (measured with rocprofv3: the
    default stream and the 8th stream created in a process both sat on queue 4, and a collective kernel on the latter ran
    between, not beside, the backward kernels -- profiles/r03_dp_streams.txt).  So the stream is CHOSEN by a measurement: a
    stand-in kernel that holds a few workgroups for `hold_us` is launched on the candidate and on the compute stream; if the
    pair takes about one hold time they overlap (minimum of three timed repetitions: a busy GPU only ever makes a pair look
    slower).  The first candidate has the requested priority class ("high": its own queue on this runtime; "low": the lowest
    class, mi3d_stream_create).  The choice is cached per (device, priority, role): every step object of a process shares it
    and the probe runs once.  When no candidate overlaps, the last one is returned with mi3d_concurrent = False and a
    RuntimeWarning: the step is still correct, only nothing will run beside the compute stream."""
    import warnings
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), priority, role, int(cus_per_xcd))
    if key in _STREAM_CACHE:
        return _STREAM_CACHE[key]
    main = torch.cuda.current_stream(device)
    buf = torch.zeros(1024, dtype=torch.float32, device=device)
    best = None
    for i in range(candidates):
        if i == 0 and cus_per_xcd > 0:
            c = _masked_stream(device, cus_per_xcd)
        elif i == 0 and priority == "low":
            c = _priority_stream(device, +1)
        elif i == 0 and priority == "high":
            c = torch.cuda.Stream(device=device, priority=-1)
        else:
            c = torch.cuda.Stream(device=device)
        times = []
        for rep in range(4):                       # first pass loads the code object
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record(main)
            c.wait_stream(main)
            call("mi3d_debug_occupy_cus", 4, hold_us, buf.data_ptr(), buf.numel(), c.cuda_stream)
            call("mi3d_debug_occupy_cus", 4, hold_us, buf.data_ptr(), buf.numel(), main.cuda_stream)
            main.wait_stream(c)
            e1.record(main)
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        c.mi3d_overlap_us = min(times[1:])
        c.mi3d_concurrent = c.mi3d_overlap_us < 1.5 * hold_us
        best = c
        if c.mi3d_concurrent:
            break
    if not best.mi3d_concurrent:
        warnings.warn(f"concurrent_stream: none of {candidates} candidate streams ran beside the compute stream on {dev} "
                      f"(pair time {best.mi3d_overlap_us:.0f} us for 2 x {hold_us} us): aux-stream work will serialise",
                      RuntimeWarning, stacklevel=2)
    _STREAM_CACHE[key] = best
    return best


class ArenaAdamW(torch.optim.Optimizer):
    """`param_groups` surface of the arena optimizer, so that torch.optim.lr_scheduler objects (the reference's
    ReduceLROnPlateau(mode='max', patience=10, factor=0.1, min_lr=1e-6), train_unet.py:381,442) can drive the learning
    rate: `scheduler = ReduceLROnPlateau(step.optimizer, ...); scheduler.step(val_dice)`.  The update itself is the
    fused kernel inside TrainStep.step(); calling .step() here is an error."""

    def __init__(self, params, lr, betas, eps, weight_decay):
        super().__init__(list(params), dict(lr=float(lr), betas=tuple(betas), eps=float(eps),
                                            weight_decay=float(weight_decay)))

    def step(self, closure=None):
        raise Mi3dError("ArenaAdamW.step(): the AdamW update runs inside TrainStep.step() / DannStep.step()")


class _Graphs:
    """hipGraph executables of one static state: variant -> [(graph_exec, comm_fn_or_None), ...]"""

    def __init__(self):
        self.by_variant = {}
        self.hyper = None

    def drop(self):
        for segs in self.by_variant.values():
            for g, _ in segs:
                if g is not None and g.value:
                    _lib.lib().mi3d_graph_destroy(g)
        self.by_variant = {}


class _StepBase:
    """Arena / communication / graph machinery shared by TrainStep and DannStep."""

    def _init_common(self, model, lr, weight_decay, betas, eps, grad_accum, process_group, compute_dtype, use_graph,
                     force_comm):
        self.model = model
        self.accum = int(grad_accum)
        if self.accum < 1:
            raise Mi3dError("grad_accum must be >= 1")
        self.micro = 0
        self.dtype = compute_dtype
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        p0 = next(model.parameters())
        _lib.require_cuda(p0, type(self).__name__)
        self.device = p0.device
        self.arena = ParamArena(model.parameters(), self.device)
        self.optimizer = ArenaAdamW(model.parameters(), lr, betas, eps, weight_decay)
        n_levels = len(model.encoder)
        # MI3D_FORCE_COMM=1 / force_comm: drive the bucket path (side stream, all-reduce, joins, segmented graphs) even
        # at world 1 -- a 1-rank RCCL group on the 1-GPU box exercises the real code path
        self.force_comm = bool(force_comm) or os.environ.get("MI3D_FORCE_COMM", "0") == "1"
        self.do_comm = self.world > 1 or (self.force_comm and dist.is_available() and dist.is_initialized())
        self.comm = DataParallelComm(self.arena, n_levels, process_group, force=self.do_comm,
                                     fine_buckets=os.environ.get("MI3D_FINE_BUCKETS", "0") == "1")
        # the exchange stream must sit on another hardware queue than the compute stream (see concurrent_stream)
        self.comm_stream = concurrent_stream(self.device, role="comm") if self.do_comm else None
        self._mark_handles = []          # exchange-mark events (mi3d_unet_backward_marks)
        self.use_graph = bool(use_graph)
        self._statics = {}
        self._static = None
        self.inv_accum = torch.full((), 1.0 / self.accum, dtype=torch.float32, device=self.device)
        self._closed = False

    # ---- optimizer surface
    @property
    def lr(self):
        return self.optimizer.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        """Learning rate (also reachable through `self.optimizer.param_groups`, i.e. torch LR schedulers).  The value
        is a kernel argument: captured step graphs are re-captured on the next step when it changed."""
        for g in self.optimizer.param_groups:
            g["lr"] = float(value)

    def _hyper(self):
        g = self.optimizer.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def _drop_graphs(self):
        for st in self._statics.values():
            st["graphs"].drop()

    def close(self):
        """Destroy captured graph executables and events (also run by __del__)."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        try:
            self._drop_graphs()
            for e in getattr(self, "_mark_handles", []):
                _lib.lib().mi3d_event_destroy(e)
            self._mark_handles = []
            self._close_extra()
        except Exception:      # noqa: BLE001  (interpreter shutdown: the library may already be gone)
            pass

    def _close_extra(self):
        pass

    def __del__(self):
        self.close()

    # ---- DDP construction semantics: rank 0's parameters and buffers win
    def broadcast_parameters(self):
        self.comm.broadcast_parameters(self.model.buffers())

    def sync_buffers(self):
        self.comm.sync_buffers(self.model.buffers())

    # ---- exchange points
    def _on_comm_stream(self, fn):
        """Run fn on the communication stream, ordered after everything enqueued so far on the compute stream; the
        compute stream keeps going."""
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream())
        aux = getattr(self, "aux_stream", None)
        if aux is not None:        # gradients the deferred weight-gradient kernels are still writing (TrainStep aux_wgrad)
            cs.wait_stream(aux)
        with torch.cuda.stream(cs):
            fn()

    def _join_comm(self):
        torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _param_versions(self):
        """Sum of the version counters of the model's Parameters (each is a view of the arena with its OWN counter)."""
        return sum(p._version for p in self.arena.params)

    def _adamw(self, arena, ranges, hyper, s):
        lr, b1, b2, eps, wd = hyper
        for k, (lo, hi) in enumerate(ranges):
            call("mi3d_adamw_apply", arena.p.data_ptr() + 4 * lo, arena.g.data_ptr() + 4 * lo,
                 arena.m.data_ptr() + 4 * lo, arena.v.data_ptr() + 4 * lo, hi - lo, lr, b1, b2, eps, wd, 1.0,
                 ptr(arena.step), int(k == len(ranges) - 1), s)

    @staticmethod
    def _trainable_ranges(arena, trainable):
        ranges = []
        for i, t in enumerate(trainable):
            if not t:
                continue
            lo, hi = arena.range_of(i, i + 1)
            if ranges and ranges[-1][1] == lo:
                ranges[-1] = (ranges[-1][0], hi)
            else:
                ranges.append((lo, hi))
        return ranges

    # ---- graph capture / replay.  `variant` = (first micro-step of an accumulation window, boundary micro-step)
    def _variant(self):
        return (self.micro % self.accum == 0, (self.micro + 1) % self.accum == 0)

    def _run(self, st, last_batch=False):
        """last_batch: the final batch of an epoch.  The reference's loaders are accelerator.prepare()'d (train_unet.py:384),
        so accelerate's accumulate() forces gradient sync + optimizer.step on the LAST batch of every epoch and restarts its
        step count (GradientState.end_of_dataloader; train_dann.py:287 does the same by hand): the micro-step becomes a
        boundary whatever its position in the window, and the next epoch starts a fresh window."""
        self._check_mode()
        variant = self._variant()
        bump = 1
        if last_batch and not variant[1]:
            variant = (variant[0], True)
            bump = self.accum - (self.micro % self.accum)
        if not self.use_graph:
            self._enqueue(st, variant, lambda fn: fn())
            self.micro += bump
            return
        gr = st["graphs"]
        hyper = self._hyper()
        if gr.hyper != hyper:
            gr.drop()
            gr.hyper = hyper
        segs = gr.by_variant.get(variant)
        if segs is None:
            segs = gr.by_variant[variant] = self._capture(st, variant)
        for g, fn in segs:
            call("mi3d_graph_launch", g, stream_ptr())
            if fn is not None:
                fn()
        self.micro += bump

    def _mutable_state(self):
        """Tensors a step execution modifies (snapshot/restore around the capture warm-up)."""
        a = self.arena
        return [a.p, a.g, a.m, a.v, a.step] + list(self.model.buffers()) + [engine._rng_state(self.model, self.device)]

    def _capture(self, st, variant):
        """Capture one micro-step into hipGraph(s).  A warm-up execution is needed first (lazy code-object loading
        must not happen inside the capture); it runs on a snapshot of all mutable state, which is restored afterwards,
        so capturing is invisible to the training trajectory.  Communication functions are not captured: they cut the
        graph, and replay launches them between the segments."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        segs = []
        with torch.cuda.stream(s):
            state = self._mutable_state()
            snap = [t.clone() for t in state]
            self._enqueue(st, variant, lambda fn: fn())
            if self.do_comm:
                self._join_comm()
                self.comm_stream.synchronize()
            s.synchronize()
            for t, c in zip(state, snap):
                t.copy_(c)
            s.synchronize()
        # The capture itself runs on a PRIVATE stream that never carries a collective.  The warm-up above issued the last bucket's
        # all-reduce on `s` (and torch hands out `s` from a pool other step objects' exchange streams come from): the process
        # group's watchdog thread polls the end events of such work for up to its 100 ms period, and a poll that meets a stream
        # in capture ends the capture with hipErrorCapturedEvent (seen once in ~10 runs of the forced-exchange test).
        cs = _capture_stream(self.device)
        cs.wait_stream(s)
        with torch.cuda.stream(cs):
            def cut(fn):
                g = C.c_void_p()
                call("mi3d_graph_end", cs.cuda_stream, C.byref(g))
                segs.append((g, fn))
                call("mi3d_graph_begin", cs.cuda_stream)

            call("mi3d_graph_begin", cs.cuda_stream)
            try:
                self._enqueue(st, variant, cut)
            finally:
                g = C.c_void_p()
                call("mi3d_graph_end", cs.cuda_stream, C.byref(g))
                segs.append((g, None))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.current_stream().wait_stream(cs)
        return segs


class TrainStep(_StepBase):
    def __init__(self, model, loss="combined", lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8,
                 grad_accum=1, kd_teacher=None, kd_alpha=0.7, kd_temperature=2.0, process_group=None,
                 compute_dtype=None, use_graph=False, aux_wgrad=None, reference_zero_grad_quirk=False,
                 force_comm=False, overlap_teacher=True, keep_logits=False, aux_cus=None):
        """aux_wgrad: the backward's critical path is the input-gradient chain alone -- the weight gradients of the decoder's
        full-resolution convs and of all deep-level convs run on a second stream (include/mi3d.h, mi3d_unet_backward: aux_stream),
        forked from the chain at most three times and joined in front of the optimizer; bit-identical results.  None (default):
        on for eager launches without a gradient exchange, off under use_graph -- this runtime's hipGraph executor spreads a forked graph over three hardware
        queues and replays it at half speed (4.4 vs 2.2 ms, profiles/r04_defer_graph_streams.txt), eager launches do not.
        keep_logits: the step folds the 1x1x1 head into the loss and never writes the logits (mi3d_unet_forward_loss); True
        keeps a copy in the static buffer `logits` for callers that read them after step().
        reference_zero_grad_quirk: train_unet.py:222 / finetune_ct.py:161 call optimizer.zero_grad() INSIDE
        accelerator.accumulate(), where accelerate only really zeroes on the boundary micro-step -> the gradient the
        reference applies is grad(last micro-batch)/accum.  False (default): accumulate all micro-batches
        (what distill_unet.py:114-115 does and what the flag's name promises); True: reproduce the reference's
        train/finetune behaviour bit for bit (non-boundary micro-steps still run forward, BN statistics and metrics)."""
        cfg = _loss_cfg(loss, kd_alpha if kd_teacher is not None else None, kd_temperature)     # validates the name first
        self._init_common(model, lr, weight_decay, betas, eps, grad_accum, process_group, compute_dtype, use_graph,
                          force_comm)
        self.teacher = kd_teacher
        self.loss_kind = loss
        self.cfg = cfg
        self.keep_logits = bool(keep_logits)
        # evaluate(): train_unet.py:259-305 uses the training loss_fn; distill_unet.py:149 uses combined_loss
        self.eval_cfg = _loss_cfg("combined" if kd_teacher is not None else loss)
        self.quirk = bool(reference_zero_grad_quirk) and kd_teacher is None
        if kd_teacher is not None:
            self._check_teacher(kd_teacher)
        # distillation (distill_unet.py:107-112): the frozen teacher's forward does not depend on the student's -- it runs on
        # its own stream beside the student forward (fork after the batch is in place, join in front of the loss; both
        # branches are captured into the one step graph).  Separate workspaces and logits: results are bitwise those of the
        # serial order.  Both forwards are long chains that leave most CUs idle at the deep levels, so they interleave.
        self.kd_stream = concurrent_stream(self.device, role="kd") if (kd_teacher is not None and overlap_teacher) else None
        # second compute stream: the deferred weight-gradient kernels run beside the data-gradient chain (mi3d_unet_backward)
        # (priority classes measured at 96^3, eager: high 2.111 / normal 2.121 / low 2.124 ms -- profiles/r04_experiments_aux_wgrad.txt)
        if aux_wgrad is None:
            # not with a gradient exchange either: the deferred deep-level weight gradients are 95 % of the gradient bytes and
            # finish ~150 us before the end of the backward, so the big all-reduce bucket (dp.bucket_ranges) would lose the
            # 0.45 ms of encoder backward it hides under; the exchange is worth more than the ~20 us the deferral buys
            aux_wgrad = not use_graph and not self.do_comm
        self.aux_stream = (concurrent_stream(self.device, priority=os.environ.get("MI3D_AUX_PRIO", "high"),
                                             cus_per_xcd=int(aux_cus if aux_cus is not None else os.environ.get("MI3D_AUX_CUS", "0")))
                           if aux_wgrad else None)
        self._events = None
        self._event_handles = []
        if aux_wgrad:
            for _ in range(4):
                e = C.c_void_p()
                call("mi3d_event_create", C.byref(e))
                self._event_handles.append(e.value)
            self._events = ptr_table(self._event_handles)
        if self.world > 1:
            self.broadcast_parameters()

    def _close_extra(self):
        for e in self._event_handles:
            _lib.lib().mi3d_event_destroy(e)
        self._event_handles = []

    def _check_teacher(self, teacher):
        """distill_unet.py:20-29 loads the teacher with map_location: it must end up on the student's device with
        the student's architecture, else the kernels would dereference foreign addresses."""
        sp, tp = list(self.model.parameters()), list(teacher.parameters())
        sb, tb = list(self.model.buffers()), list(teacher.buffers())
        if len(sp) != len(tp) or len(sb) != len(tb):
            raise Mi3dError(f"kd_teacher has {len(tp)} parameters / {len(tb)} buffers, the student {len(sp)} / {len(sb)}: "
                            "teacher and student must be the same UNet3D architecture (distill_unet.py:214-215)")
        for kind, mine, theirs in (("parameter", sp, tp), ("buffer", sb, tb)):
            for i, (a, b) in enumerate(zip(mine, theirs)):
                if b.device != self.device:
                    raise Mi3dError(f"kd_teacher {kind} {i} is on {b.device}, the student on {self.device}: move the "
                                    "teacher with .to(device) first (no CPU fallback)")
                if tuple(a.shape) != tuple(b.shape) or a.dtype != b.dtype:
                    raise Mi3dError(f"kd_teacher {kind} {i}: shape/dtype {tuple(b.shape)}/{b.dtype} != student's "
                                    f"{tuple(a.shape)}/{a.dtype}")
                if not b.is_contiguous():
                    raise Mi3dError(f"kd_teacher {kind} {i} is not contiguous")

    def reset_optimizer(self):
        """What rebuilding ``optim.AdamW`` does in the reference when the encoder is (un)frozen
        (train_unet.py:413-431, finetune_ct.py:374-381): moments and step count start from zero."""
        self.arena.m.zero_()
        self.arena.v.zero_()
        self.arena.step.zero_()
        self._drop_graphs()
        self._statics = {}
        self._static = None

    # ---- static state for one (shape, dtype, trainable set, mode)
    def _prepare(self, x, mode="train"):
        dt = self.dtype or engine.resolve_compute_dtype(self.model)
        # frozen parameters (requires_grad=False: encoder/bottleneck freezing of train_unet.py:31-43, finetune_ct.py:270-304)
        # are part of the static state: they get no gradient, no optimizer update, and trailing all-frozen backward
        # segments are not run at all
        trainable = tuple(bool(p.requires_grad) for p in self.arena.params)
        key = (tuple(x.shape), dt, trainable, mode)
        st = self._statics.get(key)
        if st is not None:
            if mode == "train":
                self._static = st
            return st
        desc = engine.build_desc(self.model, x, dt)
        L = desc.n_levels
        lib = _lib.lib()
        st = {"key": key, "desc": desc, "L": L, "trainable": trainable, "graphs": _Graphs()}
        st["ws_bytes"] = lib.mi3d_unet_workspace_bytes(C.byref(desc))
        if st["ws_bytes"] == 0:
            _lib.check(-1, "mi3d_unet_workspace_bytes")
        dev = self.device
        st["ws"] = torch.empty(st["ws_bytes"], dtype=torch.uint8, device=dev)
        n, c = desc.N, desc.out_channels
        st["x"] = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev)
        st["y"] = torch.empty((n, desc.D * desc.H * desc.W), dtype=torch.int64, device=dev)
        st["logits"] = torch.empty((n, c, desc.D, desc.H, desc.W), dtype=torch.float32, device=dev)
        st["loss"] = torch.empty((), dtype=torch.float32, device=dev)
        st["coef"] = torch.empty(_lib.LOSS_COEF_FLOATS, dtype=torch.float32, device=dev)
        st["loss_ws"] = torch.empty(lib.mi3d_seg_loss_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["metrics"] = torch.empty(4, dtype=torch.float32, device=dev)       # loss, iou, dice, acc
        st["met_ws"] = torch.empty(lib.mi3d_seg_metrics_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["ptab"] = ptr_table([p.data_ptr() for p in self.arena.params])
        st["btab"] = ptr_table([b.data_ptr() for b in self.model.buffers()])
        if mode == "train":
            st["dlogits"] = torch.empty_like(st["logits"])
            st["fused_head"] = (not _lib.get_route("no_head_loss") and
                                lib.mi3d_unet_head_loss_supported(C.byref(desc), C.byref(self.cfg)) == 1)
            st["ndrop"] = lib.mi3d_unet_dropout_count(C.byref(desc))
            st["drop"] = torch.empty(st["ndrop"], dtype=torch.float32, device=dev)
            st["gtab"] = ptr_table([gp if t else None for gp, t in zip(self.arena.grad_ptrs(), trainable)])
            st["opt_ranges"] = self._trainable_ranges(self.arena, trainable)
            nseg, last = 2 * L + 2, -1
            r = (C.c_int * 4)()
            seg_train = []
            for seg in range(nseg):
                _lib.check(lib.mi3d_unet_segment_params(C.byref(desc), seg, r), "mi3d_unet_segment_params")
                idx = list(range(r[0], r[1])) + (list(range(r[2], r[3])) if r[2] >= 0 else [])
                seg_train.append(any(trainable[i] for i in idx))
                if seg_train[-1]:
                    last = seg
            st["nseg_run"] = last + 1
            # exchange schedule: bucket -> the segment after which it is complete AND will still be run; buckets with
            # no trainable parameter are not reduced at all
            sched = {}
            for seg, (lo, hi) in self.comm.buckets.items():
                live = any(t and lo <= o < hi for o, t in zip(self.arena.offsets, trainable))
                if live and last >= 0:
                    sched.setdefault(min(seg, last), []).append(seg)
            if os.environ.get("MI3D_COMM_SERIAL") and sched and last >= 0:
                # every bucket behind the last segment, on the compute stream: no second hardware queue in the step at all (any
                # kernel on another queue costs this step 100-250 us on this runtime, profiles/r04_experiments_second_queue.txt),
                # at the price of an all-reduce that hides under nothing
                sched = {last: [b for sg in sorted(sched) for b in sched[sg]]}
            st["comm_after"] = sched
            if self.teacher is not None:
                st["t_ws"] = torch.empty(st["ws_bytes"], dtype=torch.uint8, device=dev)
                st["t_logits"] = torch.empty_like(st["logits"])
                st["t_ptab"] = ptr_table([p.data_ptr() for p in self.teacher.parameters()])
                st["t_btab"] = ptr_table([b.data_ptr() for b in self.teacher.buffers()])
            if st["fused_head"] and not self.keep_logits:
                # the fused head never writes logits / dlogits: there is no buffer to read stale values from (ts._static["logits"]
                # is None; keep_logits=True keeps a copy)
                st["logits"] = None
                st["dlogits"] = None
            self._static = st
        self._statics[key] = st
        return st

    def _check_mode(self):
        if not self.model.training:
            raise Mi3dError("TrainStep.step() on a model in eval mode: the backward kernels use batch statistics "
                            "(train_unet.py:208 calls model.train() first); use evaluate() for eval-mode passes")

    # ---- the kernel sequence of one micro-step (everything on the current stream except the all-reduces)
    def _enqueue(self, st, variant, comm):
        first, boundary = variant
        desc = st["desc"]
        s = stream_ptr()
        model = self.model
        # Q2: see reference_zero_grad_quirk in __init__
        if self.quirk:
            accumulate, run_backward = 0, boundary
        else:
            accumulate, run_backward = (0 if first else 1), True
        drop = None
        p = float(getattr(model, "dropout_rate", 0.0))
        injected = getattr(model, "_mi3d_injected_drop_scales", None)
        if injected is not None:
            if injected.numel() != st["ndrop"]:
                raise Mi3dError(f"injected dropout scales have {injected.numel()} entries, plan needs {st['ndrop']}")
            st["drop"].copy_(injected.reshape(-1).to(self.device))
            drop = st["drop"]
        elif p > 0.0:
            call("mi3d_dropout_scales", ptr(st["drop"]), st["ndrop"], p, ptr(engine._rng_state(model, self.device)), s)
            drop = st["drop"]
        t_logits = None
        if self.teacher is not None and self.kd_stream is not None:
            ks = self.kd_stream
            ks.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ks):
                call("mi3d_unet_infer", C.byref(desc), ptr(st["x"]), st["t_ptab"], st["t_btab"],
                     ptr(st["t_logits"]), None, ptr(st["t_ws"]), st["ws_bytes"], ks.cuda_stream)
        n, c, v = desc.N, desc.out_channels, desc.D * desc.H * desc.W
        fused = st["fused_head"]
        keep = ptr(st["logits"]) if self.keep_logits else None
        # weight packs the last optimizer tail already refreshed on the aux stream (valid only while nobody has written the
        # parameters through torch since: load_state_dict, a torch optimizer, any in-place op on a Parameter bump its version
        # counter; writes through `.data` or raw pointers do not -- set ts._static["prepacked"] = None after such a write)
        pre = st.get("prepacked")
        desc.prepacked_from = (pre[0] if (pre and pre[2] == getattr(self, "_param_epoch", 0) and pre[1] == self._param_versions())
                               else 0)        # (the epoch: another static state of this object -- another batch shape -- may have stepped since)
        if fused and self.teacher is None:
            # 1x1x1 head + loss + metrics in one pass: the logits are never written (mi3d.h, mi3d_unet_forward_loss)
            call("mi3d_unet_forward_loss", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], ptr(drop), 1, ptr(st["y"]), None,
                 C.byref(self.cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
                 ptr(st["met_ws"]), keep, None, ptr(st["ws"]), st["ws_bytes"], s)
        elif fused:
            # distillation: the student's body now, its head + loss after the join with the teacher's stream (below)
            call("mi3d_unet_forward", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], ptr(drop), 1,
                 None, None, ptr(st["ws"]), st["ws_bytes"], s)
        else:
            call("mi3d_unet_forward", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], ptr(drop), 1,
                 ptr(st["logits"]), None, ptr(st["ws"]), st["ws_bytes"], s)
        if self.teacher is not None:
            if self.kd_stream is not None:
                torch.cuda.current_stream().wait_stream(self.kd_stream)
            else:
                # frozen teacher (distill_unet.py:109-111): the BatchNorm-folded inference forward
                call("mi3d_unet_infer", C.byref(desc), ptr(st["x"]), st["t_ptab"], st["t_btab"],
                     ptr(st["t_logits"]), None, ptr(st["t_ws"]), st["ws_bytes"], s)
            t_logits = st["t_logits"]
        if fused and self.teacher is not None:
            call("mi3d_unet_head_loss_forward", C.byref(desc), st["ptab"], ptr(st["y"]), ptr(t_logits), C.byref(self.cfg),
                 ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]), ptr(st["met_ws"]), keep,
                 ptr(st["ws"]), st["ws_bytes"], s)
        # loss + metrics (SURVEY Q1 loop bound D) of the same logits in one pass (replaces 3 argmaxes + 2(D-1) host syncs)
        if not fused:
            call("mi3d_seg_loss_metrics_forward", ptr(st["logits"]), ptr(st["y"]), ptr(t_logits), n, c, desc.D, v,
                 C.byref(self.cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
                 ptr(st["met_ws"]), s)
        # the four scalar gathers fused into one 4-float all-reduce; it rides on the first gradient exchange
        # point (one fork of the comm stream less) and is in flight under the rest of the backward
        met = st["metrics"]
        met_pending = self.do_comm
        joined = False
        join_fn = self._join_comm
        if run_backward:
            if not fused:
                call("mi3d_seg_loss_backward", ptr(st["logits"]), ptr(st["y"]), ptr(t_logits), n, c, v, C.byref(self.cfg),
                     ptr(st["coef"]), ptr(self.inv_accum), ptr(st["dlogits"]), s)
            nseg = st["nseg_run"]
            do_comm = self.do_comm and boundary
            aux = self.aux_stream.cuda_stream if self.aux_stream is not None else None
            aux_open = False        # aux-stream work of an earlier call that nothing on the compute stream has waited for yet
            # optimizer tail on the aux stream (include/mi3d.h, mi3d_unet_chain_tail_blocks): without a gradient exchange the whole
            # backward is ONE C call; with aux_join = 0 the aux stream ends up ordered after every gradient except those of the
            # leading `tail_k` encoder blocks, which the compute stream produces last
            tail_k = 0
            if (aux is not None and boundary and not do_comm and not self.use_graph and nseg == 2 * st["L"] + 2
                    and _lib.get_route("opt_tail")):
                tail_k = _lib.lib().mi3d_unet_chain_tail_blocks(C.byref(desc))
            # one C call per run of segments between exchange steps: kernels of adjacent segments share launches (a
            # weight-gradient slab sum rides in the next BatchNorm reduction), which a call boundary would cut
            start = 0
            # MI3D_COMM_CUS=n: the segments launched while a gradient exchange is in flight size their persistent grids for
            # 256 - n CUs (the collective kernel holds the others: see mi3d_set_cu_budget)
            budget = int(os.environ.get("MI3D_COMM_CUS", "0")) if do_comm else 0
            in_flight = False
            # exchange marks (include/mi3d.h, mi3d_unet_backward_marks): an eagerly launched step keeps the backward ONE call;
            # the library records an event when a bucket's gradients are complete and the exchange stream waits for THAT
            mid = [(sg, tuple(st["comm_after"][sg])) for sg in range(nseg - 1) if do_comm and st["comm_after"].get(sg)]
            use_marks = bool(mid) and not self.use_graph and aux is None and not budget and len(mid) <= 4 and \
                not os.environ.get("MI3D_NO_MARKS")
            if use_marks:
                while len(self._mark_handles) < len(mid):
                    e = C.c_void_p()
                    call("mi3d_event_create", C.byref(e))
                    self._mark_handles.append(e.value)
                call("mi3d_unet_backward_marks", (C.c_int * len(mid))(*[sg for sg, _ in mid]),
                     ptr_table(self._mark_handles[:len(mid)]), len(mid))
            for seg in range(nseg):
                last = seg == nseg - 1
                exch = st["comm_after"].get(seg) if do_comm else None
                if use_marks and not last:
                    continue
                if last or exch:
                    if budget and in_flight:
                        call("mi3d_set_cu_budget", budget)
                    # the compute stream joins the aux stream at the end of the LAST call only (in front of the optimizer); a call
                    # that is followed by a gradient exchange leaves the join to the exchange stream (_on_comm_stream), so the
                    # deep-level weight gradients keep running under the next segments.  Segmented graphs end a capture at every
                    # exchange: there every call joins
                    join = 1 if ((last and not tail_k) or self.use_graph or aux is None) else 0
                    aux_open = aux_open or (aux is not None and not join and not tail_k)
                    if fused:
                        call("mi3d_unet_backward_loss", C.byref(desc), ptr(st["x"]), st["ptab"], st["gtab"], ptr(drop),
                             ptr(st["y"]), ptr(t_logits), C.byref(self.cfg), ptr(st["coef"]), ptr(self.inv_accum), None, 1.0,
                             accumulate, start, seg + 1, ptr(st["ws"]), st["ws_bytes"], s, aux, self._events, join)
                    else:
                        call("mi3d_unet_backward", C.byref(desc), ptr(st["x"]), st["ptab"], st["gtab"], ptr(drop),
                             ptr(st["dlogits"]), None, 1.0, accumulate, start, seg + 1, ptr(st["ws"]), st["ws_bytes"], s, aux,
                             self._events, join)
                    if budget and in_flight:
                        call("mi3d_set_cu_budget", 0)
                    in_flight = in_flight or bool(exch)
                    start = seg + 1
                    if use_marks:
                        cs = self.comm_stream
                        for i, (sg, b) in enumerate(mid):
                            with_met, met_pending = met_pending, False
                            call("mi3d_stream_wait_event", cs.cuda_stream, self._mark_handles[i])
                            with torch.cuda.stream(cs):
                                if with_met:
                                    self.comm.average_(met)
                                for k in b:
                                    self.comm.reduce_bucket(k)
                    if exch and not last:
                        with_met, met_pending = met_pending, False
                        comm(lambda b=tuple(exch), wm=with_met: self._on_comm_stream(
                            lambda: ([self.comm.average_(met)] if wm else []) + [self.comm.reduce_bucket(k) for k in b]))
                    elif exch:
                        # the exchange behind the LAST segment has nothing left to hide under: it runs on the compute stream
                        # itself, after the join with the exchange stream (collectives of one communicator never overlap each
                        # other) -- one fork / join pair per step instead of two (a cross-queue dependency costs ~12 us)
                        with_met, met_pending = met_pending, False
                        joined = True
                        comm(join_fn)
                        comm(lambda b=tuple(exch), wm=with_met: ([self.comm.average_(met)] if wm else []) +
                             [self.comm.reduce_bucket(k) for k in b])
            if aux_open:       # the last call may have had nothing for the aux stream: join what earlier calls left there
                torch.cuda.current_stream().wait_stream(self.aux_stream)
        if met_pending:
            comm(lambda: self._on_comm_stream(lambda: self.comm.average_(met)))
        if self.do_comm and not joined:
            comm(join_fn)
        if boundary and run_backward and self.aux_stream is not None and tail_k > 0:
            # AdamW over everything but the leading blocks + the re-pack of those weights for the next forward run on the aux
            # stream, behind the deferred weight gradients and beside the compute stream's last (full-resolution encoder) backward
            # kernels; the compute stream joins and updates the leading blocks.  Same arithmetic per parameter, same step count.
            cut = self.arena.offsets[8 * tail_k]
            r_a = [(max(lo, cut), hi) for lo, hi in st["opt_ranges"] if hi > cut]
            r_b = [(lo, min(hi, cut)) for lo, hi in st["opt_ranges"] if lo < cut]
            hyper = self._hyper()
            with torch.cuda.stream(self.aux_stream):
                a = self.aux_stream.cuda_stream
                lr, b1, b2, eps, wd = hyper
                for lo, hi in r_a:
                    call("mi3d_adamw_apply", self.arena.p.data_ptr() + 4 * lo, self.arena.g.data_ptr() + 4 * lo,
                         self.arena.m.data_ptr() + 4 * lo, self.arena.v.data_ptr() + 4 * lo, hi - lo, lr, b1, b2, eps, wd, 1.0,
                         ptr(self.arena.step), 0, a)
                call("mi3d_unet_pack_from", C.byref(desc), st["ptab"], ptr(st["ws"]), st["ws_bytes"], tail_k, a)
            torch.cuda.current_stream().wait_stream(self.aux_stream)
            self._adamw(self.arena, r_b if r_b else [(0, 0)], hyper, s)
            self._param_epoch = getattr(self, "_param_epoch", 0) + 1
            st["prepacked"] = (tail_k, self._param_versions(), self._param_epoch)
        elif boundary:
            self._adamw(self.arena, st["opt_ranges"], self._hyper(), s)
            self._param_epoch = getattr(self, "_param_epoch", 0) + 1
            st["prepacked"] = None

    def step(self, images, labels, last_batch=False):
        """One micro-step on (images (N,Cin,D,H,W) float, labels (N,1,D,H,W) int64).  Returns a device float32[4]
        tensor {loss, iou, dice, acc} (already averaged over ranks when world > 1).
        last_batch=True on the final batch of an epoch: the optimizer steps there even inside an accumulation window and
        the next epoch starts a new window, as accelerate does for the reference's prepared loaders (see _run)."""
        st = self._prepare(images)
        st["x"].copy_(images, non_blocking=True)
        st["y"].copy_(labels.reshape(st["y"].shape), non_blocking=True)
        return self.step_static(last_batch=last_batch)

    def step_static(self, last_batch=False):
        """Run on the data already resident in the static buffers (bench path: inputs in HBM when timing starts)."""
        st = self._static
        if st is None:
            raise Mi3dError("step_static() before any step()/load_batch()")
        self._run(st, last_batch=last_batch)
        return st["metrics"]

    def load_batch(self, images, labels):
        st = self._prepare(images)
        st["x"].copy_(images)
        st["y"].copy_(labels.reshape(st["y"].shape))

    @torch.no_grad()
    def evaluate(self, images, labels):
        """model.eval() forward + loss + metrics (train_unet.py:259-305: the TRAINING loss_fn; distill_unet.py:136-160:
        combined_loss), averaged over ranks like the reference's gather().mean(); returns device float32[4].  Has its own
        static state: a validation batch of another shape (the reference validates with batch 1) does not disturb the
        captured training graph.  Under data parallelism rank 0's BatchNorm buffers are broadcast first: DDP broadcasts them at
        every forward, so the reference validates with rank 0's running statistics on every rank."""
        if self.do_comm:
            self.sync_buffers()
        st = self._prepare(images, mode="eval")
        st["x"].copy_(images)
        st["y"].copy_(labels.reshape(st["y"].shape))
        desc = st["desc"]
        s = stream_ptr()
        n, c, v = desc.N, desc.out_channels, desc.D * desc.H * desc.W
        fused = (not _lib.get_route("no_head_loss") and not self.keep_logits and
                 _lib.lib().mi3d_unet_head_loss_supported(C.byref(desc), C.byref(self.eval_cfg)) == 1)
        if fused:       # head + loss + metrics on the decoder output: the validation logits are never written either
            call("mi3d_unet_infer", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], None, None,
                 ptr(st["ws"]), st["ws_bytes"], s)
            call("mi3d_unet_head_loss_forward", C.byref(desc), st["ptab"], ptr(st["y"]), None, C.byref(self.eval_cfg),
                 ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]), ptr(st["met_ws"]), None,
                 ptr(st["ws"]), st["ws_bytes"], s)
        else:
            call("mi3d_unet_infer", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], ptr(st["logits"]), None,
                 ptr(st["ws"]), st["ws_bytes"], s)
            call("mi3d_seg_loss_metrics_forward", ptr(st["logits"]), ptr(st["y"]), None, n, c, desc.D, v,
                 C.byref(self.eval_cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
                 ptr(st["met_ws"]), s)
        out = st["metrics"].clone()
        self.comm.average_(out)
        return out


class DannStep(_StepBase):
    """Native DANN micro-step (train_dann.py:233-300), single-GPU or data-parallel.

    Kernel sequence: source forward (workspace A, GAP -> feat[0:N]) ; target forward (workspace B, the FULL net as the
    reference runs it, Q4; GAP -> feat[N:2N]) ; loss + metrics on the source logits ; discriminator MLP forward on the
    2N feature rows (rows are independent, so the reference's two calls are one) ; row CE with labels [0]*N + [1]*N ;
    discriminator backward (its parameter gradients carry the lambda of `total = task + lambda*domain`) ; U-Net
    backward of the source graph (dlogits and dgap, gap_scale = -lambda = the gradient reversal, so the encoder sees
    -lambda^2 dL_dom/df, Q3) interleaved segment by segment with the backward of the target graph (dgap only:
    bottleneck + encoder, accumulating into the same gradient arena) ; two arena AdamW updates.
    Under DP the discriminator arena is one more bucket, in flight under the whole U-Net backward."""

    def __init__(self, seg_model, disc_model, loss="ce_tversky", lambda_domain=0.1, lr=1e-3, weight_decay=0.01,
                 betas=(0.9, 0.999), eps=1e-8, grad_accum=1, process_group=None, compute_dtype=None, use_graph=False,
                 force_comm=False, overlap_forwards=True):
        self._init_common(seg_model, lr, weight_decay, betas, eps, grad_accum, process_group, compute_dtype, use_graph,
                          force_comm)
        # The target forward does not depend on the source forward (train_dann.py:268-272): it runs on its own stream beside it.
        # Both are forwards of ONE model in train mode, i.e. both update the same BatchNorm running statistics, source first:
        # the target forward runs with the update DEFERRED (mi3d_unet_forward training = 2 publishes its batch statistics as
        # doubles) and mi3d_unet_bn_apply_deferred applies it after the join -- buffers bit-identical to the serial order.
        self.fwd_stream = concurrent_stream(self.device, role="fwd") if overlap_forwards else None
        self.disc = disc_model
        self.lam = float(lambda_domain)
        self.cfg = _loss_cfg(loss)
        self.eval_cfg = self.cfg
        self.lins = [disc_model.net[0], disc_model.net[3], disc_model.net[6], disc_model.net[8]]
        self.disc_p = [float(disc_model.net[2].p), float(disc_model.net[5].p)]
        for p in disc_model.parameters():
            _lib.require_cuda(p, "DannStep discriminator")
        self.disc_arena = ParamArena(disc_model.parameters(), self.device)
        # train_dann.py:421-422: both optimizers share lr / weight decay
        self.disc_optimizer = ArenaAdamW(disc_model.parameters(), lr, betas, eps, weight_decay)
        if self.world > 1:
            self.broadcast_parameters()
            dist.broadcast(self.disc_arena.p, src=0, group=process_group)

    def _hyper(self):
        g = self.disc_optimizer.param_groups[0]
        return super()._hyper() + (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                   float(g["weight_decay"]))

    def _mutable_state(self):
        d = self.disc_arena
        return super()._mutable_state() + [d.p, d.g, d.m, d.v, d.step, self._disc_rng()]

    def _disc_rng(self):
        st = getattr(self.disc, "_mi3d_rng_state", None)
        if st is None or st.device != self.device:
            st = torch.tensor([(torch.initial_seed() + 0x5DEECE66D) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64,
                              device=self.device)
            self.disc._mi3d_rng_state = st
        return st

    def _check_mode(self):
        if not self.model.training or not self.disc.training:
            raise Mi3dError("DannStep.step() needs both models in train mode (train_dann.py:226-227)")

    def _prepare(self, xs, xt):
        if tuple(xs.shape) != tuple(xt.shape):
            raise Mi3dError(f"source batch {tuple(xs.shape)} and target batch {tuple(xt.shape)} must have one shape "
                            "(train_dann.py:399-400: both loaders use args.batch_size)")
        dt = self.dtype or engine.resolve_compute_dtype(self.model)
        trainable = tuple(bool(p.requires_grad) for p in self.arena.params)
        key = (tuple(xs.shape), dt, trainable)
        st = self._statics.get(key)
        if st is not None:
            self._static = st
            return st
        if not all(trainable):
            raise Mi3dError("DannStep: frozen segmentation parameters are not supported")
        desc = engine.build_desc(self.model, xs, dt)
        lib = _lib.lib()
        L = desc.n_levels
        dev = self.device
        st = {"key": key, "desc": desc, "L": L, "graphs": _Graphs()}
        st["ws_bytes"] = lib.mi3d_unet_workspace_bytes(C.byref(desc))
        if st["ws_bytes"] == 0:
            _lib.check(-1, "mi3d_unet_workspace_bytes")
        n, c = desc.N, desc.out_channels
        F = 2 * desc.features[L - 1]
        if self.lins[0].in_features != F:
            raise Mi3dError(f"discriminator expects {self.lins[0].in_features} features, the bottleneck has {F}")
        for k in ("ws_s", "ws_t"):
            st[k] = torch.empty(st["ws_bytes"], dtype=torch.uint8, device=dev)
        st["xs"] = torch.empty(tuple(xs.shape), dtype=torch.float32, device=dev)
        st["xt"] = torch.empty(tuple(xs.shape), dtype=torch.float32, device=dev)
        st["y"] = torch.empty((n, desc.D * desc.H * desc.W), dtype=torch.int64, device=dev)
        st["logits"] = torch.empty((n, c, desc.D, desc.H, desc.W), dtype=torch.float32, device=dev)
        st["dlogits"] = torch.empty_like(st["logits"])
        st["fused_head"] = (not _lib.get_route("no_head_loss") and
                            lib.mi3d_unet_head_loss_supported(C.byref(desc), C.byref(self.cfg)) == 1)
        st["coef"] = torch.empty(_lib.LOSS_COEF_FLOATS, dtype=torch.float32, device=dev)
        st["loss_ws"] = torch.empty(lib.mi3d_seg_loss_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["met_ws"] = torch.empty(lib.mi3d_seg_metrics_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["metrics"] = torch.empty(5, dtype=torch.float32, device=dev)       # task, iou, dice, acc, domain
        st["ndrop"] = lib.mi3d_unet_dropout_count(C.byref(desc))
        st["drop_s"] = torch.empty(st["ndrop"], dtype=torch.float32, device=dev)
        st["drop_t"] = torch.empty(st["ndrop"], dtype=torch.float32, device=dev)
        st["feat"] = torch.empty((2 * n, F), dtype=torch.float32, device=dev)
        st["dfeat"] = torch.empty((2 * n, F), dtype=torch.float32, device=dev)
        widths = [l.out_features for l in self.lins]
        st["acts"] = [torch.empty((2 * n, w), dtype=torch.float32, device=dev) for w in widths]
        st["gacts"] = [torch.empty((2 * n, w), dtype=torch.float32, device=dev) for w in widths]
        st["lin_ws"] = torch.empty(2 * n * max(widths), dtype=torch.float32, device=dev)
        st["ddrop"] = [torch.empty((2 * n, widths[i]), dtype=torch.float32, device=dev) for i in (0, 1)]
        st["dlabels"] = torch.cat([torch.zeros(n, dtype=torch.int64), torch.ones(n, dtype=torch.int64)]).to(dev)
        st["ptab"] = ptr_table([p.data_ptr() for p in self.arena.params])
        st["gtab"] = ptr_table(self.arena.grad_ptrs())
        bufs = list(self.model.buffers())
        st["btab"] = ptr_table([b.data_ptr() for b in bufs])
        # side buffers of the deferred BatchNorm update: double[2C] per layer, addressed through the running_mean slots
        rms = [b for i, b in enumerate(bufs) if i % 3 == 0]
        st["side"] = torch.zeros(sum(2 * b.numel() for b in rms), dtype=torch.float64, device=dev)
        side_ptrs, off = [], 0
        for i, b in enumerate(bufs):
            if i % 3 == 0:
                side_ptrs.append(st["side"].data_ptr() + 8 * off)
                off += 2 * b.numel()
            else:
                side_ptrs.append(None)
        st["btab_side"] = ptr_table(side_ptrs)
        st["opt_ranges"] = [(0, self.arena.numel)]
        self._statics[key] = st
        self._static = st
        return st

    def _enqueue(self, st, variant, comm):
        first, boundary = variant
        desc, L = st["desc"], st["L"]
        s = stream_ptr()
        seg, disc = self.model, self.disc
        accumulate = 0 if first else 1
        n, c, v = desc.N, desc.out_channels, desc.D * desc.H * desc.W
        lam = self.lam
        # Dropout3d masks: one independent draw per forward (source, then target), like two module calls
        p = float(getattr(seg, "dropout_rate", 0.0))
        injected = getattr(seg, "_mi3d_injected_drop_scales", None)       # tests: (source scales, target scales)
        drop_s = drop_t = None
        if injected is not None:
            st["drop_s"].copy_(injected[0].reshape(-1))
            st["drop_t"].copy_(injected[1].reshape(-1))
            drop_s, drop_t = st["drop_s"], st["drop_t"]
        elif p > 0.0:
            rng = engine._rng_state(seg, self.device)
            call("mi3d_dropout_scales", ptr(st["drop_s"]), st["ndrop"], p, ptr(rng), s)
            call("mi3d_dropout_scales", ptr(st["drop_t"]), st["ndrop"], p, ptr(rng), s)
            drop_s, drop_t = st["drop_s"], st["drop_t"]
        feat = st["feat"]
        F = feat.shape[1]
        fs = self.fwd_stream
        if fs is not None:
            fs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(fs):
                call("mi3d_unet_forward", C.byref(desc), ptr(st["xt"]), st["ptab"], st["btab_side"], ptr(drop_t), 2 | 4,
                     None, feat.data_ptr() + 4 * n * F, ptr(st["ws_t"]), st["ws_bytes"], fs.cuda_stream)
        fused = st["fused_head"]
        beside = 4       # two forwards side by side: thin BatchNorm consumers (include/mi3d.h); also in the serial order, so that
        #                  both orders run the same kernels and stay bitwise equal
        if fused:       # source head + loss + metrics in one pass, logits never written (mi3d_unet_forward_loss)
            call("mi3d_unet_forward_loss", C.byref(desc), ptr(st["xs"]), st["ptab"], st["btab"], ptr(drop_s), 1 | beside, ptr(st["y"]), None,
                 C.byref(self.cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
                 ptr(st["met_ws"]), None, feat.data_ptr(), ptr(st["ws_s"]), st["ws_bytes"], s)
        else:
            call("mi3d_unet_forward", C.byref(desc), ptr(st["xs"]), st["ptab"], st["btab"], ptr(drop_s), 1 | beside,
                 ptr(st["logits"]), feat.data_ptr(), ptr(st["ws_s"]), st["ws_bytes"], s)
        if fs is not None:
            torch.cuda.current_stream().wait_stream(fs)
            call("mi3d_unet_bn_apply_deferred", C.byref(desc), st["btab"], st["btab_side"], s)
        else:
            call("mi3d_unet_forward", C.byref(desc), ptr(st["xt"]), st["ptab"], st["btab"], ptr(drop_t), 1 | beside,
                 None, feat.data_ptr() + 4 * n * F, ptr(st["ws_t"]), st["ws_bytes"], s)      # the target logits are never read
        if not fused:
            call("mi3d_seg_loss_metrics_forward", ptr(st["logits"]), ptr(st["y"]), None, n, c, desc.D, v,
                 C.byref(self.cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
                 ptr(st["met_ws"]), s)
            call("mi3d_seg_loss_backward", ptr(st["logits"]), ptr(st["y"]), None, n, c, v, C.byref(self.cfg),
                 ptr(st["coef"]), ptr(self.inv_accum), ptr(st["dlogits"]), s)
        # ---- discriminator on the 2N feature rows (train_dann.py:34-49,276-283)
        dd = [None, None, None, None]
        dinj = getattr(disc, "_mi3d_injected_drop_scales", None)          # tests: [mask after net.1, mask after net.4]
        for i in (0, 1):
            if dinj is not None:
                st["ddrop"][i].copy_(dinj[i])
                dd[i] = st["ddrop"][i]
            elif self.disc_p[i] > 0.0:
                call("mi3d_dropout_scales", ptr(st["ddrop"][i]), st["ddrop"][i].numel(), self.disc_p[i],
                     ptr(self._disc_rng()), s)
                dd[i] = st["ddrop"][i]
        relus = (1, 1, 1, 0)
        inp = feat
        for i, l in enumerate(self.lins):
            call("mi3d_linear_forward", ptr(inp), ptr(l.weight), ptr(l.bias), ptr(st["acts"][i]), 2 * n, l.in_features,
                 l.out_features, relus[i], ptr(dd[i]), s)
            inp = st["acts"][i]
        # domain_loss = CE/accum (train_dann.py:283); its gradient enters `total` with weight lambda (:285)
        call("mi3d_softmax_ce_rows", ptr(st["acts"][3]), ptr(st["dlabels"]), 2 * n, 2, ptr(st["metrics"][4:]),
             ptr(st["gacts"][3]), lam / self.accum, s)
        da = self.disc_arena
        dgp = da.grad_ptrs()
        for i in (3, 2, 1, 0):
            l = self.lins[i]
            x_in = feat if i == 0 else st["acts"][i - 1]
            gx = st["dfeat"] if i == 0 else st["gacts"][i - 1]
            call("mi3d_linear_backward", ptr(x_in), ptr(l.weight), ptr(st["acts"][i]), ptr(st["gacts"][i]), 2 * n,
                 l.in_features, l.out_features, relus[i], ptr(dd[i]), ptr(gx), dgp[2 * i], dgp[2 * i + 1], accumulate, 1.0,
                 ptr(st["lin_ws"]), s)
        do_comm = self.do_comm and boundary
        if self.do_comm:
            met = st["metrics"]
            comm(lambda: self._on_comm_stream(lambda: self.comm.average_(met)))
        if do_comm:
            dg = da.g
            comm(lambda: self._on_comm_stream(lambda: self.comm.average_(dg)))
        # ---- one backward over both graphs, segment by segment: gradient reversal = gap_scale -lambda
        nseg = 2 * L + 2
        dgs, dgt = st["dfeat"].data_ptr(), st["dfeat"].data_ptr() + 4 * n * F
        start = 0
        for sg in range(nseg):
            last = sg == nseg - 1
            exch = sg in self.comm.buckets and do_comm
            if last or exch:
                if fused:
                    call("mi3d_unet_backward_loss", C.byref(desc), ptr(st["xs"]), st["ptab"], st["gtab"], ptr(drop_s), ptr(st["y"]),
                         None, C.byref(self.cfg), ptr(st["coef"]), ptr(self.inv_accum), dgs, -lam, accumulate, start, sg + 1,
                         ptr(st["ws_s"]), st["ws_bytes"], s, None, None, 1)
                else:
                    call("mi3d_unet_backward", C.byref(desc), ptr(st["xs"]), st["ptab"], st["gtab"], ptr(drop_s),
                         ptr(st["dlogits"]), dgs, -lam, accumulate, start, sg + 1, ptr(st["ws_s"]), st["ws_bytes"], s, None, None, 1)
                t0 = max(start, L + 1)            # the target graph has no decoder part
                if sg + 1 > t0:
                    call("mi3d_unet_backward", C.byref(desc), ptr(st["xt"]), st["ptab"], st["gtab"], ptr(drop_t),
                         None, dgt, -lam, 1, t0, sg + 1, ptr(st["ws_t"]), st["ws_bytes"], s, None, None, 1)
                start = sg + 1
                if exch:
                    comm(lambda k=sg: self._on_comm_stream(lambda: self.comm.reduce_bucket(k)))
        if self.do_comm:
            comm(self._join_comm)
        if boundary:
            h = self._hyper()
            self._adamw(self.arena, st["opt_ranges"], h[:5], s)
            self._adamw(self.disc_arena, [(0, self.disc_arena.numel)], h[5:], s)

    def load_batch(self, source_images, source_labels, target_images):
        st = self._prepare(source_images, target_images)
        st["xs"].copy_(source_images)
        st["xt"].copy_(target_images)
        st["y"].copy_(source_labels.reshape(st["y"].shape))

    def step(self, source_images, source_labels, target_images, last_batch=False):
        """One micro-step.  Returns device float32[5] = {task_loss, iou, dice, acc, domain_loss} (source-domain metrics,
        train_dann.py:291-299; losses un-divided by the accumulation count like the reference's running sums).
        last_batch: train_dann.py:287 also steps on the final batch of an epoch."""
        st = self._prepare(source_images, target_images)
        st["xs"].copy_(source_images, non_blocking=True)
        st["xt"].copy_(target_images, non_blocking=True)
        st["y"].copy_(source_labels.reshape(st["y"].shape), non_blocking=True)
        return self.step_static(last_batch=last_batch)

    def step_static(self, last_batch=False):
        st = self._static
        if st is None:
            raise Mi3dError("step_static() before any step()/load_batch()")
        self._run(st, last_batch=last_batch)       # a partial window on the last batch: the (first, True) variant (captured like the others)
        return st["metrics"]

    @torch.no_grad()
    def evaluate(self, images, labels):
        """train_dann.py:304-327: eval-mode source-domain validation with the training loss_fn."""
        from . import unet_dann  # noqa: F401
        was = self.model.training
        self.model.eval()
        try:
            logits, _ = self.model(images, return_features=False)
        finally:
            self.model.train(was)
        n, c, v, lab = M._prep(logits, labels)
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        coef = torch.empty(_lib.LOSS_COEF_FLOATS, dtype=torch.float32, device=self.device)
        lws = torch.empty(_lib.lib().mi3d_seg_loss_workspace_bytes(c), dtype=torch.uint8, device=self.device)
        mws = torch.empty(_lib.lib().mi3d_seg_metrics_workspace_bytes(c), dtype=torch.uint8, device=self.device)
        call("mi3d_seg_loss_metrics_forward", ptr(logits), ptr(lab), None, n, c, logits.shape[2], v, C.byref(self.eval_cfg),
             ptr(out), ptr(coef), ptr(out[1:]), ptr(lws), ptr(mws), stream_ptr())
        self.comm.average_(out)
        return out
