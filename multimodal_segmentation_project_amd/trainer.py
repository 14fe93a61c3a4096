"""TrainStep — the reference's per-step hot loop (train_unet.py:220-252, distill_unet.py:107-133) as a native,
hipGraph-capturable kernel sequence with data-parallel gradient all-reduce over RCCL.

    zero_grad -> model(images) -> loss_fn -> backward(loss/accum) -> [all-reduce grads / world] -> AdamW.step
    -> calculate_{iou,dice,accuracy} -> gather(loss, iou, dice, acc).mean()

Design (MI355X-first, not a DDP translation):
  * parameters, gradients and AdamW moments live in four flat fp32 arenas (22.6 MB each); nn.Parameters are
    views, so state_dict()/load_state_dict()/torch optimizers keep working, while the optimizer is ONE kernel
    and gradient all-reduce operates on contiguous arena ranges (no bucket copy-in/copy-out).
  * backward runs as 2L+2 C calls (segments); after the decoder, the bottleneck and encoder.L-1 segments the
    finished arena range is all-reduced on a side stream (RCCL over xGMI) while the remaining, bandwidth-heavy
    full-resolution encoder backward runs (82 % of gradient bytes are ready mid-backward).
  * BatchNorm statistics are per-GPU local (DDP + BatchNorm3d semantics); the per-forward buffer broadcast of
    DDP is replaced by `sync_buffers()` before eval/checkpoint; the four scalar gathers (C4) are
    one 4-float all-reduce.
  * no host synchronisation inside step(); results are device tensors.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib, engine
from ._lib import LossCfg, call, ptr, ptr_table, stream_ptr
from . import metrics as M
from .dp import DataParallelComm, ParamArena


def _loss_cfg(kind, kd_alpha=None, temperature=2.0):
    if kd_alpha is not None:      # distillation_loss, utils/metrics.py:169-190
        return M._cfg(0.3 * kd_alpha, 2, 0.7 * kd_alpha, 0.7, 0.3, 1e-6, 1.0 - kd_alpha, temperature)
    table = {
        "combined": (1.0, 1, 1.0, 0.0, 0.0, 1e-5), "dice": (0.0, 1, 1.0, 0.0, 0.0, 1e-5),
        "tversky": (0.0, 2, 1.0, 0.5, 0.5, 1e-6), "ce_tversky": (0.3, 2, 0.7, 0.5, 0.5, 1e-6),
        "ce": (1.0, 0, 0.0, 0.0, 0.0, 1e-6),
    }
    return M._cfg(*table.get(kind, table["combined"]))


class TrainStep:
    def __init__(self, model, loss="combined", lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8,
                 grad_accum=1, kd_teacher=None, kd_alpha=0.7, kd_temperature=2.0, process_group=None,
                 compute_dtype=None, use_graph=False, two_stream=False):
        self.model = model
        self.teacher = kd_teacher
        self.cfg = _loss_cfg(loss, kd_alpha if kd_teacher is not None else None, kd_temperature)
        self._graph = None
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.accum = int(grad_accum)
        self.micro = 0
        self.dtype = compute_dtype
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.device = next(model.parameters()).device
        _lib.require_cuda(next(model.parameters()), "TrainStep")
        self.arena = ParamArena(model.parameters(), self.device)
        n_levels = len(model.encoder)
        self.comm = DataParallelComm(self.arena, n_levels, process_group)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.world > 1 else None
        # second compute stream: weight-gradient kernels run beside the data-gradient chain (mi3d_unet_backward)
        self.aux_stream = torch.cuda.Stream(device=self.device) if two_stream else None
        self._events = None
        if two_stream:
            evs = []
            for _ in range(4):
                e = C.c_void_p()
                call("mi3d_event_create", C.byref(e))
                evs.append(e.value)
            self._events = ptr_table(evs)
        self.use_graph = bool(use_graph) and self.world == 1
        self._graph = None
        self._static = None
        self.inv_accum = torch.full((), 1.0 / self.accum, dtype=torch.float32, device=self.device)
        if self.world > 1:
            self.broadcast_parameters()

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        """Learning rate (ReduceLROnPlateau of train_unet.py:381,442 is host policy: assign the new value here).  The
        value is a kernel argument, so a captured step graph is dropped and re-captured on the next step."""
        self._lr = float(value)
        self._graph = None

    def reset_optimizer(self):
        """What rebuilding ``optim.AdamW`` does in the reference when the encoder is (un)frozen
        (train_unet.py:413-431, finetune_ct.py:374-381): moments and step count start from zero."""
        self.arena.m.zero_()
        self.arena.v.zero_()
        self.arena.step.zero_()
        self._graph = None
        self._static = None

    # ---- DDP construction semantics: rank 0's parameters and buffers win
    def broadcast_parameters(self):
        self.comm.broadcast_parameters(self.model.buffers())

    def sync_buffers(self):
        self.comm.sync_buffers(self.model.buffers())

    # ---- static state for one (shape, dtype)
    def _prepare(self, x):
        dt = self.dtype or engine.resolve_compute_dtype(self.model)
        desc = engine.build_desc(self.model, x, dt)
        # frozen parameters (requires_grad=False: encoder/bottleneck freezing of train_unet.py:31-43, finetune_ct.py:270-304)
        # are part of the static state: they get no gradient, no optimizer update, and trailing all-frozen backward
        # segments are not run at all
        trainable = tuple(bool(p.requires_grad) for p in self.arena.params)
        key = (tuple(x.shape), dt, trainable)
        if self._static is not None and self._static["key"] == key:
            return self._static
        L = desc.n_levels
        lib = _lib.lib()
        st = {"key": key, "desc": desc, "L": L, "trainable": trainable}
        st["ws_bytes"] = lib.mi3d_unet_workspace_bytes(C.byref(desc))
        if st["ws_bytes"] == 0:
            _lib.check(-1, "mi3d_unet_workspace_bytes")
        dev = self.device
        st["ws"] = torch.empty(st["ws_bytes"], dtype=torch.uint8, device=dev)
        n, c = desc.N, desc.out_channels
        st["x"] = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev)
        st["y"] = torch.empty((n, desc.D * desc.H * desc.W), dtype=torch.int64, device=dev)
        st["logits"] = torch.empty((n, c, desc.D, desc.H, desc.W), dtype=torch.float32, device=dev)
        st["dlogits"] = torch.empty_like(st["logits"])
        st["loss"] = torch.empty((), dtype=torch.float32, device=dev)
        st["coef"] = torch.empty(_lib.LOSS_COEF_FLOATS, dtype=torch.float32, device=dev)
        st["loss_ws"] = torch.empty(lib.mi3d_seg_loss_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["metrics"] = torch.empty(4, dtype=torch.float32, device=dev)       # loss, iou, dice, acc
        st["met_ws"] = torch.empty(lib.mi3d_seg_metrics_workspace_bytes(c), dtype=torch.uint8, device=dev)
        st["ndrop"] = lib.mi3d_unet_dropout_count(C.byref(desc))
        st["drop"] = torch.empty(st["ndrop"], dtype=torch.float32, device=dev)
        st["ptab"] = ptr_table([p.data_ptr() for p in self.arena.params])
        st["gtab"] = ptr_table([gp if t else None for gp, t in zip(self.arena.grad_ptrs(), trainable)])
        # contiguous trainable arena ranges for the optimizer, and the number of backward segments that still matter
        a = self.arena
        ranges = []
        for i, t in enumerate(trainable):
            if not t:
                continue
            lo, hi = a.range_of(i, i + 1)
            if ranges and ranges[-1][1] == lo:
                ranges[-1] = (ranges[-1][0], hi)
            else:
                ranges.append((lo, hi))
        st["opt_ranges"] = ranges
        nseg, last = 2 * L + 2, -1
        r = (C.c_int * 4)()
        for seg in range(nseg):
            _lib.check(lib.mi3d_unet_segment_params(C.byref(desc), seg, r), "mi3d_unet_segment_params")
            idx = list(range(r[0], r[1])) + (list(range(r[2], r[3])) if r[2] >= 0 else [])
            if any(trainable[i] for i in idx):
                last = seg
        st["nseg_run"] = last + 1
        st["btab"] = ptr_table([b.data_ptr() for b in self.model.buffers()])
        if self.teacher is not None:
            st["t_ws"] = torch.empty(st["ws_bytes"], dtype=torch.uint8, device=dev)
            st["t_logits"] = torch.empty_like(st["logits"])
            st["t_ptab"] = ptr_table([p.data_ptr() for p in self.teacher.parameters()])
            st["t_btab"] = ptr_table([b.data_ptr() for b in self.teacher.buffers()])
        self._static = st
        self._graph = None
        return st

    # ---- the kernel sequence of one micro-step (everything on the current stream except the all-reduces)
    def _enqueue(self, st):
        desc, L = st["desc"], st["L"]
        s = stream_ptr()
        model = self.model
        boundary = (self.micro + 1) % self.accum == 0
        # Q2: train_unet.py:222 zeroes inside accumulate() -> only the boundary micro-batch's gradient
        # survives; distillation (distill_unet.py:114-115) accumulates properly.  We accumulate properly in both
        # and document the reference quirk instead of reproducing a bug.
        accumulate = 0 if self.micro % self.accum == 0 else 1
        drop = None
        p = float(getattr(model, "dropout_rate", 0.0))
        if model.training and p > 0.0:
            call("mi3d_dropout_scales", ptr(st["drop"]), st["ndrop"], p, ptr(engine._rng_state(model, self.device)), s)
            drop = st["drop"]
        call("mi3d_unet_forward", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], ptr(drop), int(model.training),
             ptr(st["logits"]), None, ptr(st["ws"]), st["ws_bytes"], s)
        t_logits = None
        if self.teacher is not None:
            call("mi3d_unet_forward", C.byref(desc), ptr(st["x"]), st["t_ptab"], st["t_btab"], None, 0,
                 ptr(st["t_logits"]), None, ptr(st["t_ws"]), st["ws_bytes"], s)
            t_logits = st["t_logits"]
        n, c, v = desc.N, desc.out_channels, desc.D * desc.H * desc.W
        # loss + metrics (SURVEY Q1 loop bound D) of the same logits in one pass (replaces 3 argmaxes + 2(D-1) host syncs)
        call("mi3d_seg_loss_metrics_forward", ptr(st["logits"]), ptr(st["y"]), ptr(t_logits), n, c, desc.D, v,
             C.byref(self.cfg), ptr(st["metrics"]), ptr(st["coef"]), ptr(st["metrics"][1:]), ptr(st["loss_ws"]),
             ptr(st["met_ws"]), s)
        call("mi3d_seg_loss_backward", ptr(st["logits"]), ptr(st["y"]), ptr(t_logits), n, c, v, C.byref(self.cfg),
             ptr(st["coef"]), ptr(self.inv_accum), ptr(st["dlogits"]), s)
        nseg = st["nseg_run"]
        do_comm = self.world > 1 and boundary
        aux = self.aux_stream.cuda_stream if self.aux_stream is not None else None
        # one C call per run of segments between exchange steps: kernels of adjacent segments share launches (a
        # weight-gradient slab sum rides in the next BatchNorm reduction), which a call boundary would cut
        start = 0
        for seg in range(nseg):
            last = seg == nseg - 1
            if last or (do_comm and seg in self.comm.buckets):
                call("mi3d_unet_backward", C.byref(desc), ptr(st["x"]), st["ptab"], st["gtab"], ptr(drop),
                     ptr(st["dlogits"]), None, 1.0, accumulate, start, seg + 1, ptr(st["ws"]), st["ws_bytes"], s, aux,
                     self._events)
                start = seg + 1
                if do_comm and seg in self.comm.buckets:
                    self._allreduce_bucket(seg)
        if do_comm:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if boundary:
            a = self.arena
            rng = st["opt_ranges"]
            for k, (lo, hi) in enumerate(rng):
                call("mi3d_adamw_apply", a.p.data_ptr() + 4 * lo, a.g.data_ptr() + 4 * lo, a.m.data_ptr() + 4 * lo,
                     a.v.data_ptr() + 4 * lo, hi - lo, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, 1.0,
                     ptr(a.step), int(k == len(rng) - 1), s)
        self.micro += 1

    def _allreduce_bucket(self, seg):
        """Average one finished gradient range over ranks on the communication stream, ordered after the kernels
        enqueued so far; the compute stream keeps running the remaining backward segments."""
        cs = self.comm_stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            self.comm.reduce_bucket(seg)

    def step(self, images, labels):
        """One micro-step on (images (N,Cin,D,H,W) float, labels (N,1,D,H,W) int64).  Returns a device float32[4]
        tensor {loss, iou, dice, acc} (already averaged over ranks when world > 1)."""
        st = self._prepare(images)
        st["x"].copy_(images, non_blocking=True)
        st["y"].copy_(labels.reshape(st["y"].shape), non_blocking=True)
        return self.step_static()

    def step_static(self):
        """Run on the data already resident in the static buffers (bench path: inputs in HBM when timing starts)."""
        st = self._static
        if st is None:
            raise _lib.Mi3dError("step_static() before any step()/load_batch()")
        if self.use_graph and self.accum == 1:
            if self._graph is None:
                self._capture(st)
            call("mi3d_graph_launch", self._graph, stream_ptr())
            self.micro += 1
        else:
            self._enqueue(st)
        self.comm.average_(st["metrics"])     # the four scalar gathers fused into one 4-float all-reduce
        return st["metrics"]

    def load_batch(self, images, labels):
        st = self._prepare(images)
        st["x"].copy_(images)
        st["y"].copy_(labels.reshape(st["y"].shape))

    def _capture(self, st):
        """Capture one micro-step into a hipGraph.  A warm-up execution is needed first (lazy code-object loading must
        not happen inside the capture); it runs on a snapshot of all mutable state, which is restored afterwards, so
        capturing is invisible to the training trajectory."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            a = self.arena
            snap = [t.clone() for t in (a.p, a.g, a.m, a.v, a.step)]
            bufs = [b.clone() for b in self.model.buffers()]
            rng = engine._rng_state(self.model, self.device).clone()
            micro = self.micro
            self._enqueue(st)
            s.synchronize()
            for t, c in zip((a.p, a.g, a.m, a.v, a.step), snap):
                t.copy_(c)
            for b, c in zip(self.model.buffers(), bufs):
                b.copy_(c)
            engine._rng_state(self.model, self.device).copy_(rng)
            self.micro = micro
            s.synchronize()
            call("mi3d_graph_begin", s.cuda_stream)
            try:
                self._enqueue(st)
            finally:
                g = C.c_void_p()
                call("mi3d_graph_end", s.cuda_stream, C.byref(g))
            self._graph = g
            self.micro = micro             # the captured enqueue did not execute
        torch.cuda.current_stream().wait_stream(s)

    @torch.no_grad()
    def evaluate(self, images, labels):
        """model.eval() forward + loss + metrics (train_unet.py:259-305); returns device float32[4]."""
        st = self._prepare(images)
        st["x"].copy_(images)
        st["y"].copy_(labels.reshape(st["y"].shape))
        desc = st["desc"]
        s = stream_ptr()
        n, c, v = desc.N, desc.out_channels, desc.D * desc.H * desc.W
        call("mi3d_unet_forward", C.byref(desc), ptr(st["x"]), st["ptab"], st["btab"], None, 0, ptr(st["logits"]), None,
             ptr(st["ws"]), st["ws_bytes"], s)
        cfg = _loss_cfg("combined")
        call("mi3d_seg_loss_forward", ptr(st["logits"]), ptr(st["y"]), None, n, c, v, C.byref(cfg), ptr(st["metrics"]),
             ptr(st["coef"]), ptr(st["loss_ws"]), s)
        call("mi3d_seg_metrics", ptr(st["logits"]), ptr(st["y"]), n, c, desc.D, v, ptr(st["metrics"][1:]),
             ptr(st["met_ws"]), s)
        return st["metrics"].clone()
