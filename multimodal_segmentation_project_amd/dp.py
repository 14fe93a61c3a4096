"""Data-parallel plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests).

Replaces what accelerate/DDP does for the reference: C1 parameter/buffer broadcast at construction,
C2 gradient averaging, C3 BatchNorm-buffer authority of rank 0, C4 metric averaging — re-designed for a fully
connected xGMI node instead of translated from DDP's bucket machinery:
  * gradients live in ONE flat fp32 arena; a "bucket" is a contiguous arena range, reduced in place;
  * two buckets keyed by the backward segment that completes them: [encoder.L-1 .. final_conv] (97 % of the bytes) after
    encoder.L-1's backward, in flight while the bandwidth-heavy encoder.L-2..0 backward runs; [encoder.0..L-2] at the end
    (four readiness-ordered buckets are available as fine_buckets=True; fewer collective calls measured faster);
  * BatchNorm statistics stay per-rank (DDP + BatchNorm3d semantics, NOT SyncBN).
"""
import os

import torch
import torch.distributed as dist


class ParamArena:
    """Flat fp32 storage for parameters / gradients / AdamW moments; nn.Parameters become views (state_dict keeps
    working, the optimizer is one kernel, all-reduce needs no copy-in/copy-out)."""

    def __init__(self, params, device):
        self.params = list(params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64          # 256-B aligned slots
        self.numel = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.step = torch.zeros(1, dtype=torch.int64, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g[o:o + p.numel()].view(p.shape)

    def grad_ptrs(self):
        return [self.g.data_ptr() + 4 * o for o in self.offsets]

    def range_of(self, first, last):
        """arena element range covering params[first:last]"""
        end = self.offsets[last] if last < len(self.params) else self.numel
        return self.offsets[first], end


def bucket_ranges(arena, n_levels, fine=False):
    """{backward segment index -> (lo, hi) arena range complete after that segment}.  Parameter table order is the
    reference's parameters() order: encoder.0..L-1, bottleneck, upconvs, decoder, final_conv (include/mi3d.h).

    Default: TWO exchanges.  [encoder.L-1, bottleneck, upconvs, decoder, final_conv] (one contiguous range, 97 % of the
    bytes for the default net) goes out after encoder.L-1's backward segment, while the bandwidth-heavy encoder.L-2..0
    backward (0.45 ms at 96^3) still runs; [encoder.0..L-2] at the end.  Every collective call costs ~25 us of HOST time
    in an eagerly launched step (measured: 5 calls = +0.12 ms/step on the 1-rank RCCL path), so fewer, larger exchanges win
    over the finest readiness order (fine=True: four buckets after decoder / bottleneck / encoder.L-1 / the end)."""
    L = n_levels
    npar = len(arena.params)
    if not fine:
        if L > 1:
            return {L + 2: arena.range_of(8 * (L - 1), npar), 2 * L + 1: arena.range_of(0, 8 * (L - 1))}
        return {2 * L + 1: arena.range_of(0, npar)}
    buckets = {L: arena.range_of(8 * (L + 1), npar),            # upconvs + decoder + final_conv
               L + 1: arena.range_of(8 * L, 8 * L + 8)}         # bottleneck
    if L > 1:
        buckets[L + 2] = arena.range_of(8 * (L - 1), 8 * L)     # encoder.L-1
        buckets[2 * L + 1] = arena.range_of(0, 8 * (L - 1))     # encoder.0 .. L-2
    else:
        buckets[2 * L + 1] = arena.range_of(0, 8)
    return buckets


class DataParallelComm:
    def __init__(self, arena, n_levels, group=None, force=False, fine_buckets=False):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # force: issue the collectives even in a 1-rank group (MI3D_FORCE_COMM: the RCCL path on a 1-GPU box)
        self.enabled = self.world > 1 or (bool(force) and dist.is_available() and dist.is_initialized())
        self.buckets = bucket_ranges(arena, n_levels, fine=fine_buckets)
        self.backend = dist.get_backend(group) if self.enabled else None
        # MI3D_EMULATE_COMM="<workgroups>:<microseconds>": every gradient exchange also launches a stand-in kernel that holds
        # that many CU slots for that long on the communication stream (1-GPU box: a 1-rank all-reduce is a copy and occupies
        # nothing, so the cost of a RESIDENT collective beside the backward kernels cannot be seen otherwise)
        self.emulate = None
        em = os.environ.get("MI3D_EMULATE_COMM", "")
        if em and self.enabled:
            w, _, u = em.partition(":")
            self.emulate = (int(w), int(u or 250))

    def broadcast_parameters(self, buffers=()):
        """C1: rank 0's parameters (one flat broadcast) and buffers win."""
        if self.enabled:
            dist.broadcast(self.arena.p, src=0, group=self.group)
            self.sync_buffers(buffers)

    def sync_buffers(self, buffers):
        """C3 replacement: rank 0's BN running statistics are authoritative; call before eval / checkpoint.  ONE
        coalesced broadcast (like DDP's): all buffers (fp32 statistics and int64 counters) travel as one byte string."""
        if not self.enabled:
            return
        buffers = [b for b in buffers]
        if not buffers:
            return
        flat = torch.cat([b.detach().reshape(-1).view(torch.uint8) for b in buffers])
        dist.broadcast(flat, src=0, group=self.group)
        off = 0
        with torch.no_grad():
            for b in buffers:
                nb = b.numel() * b.element_size()
                b.copy_(flat[off:off + nb].view(b.dtype).view(b.shape))
                off += nb

    def average_(self, t):
        if not self.enabled:
            return
        if self.backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:                                                   # gloo has no AVG
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)

    def reduce_bucket(self, seg):
        """C2: average the gradient range completed by backward segment `seg` (no-op if none)."""
        r = self.buckets.get(seg)
        if r is not None and self.enabled:
            self.average_(self.arena.g[r[0]:r[1]])
            if self.emulate is not None and self.arena.g.is_cuda:
                from ._lib import call, stream_ptr
                if getattr(self, "_emulate_buf", None) is None:          # the stand-in's own scratch, never the gradient arena
                    self._emulate_buf = torch.zeros(1 << 20, dtype=torch.float32, device=self.arena.g.device)
                usec = max(1, int(self.emulate[1] * (r[1] - r[0]) / self.arena.numel))      # duration ~ bytes exchanged
                call("mi3d_debug_occupy_cus", self.emulate[0], usec, self._emulate_buf.data_ptr(), self._emulate_buf.numel(),
                     stream_ptr())
