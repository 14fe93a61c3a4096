"""Host-side engine: binds the reference's nn.Module surface to the whole-network HIP plan (csrc/plan.hip).

One autograd node per network forward: ``_UNetFn.forward`` makes ONE C call that launches every forward kernel,
``_UNetFn.backward`` makes one C call per backward segment (so data-parallel gradient all-reduces can be
interleaved, see dp.py).  PyTorch only owns memory (params, buffers, logits, the workspace) and the stream.

Reference surface mirrored here: models/unet.py:64-90 and models/unet_dann.py:65-98 (forward), the autograd
backward triggered by train_unet.py:225 / train_dann.py:286.
"""
import ctypes as C
import threading

import torch

from . import _lib
from ._lib import UNetDesc, call, ptr, ptr_table, stream_ptr

_state = threading.local()
_default_compute_dtype = None


def set_compute_dtype(dtype):
    """Process-wide internal activation dtype: torch.float32 (exact path), torch.bfloat16 (fast path) or None
    (= follow torch.autocast, fp32 outside it).  Mirrors accelerate's mixed_precision flag (train_unet.py:533)."""
    global _default_compute_dtype
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise _lib.Mi3dError(f"unsupported compute dtype {dtype}")
    _default_compute_dtype = dtype


def resolve_compute_dtype(model):
    dt = getattr(model, "compute_dtype", None)
    if dt is None:
        dt = _default_compute_dtype
    if dt is None:
        if torch.is_autocast_enabled():
            # fp16 autocast (the reference's shipped setting, run_training.sh:28) maps to bf16 here: same MFMA
            # rate on gfx950, fp32 range, no GradScaler needed (loss scaling still passes through: backward is linear)
            dt = torch.bfloat16
        else:
            dt = torch.float32
    return dt


def build_desc(model, x, dtype):
    if x.dim() != 5:
        raise _lib.Mi3dError(f"expected a (N,C,D,H,W) input, got shape {tuple(x.shape)}")
    feats = [blk.double_conv[0].out_channels for blk in model.encoder]
    if len(feats) > _lib.MAX_LEVELS:
        raise _lib.Mi3dError(f"{len(feats)} levels > {_lib.MAX_LEVELS}")
    d = UNetDesc()
    d.in_channels = model.encoder[0].double_conv[0].in_channels
    d.out_channels = model.final_conv.out_channels
    d.n_levels = len(feats)
    for i, f in enumerate(feats):
        d.features[i] = f
    d.N, d.D, d.H, d.W = x.shape[0], x.shape[2], x.shape[3], x.shape[4]
    if x.shape[1] != d.in_channels:
        raise _lib.Mi3dError(f"input has {x.shape[1]} channels, model expects {d.in_channels}")
    d.dtype = _lib.dtype_code(dtype)
    bn = model.encoder[0].double_conv[1]
    if bn.momentum is None:
        raise _lib.Mi3dError("BatchNorm3d(momentum=None) (cumulative moving average) is not supported; the reference "
                             "uses the default momentum 0.1 (models/unet.py:12,16)")
    d.bn_momentum = bn.momentum
    d.bn_eps = bn.eps
    return d


class _Hold:
    """Per-call context handed to the autograd node (not a tensor)."""
    __slots__ = ("desc", "training", "buffers", "want_gap", "ws_bytes", "n_enc_params", "segment_hook")


def _rng_state(model, device):
    st = getattr(model, "_mi3d_rng_state", None)
    if st is None or st.device != device:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        model._mi3d_rng_state = st
    return st


def make_drop_scales(model, desc, p, device, injected=None):
    """Dropout3d channel scales for all 2*(2L+1) dropout layers (models/unet.py:14,18) in ONE kernel launch."""
    n = _lib.lib().mi3d_unet_dropout_count(C.byref(desc))
    if injected is not None:
        if injected.numel() != n:
            raise _lib.Mi3dError(f"injected dropout scales have {injected.numel()} entries, plan needs {n}")
        return injected.to(device=device, dtype=torch.float32).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=device)
    call("mi3d_dropout_scales", ptr(out), n, float(p), ptr(_rng_state(model, device)), stream_ptr())
    return out


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hold, x, drop, *params):
        desc = hold.desc
        dev = x.device
        ws = torch.empty(hold.ws_bytes, dtype=torch.uint8, device=dev)
        logits = torch.empty((desc.N, desc.out_channels, desc.D, desc.H, desc.W), dtype=torch.float32, device=dev)
        gap = None
        if hold.want_gap:
            gap = torch.empty((desc.N, 2 * desc.features[desc.n_levels - 1]), dtype=torch.float32, device=dev)
        ptab = ptr_table([p.data_ptr() for p in params])
        btab = ptr_table([b.data_ptr() for b in hold.buffers])
        call("mi3d_unet_forward", C.byref(desc), ptr(x), ptab, btab, ptr(drop), int(hold.training), ptr(logits),
             ptr(gap), ptr(ws), hold.ws_bytes, stream_ptr())
        ctx.hold, ctx.ws, ctx.drop = hold, ws, drop
        ctx.save_for_backward(x, *params)
        ctx.set_materialize_grads(False)
        if gap is None:
            return logits
        return logits, gap

    @staticmethod
    def backward(ctx, dlogits, dgap=None):
        hold = ctx.hold
        desc = hold.desc
        x, *params = ctx.saved_tensors
        if dlogits is None and dgap is None:
            return (None, None, None) + (None,) * len(params)
        if not hold.training:
            # the backward kernels implement train-mode BatchNorm (batch statistics); an eval-mode forward normalised
            # with running statistics, whose gradient is a different formula
            raise _lib.Mi3dError("backward through a UNet3D forward that ran in eval mode is not supported: call "
                                 "model.train() (train_unet.py:208) or wrap evaluation in torch.no_grad()")
        if dlogits is not None:
            dlogits = dlogits.contiguous().float()
        if dgap is not None:
            dgap = dgap.contiguous().float()
        # without a logits gradient only encoder + bottleneck parameters receive gradient (DANN target pass)
        n_live = len(params) if dlogits is not None else hold.n_enc_params
        grads = [torch.empty_like(p) if (i < n_live and ctx.needs_input_grad[3 + i]) else None
                 for i, p in enumerate(params)]
        ptab = ptr_table([p.data_ptr() for p in params])
        gtab = ptr_table([g.data_ptr() if g is not None else None for g in grads])
        nseg = 2 * desc.n_levels + 2
        hook = hold.segment_hook
        for seg in range(nseg):
            call("mi3d_unet_backward", C.byref(desc), ptr(x), ptab, gtab, ptr(ctx.drop), ptr(dlogits), ptr(dgap),
                 1.0, 0, seg, seg + 1, ptr(ctx.ws), hold.ws_bytes, stream_ptr(), None, None, 1)
            if hook is not None:
                hook(seg, grads)
        ctx.ws = None
        return (None, None, None) + tuple(grads)


def _infer(model, x, desc, hold, params, want_gap):
    dev = x.device
    ws = torch.empty(hold.ws_bytes, dtype=torch.uint8, device=dev)
    logits = torch.empty((desc.N, desc.out_channels, desc.D, desc.H, desc.W), dtype=torch.float32, device=dev)
    gap = None
    if want_gap:
        gap = torch.empty((desc.N, 2 * desc.features[desc.n_levels - 1]), dtype=torch.float32, device=dev)
    ptab = ptr_table([p.data_ptr() for p in params])
    btab = ptr_table([b.data_ptr() for b in hold.buffers])
    call("mi3d_unet_infer", C.byref(desc), ptr(x), ptab, btab, ptr(logits), ptr(gap), ptr(ws), hold.ws_bytes, stream_ptr())
    if model.output_activation is not None:
        logits = model.output_activation(logits)
    return logits, gap


def unet_forward(model, x, want_gap=False):
    """Shared body of unet.UNet3D.forward and unet_dann.UNet3D.forward."""
    _lib.require_cuda(x, "UNet3D.forward")
    params = list(model.parameters())
    for p in params:
        _lib.require_cuda(p, "UNet3D parameter")
    x = x.detach().contiguous().float() if not x.requires_grad else x.contiguous().float()
    dtype = resolve_compute_dtype(model)
    desc = build_desc(model, x, dtype)
    hold = _Hold()
    hold.desc = desc
    hold.training = bool(model.training)
    hold.buffers = list(model.buffers())
    hold.want_gap = bool(want_gap)
    hold.ws_bytes = _lib.lib().mi3d_unet_workspace_bytes(C.byref(desc))
    if hold.ws_bytes == 0:
        _lib.check(-1, "mi3d_unet_workspace_bytes")
    hold.n_enc_params = 8 * (desc.n_levels + 1)
    hold.segment_hook = getattr(model, "_mi3d_segment_hook", None)
    expect = _lib.lib().mi3d_unet_num_params(C.byref(desc))
    if len(params) != expect or len(hold.buffers) != _lib.lib().mi3d_unet_num_buffers(C.byref(desc)):
        raise _lib.Mi3dError(f"module has {len(params)} parameters / {len(hold.buffers)} buffers, plan expects {expect}")
    if not model.training and not (torch.is_grad_enabled() and any(q.requires_grad for q in params)):
        # inference (train_unet.py:259-305 evaluate, test_model.py:242-251, the distillation teacher): BatchNorm folded
        # into the convs, no saved activations, no autograd node
        return _infer(model, x, desc, hold, params, want_gap)
    drop = None
    p = float(getattr(model, "dropout_rate", 0.0))
    injected = getattr(model, "_mi3d_injected_drop_scales", None)
    if model.training and (p > 0.0 or injected is not None):
        drop = make_drop_scales(model, desc, p, x.device, injected)
    out = _UNetFn.apply(hold, x, drop, *params)
    if want_gap:
        logits, gap = out
    else:
        logits, gap = out, None
    if model.output_activation is not None:
        logits = model.output_activation(logits)
    return logits, gap
