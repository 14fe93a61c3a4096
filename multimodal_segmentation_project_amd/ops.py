"""Stand-alone DoubleConv on the per-operator C-ABI entry points (mi3d_conv3_*, mi3d_bn_relu_drop_*, layout
converters).  The whole-network path (engine.py) does not go through here; this serves callers that use a
DoubleConv by itself (reference models/unet.py:6-22) and the per-operator parity tests.
"""
import torch

from . import _lib, engine
from ._lib import call, ptr, stream_ptr


def _tdtype(code):
    return torch.float32 if code == _lib.DTYPE_F32 else torch.bfloat16


class _DoubleConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, x, drops, *params):
        dt, training, bufs, momentum, eps = cfg
        T = _tdtype(dt)
        n, cin, d, h, w = x.shape
        v, m = d * h * w, n * d * h * w
        dev = x.device
        s = stream_ptr()
        xcl = torch.empty((n, d, h, w, cin), dtype=T, device=dev)
        call("mi3d_ncdhw_to_ndhwc", dt, ptr(x), ptr(xcl), cin, cin, n, v, s)
        saved, inp, ci = [xcl], xcl, cin
        for half in range(2):
            wgt, bias, gamma, beta = params[4 * half:4 * half + 4]
            co = wgt.shape[0]
            wsb = _lib.lib().mi3d_conv3_workspace_bytes(ci, co, n, d, h, w)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            y = torch.empty((n, d, h, w, co), dtype=T, device=dev)
            call("mi3d_conv3_forward", dt, dt, ptr(inp), ci, ci, ptr(wgt), ptr(bias), ptr(y), co, co, n, d, h, w,
                 ptr(ws), wsb, s)
            z = torch.empty_like(y)
            stat = torch.empty(4 * co, dtype=torch.float32, device=dev)
            bws = torch.empty(_lib.lib().mi3d_bn_workspace_bytes(co), dtype=torch.uint8, device=dev)
            rm, rv, nbt = bufs[3 * half:3 * half + 3]
            call("mi3d_bn_relu_drop_forward", dt, ptr(y), co, co, m, v, ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                 ptr(nbt), momentum, eps, int(training), ptr(drops[half]), ptr(z), co, ptr(stat), ptr(bws), s)
            saved += [y, z, stat]
            inp, ci = z, co
        out = torch.empty((n, ci, d, h, w), dtype=torch.float32, device=dev)
        call("mi3d_ndhwc_to_ncdhw", dt, ptr(inp), ci, ptr(out), ci, n, v, s)
        ctx.cfg, ctx.drops, ctx.geo = cfg, drops, (n, cin, d, h, w)
        ctx.save_for_backward(*saved, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        dt = ctx.cfg[0]
        T = _tdtype(dt)
        n, cin, d, h, w = ctx.geo
        v, m = d * h * w, n * d * h * w
        xcl, y0, z0, st0, y1, z1, st1, *params = ctx.saved_tensors
        dev = gout.device
        s = stream_ptr()
        co = y1.shape[-1]
        g = torch.empty((n, d, h, w, co), dtype=T, device=dev)
        call("mi3d_ncdhw_to_ndhwc", dt, ptr(gout.contiguous().float()), ptr(g), co, co, n, v, s)
        grads = [None] * 8
        for half, (inp, y, st) in ((1, (z0, y1, st1)), (0, (xcl, y0, st0))):
            wgt = params[4 * half]
            ci = inp.shape[-1]
            bws = torch.empty(_lib.lib().mi3d_bn_workspace_bytes(co), dtype=torch.uint8, device=dev)
            dy = torch.empty_like(y)
            dgam, dbet = torch.empty(co, device=dev), torch.empty(co, device=dev)
            call("mi3d_bn_relu_drop_backward", dt, ptr(g), co, ptr(y), co, co, m, v, ptr(st), ptr(ctx.drops[half]),
                 ptr(dy), co, ptr(dgam), ptr(dbet), 0, ptr(bws), s)
            wsb = _lib.lib().mi3d_conv3_workspace_bytes(ci, co, n, d, h, w)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            dx = torch.empty((n, d, h, w, ci), dtype=T, device=dev)
            dW, db = torch.empty_like(wgt), torch.empty(co, device=dev)
            call("mi3d_conv3_backward", dt, dt, ptr(inp), ci, ci, ptr(wgt), ptr(dy), co, co, ptr(dx), ci, ptr(dW),
                 ptr(db), 0, n, d, h, w, ptr(ws), wsb, s)
            grads[4 * half:4 * half + 4] = [dW, db, dgam, dbet]
            g = dx
        gx = torch.empty((n, cin, d, h, w), dtype=torch.float32, device=dev)
        call("mi3d_ndhwc_to_ncdhw", dt, ptr(g), cin, ptr(gx), cin, n, v, s)
        return (None, gx, None) + tuple(grads)


def double_conv_forward(mod, x):
    _lib.require_cuda(x, "DoubleConv.forward")
    seq = mod.double_conv
    dt = _lib.dtype_code(engine.resolve_compute_dtype(mod))
    params = [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[4].weight, seq[4].bias, seq[5].weight,
              seq[5].bias]
    bufs = [seq[1].running_mean, seq[1].running_var, seq[1].num_batches_tracked, seq[5].running_mean,
            seq[5].running_var, seq[5].num_batches_tracked]
    drops = [None, None]
    if mod.training:
        for i, di in ((0, 3), (1, 7)):
            p = float(seq[di].p)
            if p > 0.0:
                dsc = torch.empty((x.shape[0], seq[4].out_channels), dtype=torch.float32, device=x.device)
                call("mi3d_dropout_scales", ptr(dsc), dsc.numel(), p, ptr(engine._rng_state(mod, x.device)), stream_ptr())
                drops[i] = dsc
    mom = 0.1 if seq[1].momentum is None else seq[1].momentum
    cfg = (dt, bool(mod.training), bufs, float(mom), float(seq[1].eps))
    return _DoubleConvFn.apply(cfg, x.contiguous().float(), drops, *params)
