"""DANN pieces of the reference's train_dann.py on the MI355X HIP path.

  GradientReversal / grad_reverse   train_dann.py:22-32   forward = view, backward = -lambda * g
  DomainDiscriminator               train_dann.py:34-49   Linear(in,256)-ReLU-Drop(.2)-Linear(256,128)-ReLU-Drop(.2)
                                                          -Linear(128,64)-ReLU-Linear(64,2); keys net.{0,3,6,8}.*
  domain_ce                         train_dann.py:283     nn.CrossEntropyLoss() on the concatenated predictions
The MLP (<= 0.2 MFLOP per row) runs as 4 + 4 small HIP launches inside ONE autograd node.
"""
import torch
import torch.nn as nn

from . import _lib
from . import nn as mnn
from ._lib import call, ptr, stream_ptr


class GradientReversal(torch.autograd.Function):
    """Identity forward; the gradient comes back multiplied by -lambda (one HIP launch).  The native DannStep does not
    even launch that: it hands -lambda to the U-Net backward as its `gap_scale` argument."""

    @staticmethod
    def forward(ctx, features, lambda_):
        ctx.neg_scale = -float(lambda_)
        return features.view_as(features)

    @staticmethod
    def backward(ctx, g):
        _lib.require_cuda(g, "GradientReversal.backward")
        g = g.contiguous().float()
        out = torch.empty_like(g)
        call("mi3d_scale", ptr(g), ptr(out), g.numel(), ctx.neg_scale, None, stream_ptr())
        return out, None


def grad_reverse(x, lambda_):
    return GradientReversal.apply(x, lambda_)


class _DiscFn(torch.autograd.Function):
    """y = L8(relu(L6(drop(relu(L3(drop(relu(L0 x))))))))"""

    @staticmethod
    def forward(ctx, x, drops, *wb):
        x = x.contiguous().float()
        m = x.shape[0]
        acts = [x]
        relus = (1, 1, 1, 0)
        for i in range(4):
            w, b = wb[2 * i], wb[2 * i + 1]
            y = torch.empty((m, w.shape[0]), dtype=torch.float32, device=x.device)
            call("mi3d_linear_forward", ptr(acts[-1]), ptr(w), ptr(b), ptr(y), m, w.shape[1], w.shape[0], relus[i],
                 ptr(drops[i]), stream_ptr())
            acts.append(y)
        ctx.drops, ctx.relus = drops, relus
        ctx.save_for_backward(*acts, *wb)
        return acts[-1]

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        acts, wb = saved[:5], saved[5:]
        m = acts[0].shape[0]
        g = gy.contiguous().float()
        grads = [None] * 8
        for i in (3, 2, 1, 0):
            w = wb[2 * i]
            gx = torch.empty_like(acts[i])
            gw, gb = torch.empty_like(w), torch.empty_like(wb[2 * i + 1])
            ws = torch.empty(m * w.shape[0], dtype=torch.float32, device=g.device)
            call("mi3d_linear_backward", ptr(acts[i]), ptr(w), ptr(acts[i + 1]), ptr(g), m, w.shape[1], w.shape[0],
                 ctx.relus[i], ptr(ctx.drops[i]), ptr(gx), ptr(gw), ptr(gb), 0, 1.0, ptr(ws), stream_ptr())
            grads[2 * i], grads[2 * i + 1] = gw, gb
            g = gx
        return (g, None) + tuple(grads)


class DomainDiscriminator(nn.Module):
    def __init__(self, in_features, hidden_dim=128):   # hidden_dim is unused in the reference too
        super().__init__()
        self.net = nn.Sequential(
            mnn.Linear(in_features, 256), mnn.ReLU(), mnn.Dropout(0.2),
            mnn.Linear(256, 128), mnn.ReLU(), mnn.Dropout(0.2),
            mnn.Linear(128, 64), mnn.ReLU(),
            mnn.Linear(64, 2),
        )

    def forward(self, x):
        _lib.require_cuda(x, "DomainDiscriminator.forward")
        lins = [self.net[0], self.net[3], self.net[6], self.net[8]]
        drops = [None, None, None, None]
        injected = getattr(self, "_mi3d_injected_drop_scales", None)
        if self.training and injected is not None:
            for i in (0, 1):
                drops[i] = injected[i].to(device=x.device, dtype=torch.float32).contiguous()
        elif self.training:
            st = getattr(self, "_mi3d_rng_state", None)
            if st is None or st.device != x.device:
                st = torch.tensor([(torch.initial_seed() + 0x5DEECE66D) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64,
                                  device=x.device)
                self._mi3d_rng_state = st
            for i, di in ((0, 2), (1, 5)):
                p = float(self.net[di].p)
                if p > 0.0:
                    d = torch.empty((x.shape[0], lins[i].out_features), dtype=torch.float32, device=x.device)
                    call("mi3d_dropout_scales", ptr(d), d.numel(), p, ptr(st), stream_ptr())
                    drops[i] = d
        wb = []
        for l in lins:
            wb += [l.weight, l.bias]
        return _DiscFn.apply(x, drops, *wb)


class _RowCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lg = logits.contiguous().float()
        m, c = lg.shape
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        d = torch.empty_like(lg)
        call("mi3d_softmax_ce_rows", ptr(lg), ptr(labels.contiguous().long()), m, c, ptr(loss), ptr(d), 1.0, stream_ptr())
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty_like(d)
        call("mi3d_scale", ptr(d), ptr(out), d.numel(), 1.0, ptr(g), stream_ptr())
        return out, None


def domain_ce(domain_preds, domain_labels):
    """nn.CrossEntropyLoss()(preds, labels) for the (2N, 2) domain head [train_dann.py:283]."""
    _lib.require_cuda(domain_preds, "domain_ce")
    return _RowCEFn.apply(domain_preds, domain_labels)
