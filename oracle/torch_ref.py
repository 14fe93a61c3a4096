"""Functional vanilla-torch restatement of the reference's network / step math (TEST INFRASTRUCTURE).

Written independently of the reference's classes: everything is a pure function over a
``state_dict`` with the reference's key names, so it can run in fp32 or fp64 on CPU and be
differentiated by autograd.  Pinned against tests/golden/*.npz (outputs of the reference).

Follows (paths relative to /root/reference):
  models/unet.py:6-22,64-90      conv3-BN-ReLU-Dropout3d x2; pool; upconv + cat(skip, x); final 1x1x1
  models/unet_dann.py:77-79      GAP of the bottleneck
  utils/metrics.py:14-40,137-190 loss family
  train_dann.py:22-49,268-285    GRL, discriminator MLP, DANN total loss
"""
import torch
import torch.nn.functional as F


def _half(sd, pre, ci, bi, x, train, drop_scale, updates):
    x = F.conv3d(x, sd[f"{pre}.{ci}.weight"], sd[f"{pre}.{ci}.bias"], padding=1)
    rm, rv = sd[f"{pre}.{bi}.running_mean"], sd[f"{pre}.{bi}.running_var"]
    if train:
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        x = F.batch_norm(x, rm2, rv2, sd[f"{pre}.{bi}.weight"], sd[f"{pre}.{bi}.bias"], True, 0.1, 1e-5)
        updates[f"{pre}.{bi}.running_mean"] = rm2
        updates[f"{pre}.{bi}.running_var"] = rv2
        updates[f"{pre}.{bi}.num_batches_tracked"] = sd[f"{pre}.{bi}.num_batches_tracked"] + 1
    else:
        x = F.batch_norm(x, rm, rv, sd[f"{pre}.{bi}.weight"], sd[f"{pre}.{bi}.bias"], False, 0.1, 1e-5)
    x = torch.relu(x)
    if drop_scale is not None:
        x = x * drop_scale[:, :, None, None, None]
    return x


def double_conv(sd, pre, x, train=True, drop_scales=(None, None), updates=None):
    updates = {} if updates is None else updates
    x = _half(sd, pre + ".double_conv", 0, 1, x, train, drop_scales[0], updates)
    x = _half(sd, pre + ".double_conv", 4, 5, x, train, drop_scales[1], updates)
    return x


def unet3d_forward(sd, x, train=True, n_levels=None, drop_scales=None, return_features=False):
    """Returns (logits, gap_or_None, buffer_updates).  drop_scales: dict block-name -> (s1, s2) of (N,C)."""
    if n_levels is None:
        n_levels = len({k.split(".")[1] for k in sd if k.startswith("encoder.")})
    updates, skips = {}, []
    ds = drop_scales or {}
    for l in range(n_levels):
        x = double_conv(sd, f"encoder.{l}", x, train, ds.get(f"encoder.{l}", (None, None)), updates)
        skips.append(x)
        x = F.max_pool3d(x, 2, 2)
    x = double_conv(sd, "bottleneck", x, train, ds.get("bottleneck", (None, None)), updates)
    gap = x.mean(dim=(2, 3, 4)) if return_features else None
    for i in range(n_levels):
        x = F.conv_transpose3d(x, sd[f"upconvs.{i}.weight"], sd[f"upconvs.{i}.bias"], stride=2)
        skip = skips[n_levels - 1 - i]
        if x.shape != skip.shape:
            x = F.interpolate(x, size=skip.shape[2:])
        x = torch.cat((skip, x), dim=1)
        x = double_conv(sd, f"decoder.{i}", x, train, ds.get(f"decoder.{i}", (None, None)), updates)
    x = F.conv3d(x, sd["final_conv.weight"], sd["final_conv.bias"])
    return x, gap, updates


LOSS_KINDS = {
    "combined": (1.0, 1, 1.0, 0.0, 0.0, 1e-5),
    "dice": (0.0, 1, 1.0, 0.0, 0.0, 1e-5),
    "tversky": (0.0, 2, 1.0, 0.5, 0.5, 1e-6),
    "ce_tversky": (0.3, 2, 0.7, 0.5, 0.5, 1e-6),
    "ce_tversky_default": (0.3, 2, 0.7, 0.7, 0.3, 1e-6),
}


def seg_loss(logits, labels, kind="combined", teacher=None, kd_alpha=None, temperature=2.0):
    w_ce, rk, w_reg, a, b, eps = LOSS_KINDS[kind]
    w_kd = 0.0
    if teacher is not None:
        w_ce, rk, w_reg, a, b, eps = LOSS_KINDS["ce_tversky_default"]
        w_ce, w_reg, w_kd = w_ce * kd_alpha, w_reg * kd_alpha, 1.0 - kd_alpha
    t = labels.reshape(labels.shape[0], *labels.shape[2:])
    C = logits.shape[1]
    logp = F.log_softmax(logits, dim=1)
    p = logp.exp()
    onehot = F.one_hot(t, C).movedim(-1, 1).to(logits.dtype)
    ce = -(logp * onehot).sum(dim=1).mean()
    dims = [0] + list(range(2, logits.dim()))
    I = (p * onehot).sum(dim=dims)
    P = p.sum(dim=dims)
    T = onehot.sum(dim=dims)
    if rk == 1:
        reg = (1 - (2 * I + eps) / (P + T + eps))[1:].mean()
    else:
        fp = (p * (1 - onehot)).sum(dim=dims)
        fn = ((1 - p) * onehot).sum(dim=dims)
        reg = (1 - (I + eps) / (I + a * fp + b * fn + eps))[1:].mean()
    loss = w_ce * ce + w_reg * reg
    if w_kd:
        ls = F.log_softmax(logits / temperature, dim=1)
        lt = F.log_softmax(teacher / temperature, dim=1)
        loss = loss + w_kd * temperature ** 2 * (lt.exp() * (lt - ls)).mean()
    return loss


def disc_forward(sd, x, drop_scales=(None, None)):
    """DomainDiscriminator MLP (train_dann.py:34-49): keys net.{0,3,6,8}.{weight,bias}."""
    x = torch.relu(F.linear(x, sd["net.0.weight"], sd["net.0.bias"]))
    if drop_scales[0] is not None:
        x = x * drop_scales[0]
    x = torch.relu(F.linear(x, sd["net.3.weight"], sd["net.3.bias"]))
    if drop_scales[1] is not None:
        x = x * drop_scales[1]
    x = torch.relu(F.linear(x, sd["net.6.weight"], sd["net.6.bias"]))
    return F.linear(x, sd["net.8.weight"], sd["net.8.bias"])


class _GRL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.lam * g, None


def dann_total_loss(seg_sd, disc_sd, xs, ys, xt, lam, kind="combined"):
    """train_dann.py:268-285 (fp32 branch, accumulation 1).  Returns (total, task, domain, updates)."""
    so, sf, up1 = unet3d_forward(seg_sd, xs, True, return_features=True)
    task = seg_loss(so, ys, kind)
    sd2 = dict(seg_sd)
    sd2.update(up1)      # the target forward sees the running stats already updated by the source forward
    _, tf, up2 = unet3d_forward(sd2, xt, True, return_features=True)
    sp = disc_forward(disc_sd, _GRL.apply(sf, lam))
    tp = disc_forward(disc_sd, _GRL.apply(tf, lam))
    dl = F.cross_entropy(torch.cat([sp, tp]), torch.cat([torch.zeros(len(sp), dtype=torch.long),
                                                         torch.ones(len(tp), dtype=torch.long)]))
    return task + lam * dl, task, dl, up2, so


def train_loop(sd, batches, accum=1, zero_grad_quirk=False, lr=1e-3, wd=0.01, kind="combined", metrics_fn=None):
    """The step loop of train_unet.py:220-226 under accelerate's accumulate() (gradient_accumulation_steps = accum),
    AdamW defaults of train_unet.py:378.  zero_grad_quirk=True is what the reference does in train_unet.py /
    finetune_ct.py (optimizer.zero_grad() at the top of every micro-step is only honoured by accelerate on the boundary
    micro-step: the earlier micro-batches' gradients are discarded); False accumulates all micro-batches
    (distill_unet.py:114-115).  Returns (state_dict after the loop, per-micro-step losses, last gradients)."""
    sd = {k: v.detach().clone() for k, v in sd.items()}
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    opt = torch.optim.AdamW([sd[k] for k in names], lr=lr, weight_decay=wd)
    losses = []
    for i, (x, y) in enumerate(batches):
        boundary = (i + 1) % accum == 0
        if (zero_grad_quirk and boundary) or (not zero_grad_quirk and i % accum == 0):
            opt.zero_grad()
        logits, _, updates = unet3d_forward(sd, x, train=True)
        loss = seg_loss(logits, y, kind)
        (loss / accum).backward()
        for k, v in updates.items():
            sd[k] = v
        if boundary:
            opt.step()
        losses.append(float(loss))
    grads = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else None) for k in names}
    return {k: v.detach() for k, v in sd.items()}, losses, grads


def plan_drop_scales(drop, n_levels):
    """dict block-name -> (s1, s2) of (N, C)  ->  the flat layout of mi3d_unet_dropout_count (include/mi3d.h):
    blocks [encoder.0..L-1, bottleneck, decoder.0..L-1], first then second Dropout3d of the block, each [N][C]."""
    blocks = [f"encoder.{l}" for l in range(n_levels)] + ["bottleneck"] + [f"decoder.{i}" for i in range(n_levels)]
    return torch.cat([drop[b][h].reshape(-1).float() for b in blocks for h in (0, 1)])
