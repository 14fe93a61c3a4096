#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING the reference.

Runs only in the build container (needs /root/reference, CPU torch).  Nothing of
the reference is copied: the script imports its modules, feeds them seeded
synthetic tensors and stores inputs + outputs as small .npz files.  The GPU box
never runs this script; tests there read the committed .npz files.

Reference entry points exercised (file:line relative to /root/reference):
  models/unet.py:6-22 (DoubleConv), :24-90 (UNet3D)
  models/unet_dann.py:65-98 (forward with return_features)
  utils/metrics.py:14-40,65-129,137-190 (losses and metrics)
  train_unet.py:178-205 (get_loss_fn variants)
  train_dann.py:22-49 (GradientReversal, DomainDiscriminator), :268-285 (DANN step math)

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _import_reference():
    sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    # utils/dataloader.py needs nibabel + monai for FILE I/O only; they are not
    # installed (ordinary ModuleNotFoundError).  Empty stand-in modules let
    # train_unet.py / train_dann.py import so their pure-torch symbols are usable.
    for name in ("nibabel", "monai", "monai.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    class _AnyTransform:
        def __init__(self, *a, **k):
            pass

    mt = sys.modules["monai.transforms"]
    def _any(name):                               # PEP 562: any transform name resolves to a dummy
        if name.startswith("__"):                 # ... but module dunders (__file__, __path__: inspect.getmodule) do not exist
            raise AttributeError(name)
        return _AnyTransform
    mt.__getattr__ = _any
    import models.unet as ref_unet
    import models.unet_dann as ref_unet_dann
    import utils.metrics as ref_metrics
    import train_unet as ref_train_unet
    import train_dann as ref_train_dann
    return ref_unet, ref_unet_dann, ref_metrics, ref_train_unet, ref_train_dann


def npy(t):
    return t.detach().cpu().numpy()


def synth(n, s, seed, blocky=False, dims=None):
    """SURVEY §8(d) synthetic inputs: seeded randn image, uniform or blocky labels."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = dims if dims else (s, s, s)
    x = torch.randn(n, 1, d, h, w, generator=g)
    y = torch.randint(0, 4, (n, 1, d, h, w), generator=g)
    if blocky:
        zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(h), torch.arange(w), indexing="ij")
        lab = ((zz // max(d // 4, 1)) + (yy // max(h // 4, 1)) + (xx // max(w // 4, 1))) % 4
        y = lab[None, None].expand(n, 1, d, h, w).contiguous().long()
        x = y.float() / 3.0 + 0.1 * x
    return x, y


def param_digest(sd):
    """Per-tensor (sum, abs-sum) — lets a test prove its seeded init equals the reference's."""
    keys = sorted(sd.keys())
    dig = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    return keys, dig


def gen_small_unet(ref_unet, ref_metrics, out):
    """Complete fixture (all weights, all grads) for a small non-cubic, odd-channel net."""
    torch.manual_seed(7)
    m = ref_unet.UNet3D(in_channels=2, out_channels=3, features=[4, 8], dropout_rate=0.0)
    # make BN affine/bias non-trivial so parity exercises them
    g = torch.Generator().manual_seed(71)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 2, 8, 12, 4, generator=g)
    y = torch.randint(0, 3, (2, 1, 8, 12, 4), generator=g)
    m.train()
    logits = m(x)
    loss = ref_metrics.combined_loss(logits, y)
    loss.backward()
    d = {"x": npy(x), "y": npy(y), "logits": npy(logits), "loss": npy(loss)}
    for k, v in sd0.items():
        d["sd0/" + k] = npy(v)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["sd1/" + k] = npy(v)
    for k, p in m.named_parameters():
        d["grad/" + k] = npy(p.grad)
    m.eval()
    with torch.no_grad():
        d["logits_eval"] = npy(m(x))
    np.savez_compressed(os.path.join(out, "small_unet.npz"), **d)


def gen_doubleconv(ref_unet, out):
    torch.manual_seed(11)
    m = ref_unet.DoubleConv(3, 5, dropout_rate=0.0)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn(p.shape, generator=g))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 6, 5, 7, generator=g, requires_grad=True)
    go = torch.randn(2, 5, 6, 5, 7, generator=g)
    m.train()
    o = m(x)
    o.backward(go)
    d = {"x": npy(x), "go": npy(go), "out": npy(o), "gx": npy(x.grad)}
    for k, v in sd0.items():
        d["sd0/" + k] = npy(v)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["sd1/" + k] = npy(v)
    for k, p in m.named_parameters():
        d["grad/" + k] = npy(p.grad)
    np.savez_compressed(os.path.join(out, "doubleconv.npz"), **d)


def gen_losses(ref_metrics, ref_train_unet, out):
    g = torch.Generator().manual_seed(21)
    cases = {}
    logits = 2.0 * torch.randn(2, 4, 8, 8, 8, generator=g)
    labels = torch.randint(0, 4, (2, 1, 8, 8, 8), generator=g)
    cases["uniform"] = (logits, labels)
    lab2 = labels.clone()
    lab2[lab2 == 2] = 0  # class 2 absent
    cases["absent2"] = (1.5 * torch.randn(2, 4, 8, 8, 8, generator=g), lab2)
    cases["single0"] = (torch.randn(1, 4, 4, 6, 8, generator=g), torch.zeros(1, 1, 4, 6, 8, dtype=torch.long))
    onehot_lab = torch.randint(0, 4, (1, 1, 4, 4, 4), generator=g)
    onehot = torch.full((1, 4, 4, 4, 4), -20.0)
    onehot.scatter_(1, onehot_lab, 20.0)
    cases["onehot"] = (onehot, onehot_lab)
    cases["c3_noncubic"] = (torch.randn(2, 3, 3, 5, 7, generator=g), torch.randint(0, 3, (2, 1, 3, 5, 7), generator=g))

    fns = {
        "combined": ref_metrics.combined_loss,
        "tversky55": lambda p, t: ref_metrics.tversky_loss(p, t, alpha=0.5, beta=0.5),
        "ce_tversky73": ref_metrics.combined_ce_tversky_loss,
        "ce_tversky55": ref_train_unet.get_loss_fn("ce_tversky"),
        "dice": ref_train_unet.get_loss_fn("dice"),
        "tversky_fn": ref_train_unet.get_loss_fn("tversky"),
        "default_fn": ref_train_unet.get_loss_fn("whatever"),
    }
    d = {}
    for cname, (lg, lb) in cases.items():
        d[f"{cname}/logits"] = npy(lg)
        d[f"{cname}/labels"] = npy(lb)
        for fname, fn in fns.items():
            z = lg.clone().requires_grad_(True)
            l = fn(z, lb)
            l.backward()
            d[f"{cname}/{fname}/loss"] = npy(l)
            d[f"{cname}/{fname}/grad"] = npy(z.grad)
        # fp64 version of the default loss (tight reference for the closed-form gradient)
        z = lg.double().clone().requires_grad_(True)
        l = ref_metrics.combined_loss(z, lb)
        l.backward()
        d[f"{cname}/combined64/loss"] = npy(l)
        d[f"{cname}/combined64/grad"] = npy(z.grad)
        # distillation
        t_logits = lg + 0.7 * torch.randn(lg.shape, generator=g)
        d[f"{cname}/teacher"] = npy(t_logits)
        for alpha, temp in ((0.7, 2.0), (0.3, 4.0)):
            z = lg.clone().requires_grad_(True)
            l = ref_metrics.distillation_loss(z, t_logits, lb, alpha, temp)
            l.backward()
            d[f"{cname}/distill_a{alpha}_t{temp}/loss"] = npy(l)
            d[f"{cname}/distill_a{alpha}_t{temp}/grad"] = npy(z.grad)
        # metrics (Q1: loop bound is the first spatial dim)
        for mname in ("calculate_iou", "calculate_dice", "calculate_accuracy"):
            v = getattr(ref_metrics, mname)(lg, lb)
            d[f"{cname}/{mname}"] = np.asarray(float(v), dtype=np.float64)
    # Q1 edge: D=2 < C=4 → only class 1 is scored
    lg = torch.randn(2, 4, 2, 8, 8, generator=g)
    lb = torch.randint(0, 4, (2, 1, 2, 8, 8), generator=g)
    d["q1_d2/logits"], d["q1_d2/labels"] = npy(lg), npy(lb)
    for mname in ("calculate_iou", "calculate_dice", "calculate_accuracy"):
        d[f"q1_d2/{mname}"] = np.asarray(float(getattr(ref_metrics, mname)(lg, lb)), dtype=np.float64)
    # no foreground class present at all → python 0 / 0.0
    lb0 = torch.zeros(1, 1, 4, 4, 4, dtype=torch.long)
    lg0 = torch.randn(1, 4, 4, 4, 4, generator=g)
    d["nofg/logits"], d["nofg/labels"] = npy(lg0), npy(lb0)
    for mname in ("calculate_iou", "calculate_dice", "calculate_accuracy"):
        d[f"nofg/{mname}"] = np.asarray(float(getattr(ref_metrics, mname)(lg0, lb0)), dtype=np.float64)
    np.savez_compressed(os.path.join(out, "losses_metrics.npz"), **d)


SLICE_KEYS = [
    "encoder.0.double_conv.0.weight", "encoder.0.double_conv.4.weight", "encoder.1.double_conv.0.weight",
    "bottleneck.double_conv.5.bias", "upconvs.0.bias", "upconvs.3.weight", "decoder.3.double_conv.0.weight",
    "decoder.3.double_conv.5.weight", "decoder.3.double_conv.5.bias", "final_conv.weight", "final_conv.bias",
    "encoder.0.double_conv.1.weight", "encoder.2.double_conv.5.bias", "decoder.0.double_conv.1.weight",
]


def gen_default_unet(ref_unet, ref_metrics, out):
    """Default architecture (5.65 M params): seeded init (digest stored), logits, loss, grads digest."""
    d = {}
    for tag, (n, s) in {"s16n2": (2, 16), "s32n1": (1, 32)}.items():
        torch.manual_seed(0)
        m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
        keys, dig = param_digest(m.state_dict())
        d["param_keys"] = np.array(keys)
        d["param_digest"] = dig
        d["param_shapes"] = np.array([str(tuple(m.state_dict()[k].shape)) for k in keys])
        d["param_dtypes"] = np.array([str(m.state_dict()[k].dtype) for k in keys])
        x, y = synth(n, s, 1234)
        m.train()
        logits = m(x)
        loss = ref_metrics.combined_loss(logits, y)
        loss.backward()
        d[f"{tag}/logits"] = npy(logits)
        d[f"{tag}/loss"] = npy(loss)
        d[f"{tag}/dice"] = npy(ref_metrics.calculate_dice(logits, y))
        d[f"{tag}/iou"] = npy(ref_metrics.calculate_iou(logits, y))
        d[f"{tag}/acc"] = npy(ref_metrics.calculate_accuracy(logits, y))
        names = [k for k, _ in m.named_parameters()]
        d["grad_names"] = np.array(names)
        d[f"{tag}/grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
        for k, p in m.named_parameters():
            if k in SLICE_KEYS:
                d[f"{tag}/grad/{k}"] = npy(p.grad)
        bn = {k: v for k, v in m.state_dict().items() if "running" in k}
        d[f"{tag}/bn_keys"] = np.array(sorted(bn.keys()))
        d[f"{tag}/bn_after1"] = np.concatenate([npy(bn[k]).ravel() for k in sorted(bn.keys())])
        # the reference's own mixed-precision path (accelerate mixed_precision='bf16' == torch.autocast, train_unet.py:533)
        # on the same weights/inputs: its deviation from fp32 is the yardstick for the bf16 HIP path's tolerance
        fp32_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        torch.manual_seed(0)
        m2 = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
        m2.train()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lg2 = m2(x)
        lg2 = lg2.float()
        loss2 = ref_metrics.combined_loss(lg2, y)
        loss2.backward()
        rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        d[f"{tag}/autocast_bf16/logits_relerr"] = np.asarray(rel(lg2.detach(), logits.detach()))
        d[f"{tag}/autocast_bf16/loss"] = npy(loss2)
        d[f"{tag}/autocast_bf16/dice"] = npy(ref_metrics.calculate_dice(lg2, y))
        d[f"{tag}/autocast_bf16/grad_relerr"] = np.array([rel(p.grad, fp32_grads[k]) for k, p in m2.named_parameters()])
    # 5-step AdamW trajectory on a learnable (blocky) 16^3 N=2 problem; train_unet.py:378 defaults
    torch.manual_seed(0)
    m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    x, y = synth(2, 16, 1234, blocky=True)
    traj, dices = [], []
    m.train()
    for _ in range(5):
        opt.zero_grad()
        o = m(x)
        l = ref_metrics.combined_loss(o, y)
        l.backward()
        opt.step()
        traj.append(float(l))
        dices.append(float(ref_metrics.calculate_dice(o, y)))
    d["traj/loss"] = np.array(traj)
    d["traj/dice"] = np.array(dices)
    _, dig = param_digest(m.state_dict())
    d["traj/param_digest_after5"] = dig
    np.savez_compressed(os.path.join(out, "default_unet.npz"), **d)


def gen_dann(ref_unet_dann, ref_metrics, ref_train_dann, out):
    d = {}
    # GRL
    g = torch.Generator().manual_seed(31)
    f = torch.randn(3, 5, generator=g, requires_grad=True)
    r = ref_train_dann.grad_reverse(f, 0.2)
    go = torch.randn(3, 5, generator=g)
    r.backward(go)
    d["grl/x"], d["grl/out"], d["grl/go"], d["grl/gx"] = npy(f), npy(r), npy(go), npy(f.grad)
    # discriminator (eval → dropout off) forward/backward
    torch.manual_seed(3)
    disc = ref_train_dann.DomainDiscriminator(256)
    dkeys, ddig = param_digest(disc.state_dict())
    d["disc/param_keys"], d["disc/param_digest"] = np.array(dkeys), ddig
    disc.eval()
    feats = torch.randn(4, 256, generator=g, requires_grad=True)
    lab = torch.tensor([0, 0, 1, 1])
    pred = disc(feats)
    l = torch.nn.CrossEntropyLoss()(pred, lab)
    l.backward()
    d["disc/feats"], d["disc/pred"], d["disc/loss"], d["disc/gfeats"] = npy(feats), npy(pred), npy(l), npy(feats.grad)
    for k, p in disc.named_parameters():
        d["disc/grad/" + k] = npy(p.grad)
    # one full DANN step (train_dann.py:268-285), dropout disabled in both nets for determinism
    torch.manual_seed(0)
    seg = ref_unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    torch.manual_seed(3)
    disc = ref_train_dann.DomainDiscriminator(256)
    for mod in disc.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    seg.train(); disc.train()
    xs, ys = synth(2, 16, 1234)
    xt, _ = synth(2, 16, 4321)
    xs = xs.clamp(0, 1)                      # "CT-like"
    xt = (xt - xt.min()) / (xt.max() - xt.min())  # "MRI-like"
    lam = 0.2
    so, sf = seg(xs, return_features=True)
    task = ref_metrics.combined_loss(so, ys)
    _, tf = seg(xt, return_features=True)
    sp = disc(ref_train_dann.grad_reverse(sf, lam))
    tp = disc(ref_train_dann.grad_reverse(tf, lam))
    dl = torch.nn.CrossEntropyLoss()(torch.cat([sp, tp]), torch.tensor([0, 0, 1, 1]))
    total = task + lam * dl
    total.backward()
    d["step/xs"], d["step/ys"], d["step/xt"] = npy(xs), npy(ys), npy(xt)
    d["step/task"], d["step/domain"], d["step/total"] = npy(task), npy(dl), npy(total)
    d["step/sfeat"], d["step/tfeat"] = npy(sf), npy(tf)
    d["step/logits"] = npy(so)
    d["step/seg_grad_names"] = np.array([k for k, _ in seg.named_parameters()])
    d["step/seg_grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in seg.named_parameters()])
    for k, p in seg.named_parameters():
        if k in SLICE_KEYS:
            d["step/seg_grad/" + k] = npy(p.grad)
    for k, p in disc.named_parameters():
        d["step/disc_grad/" + k] = npy(p.grad)
    bn = {k: v for k, v in seg.state_dict().items() if "running" in k}
    d["step/bn_after"] = np.concatenate([npy(bn[k]).ravel() for k in sorted(bn.keys())])
    # unet_dann surface: return_features=False → (logits, None)
    seg.eval()
    with torch.no_grad():
        o, none = seg(xs)
    assert none is None
    d["eval/logits"] = npy(o)
    np.savez_compressed(os.path.join(out, "dann.npz"), **d)


def gen_distill(ref_unet, ref_metrics, out):
    """distill_unet.py:107-115 step: student(train) + teacher(eval,no_grad) + distillation_loss."""
    d = {}
    torch.manual_seed(0)
    student = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    torch.manual_seed(1)
    teacher = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    # give the teacher non-trivial running stats so eval-mode BN is exercised
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, b in teacher.named_buffers():
            if k.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if k.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    d["teacher_bn_keys"] = np.array(sorted(k for k, _ in teacher.named_buffers() if "running" in k))
    d["teacher_bn"] = np.concatenate([npy(dict(teacher.named_buffers())[k]).ravel() for k in d["teacher_bn_keys"]])
    student.train(); teacher.eval()
    x, y = synth(2, 16, 1234)
    s = student(x)
    with torch.no_grad():
        t = teacher(x)
    l = ref_metrics.distillation_loss(s, t, y, 0.7, 2.0)
    l.backward()
    d["teacher_logits"], d["student_logits"], d["loss"] = npy(t), npy(s), npy(l)
    d["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in student.named_parameters()])
    np.savez_compressed(os.path.join(out, "distill.npz"), **d)


# ---------------------------------------------------------------------------------------------------------------
# Round-2 fixtures: Dropout3d with real masks, the BASELINE configs at their real sizes (summaries only: losses,
# metrics, gradient norms + leading slices, BN buffers, logit slices; each file a few hundred KB), the reference's
# own step loops executed as they are (train_unet.train_one_epoch under accelerate's accumulate(): Q2;
# train_dann.train_one_epoch_dann: two optimizers), and its evaluate().
# ---------------------------------------------------------------------------------------------------------------
# leading-index gradient slices incl. the level-4 tensors (bottleneck, decoder.0, upconvs.0) the round-1 SLICE_KEYS missed
GRAD_SLICES = {
    "encoder.0.double_conv.0.weight": None, "encoder.0.double_conv.4.weight": None, "encoder.1.double_conv.0.weight": 8,
    "encoder.3.double_conv.4.weight": 1, "bottleneck.double_conv.0.weight": 2, "bottleneck.double_conv.4.weight": 1,
    "bottleneck.double_conv.1.weight": None, "bottleneck.double_conv.1.bias": None, "bottleneck.double_conv.5.bias": None,
    "upconvs.0.weight": 2, "upconvs.0.bias": None, "decoder.0.double_conv.0.weight": 1, "decoder.0.double_conv.4.weight": 2,
    "decoder.0.double_conv.1.weight": None, "upconvs.3.weight": None, "upconvs.3.bias": None,
    "decoder.3.double_conv.0.weight": None, "decoder.3.double_conv.4.weight": None, "decoder.3.double_conv.5.weight": None,
    "decoder.3.double_conv.5.bias": None, "final_conv.weight": None, "final_conv.bias": None,
}


def _summ(d, pre, model, logits=None):
    """Summary of a model after backward (+ maybe an optimizer step): gradient norms, leading slices, BN buffers."""
    names = [k for k, _ in model.named_parameters()]
    d[pre + "grad_names"] = np.array(names)
    d[pre + "grad_norms"] = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0
                                      for _, p in model.named_parameters()])
    for k, p in model.named_parameters():
        if k in GRAD_SLICES and p.grad is not None:
            n = GRAD_SLICES[k]
            d[pre + "grad/" + k] = npy(p.grad if n is None else p.grad[:n])
    bn = {k: v for k, v in model.state_dict().items() if "running" in k}
    d[pre + "bn_keys"] = np.array(sorted(bn.keys()))
    d[pre + "bn_after"] = np.concatenate([npy(bn[k]).ravel() for k in sorted(bn.keys())])
    if logits is not None:
        s = logits.shape[-1]
        a, b = s // 2 - 2, s // 2 + 2
        d[pre + "logits_center"] = npy(logits[:, :, a:b, a:b, a:b])
        d[pre + "logits_corner"] = npy(logits[:, :, :3, :3, :3])
        d[pre + "logits_absmean"] = np.asarray(float(logits.double().abs().mean()))
        d[pre + "logits_mean_per_class"] = npy(logits.double().mean(dim=(0, 2, 3, 4)))


def _record_dropout_masks(model, cls):
    """Forward hooks on every dropout module: mask scale per (n, c) = out/in (0 or 1/(1-p)).  Returns (handles, store);
    store[name] is a list (one entry per forward call)."""
    store, handles = {}, []

    def mk(name):
        def hook(mod, inp, out):
            x, o = inp[0].detach(), out.detach()
            if x.dim() == 5:
                num, den = o.double().abs().sum(dim=(2, 3, 4)), x.double().abs().sum(dim=(2, 3, 4))
            else:
                num, den = o.double().abs(), x.double().abs()
            keep = 1.0 / (1.0 - mod.p) if mod.p < 1 else 0.0
            sc = torch.where(den > 0, num / den.clamp_min(1e-300), torch.full_like(den, float("nan")))
            # entries whose input was all-zero are undetermined by out/in: any value reproduces the output
            sc = torch.where(torch.isnan(sc), torch.full_like(sc, keep), sc)
            store.setdefault(name, []).append(sc.float())
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, cls):
            handles.append(mod.register_forward_hook(mk(name)))
    return handles, store


def gen_dropout(ref_unet, ref_metrics, out):
    """Dropout3d with REAL masks (models/unet.py:14,18, p = 0.5 and the shipped 0.1): the masks the reference drew are
    recorded by hooks, so the oracle / HIP path can replay them.  Complete small net (all grads) + default net summary."""
    d = {}
    torch.manual_seed(17)
    m = ref_unet.UNet3D(in_channels=2, out_channels=3, features=[4, 8], dropout_rate=0.5)
    g = torch.Generator().manual_seed(171)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    for k, v in m.state_dict().items():
        d["small/sd0/" + k] = npy(v.clone())       # clone: the forward updates the BN buffers in place
    x = torch.randn(3, 2, 8, 4, 12, generator=g)
    y = torch.randint(0, 3, (3, 1, 8, 4, 12), generator=g)
    m.train()
    handles, store = _record_dropout_masks(m, torch.nn.Dropout3d)
    torch.manual_seed(99)
    logits = m(x)
    loss = ref_metrics.combined_loss(logits, y)
    loss.backward()
    for h in handles:
        h.remove()
    d["small/x"], d["small/y"], d["small/logits"], d["small/loss"] = npy(x), npy(y), npy(logits), npy(loss)
    for name, lst in store.items():
        d["small/mask/" + name] = npy(lst[0])
    for k, p in m.named_parameters():
        d["small/grad/" + k] = npy(p.grad)
    for k, v in m.state_dict().items():
        if "running" in k:
            d["small/sd1/" + k] = npy(v)
    # default net, N=2, 16^3, p=0.1 (the _ct_ scripts' setting): summary
    torch.manual_seed(0)
    m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.1)
    m.train()
    x, y = synth(2, 16, 1234)
    handles, store = _record_dropout_masks(m, torch.nn.Dropout3d)
    torch.manual_seed(5)
    logits = m(x)
    loss = ref_metrics.combined_loss(logits, y)
    loss.backward()
    for h in handles:
        h.remove()
    for name, lst in store.items():
        d["default/mask/" + name] = npy(lst[0])
    d["default/logits"], d["default/loss"] = npy(logits), npy(loss)
    _summ(d, "default/", m)
    np.savez_compressed(os.path.join(out, "dropout.npz"), **d)


def gen_oddsize(ref_unet, ref_metrics, out):
    """Volume sides NOT divisible by 2^levels: MaxPool3d floors and UNet3D.forward nearest-resizes the upsampled tensor to
    the skip's shape (models/unet.py:81-83).  Complete small-net fixture (6x10x7, two levels: every side hits the resize)."""
    d = {}
    torch.manual_seed(23)
    m = ref_unet.UNet3D(in_channels=2, out_channels=3, features=[4, 8], dropout_rate=0.0)
    g = torch.Generator().manual_seed(231)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    for k, v in m.state_dict().items():
        d["sd0/" + k] = npy(v.clone())
    x = torch.randn(2, 2, 6, 10, 7, generator=g)
    y = torch.randint(0, 3, (2, 1, 6, 10, 7), generator=g)
    m.train()
    logits = m(x)
    loss = ref_metrics.combined_loss(logits, y)
    loss.backward()
    d["x"], d["y"], d["logits"], d["loss"] = npy(x), npy(y), npy(logits), npy(loss)
    for k, p in m.named_parameters():
        d["grad/" + k] = npy(p.grad)
    for k, v in m.state_dict().items():
        if "running" in k:
            d["sd1/" + k] = npy(v.clone())
    m.eval()
    with torch.no_grad():
        d["logits_eval"] = npy(m(x))
    np.savez_compressed(os.path.join(out, "oddsize.npz"), **d)


def gen_preproc(out):
    """the reference's CombinedDataset.__getitem__ (utils/dataloader.py:148-200) EXECUTED on in-memory
    volumes: the nibabel stand-in's load() hands back seeded arrays instead of reading NIfTI files, everything after it
    (CT window, MRI z-score / percentile clip / min-max, AMOS and CHAOS label remaps) is the reference's own code."""
    import tempfile
    import utils.dataloader as ref_dl
    rng = np.random.default_rng(5)
    shape = (12, 20, 16)
    vols = {}

    class _Img:
        def __init__(self, a):
            self.a = a

        def get_fdata(self):
            return self.a.astype(np.float64)

    root = tempfile.mkdtemp()
    names = ["amos_ct", "amos_mri", "chaos_mri", "ts_ct", "btcv"]
    for ds in names:
        for sub in ("images", "labels"):
            os.makedirs(os.path.join(root, ds, sub))
        ip, lp = os.path.join(root, ds, "images", "v0.nii.gz"), os.path.join(root, ds, "labels", "v0.nii.gz")
        open(ip, "w").close(); open(lp, "w").close()
        if ds.endswith("_ct"):
            img = rng.normal(40.0, 250.0, shape).astype(np.float32)              # HU-like, both window edges exceeded
        else:
            img = (rng.gamma(2.0, 120.0, shape) + 30.0 * rng.standard_normal(shape)).astype(np.float32)   # skewed MRI-like
            img[rng.random(shape) < 0.002] *= 8.0                                # outliers above the 99th percentile
        if ds.startswith("amos"):
            lab = rng.integers(0, 16, shape)
        elif ds.startswith("chaos"):
            lab = rng.choice([0, 54, 55, 63, 70, 71, 109, 110, 126, 135, 136, 174, 175, 189, 200, 201, 239, 240, 252, 255], shape)
        else:
            lab = rng.integers(0, 4, shape)
        vols[ip], vols[lp] = img, lab.astype(np.float64)
    sys.modules["nibabel"].load = lambda path: _Img(vols[path])
    ref_dl.nib.load = sys.modules["nibabel"].load
    ds_obj = ref_dl.CombinedDataset(root, transform=None)
    d = {}
    for i, smp in enumerate(ds_obj.samples):
        name = smp["dataset_name"]
        img_t, lab_t = ds_obj[i]
        d[f"{name}/image_in"] = vols[smp["image_path"]]
        d[f"{name}/label_in"] = vols[smp["label_path"]].astype(np.int64)
        d[f"{name}/image_out"] = npy(img_t)[0]
        d[f"{name}/label_out"] = npy(lab_t)[0]
    d["names"] = np.array(sorted(n for n in names))
    np.savez_compressed(os.path.join(out, "preproc.npz"), **d)


def _autocast_yardstick(d, pre, ref_unet, ref_metrics, x, y, fp32_logits, fp32_grads, seed=0):
    """The reference's own bf16 autocast run (accelerate mixed_precision='bf16') vs its fp32 run on the same inputs."""
    torch.manual_seed(seed)
    m2 = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    m2.train()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lg2 = m2(x)
    lg2 = lg2.float()
    loss2 = ref_metrics.combined_loss(lg2, y)
    loss2.backward()
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    d[pre + "logits_relerr"] = np.asarray(rel(lg2.detach(), fp32_logits))
    d[pre + "loss"] = npy(loss2)
    d[pre + "dice"] = npy(ref_metrics.calculate_dice(lg2, y))
    d[pre + "iou"] = npy(ref_metrics.calculate_iou(lg2, y))
    d[pre + "acc"] = npy(ref_metrics.calculate_accuracy(lg2, y))
    d[pre + "grad_relerr"] = np.array([rel(p.grad, fp32_grads[k]) for k, p in m2.named_parameters()])


def gen_config2(ref_unet, ref_metrics, out):
    """BASELINE config 2 at its real size: UNet3D 96^3, N=2 (fp32 reference + the reference's autocast-bf16 run)."""
    d = {}
    torch.manual_seed(0)
    m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    x, y = synth(2, 96, 1234)
    m.train()
    logits = m(x)
    loss = ref_metrics.combined_loss(logits, y)
    loss.backward()
    d["loss"] = npy(loss)
    d["dice"], d["iou"], d["acc"] = (npy(ref_metrics.calculate_dice(logits, y)), npy(ref_metrics.calculate_iou(logits, y)),
                                     npy(ref_metrics.calculate_accuracy(logits, y)))
    _summ(d, "", m, logits.detach())
    fp32_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    _autocast_yardstick(d, "autocast_bf16/", ref_unet, ref_metrics, x, y, logits.detach(), fp32_grads)
    np.savez_compressed(os.path.join(out, "config2_96.npz"), **d)


def gen_config5(ref_unet, ref_metrics, out):
    """BASELINE config 5: distillation step (distill_unet.py:107-115) at 128^3, N=2, alpha 0.7, T 2.0."""
    d = {}
    torch.manual_seed(0)
    student = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    torch.manual_seed(1)
    teacher = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, b in teacher.named_buffers():
            if k.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if k.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    student.train(); teacher.eval()
    x, y = synth(2, 128, 1234)
    s = student(x)
    with torch.no_grad():
        t = teacher(x)
    l = ref_metrics.distillation_loss(s, t, y, 0.7, 2.0)
    l.backward()
    d["loss"] = npy(l)
    d["dice"] = npy(ref_metrics.calculate_dice(s, y))
    _summ(d, "student/", student, s.detach())
    tt = {}
    _summ_logits = {}
    a, b = 62, 66
    d["teacher/logits_center"] = npy(t[:, :, a:b, a:b, a:b])
    d["teacher/logits_absmean"] = np.asarray(float(t.double().abs().mean()))
    d["teacher/logits_mean_per_class"] = npy(t.double().mean(dim=(0, 2, 3, 4)))
    np.savez_compressed(os.path.join(out, "config5_128.npz"), **d)


def _ct_like(x):
    """images in HU-like units pushed through the reference's CT window (utils/dataloader.py:111-117)"""
    hu = 200.0 * x.numpy()
    hu = np.clip(hu, -160, 240)
    return torch.from_numpy(((hu - (-160)) / (240 - (-160))).astype(np.float32))


def _mri_like(x):
    """utils/dataloader.py:128-144 on each volume: z-score, 1-99 percentile clip, min-max"""
    outs = []
    for v in x.numpy():
        im = (v - np.mean(v)) / (np.std(v) + 1e-8)
        low, high = np.percentile(im, [1, 99])
        im = np.clip(im, low, high)
        outs.append(((im - low) / (high - low + 1e-8)).astype(np.float32))
    return torch.from_numpy(np.stack(outs))


class _Args:
    pass


def gen_config4(ref_unet_dann, ref_train_unet, ref_train_dann, out):
    """BASELINE config 4 per rank: the reference's own train_one_epoch_dann (train_dann.py:225-301) on ONE batch of
    N=2 source (CT-like, labelled) + N=2 target (MRI-like) 96^3 volumes, lambda 0.2, loss ce_tversky (its default),
    AdamW lr 1e-3 wd 0.01 for both nets (train_dann.py:421-422), Dropout masks of the discriminator recorded."""
    d = {}
    torch.manual_seed(0)
    seg = ref_unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    torch.manual_seed(3)
    disc = ref_train_dann.DomainDiscriminator(256)
    xs, ys = synth(2, 96, 1234)
    xt, _ = synth(2, 96, 4321)
    xs, xt = _ct_like(xs), _mri_like(xt)
    opt_s = torch.optim.AdamW(seg.parameters(), lr=1e-3, weight_decay=0.01)
    opt_d = torch.optim.AdamW(disc.parameters(), lr=1e-3, weight_decay=0.01)
    args = _Args()
    args.epochs, args.gradient_accumulation_steps = 1, 1
    handles, store = _record_dropout_masks(disc, torch.nn.Dropout)
    feats = []
    h2 = disc.register_forward_hook(lambda mod, inp, outp: feats.append((inp[0].detach().clone(), outp.detach().clone())))
    torch.manual_seed(77)
    res = ref_train_dann.train_one_epoch_dann(seg, disc, ([(xs, ys)], [(xt, torch.zeros(1))]), opt_s, opt_d,
                                              torch.device("cpu"), 0, args, ref_train_unet.get_loss_fn("ce_tversky"), 0.2)
    for h in handles + [h2]:
        h.remove()
    d["task"], d["domain"], d["dice"], d["iou"], d["acc"] = [np.asarray(float(v)) for v in res]
    d["xs_stats"] = np.array([float(xs.mean()), float(xs.std()), float(xt.mean()), float(xt.std())])
    d["sfeat"], d["tfeat"] = npy(feats[0][0]), npy(feats[1][0])
    d["spred"], d["tpred"] = npy(feats[0][1]), npy(feats[1][1])
    for name, lst in store.items():
        d["disc_mask/" + name] = np.stack([npy(v) for v in lst])       # [2 calls (source, target)][N][features]
    _summ(d, "seg/", seg)
    d["disc/grad_names"] = np.array([k for k, _ in disc.named_parameters()])
    d["disc/grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in disc.named_parameters()])
    for k, p in disc.named_parameters():
        d["disc/grad/" + k] = npy(p.grad if p.numel() <= 4096 else p.grad[:4])
    _, dig = param_digest(seg.state_dict())
    d["seg/param_digest_after"] = dig
    _, dig = param_digest(disc.state_dict())
    d["disc/param_digest_after"] = dig
    np.savez_compressed(os.path.join(out, "config4_dann96.npz"), **d)


def gen_loops(ref_unet, ref_unet_dann, ref_train_unet, ref_train_dann, out):
    """The reference's step loops executed as they are, at 32^3 (a 16^3 input has a 1x1x1 bottleneck whose
    BatchNorm over 2 values amplifies fp32 roundoff):
      accum/   train_unet.train_one_epoch (train_unet.py:207-257) under Accelerator(gradient_accumulation_steps=2): 4
               micro-batches = 2 optimizer steps.  Pins Q2 (zero_grad inside accumulate(): only the boundary
               micro-batch's gradient survives, scaled 1/accum).
      eval/    train_unet.evaluate (:259-305) with the ce_tversky loss on 2 batches of 1.
      dann/    train_dann.train_one_epoch_dann with gradient_accumulation_steps=2 over 2 batches (one optimizer step)."""
    import tempfile
    from accelerate import Accelerator
    d = {}
    tmp = tempfile.mkdtemp()
    args = _Args()
    args.epochs, args.experiment_dir, args.experiment_name = 1, tmp, "x"
    os.makedirs(os.path.join(tmp, "x", "logs"), exist_ok=True)
    acc = Accelerator(gradient_accumulation_steps=2, cpu=True)
    torch.manual_seed(0)
    m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    m, opt = acc.prepare(m, opt)
    loader = [synth(2, 32, 500 + i, blocky=(i % 2 == 0)) for i in range(4)]
    loss_fn = ref_train_unet.get_loss_fn("combined")
    res = ref_train_unet.train_one_epoch(m, loader, opt, acc, 0, args, loss_fn)
    d["accum/result"] = np.array([float(v) for v in res])
    mm = acc.unwrap_model(m)
    _summ(d, "accum/", mm)
    _, dig = param_digest(mm.state_dict())
    d["accum/param_digest_after"] = dig
    # the same loop stopped after the FIRST accumulation window (2 micro-batches, one optimizer step): its gradients are
    # grad(second micro-batch)/2 at the initial parameters -- free of AdamW's sign-like nonlinearity
    acc1 = Accelerator(gradient_accumulation_steps=2, cpu=True)
    torch.manual_seed(0)
    m1 = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    opt1 = torch.optim.AdamW(m1.parameters(), lr=1e-3, weight_decay=0.01)
    m1, opt1 = acc1.prepare(m1, opt1)
    res1 = ref_train_unet.train_one_epoch(m1, loader[:2], opt1, acc1, 0, args, loss_fn)
    d["accum_first/result"] = np.array([float(v) for v in res1])
    _summ(d, "accum_first/", acc1.unwrap_model(m1))
    # evaluate() on the model as trained above, ce_tversky loss
    ev_loader = [synth(1, 32, 600 + i, blocky=True) for i in range(2)]
    res = ref_train_unet.evaluate(m, ev_loader, acc, 0, args, ref_train_unet.get_loss_fn("ce_tversky"))
    d["eval/result"] = np.array([float(v) for v in res])
    per = []
    mm.eval()
    with torch.no_grad():
        for xx, yy in ev_loader:
            per.append(float(ref_train_unet.get_loss_fn("ce_tversky")(mm(xx), yy)))
    d["eval/per_batch_loss"] = np.array(per)
    # DANN loop with accumulation 2 (train_dann.py:237-239,286-289): both micro-batches accumulate, one step
    torch.manual_seed(0)
    seg = ref_unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    torch.manual_seed(3)
    disc = ref_train_dann.DomainDiscriminator(256)
    for mod in disc.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    opt_s = torch.optim.AdamW(seg.parameters(), lr=1e-3, weight_decay=0.01)
    opt_d = torch.optim.AdamW(disc.parameters(), lr=1e-3, weight_decay=0.01)
    a2 = _Args()
    a2.epochs, a2.gradient_accumulation_steps = 1, 2
    src = [synth(2, 32, 700 + i) for i in range(2)]
    tgt = [(synth(2, 32, 800 + i)[0], torch.zeros(1)) for i in range(2)]
    res = ref_train_dann.train_one_epoch_dann(seg, disc, (src, tgt), opt_s, opt_d, torch.device("cpu"), 0, a2,
                                              ref_train_unet.get_loss_fn("combined"), 0.2)
    d["dann/result"] = np.array([float(v) for v in res])      # task, domain, dice, iou, acc (means over the 2 batches)
    _summ(d, "dann/seg/", seg)
    d["dann/disc_grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in disc.named_parameters()])
    _, dig = param_digest(seg.state_dict())
    d["dann/seg_param_digest_after"] = dig
    _, dig = param_digest(disc.state_dict())
    d["dann/disc_param_digest_after"] = dig
    # encoder freezing exactly as train_unet.py:31-50,413-431 does it: freeze_encoder + update_optimizer_for_frozen_encoder,
    # two steps; then unfreeze_encoder + a fresh AdamW over all parameters, one step
    accf = Accelerator(gradient_accumulation_steps=1, cpu=True)
    torch.manual_seed(0)
    mf = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    optf = torch.optim.AdamW(mf.parameters(), lr=1e-3, weight_decay=0.01)
    mf, optf = accf.prepare(mf, optf)
    raw = accf.unwrap_model(mf)
    ref_train_unet.freeze_encoder(raw)
    optf = accf.prepare(ref_train_unet.update_optimizer_for_frozen_encoder(raw, optf, 1e-3))
    fl = [synth(2, 32, 900 + i, blocky=True) for i in range(3)]
    res = ref_train_unet.train_one_epoch(mf, fl[:2], optf, accf, 0, args, loss_fn)
    d["freeze/result_frozen"] = np.array([float(v) for v in res])
    _, dig = param_digest(raw.state_dict())
    d["freeze/param_digest_frozen"] = dig
    d["freeze/frozen_grad_is_none"] = np.array([p.grad is None for p in raw.encoder.parameters()])
    ref_train_unet.unfreeze_encoder(raw)
    optf = accf.prepare(torch.optim.AdamW(raw.parameters(), lr=1e-3, weight_decay=0.01))
    res = ref_train_unet.train_one_epoch(mf, fl[2:], optf, accf, 0, args, loss_fn)
    d["freeze/result_unfrozen"] = np.array([float(v) for v in res])
    _, dig = param_digest(raw.state_dict())
    d["freeze/param_digest_unfrozen"] = dig
    np.savez_compressed(os.path.join(out, "loops.npz"), **d)


# ---------------------------------------------------------------------------------------------- data-parallel (A15)
def _dp2_data(kind):
    """Per-rank micro-batches of the dp2 fixture, regenerated from seeds by the tests (nothing but seeds is stored)."""
    if kind == "plain":          # 2 micro-batches per rank, N = 2, 32^3: rank r sees seeds 1000 + 10 r + i
        return [[synth(2, 32, 1000 + 10 * r + i, blocky=(i % 2 == 0)) for i in range(2)] for r in range(2)]
    # "loader": ONE dataset of 6 batches of N = 2 (seeds 1100 + b); accelerate's BatchSamplerShard deals batch b to rank b % 2
    return [synth(2, 32, 1100 + b, blocky=(b % 2 == 1)) for b in range(6)]


def _dp2_worker(rank, out_path):
    """One of the two CPU ranks (gloo): the reference's train_one_epoch under accelerate exactly as train_unet.py:309-312,
    384-386 sets it up -- Accelerator(...) -> prepare(model, optimizer[, loader]) = DDP wrap -> accumulate / backward /
    gather.  Captures what a rank observes: the returned means, its (all-reduced) gradients, its BatchNorm buffers."""
    import tempfile
    from accelerate import Accelerator
    try:
        import datasets  # noqa: F401  (accelerate.prepare(DataLoader) imports it; it must probe for nibabel BEFORE the stand-ins exist)
    except ImportError:
        pass
    ref_unet, _, _, ref_train_unet, _ = _import_reference()
    torch.set_num_threads(4)
    d = {}
    tmp = tempfile.mkdtemp()
    args = _Args()
    args.epochs, args.experiment_dir, args.experiment_name = 1, tmp, "x"
    os.makedirs(os.path.join(tmp, "x", "logs"), exist_ok=True)
    loss_fn = ref_train_unet.get_loss_fn("combined")

    def fresh(accum):
        acc = Accelerator(gradient_accumulation_steps=accum, cpu=True)
        assert acc.num_processes == 2 and acc.process_index == rank, (acc.num_processes, acc.process_index)
        torch.manual_seed(0)
        m = ref_unet.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        return acc, m, opt

    # ---- plain/: accumulation 1, each rank iterates its own list of micro-batches (the reference's loop with a sharded
    # sampler); stopped after ONE step (gradients = mean over ranks at the initial parameters) and after TWO steps
    data = _dp2_data("plain")[rank]
    for tag, nb in (("plain1/", 1), ("plain2/", 2)):
        acc, m, opt = fresh(1)
        m, opt = acc.prepare(m, opt)
        res = ref_train_unet.train_one_epoch(m, data[:nb], opt, acc, 0, args, loss_fn)
        raw = acc.unwrap_model(m)
        d[tag + "result"] = np.array([float(v) for v in res])
        _summ(d, tag, raw)
        _, dig = param_digest(raw.state_dict())
        d[tag + "param_digest_after"] = dig
        acc.free_memory()
    # ---- loader/: accumulation 2 over a PREPARED DataLoader of 3 batches per rank (3 % 2 != 0): accelerate forces
    # sync + optimizer.step on the last batch of the epoch (GradientState.end_of_dataloader), so the steps land on batches
    # 2 and 3; with the reference's zero_grad-inside-accumulate (Q2) the applied gradients are g2/2 and then g3/2
    batches = _dp2_data("loader")
    xs = torch.cat([b[0] for b in batches]); ys = torch.cat([b[1] for b in batches])
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xs, ys), batch_size=2, shuffle=False)
    acc, m, opt = fresh(2)
    m, opt, dl = acc.prepare(m, opt, dl)
    seen = []
    steps = []
    raw = acc.unwrap_model(m)
    real_step = opt.optimizer.step

    def counting_step(*a, **k):
        steps.append(len(seen))
        return real_step(*a, **k)
    opt.optimizer.step = counting_step

    class _Spy:                      # records which samples this rank saw, in order (first voxel of each volume)
        def __init__(self, it): self.it = it
        def __len__(self): return len(self.it)
        def __iter__(self):
            for xb, yb in self.it:
                seen.append(float(xb.double().sum()))
                yield xb, yb
    res = ref_train_unet.train_one_epoch(m, _Spy(dl), opt, acc, 0, args, loss_fn)
    d["loader/result"] = np.array([float(v) for v in res])
    d["loader/seen_sum"] = np.array(seen)
    d["loader/step_after_batch"] = np.array(steps)
    _summ(d, "loader/", raw)
    _, dig = param_digest(raw.state_dict())
    d["loader/param_digest_after"] = dig
    np.savez_compressed(out_path, **d)


def gen_dp2(out):
    """A15 / SURVEY 2.2 C1-C4 pinned to the reference: TWO CPU processes (gloo) run train_unet.train_one_epoch under
    accelerate's DDP wrap on per-rank shards.  dp2.npz holds, per rank r, `r{r}/...`: returned means (already gathered and
    averaged over ranks by the reference), gradient norms / slices (identical on both ranks: DDP averages), BatchNorm
    buffers (rank-local statistics; DDP broadcasts rank 0's at every forward), parameter digests after the steps."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp()
    port = 29611
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ACCELERATE_USE_CPU="1", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dp2-worker", str(r),
                                       "--out", os.path.join(tmp, f"r{r}.npz")], env=env))
    for pr in procs:
        if pr.wait(timeout=1800) != 0:
            raise RuntimeError("dp2 worker failed")
    d = {}
    for r in range(2):
        for k, v in np.load(os.path.join(tmp, f"r{r}.npz"), allow_pickle=False).items():
            # DDP all-reduces the gradients: rank 1's slices (and its copies of the name tables) are rank 0's, bit for bit
            # (asserted here) -- only rank 0's are stored
            if r == 1 and ("/grad/" in k or k.endswith("/grad_names") or k.endswith("/bn_keys")):
                assert np.array_equal(v, d["r0/" + k]), k
                continue
            d[f"r{r}/{k}"] = v
    np.savez_compressed(os.path.join(out, "dp2.npz"), **d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="comma-separated fixture names (default: all)")
    ap.add_argument("--dp2-worker", type=int, default=-1, help="internal: rank of a gen_dp2 worker process")
    a = ap.parse_args()
    if a.dp2_worker >= 0:
        _dp2_worker(a.dp2_worker, a.out)
        return
    only = set(filter(None, a.only.split(",")))
    want = lambda name: not only or name in only
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(False)
    ref_unet, ref_unet_dann, ref_metrics, ref_train_unet, ref_train_dann = _import_reference()
    if want("small_unet"): gen_small_unet(ref_unet, ref_metrics, a.out)
    if want("doubleconv"): gen_doubleconv(ref_unet, a.out)
    if want("losses_metrics"): gen_losses(ref_metrics, ref_train_unet, a.out)
    if want("default_unet"): gen_default_unet(ref_unet, ref_metrics, a.out)
    if want("dann"): gen_dann(ref_unet_dann, ref_metrics, ref_train_dann, a.out)
    if want("distill"): gen_distill(ref_unet, ref_metrics, a.out)
    if want("dropout"): gen_dropout(ref_unet, ref_metrics, a.out)
    if want("oddsize"): gen_oddsize(ref_unet, ref_metrics, a.out)
    if want("preproc"): gen_preproc(a.out)
    if want("loops"): gen_loops(ref_unet, ref_unet_dann, ref_train_unet, ref_train_dann, a.out)
    if want("dp2"): gen_dp2(a.out)
    if want("config2_96"): gen_config2(ref_unet, ref_metrics, a.out)
    if want("config4_dann96"): gen_config4(ref_unet_dann, ref_train_unet, ref_train_dann, a.out)
    if want("config5_128"): gen_config5(ref_unet, ref_metrics, a.out)
    with open(os.path.join(a.out, "PROVENANCE.txt"), "w") as f:
        f.write("generated by tools/gen_golden.py from /root/reference (fransiskusbudi/multimodal_segmentation_project @ 2025-08-24)\n")
        f.write(f"torch {torch.__version__} CPU, numpy {np.__version__}, threads {torch.get_num_threads()}\n")
    for fn in sorted(os.listdir(a.out)):
        print(fn, os.path.getsize(os.path.join(a.out, fn)))


if __name__ == "__main__":
    main()
