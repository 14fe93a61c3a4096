"""Checkpoint wire format of the reference — pure host code.

The reference saves ``{epoch, model_state_dict, optimizer_state_dict, train_loss, val_loss, train_dice, val_dice,
encoder_frozen}`` (train_unet.py:477-486; DANN adds task_loss/domain_loss, train_dann.py:478-489; distillation saves
``{'model_state_dict'}`` only, distill_unet.py:256) and loads tolerantly: raw or wrapped dicts
(finetune_ct.py:246-268, train_dann.py:406-412) with an optional ``module.`` prefix (test_model.py:381-385).
``UNet3D.state_dict()`` here has the same 136 keys / shapes / dtypes, so ``.pth`` files interchange both ways.
"""
import torch


def extract_state_dict(obj):
    """state_dict out of a raw state_dict or a reference-style checkpoint dict; strips a DDP ``module.`` prefix."""
    sd = obj
    if isinstance(obj, dict) and "model_state_dict" in obj:
        sd = obj["model_state_dict"]
    if not isinstance(sd, dict):
        raise TypeError(f"not a checkpoint / state_dict: {type(obj)}")
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_model(model, path_or_obj, strict=True, map_location="cpu"):
    """finetune_ct.py:246-268 / test_model.py:381-385 equivalent.  Parameters that live in a TrainStep arena are
    updated IN PLACE (load_state_dict copies into the existing storage), so a running TrainStep keeps working."""
    obj = torch.load(path_or_obj, map_location=map_location) if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, "read") \
        else path_or_obj
    result = model.load_state_dict(extract_state_dict(obj), strict=strict)
    return obj if isinstance(obj, dict) else {}, result


def adamw_state_dict(train_step):
    """The fused arena optimizer's state in ``torch.optim.AdamW.state_dict()`` layout (one param group, per-parameter
    ``step`` / ``exp_avg`` / ``exp_avg_sq``), so the reference's ``optimizer_state_dict`` slot is filled with something
    ``torch.optim.AdamW.load_state_dict`` accepts."""
    a = train_step.arena
    step = a.step.detach().to("cpu").to(torch.float32).reshape(())
    state = {}
    for i, (p, o) in enumerate(zip(a.params, a.offsets)):
        n = p.numel()
        state[i] = {"step": step.clone(), "exp_avg": a.m[o:o + n].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": a.v[o:o + n].view(p.shape).detach().cpu().clone()}
    lr, b1, b2, eps, wd = train_step._hyper()[:5]
    group = {"lr": lr, "betas": (b1, b2), "eps": eps, "weight_decay": wd,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": True, "params": list(range(len(a.params)))}
    return {"state": state, "param_groups": [group]}


def save_checkpoint(path, model, epoch, train_loss=None, val_loss=None, train_dice=None, val_dice=None,
                    encoder_frozen=False, optimizer_state_dict=None, **extra):
    """Writes the dict of train_unet.py:477-486 (extra keys, e.g. task_loss / domain_loss for DANN, pass through)."""
    ckpt = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
            "optimizer_state_dict": optimizer_state_dict if optimizer_state_dict is not None else {},
            "train_loss": train_loss, "val_loss": val_loss, "train_dice": train_dice, "val_dice": val_dice,
            "encoder_frozen": encoder_frozen}
    ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt
