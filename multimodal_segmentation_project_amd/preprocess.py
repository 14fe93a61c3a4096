"""GPU mirror of the per-volume arithmetic of the reference's input pipeline, CombinedDataset.__getitem__
(utils/dataloader.py:148-200): what two DataLoader workers compute in numpy per 192^3 volume runs here as a handful of
HBM-bound kernels on volumes already resident on the device (file decoding / MONAI augmentation stay outside: SURVEY §8 F4).

  preprocess_ct(image)            utils/dataloader.py:111-117
  preprocess_mri(image)           utils/dataloader.py:128-144   (exact np.percentile order statistics via radix select)
  remap_labels(label, dataset)    utils/dataloader.py:162-181   (AMOS table, CHAOS ranges, ts / btcv identity)
  preprocess(image, dataset)      the modality dispatch of :153-159 ('_ct' suffix -> CT, everything else -> MRI)
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def _vol(image):
    _lib.require_cuda(image, "preprocess")
    return image.contiguous().float()


def preprocess_ct(image, window_min=-160.0, window_max=240.0):
    x = _vol(image)
    out = torch.empty_like(x)
    call("mi3d_preprocess_ct", ptr(x), ptr(out), x.numel(), float(window_min), float(window_max), stream_ptr())
    return out


def preprocess_mri(image, p_low=1.0, p_high=99.0):
    x = _vol(image)
    out = torch.empty_like(x)
    ws = torch.empty(_lib.lib().mi3d_preprocess_mri_workspace_bytes(), dtype=torch.uint8, device=x.device)
    call("mi3d_preprocess_mri", ptr(x), ptr(out), x.numel(), float(p_low), float(p_high), ptr(ws), stream_ptr())
    return out


def preprocess(image, dataset_name):
    return preprocess_ct(image) if dataset_name.lower().endswith("_ct") else preprocess_mri(image)


def remap_labels(label, dataset_name):
    _lib.require_cuda(label, "remap_labels")
    lab = label.contiguous().long()
    kind = 1 if dataset_name.startswith("amos") else 2 if dataset_name.startswith("chaos") else 0
    out = torch.empty_like(lab)
    call("mi3d_remap_labels", ptr(lab), ptr(out), lab.numel(), kind, stream_ptr())
    return out
