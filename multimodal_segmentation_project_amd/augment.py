"""GPU mirror of the reference's training-set augmentation, `combined_transform()` (utils/dataloader.py:223-262; applied
by CombinedDataset.__getitem__ at :192-196 for the training split, train_unet.py:361).

The reference composes five MONAI dictionary transforms — RandBiasFieldd, RandGaussianNoised, RandAdjustContrastd,
RandHistogramShiftd (image, prob 0.3 each) and RandCoarseDropoutd (image + label, 2 holes of 16^3, prob 0.3).  Here the
host draws the few random scalars (numpy RandomState streams laid out like MONAI's Compose: one seed per transform from
`set_random_state`, an outer stream for the wrapper's prob draw and an inner one for the transform's parameters) and
the per-voxel arithmetic runs on the device in at most three read+write passes (`mi3d_augment`, csrc/augment.hip).

    tf = combined_transform()                    # same name and dict-in / dict-out call as the reference's
    tf.set_random_state(seed)                    # MONAI Compose API
    out = tf({'image': image, 'label': label})   # (C, D, H, W) CUDA tensors (numpy / CPU inputs are uploaded)

MONAI (`monai>=1.2.0`, requirements.txt:10) is not under /root/reference and not installed: the algorithms are restated
from its published source, parity is unpinned at that boundary.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

MAX_SEED = np.iinfo(np.uint32).max + 1


class AugmentDraw:
    """One sample's drawn parameters (host values; `noise` is a float32 numpy array, a device seed, or None)."""
    __slots__ = ("bias_coeff", "noise", "noise_std", "noise_seed", "gamma", "ref_cp", "flt_cp", "hole_lo", "hole_size")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


def _n_bias_coeff(rank, degree):
    return int(np.prod([(degree + k) / k for k in range(1, rank + 1)]))


class CombinedTransform:
    """`Compose([...])` of utils/dataloader.py:224-262 with MONAI's default arguments for everything the reference does
    not set (bias degree 3, coeff_range (0, 0.1); noise std sampled from U(0, std))."""

    def __init__(self, prob=0.3, bias_degree=3, coeff_range=(0.0, 0.1), noise_mean=0.0, noise_std=0.01, gamma=(0.7, 1.5),
                 num_control_points=5, holes=2, hole_size=(16, 16, 16), fill_value=0.0, noise="device"):
        if noise not in ("device", "host"):
            raise _lib.Mi3dError("noise must be 'device' (drawn by the kernel) or 'host' (numpy stream, uploaded)")
        if not 0 <= bias_degree <= 3:
            raise _lib.Mi3dError("bias_degree must be 0..3")
        if not 2 <= num_control_points <= _lib.AUG_MAX_CP or not 0 <= holes <= _lib.AUG_MAX_HOLES:
            raise _lib.Mi3dError("num_control_points must be 2..16 and holes 0..8")
        self.prob, self.bias_degree, self.coeff_range = prob, bias_degree, coeff_range
        self.noise_mean, self.noise_std, self.gamma = noise_mean, noise_std, gamma
        self.num_control_points, self.holes, self.hole_size, self.fill_value = num_control_points, holes, hole_size, fill_value
        self.noise = noise
        self.set_random_state(None)

    def set_random_state(self, seed=None):
        """Compose.set_random_state: a uint32 seed per transform, the wrapper and the wrapped transform seeded alike."""
        if seed is None:
            seeds = [None] * 5
        else:
            r = np.random.RandomState(seed)
            seeds = [int(r.randint(MAX_SEED, dtype="uint32")) for _ in range(5)]
        self._outer = [np.random.RandomState(s) for s in seeds]
        self._inner = [np.random.RandomState(s) for s in seeds]
        return self

    def draw(self, img_shape):
        """Draw one sample's parameters for an image of shape (C, D, H, W), in the order MONAI's transforms draw them."""
        p = AugmentDraw()
        spatial = tuple(int(s) for s in img_shape[1:])
        o, r = self._outer[0], self._inner[0]
        if o.rand() < self.prob:
            r.rand()
            p.bias_coeff = r.uniform(self.coeff_range[0], self.coeff_range[1], _n_bias_coeff(3, self.bias_degree)).tolist()
        o, r = self._outer[1], self._inner[1]
        if o.rand() < self.prob:
            r.rand()
            p.noise_std = float(r.uniform(0, self.noise_std))
            if self.noise == "host":
                p.noise = r.normal(self.noise_mean, p.noise_std, size=tuple(int(s) for s in img_shape)).astype(np.float32)
            else:
                p.noise_seed = int(r.randint(MAX_SEED, dtype="uint32"))
        o, r = self._outer[2], self._inner[2]
        if o.rand() < self.prob:
            r.rand()
            p.gamma = float(r.uniform(low=self.gamma[0], high=self.gamma[1]))
        o, r = self._outer[3], self._inner[3]
        if o.rand() < self.prob:
            r.rand()
            n = int(r.randint(self.num_control_points, self.num_control_points + 1))
            ref = np.linspace(0, 1, n)
            flt = np.copy(ref)
            for i in range(1, n - 1):
                flt[i] = r.uniform(flt[i - 1], flt[i + 1])
            p.ref_cp, p.flt_cp = ref, flt
        o, r = self._outer[4], self._inner[4]
        if o.rand() < self.prob:
            r.rand()
            size = tuple(min(int(s), d) for s, d in zip(self.hole_size, spatial))
            p.hole_lo = [tuple(int(r.randint(0, d - s + 1)) if d > s else 0 for d, s in zip(spatial, size))
                         for _ in range(self.holes)]
            p.hole_size = size
        return p

    def __call__(self, sample):
        image, label = _device_volume(sample["image"], torch.float32), sample["label"]
        p = self.draw(tuple(image.shape))
        out = dict(sample)
        out["image"] = apply_image(image, p, noise_mean=self.noise_mean, bias_degree=self.bias_degree,
                                   fill_value=self.fill_value)
        if p.hole_lo is not None:                # a fresh tensor, like MONAI: the caller's label is left alone
            out["label"] = apply_label(_device_volume(label, torch.int64).clone(), p)
        return out


def combined_transform(**kw):
    return CombinedTransform(**kw)


def _device_volume(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.as_tensor(np.ascontiguousarray(x))
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.Mi3dError("augment: no GPU to upload the volume to (the MI355X HIP path has no CPU fallback)")
        x = x.cuda(non_blocking=True)
    if x.dim() != 4:
        raise _lib.Mi3dError(f"augment: expected a (C, D, H, W) volume, got shape {tuple(x.shape)}")
    return x.contiguous().to(dtype)


def _params_struct(p, noise_mean, bias_degree, fill_value):
    a = _lib.AugParams()
    if p.bias_coeff is not None:
        if len(p.bias_coeff) != _n_bias_coeff(3, bias_degree):
            raise _lib.Mi3dError(f"bias field of degree {bias_degree} takes {_n_bias_coeff(3, bias_degree)} coefficients")
        a.do_bias, a.bias_degree = 1, bias_degree
        for i, c in enumerate(p.bias_coeff):
            a.bias_coeff[i] = float(c)
    if p.noise is not None or p.noise_seed is not None:
        a.do_noise, a.noise_mean = 1, float(noise_mean)
        a.noise_std = float(p.noise_std or 0.0)
        a.noise_seed = int(p.noise_seed or 0)
    if p.gamma is not None:
        a.do_contrast, a.gamma = 1, float(p.gamma)
    if p.ref_cp is not None:
        a.do_hist, a.n_cp = 1, len(p.ref_cp)
        for i, (x, y) in enumerate(zip(p.ref_cp, p.flt_cp)):
            a.ref_cp[i], a.flt_cp[i] = float(x), float(y)
    if p.hole_lo is not None:
        a.n_holes = len(p.hole_lo)
        for i in range(3):
            a.hole_size[i] = int(p.hole_size[i])
        for q, lo in enumerate(p.hole_lo):
            for i in range(3):
                a.hole_lo[q][i] = int(lo[i])
    a.fill_value = float(fill_value)
    return a


def apply_image(image, p, noise_mean=0.0, bias_degree=3, fill_value=0.0, out=None):
    """The image half of the chain with already-drawn parameters `p` on a (C, D, H, W) float32 CUDA tensor."""
    _lib.require_cuda(image, "augment")
    x = image.contiguous().float()
    out = torch.empty_like(x) if out is None else out
    noise = None
    if p.noise is not None:
        noise = torch.as_tensor(p.noise).to(x.device, non_blocking=True).contiguous()
        if tuple(noise.shape) != tuple(x.shape):
            raise _lib.Mi3dError("augment: noise tensor shape differs from the image's")
    a = _params_struct(p, noise_mean, bias_degree, fill_value)
    nbytes = _lib.lib().mi3d_augment_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    c, d, h, w = x.shape
    call("mi3d_augment", ptr(x), ptr(out), ptr(noise), c, d, h, w, C.byref(a), ptr(ws), nbytes, stream_ptr())
    return out


def apply_label(label, p, fill=0):
    """The label half of RandCoarseDropoutd (in place on a contiguous int64 (C, D, H, W) CUDA tensor)."""
    _lib.require_cuda(label, "augment")
    if label.dtype != torch.int64 or not label.is_contiguous():
        raise _lib.Mi3dError("augment: label must be a contiguous int64 tensor")
    if p.hole_lo is None:
        return label
    n = len(p.hole_lo)
    lo = (C.c_int32 * (3 * n))(*[int(v) for q in p.hole_lo for v in q])
    size = (C.c_int32 * 3)(*[int(v) for v in p.hole_size])
    c, d, h, w = label.shape
    call("mi3d_fill_boxes_i64", ptr(label), c, d, h, w, n, lo, size, int(fill), stream_ptr())
    return label
