"""CPU restatement of the training-set augmentation chain of the reference's input pipeline (TEST INFRASTRUCTURE).

`combined_transform()` (utils/dataloader.py:223-262, paths relative to /root/reference; used at train_unet.py:361) is a
MONAI `Compose` of five dictionary transforms:
  RandBiasFieldd(image, prob .3) -> RandGaussianNoised(image, prob .3, mean 0, std .01) ->
  RandAdjustContrastd(image, prob .3, gamma (.7, 1.5)) -> RandHistogramShiftd(image, prob .3, 5 control points) ->
  RandCoarseDropoutd(image + label, 2 holes of 16^3, fill 0, prob .3)

PARITY UNPINNED.  The arithmetic lives in a third-party dependency that is neither under /root/reference nor installed
here: MONAI, `monai>=1.2.0` (requirements.txt:10, no upper pin, no lock file).  What follows restates MONAI's published
algorithm for these transforms (monai/transforms/intensity/array.py: RandBiasField, RandGaussianNoise, AdjustContrast /
RandAdjustContrast, RandHistogramShift, RandCoarseDropout; intensity/dictionary.py for the *d wrappers; utils.py:
get_random_patch, get_valid_patch_size; compose.py: Compose.set_random_state) as of the 1.2-1.3 line, with the same
numpy / torch calls MONAI makes (np.polynomial.legendre.leggrid3d, torch float32 arithmetic, numpy RandomState draws).
The reference holds no test, fixture or golden vector for this chain, so neither this file nor the device kernels can
be checked against MONAI's real output in this container; tests compare the device kernels with THIS restatement.
"""
import numpy as np
import torch

MAX_SEED = np.iinfo(np.uint32).max + 1

# ---------------------------------------------------------------------------------------------------------------------
# deterministic arithmetic (the part the device runs)


def bias_field(shape, coeff, degree=3):
    """RandBiasField._generate_random_field, rank 3: Legendre series on linspace(-1, 1) float32 coordinates, float64."""
    coeff_mat = np.zeros((degree + 1,) * 3)
    coords = [np.linspace(-1.0, 1.0, dim, dtype=np.float32) for dim in shape]
    pts = [[0, 0, 0]]
    for i in range(degree + 1):
        for j in range(degree + 1 - i):
            for k in range(degree + 1 - i - j):
                pts.append([i, j, k])
    if len(pts) > 1:
        pts = pts[1:]
    np_pts = np.stack(pts)
    coeff_mat[np_pts[:, 0], np_pts[:, 1], np_pts[:, 2]] = coeff
    return np.polynomial.legendre.leggrid3d(coords[0], coords[1], coords[2], coeff_mat)


def n_bias_coeff(rank=3, degree=3):
    return int(np.prod([(degree + k) / k for k in range(1, rank + 1)]))


def apply_bias_field(img, coeff, degree=3):
    """RandBiasField.__call__: the same field for every channel, img * exp(field) in float64, cast back to float32."""
    img = np.asarray(img, dtype=np.float32)
    field = bias_field(img.shape[1:], coeff, degree)
    fields = np.stack([field] * img.shape[0], axis=0)
    return (img * np.exp(fields)).astype(np.float32)


def add_noise(img, noise):
    """RandGaussianNoise.__call__: img + noise (float32)."""
    return (torch.as_tensor(img, dtype=torch.float32) + torch.as_tensor(noise, dtype=torch.float32)).numpy()


def adjust_contrast(img, gamma):
    """AdjustContrast.__call__ (invert_image=False, retain_stats=False): float32 torch arithmetic."""
    img = torch.as_tensor(np.asarray(img, dtype=np.float32))
    epsilon = 1e-7
    img_min = img.min()
    img_range = img.max() - img_min
    ret = ((img - img_min) / float(img_range + epsilon)) ** gamma * img_range + img_min
    return ret.numpy()


def histogram_shift(img, reference_control_points, floating_control_points):
    """RandHistogramShift.__call__ + .interp on a torch float32 tensor."""
    img_t = torch.as_tensor(np.asarray(img, dtype=np.float32))
    img_min, img_max = img_t.min(), img_t.max()
    if img_min == img_max:
        return img_t.numpy()
    xp = torch.as_tensor(np.asarray(reference_control_points), dtype=torch.float32)
    yp = torch.as_tensor(np.asarray(floating_control_points), dtype=torch.float32)
    xp = xp * (img_max - img_min) + img_min
    fp = yp * (img_max - img_min) + img_min
    x = img_t
    m = (fp[1:] - fp[:-1]) / (xp[1:] - xp[:-1])
    b = fp[:-1] - (m * xp[:-1])
    indices = torch.searchsorted(xp.reshape(-1), x.reshape(-1)) - 1
    indices = torch.clip(indices, 0, len(m) - 1)
    f = (m[indices] * x.reshape(-1) + b[indices]).reshape(x.shape)
    f[x < xp[0]] = fp[0]
    f[x > xp[-1]] = fp[-1]
    return f.numpy()


def coarse_dropout(arr, hole_lo, hole_size, fill_value=0.0):
    """RandCoarseDropout._transform_holes (dropout_holes=True): every channel, boxes [lo, lo + size)."""
    out = np.array(arr, copy=True)
    for lo in hole_lo:
        sl = (slice(None),) + tuple(slice(int(a), int(a) + int(s)) for a, s in zip(lo, hole_size))
        out[sl] = fill_value
    return out


# ---------------------------------------------------------------------------------------------------------------------
# random draws (host side; numpy RandomState streams laid out like MONAI's Compose)


def transform_seeds(seed):
    """Compose.set_random_state(seed): one uint32 seed per randomizable transform, drawn in order."""
    r = np.random.RandomState(seed)
    return [int(r.randint(MAX_SEED, dtype="uint32")) for _ in range(5)]


class Streams:
    """Per dictionary transform: an outer RandomState (the *d wrapper's prob draw) and an inner one (the array
    transform, constructed with prob=1.0) — the *d wrappers' set_random_state seeds both with the same seed."""

    def __init__(self, seed=None):
        seeds = transform_seeds(seed) if seed is not None else [None] * 5
        self.outer = [np.random.RandomState(s) for s in seeds]
        self.inner = [np.random.RandomState(s) for s in seeds]


def draw_params(streams, img_shape, prob=0.3, degree=3, coeff_range=(0.0, 0.1), noise_mean=0.0, noise_std=0.01,
                gamma=(0.7, 1.5), num_control_points=5, holes=2, hole_size=(16, 16, 16)):
    """One sample's worth of draws.  img_shape = (C, D, H, W).  Returns a dict of plain numpy values / None."""
    p = {"bias_coeff": None, "noise": None, "gamma": None, "ref_cp": None, "flt_cp": None, "hole_lo": None,
         "hole_size": None}
    spatial = tuple(img_shape[1:])
    # RandBiasFieldd
    o, r = streams.outer[0], streams.inner[0]
    if o.rand() < prob:
        r.rand()                                                         # inner RandomizableTransform.randomize, prob 1.0
        p["bias_coeff"] = r.uniform(coeff_range[0], coeff_range[1], n_bias_coeff(len(spatial), degree)).tolist()
    # RandGaussianNoised
    o, r = streams.outer[1], streams.inner[1]
    if o.rand() < prob:
        r.rand()
        std = r.uniform(0, noise_std)                                    # sample_std=True
        p["noise"] = r.normal(noise_mean, std, size=tuple(img_shape)).astype(np.float32)
    # RandAdjustContrastd
    o, r = streams.outer[2], streams.inner[2]
    if o.rand() < prob:
        r.rand()
        p["gamma"] = float(r.uniform(low=gamma[0], high=gamma[1]))
    # RandHistogramShiftd
    o, r = streams.outer[3], streams.inner[3]
    if o.rand() < prob:
        r.rand()
        n = int(r.randint(num_control_points, num_control_points + 1))
        ref = np.linspace(0, 1, n)
        flt = np.copy(ref)
        for i in range(1, n - 1):
            flt[i] = r.uniform(flt[i - 1], flt[i + 1])
        p["ref_cp"], p["flt_cp"] = ref, flt
    # RandCoarseDropoutd (image and label share the holes)
    o, r = streams.outer[4], streams.inner[4]
    if o.rand() < prob:
        r.rand()
        size = tuple(min(int(s), int(d)) for s, d in zip(hole_size, spatial))        # get_valid_patch_size
        los = []
        for _ in range(holes):
            los.append(tuple(int(r.randint(0, d - s + 1)) if d > s else 0 for d, s in zip(spatial, size)))
        p["hole_lo"], p["hole_size"] = los, size
    return p


def apply(image, label, p):
    """The chain on one (C, D, H, W) float image and its label with already-drawn parameters."""
    img = np.asarray(image, dtype=np.float32)
    if p["bias_coeff"] is not None:
        img = apply_bias_field(img, p["bias_coeff"])
    if p["noise"] is not None:
        img = add_noise(img, p["noise"])
    if p["gamma"] is not None:
        img = adjust_contrast(img, p["gamma"])
    if p["ref_cp"] is not None:
        img = histogram_shift(img, p["ref_cp"], p["flt_cp"])
    if p["hole_lo"] is not None:
        img = coarse_dropout(img, p["hole_lo"], p["hole_size"], 0.0)
        label = coarse_dropout(label, p["hole_lo"], p["hole_size"], 0)
    return img, label
