"""Time the input-pipeline kernels on one 192^3 volume (the reference's real volume size): python tools/time_augment.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_segmentation_project_amd import augment, preprocess  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    S = 192
    x = torch.rand(1, S, S, S, device="cuda")
    lab = torch.randint(0, 4, (1, S, S, S), device="cuda")
    nbytes = x.numel() * 4
    p = augment.AugmentDraw()
    p.bias_coeff = np.random.RandomState(0).uniform(0, 0.1, 20).tolist()
    p.noise_seed, p.noise_std, p.gamma = 3, 0.01, 0.9
    p.ref_cp, p.flt_cp = np.linspace(0, 1, 5), np.array([0, 0.2, 0.55, 0.8, 1.0])
    p.hole_lo, p.hole_size = [(5, 6, 7), (100, 90, 80)], (16, 16, 16)
    out = torch.empty_like(x)
    rows = []
    t = timed(lambda: augment.apply_image(x, p, out=out))
    rows.append(("augment: all five transforms (3 R + 3 W passes)", t, 6 * nbytes))
    for name, keys in (("bias + noise", ("bias_coeff", "noise_seed", "noise_std")), ("bias only", ("bias_coeff",)),
                       ("device noise only", ("noise_seed", "noise_std")), ("contrast (min/max pass + apply)", ("gamma",)),
                       ("histogram shift (min/max pass + apply)", ("ref_cp", "flt_cp"))):
        q = augment.AugmentDraw()
        for k in keys:
            setattr(q, k, getattr(p, k))
        passes = 2 if ("bias_coeff" in keys or "noise_seed" in keys) else 3
        rows.append((f"augment: {name}", timed(lambda: augment.apply_image(x, q, out=out)), passes * nbytes))
    rows.append(("augment: label holes", timed(lambda: augment.apply_label(lab, p)), 2 * 16 ** 3 * 8))
    raw = torch.randn(S, S, S, device="cuda") * 300
    rows.append(("preprocess_ct", timed(lambda: preprocess.preprocess_ct(raw)), 2 * nbytes))
    rows.append(("preprocess_mri (2 sums + 4 radix passes + apply)", timed(lambda: preprocess.preprocess_mri(raw)), 8 * nbytes))
    rows.append(("remap_labels (AMOS)", timed(lambda: preprocess.remap_labels(lab, "amos_ct")), 2 * lab.numel() * 8))
    for name, us, b in rows:
        print(f"{name:58s} {us:9.1f} us  {b / us / 1e6:8.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
