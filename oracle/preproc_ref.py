"""numpy restatement of the per-volume arithmetic of the reference's input pipeline (TEST INFRASTRUCTURE).

Follows utils/dataloader.py (paths relative to /root/reference): :111-117 preprocess_ct, :128-144 preprocess_mri,
:153-159 modality dispatch, :162-181 label remaps.  Pinned against tests/golden/preproc.npz, which tools/gen_golden.py
produced by executing the reference's own CombinedDataset.__getitem__ on in-memory volumes.
"""
import numpy as np


def preprocess_ct(image, window_min=-160, window_max=240):
    image = np.clip(image, window_min, window_max)
    return (image - window_min) / (window_max - window_min)


def preprocess_mri(image):
    mean = np.mean(image)
    std = np.std(image)
    image = (image - mean) / (std + 1e-8)
    low, high = np.percentile(image, [1, 99])
    image = np.clip(image, low, high)
    return ((image - low) / (high - low + 1e-8)).astype(np.float32)


def preprocess(image, dataset_name):
    return preprocess_ct(image) if dataset_name.lower().endswith("_ct") else preprocess_mri(image)


def remap_labels(label, dataset_name):
    if dataset_name.startswith("amos"):
        out = np.zeros_like(label)
        for old, new in {0: 0, 1: 1, 2: 3, 3: 3, 6: 2}.items():
            out[label == old] = new
        return out
    if dataset_name.startswith("chaos"):
        out = np.zeros_like(label)
        for (lo, hi), new in {(55, 70): 2, (110, 135): 3, (175, 200): 3, (240, 255): 1}.items():
            out[(label >= lo) & (label <= hi)] = new
        return out
    return label
