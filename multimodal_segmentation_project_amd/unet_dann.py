"""Drop-in mirror of the reference's models/unet_dann.py: the same network and state_dict as unet.UNet3D, but
forward(x, return_features=False) ALWAYS returns a tuple (logits, gap) with
gap = mean over (D,H,W) of the bottleneck output, shape (N, 2*features[-1]), or None (models/unet_dann.py:65-98).
"""
from . import engine
from .unet import DoubleConv, UNet3D as _BaseUNet3D  # noqa: F401  (DoubleConv re-exported like the reference file)


class UNet3D(_BaseUNet3D):
    def forward(self, x, return_features=False):
        logits, gap = engine.unet_forward(self, x, want_gap=bool(return_features))
        return logits, gap
