"""Drop-in mirror of the reference's models/unet.py (DoubleConv :6-22, UNet3D :24-90) on the MI355X HIP path.

Same constructor signatures, attribute names (.encoder/.pool/.bottleneck/.upconvs/.decoder/.final_conv), the
same 136-entry state_dict (keys, shapes, dtypes) and the same seeded initialisation; forward() returns fp32
logits (N,out,D,H,W).  All arithmetic runs in hand-written HIP kernels via csrc/plan.hip.
"""
import torch.nn as nn

from . import engine
from . import nn as mnn


class DoubleConv(nn.Module):
    """[Conv3d(k3,p1) => BatchNorm3d => ReLU => Dropout3d] x 2   (reference models/unet.py:6-22)"""

    def __init__(self, in_channels, out_channels, dropout_rate=0.1):
        super().__init__()
        self.double_conv = nn.Sequential(
            mnn.Conv3d(in_channels, out_channels, kernel_size=3, padding=1),
            mnn.BatchNorm3d(out_channels),
            mnn.ReLU(inplace=True),
            mnn.Dropout3d(p=dropout_rate),
            mnn.Conv3d(out_channels, out_channels, kernel_size=3, padding=1),
            mnn.BatchNorm3d(out_channels),
            mnn.ReLU(inplace=True),
            mnn.Dropout3d(p=dropout_rate),
        )

    def forward(self, x):
        from . import ops
        return ops.double_conv_forward(self, x)


class UNet3D(nn.Module):
    """3D U-Net (reference models/unet.py:24-90).  features must double per level (as the reference's
    ConvTranspose3d(feature*2, feature) wiring requires)."""

    def __init__(self, in_channels=1, out_channels=1, features=[16, 32, 64, 128], output_activation=None,
                 dropout_rate=0.1):
        super().__init__()
        self.encoder = nn.ModuleList()
        self.pool = mnn.MaxPool3d(kernel_size=2, stride=2)
        self.output_activation = output_activation
        self.dropout_rate = dropout_rate
        for feature in features:
            self.encoder.append(DoubleConv(in_channels, feature, dropout_rate))
            in_channels = feature
        self.bottleneck = DoubleConv(features[-1], features[-1] * 2, dropout_rate)
        self.upconvs = nn.ModuleList()
        self.decoder = nn.ModuleList()
        for feature in reversed(features):
            self.upconvs.append(mnn.ConvTranspose3d(feature * 2, feature, kernel_size=2, stride=2))
            self.decoder.append(DoubleConv(feature * 2, feature, dropout_rate))
        self.final_conv = mnn.Conv3d(features[0], out_channels, kernel_size=1)
        self.compute_dtype = None   # None: follow torch.autocast / engine.set_compute_dtype

    def forward(self, x):
        logits, _ = engine.unet_forward(self, x, want_gap=False)
        return logits
