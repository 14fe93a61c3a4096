"""Parameter-holding layer shells with the reference's names, init and state_dict layout.

They subclass the torch.nn layers the reference instantiates (models/unet.py:11-18,40,56-58,62) so that
construction order, default initialisation (=> identical weights under the same torch.manual_seed) and
state_dict keys/shapes/dtypes are exactly the reference's.  They never compute: the arithmetic of a whole
DoubleConv / UNet3D runs in the HIP plan (engine.py).  Calling an inner layer on its own is not part of the
hot path and fails loudly instead of silently running a non-HIP kernel.
"""
import torch.nn as nn

from ._lib import Mi3dError


def _standalone(self, *a, **k):
    raise Mi3dError(f"{type(self).__name__} is a parameter shell of the fused MI355X path; call the enclosing "
                    "DoubleConv / UNet3D (or DomainDiscriminator) instead")


class Conv3d(nn.Conv3d):
    forward = _standalone


class BatchNorm3d(nn.BatchNorm3d):
    forward = _standalone


class ReLU(nn.ReLU):
    forward = _standalone


class Dropout3d(nn.Dropout3d):
    forward = _standalone


class MaxPool3d(nn.MaxPool3d):
    forward = _standalone


class ConvTranspose3d(nn.ConvTranspose3d):
    forward = _standalone


class Linear(nn.Linear):
    forward = _standalone


class Dropout(nn.Dropout):
    forward = _standalone
