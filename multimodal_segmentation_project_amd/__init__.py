"""MI355X-native (gfx950, hand-written HIP) 3D U-Net training hot path — drop-in for the reference's
models/unet.py, models/unet_dann.py, utils/metrics.py and the DANN pieces of train_dann.py."""
from . import checkpoint, engine, metrics, preprocess, unet, unet_dann, dann  # noqa: F401
from .engine import set_compute_dtype  # noqa: F401
from .unet import DoubleConv, UNet3D  # noqa: F401
from .dann import DomainDiscriminator, GradientReversal, grad_reverse  # noqa: F401
from .metrics import (calculate_accuracy, calculate_dice, calculate_iou, combined_ce_tversky_loss,  # noqa: F401
                      combined_loss, distillation_loss, get_loss_fn, per_class_dice_iou, tversky_loss)

__version__ = "0.1.0"
