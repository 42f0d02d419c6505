"""pytorch_nested-unet_amd — MI355X-native UNet++ train/val hot path.

The directory name carries a hyphen (it mirrors the reference repository's name), so
import it with importlib or through the `nunet_amd` alias module at the repo root:

    import nunet_amd
    model = nunet_amd.archs.NestedUNet(1, 3, False, dtype='bf16').cuda()
"""
from . import _lib, archs, dataset, engine, losses, metrics, parallel, synth, utils  # noqa: F401
from .archs import NestedUNet, UNet  # noqa: F401
from .losses import BCEDiceLoss  # noqa: F401
from .metrics import iou_score  # noqa: F401
