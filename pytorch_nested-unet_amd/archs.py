"""Drop-in `archs` module for the UNet++ path (reference registry-by-name:
`archs.__dict__[config['arch']](num_classes, input_channels, deep_supervision)`,
trains.py:219-221, val.py:46-48).

Same constructor signatures, attribute names, parameter registration order,
default initialisation and state_dict keys as reference finished/archs1.py:14-143
(= archs_backup.py:24-152). The compute is libnunet's HIP kernels; the nn.Conv2d /
nn.BatchNorm2d children below are parameter containers only and are never called.
"""
import os

import torch
from torch import nn

from . import _lib as L
from .engine import Engine, _NetFn

__all__ = ['UNet', 'NestedUNet']


class VGGBlock(nn.Module):
    """conv3x3 -> BN -> ReLU, twice (reference finished/archs1.py:14-32)."""

    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_channels, middle_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(middle_channels)
        self.conv2 = nn.Conv2d(middle_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        raise L.NunetError("VGGBlock is a parameter container; the whole network runs as one HIP plan "
                           "(call the NestedUNet/UNet module)")


class _PlanNet(nn.Module):
    _unet = False

    def __init__(self, num_classes, input_channels, deep_supervision, kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.input_channels = input_channels
        dt = kwargs.get('dtype', os.environ.get('NUNET_DTYPE', 'fp32'))
        if dt not in L.DTYPES:
            raise ValueError("dtype must be one of fp32/bf16/fp16, got %r" % (dt,))
        self.compute_dtype = L.DTYPES[dt]
        self._engine = None

    # A device move (.cuda() / .to(dev)) swaps the parameters' storage: engine() notices (device change or the
    # parameters no longer being views of its arenas, Engine.intact()) and re-homes them at the next forward. A call
    # that moves nothing (.cuda() on a module already there, .float()) leaves the arenas - and everything that holds
    # pointers into them: a TrainStep, its captured hipGraphs - untouched.

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise L.NunetError("NestedUNet/UNet run on the MI355X only: move the module with .cuda() first "
                               "(no CPU fallback; the CPU oracle lives under oracle/ for tests)")
        if self._engine is None or self._engine.device != dev or not self._engine.intact():
            self._engine = Engine(self, dev)
        return self._engine

    def plan_for(self, input):
        if input.dim() != 4 or input.size(1) != self.input_channels:
            raise L.NunetError("expected input [N,%d,H,W], got %s" % (self.input_channels, tuple(input.shape)))
        n, _, h, w = input.shape
        return self.engine().plan(n, h, w, self.input_channels, self.num_classes,
                                  getattr(self, 'deep_supervision', False), self.compute_dtype, self._unet)

    def forward(self, input):
        eng = self.engine()
        if input.requires_grad:
            raise L.NunetError("gradient w.r.t. the input image is not part of this path")
        input = input.contiguous()
        if input.dtype != torch.float32:
            input = input.float()
        pl = self.plan_for(input)
        training = self.training
        anchor = None
        if torch.is_grad_enabled():
            for p in eng.module_params:
                if p.requires_grad:
                    anchor = p
                    break
        if anchor is not None and training:
            out = _NetFn.apply(anchor, input, self, pl, True)
        else:
            out = eng.forward(pl, input, training)
        if getattr(self, 'deep_supervision', False) and not self._unet:
            return [out[k] for k in range(out.size(0))]
        return out[0]


class UNet(_PlanNet):
    """reference finished/archs1.py:35-71."""
    _unet = True

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__(num_classes, input_channels, deep_supervision, kwargs)
        nb_filter = [32, 64, 128, 256, 512]
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv0_0 = VGGBlock(input_channels, nb_filter[0], nb_filter[0])
        for i in range(1, 5):
            setattr(self, 'conv%d_0' % i, VGGBlock(nb_filter[i - 1], nb_filter[i], nb_filter[i]))
        for i in (3, 2, 1, 0):
            setattr(self, 'conv%d_%d' % (i, 4 - i), VGGBlock(nb_filter[i] + nb_filter[i + 1], nb_filter[i], nb_filter[i]))
        self.final = nn.Conv2d(nb_filter[0], num_classes, kernel_size=1)


class NestedUNet(_PlanNet):
    """reference finished/archs1.py:74-143."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__(num_classes, input_channels, deep_supervision, kwargs)
        nb_filter = [32, 64, 128, 256, 512]
        self.deep_supervision = deep_supervision
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        # registration order = column by column of the x_{i,j} grid (archs1.py:85-103)
        for j in range(5):
            for i in range(5 - j):
                if j == 0:
                    cin = input_channels if i == 0 else nb_filter[i - 1]
                else:
                    cin = nb_filter[i] * j + nb_filter[i + 1]
                setattr(self, 'conv%d_%d' % (i, j), VGGBlock(cin, nb_filter[i], nb_filter[i]))
        if self.deep_supervision:
            for k in (1, 2, 3, 4):
                setattr(self, 'final%d' % k, nn.Conv2d(nb_filter[0], num_classes, kernel_size=1))
        else:
            self.final = nn.Conv2d(nb_filter[0], num_classes, kernel_size=1)
