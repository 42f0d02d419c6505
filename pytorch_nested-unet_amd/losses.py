"""Drop-in `losses` module (reference registry: `losses.__dict__[config['loss']]()`,
trains.py:213). BCEDiceLoss follows reference losses.py:103-117."""
import torch
from torch import nn

from . import _lib as L

__all__ = ['BCEDiceLoss', 'LovaszHingeLoss']


class _BCEDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        L.require_gpu_tensor(logits, torch.float32, "logits")
        L.require_gpu_tensor(target, torch.float32, "target")
        if logits.shape != target.shape:
            raise L.NunetError("BCEDiceLoss: logits %s vs target %s" % (tuple(logits.shape), tuple(target.shape)))
        n = logits.size(0)
        per = logits.numel() // n
        lib = L.lib()
        ws = torch.empty((lib.nunet_bce_dice_ws_bytes(n) + 3) // 4, dtype=torch.float32, device=logits.device)   # the library states the size ...
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        L.check(lib.nunet_bce_dice_fwd(L.ptr(logits), L.ptr(target), n, per, L.ptr(ws), L.nbytes(ws), L.ptr(loss), L.stream()),   # ... and checks it
                "nunet_bce_dice_fwd")
        ctx.save_for_backward(logits, target, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, target, ws = ctx.saved_tensors
        n = logits.size(0)
        per = logits.numel() // n
        g = g.contiguous().float()
        dx = torch.empty_like(logits)
        L.check(L.lib().nunet_bce_dice_bwd(L.ptr(logits), L.ptr(target), n, per, L.ptr(ws), L.nbytes(ws), L.ptr(g), L.ptr(dx),
                                           L.stream()), "nunet_bce_dice_bwd")
        return dx, None


class BCEDiceLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, input, target):
        return _BCEDiceFn.apply(input.contiguous(), target.contiguous())


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        L.require_gpu_tensor(logits, torch.float32, "logits")
        L.require_gpu_tensor(target, torch.float32, "target")
        if logits.shape != target.shape or logits.dim() != 3:
            raise L.NunetError("LovaszHingeLoss: expected [N,H,W] logits and target, got %s / %s"
                               % (tuple(logits.shape), tuple(target.shape)))
        n = logits.size(0)
        per = logits.numel() // n
        ws = torch.empty(L.lib().nunet_lovasz_ws_bytes(n, per), dtype=torch.uint8, device=logits.device)
        unit = torch.empty_like(logits)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        L.check(L.lib().nunet_lovasz_hinge_fwd(L.ptr(logits), L.ptr(target), n, per, L.ptr(ws), L.nbytes(ws), L.ptr(unit), L.ptr(loss),
                                               L.stream()), "nunet_lovasz_hinge_fwd")
        ctx.save_for_backward(unit)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (unit,) = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(unit)
        L.check(L.lib().nunet_lovasz_hinge_bwd(L.ptr(unit), L.ptr(g), unit.numel(), L.ptr(dx), L.stream()),
                "nunet_lovasz_hinge_bwd")
        return dx, None


class LovaszHingeLoss(nn.Module):
    """reference losses.py:120-129 (lovasz_hinge(per_image=True) of the channel-squeezed tensors)."""

    def __init__(self):
        super().__init__()

    def forward(self, input, target):
        input = input.squeeze(1)
        target = target.squeeze(1)
        return _LovaszFn.apply(input.contiguous(), target.contiguous())
