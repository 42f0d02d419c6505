"""Device-side counterpart of the reference sample pipeline (dataset.py:66-74: Normalize -> /255 ->
HWC->CHW; trains.py:258-259: RandomRotate90, Flip) for uint8 batches already decoded on the host.
Only uint8 crosses PCIe (4x fewer bytes than the reference's float tensors)."""
import numpy as np
import torch

from . import _lib as L

MEAN = (0.485, 0.456, 0.406)      # albumentations Normalize() defaults (trains.py:266)
STD = (0.229, 0.224, 0.225)


def preprocess_images(u8_nhwc, aug=None, mean=MEAN, std=STD):
    """uint8 [N,H,W,C] (device) -> float32 [N,C,H,W]: ((u/255 - mean)/std)/255, optional per-sample
    geometric augmentation codes (int32 [N]: rot90 count | hflip<<2 | vflip<<3)."""
    L.require_gpu_tensor(u8_nhwc, torch.uint8, "images")
    n, h, w, c = u8_nhwc.shape
    dev = u8_nhwc.device
    m = torch.tensor(np.resize(np.asarray(mean, np.float32), c), device=dev)
    s = torch.tensor(np.resize(np.asarray(std, np.float32), c), device=dev)
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    if aug is not None:
        L.require_gpu_tensor(aug, torch.int32, "aug")
        if h != w and bool((aug & 1).any()):
            raise L.NunetError("rot90 by an odd count needs square images")
    L.check(L.lib().nunet_preprocess_u8(L.ptr(u8_nhwc), n, h, w, c, L.ptr(m), L.ptr(s), L.ptr(aug), 1.0 / 255.0,
                                        L.ptr(out), L.stream()), "nunet_preprocess_u8")
    return out


def preprocess_masks(u8_nhwc, aug=None):
    """uint8 {0,255} [N,H,W,K] -> float32 [N,K,H,W] in {0,1} (dataset.py:73), same augmentation codes."""
    L.require_gpu_tensor(u8_nhwc, torch.uint8, "masks")
    n, h, w, c = u8_nhwc.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=u8_nhwc.device)
    L.check(L.lib().nunet_preprocess_u8(L.ptr(u8_nhwc), n, h, w, c, None, None, L.ptr(aug), 1.0, L.ptr(out), L.stream()),
            "nunet_preprocess_u8")
    return out


def draw_augmentation(n, generator=None, device="cuda"):
    """Per-sample codes for RandomRotate90() + Flip() (trains.py:258-259), drawn on the host."""
    k = torch.randint(0, 4, (n,), generator=generator)
    f = torch.randint(0, 4, (n,), generator=generator)       # bit0 hflip, bit1 vflip
    return (k | (f << 2)).to(torch.int32).to(device)
