"""Drop-in `metrics` (reference metrics.py:6-18): global batch IoU."""
import torch

from . import _lib as L


def iou_counts(output, target, counts=None):
    """Device-side (intersection, union) counts as a uint64[2] tensor; no host sync.
    `counts` may be passed to accumulate over several batches."""
    L.require_gpu_tensor(output, torch.float32, "output")
    L.require_gpu_tensor(target, torch.float32, "target")
    if counts is None:
        counts = torch.zeros(2, dtype=torch.int64, device=output.device)
    L.check(L.lib().nunet_iou_counts(L.ptr(output), L.ptr(target), output.numel(), L.ptr(counts), L.stream()),
            "nunet_iou_counts")
    return counts


def iou_from_counts(counts):
    smooth = 1e-5
    i, u = (int(v) for v in counts.tolist())
    return (i + smooth) / (u + smooth)


def iou_score(output, target):
    """Same value and return type (python float) as reference metrics.py:6-18;
    like the reference it synchronises with the device."""
    return iou_from_counts(iou_counts(output.detach().contiguous(), target.contiguous()))


def sigmoid_masks_u8(output):
    """uint8 masks of the evaluation driver, `(sigmoid(output) * 255).astype('uint8')` (reference val.py:100-105),
    computed on the device: [N, K, H, W] fp32 logits -> [N, K, H, W] uint8."""
    import torch
    from . import _lib as L
    out = output.detach().contiguous()
    if out.dtype != torch.float32 or not out.is_cuda:
        raise L.NunetError("sigmoid_masks_u8: CUDA fp32 logits expected")
    m = torch.empty(out.shape, dtype=torch.uint8, device=out.device)
    L.check(L.lib().nunet_sigmoid_u8(L.ptr(out), L.ptr(m), out.numel(), L.stream()), "sigmoid_u8")
    return m
