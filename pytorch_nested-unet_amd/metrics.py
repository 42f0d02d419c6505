"""Drop-in `metrics` (reference metrics.py:6-18): global batch IoU."""
import torch

from . import _lib as L


_IOU_THR = None


def iou_logit_threshold():
    """The smallest fp32 logit x with `torch.sigmoid(x) > 0.5` as the REFERENCE evaluates it (metrics.py:10-12: fp32 sigmoid
    on the host, then `> 0.5`). It is not 0: in fp32 sigmoid(x) rounds to exactly 0.5 for 0 < x <~ 6e-8, and those pixels
    are background in the reference's integer counts. Found once by bisection over fp32 bit patterns with the host's
    torch.sigmoid (padded to a full vector: torch's scalar tail can differ from its vectorised path by an ulp)."""
    global _IOU_THR
    if _IOU_THR is None:
        import numpy as np

        def fg(bits):
            xp = np.zeros(256, np.float32)
            xp[0] = np.array([bits], np.uint32).view(np.float32)[0]
            return bool(torch.sigmoid(torch.from_numpy(xp)).numpy()[0] > np.float32(0.5))
        lo, hi = 0, int(np.array([1.0], np.float32).view(np.uint32)[0])     # sigmoid(+0) = 0.5: not foreground; sigmoid(1) is
        assert not fg(lo) and fg(hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if fg(mid):
                hi = mid
            else:
                lo = mid
        _IOU_THR = float(np.array([hi], np.uint32).view(np.float32)[0])
    return _IOU_THR


def iou_counts(output, target, counts=None):
    """Device-side (intersection, union) counts as a uint64[2] tensor; no host sync.
    `counts` may be passed to accumulate over several batches."""
    L.require_gpu_tensor(output, torch.float32, "output")
    L.require_gpu_tensor(target, torch.float32, "target")
    if counts is None:
        counts = torch.zeros(2, dtype=torch.int64, device=output.device)
    L.check(L.lib().nunet_iou_counts(L.ptr(output), L.ptr(target), output.numel(), iou_logit_threshold(), L.ptr(counts), L.stream()),
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


_U8_THRESHOLDS = {}


def sigmoid_u8_thresholds(device):
    """thr[k-1], k = 1..255: the smallest fp32 logit x with `(torch.sigmoid(x) * 255).astype('uint8') >= k`, found by
    bisection over fp32 bit patterns with the host's torch.sigmoid - the function the reference's export applies
    (val.py:100-105). The device kernel counts thresholds <= x, so its bytes equal the reference's exactly."""
    import numpy as np
    import torch
    key = str(device)
    if key not in _U8_THRESHOLDS:
        def byte(x):        # the reference expression, float32 throughout; padded to 256 so that every element takes
            xp = np.zeros(256, np.float32); xp[:x.size] = x      # torch's vectorised path (its scalar tail can differ by an ulp)
            return (torch.sigmoid(torch.from_numpy(xp)).numpy() * np.float32(255)).astype("uint8").astype(np.int64)[:x.size]

        def key_of(x):      # order-preserving int64 key of a float32
            b = x.view(np.int32).astype(np.int64)
            return np.where(b >= 0, b, -(b & 0x7fffffff) - 1)

        def from_key(k):
            b = np.where(k >= 0, k, (-(k + 1)) | 0x80000000).astype(np.uint32)
            return b.view(np.float32)
        ks = np.arange(1, 256, dtype=np.int64)
        lo = np.full(255, key_of(np.array([-200.0], np.float32))[0], np.int64)   # byte(lo) = 0 < k
        hi = np.full(255, key_of(np.array([200.0], np.float32))[0], np.int64)    # byte(hi) = 255 >= k
        assert byte(from_key(lo))[0] == 0 and byte(from_key(hi))[0] == 255
        while np.any(hi - lo > 1):
            mid = (lo + hi) // 2
            ge = byte(from_key(mid)) >= ks
            hi = np.where(ge, mid, hi)
            lo = np.where(ge, lo, mid)
        _U8_THRESHOLDS[key] = torch.from_numpy(from_key(hi).copy()).to(device)
    return _U8_THRESHOLDS[key]


def sigmoid_masks_u8(output):
    """uint8 masks of the evaluation driver, `(sigmoid(output) * 255).astype('uint8')` (reference val.py:100-105),
    computed on the device, byte-exact against the host expression: [N, K, H, W] fp32 logits -> [N, K, H, W] uint8."""
    import torch
    from . import _lib as L
    out = output.detach().contiguous()
    if out.dtype != torch.float32 or not out.is_cuda:
        raise L.NunetError("sigmoid_masks_u8: CUDA fp32 logits expected")
    m = torch.empty(out.shape, dtype=torch.uint8, device=out.device)
    thr = sigmoid_u8_thresholds(out.device)
    L.check(L.lib().nunet_sigmoid_u8(L.ptr(out), L.ptr(thr), L.ptr(m), out.numel(), L.stream()), "sigmoid_u8")
    return m
