"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

CPU restatement (stock PyTorch ops + numpy, float32 or float64) of the
reference hot path of husheng876/pytorch_nested-unet:

  VGGBlock / NestedUNet      reference finished/archs1.py:14-32, 74-143
                             (= archs_backup.py:24-42, 84-152)
  BCEDiceLoss                reference losses.py:103-117
  iou_score                  reference metrics.py:6-18
  AverageMeter               reference utils.py:17-33
  train-step semantics       reference trains.py:106-147 (DS loss mean :118-124)
  SGD / cosine schedule      reference trains.py:229-239,323-324

Parity pin: this restatement is checked against golden vectors captured from
the *imported reference itself* (tests/golden/make_golden.py, run in the
authoring container where /root/reference exists); see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

NB_FILTER = (32, 64, 128, 256, 512)          # finished/archs1.py:78
BN_EPS = 1e-5                                # nn.BatchNorm2d default (archs1.py:19)
BN_MOMENTUM = 0.1


def _nodes():
    # execution order of finished/archs1.py:114-131: anti-diagonals of the grid
    return [(s - j, j) for s in range(5) for j in range(s + 1)]


def _to_t(v, dtype):
    t = torch.as_tensor(np.asarray(v))
    return t.to(dtype) if t.is_floating_point() else t


class _RoundST(torch.autograd.Function):
    """Storage-rounding emulation: the value is rounded to `dt` on the way forward and (round_grad) the gradient on the
    way back - what a tensor (and the gradient tensor of the same shape) suffers when the HIP path keeps it in 16-bit
    storage between two kernels. The arithmetic around it stays in the oracle's own precision."""

    @staticmethod
    def forward(ctx, x, dt, round_grad):
        ctx.dt, ctx.rg = dt, round_grad
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dt).to(g.dtype) if ctx.rg else g), None, None


class OracleNet:
    """Functional NestedUNet over a reference-format state dict.

    `state` maps reference state_dict names -> tensors/arrays. Parameters are
    held as leaf tensors with requires_grad so that torch autograd provides the
    reference backward (the reference itself relies on autograd, trains.py:132).
    """

    def __init__(self, state, num_classes=1, input_channels=3, deep_supervision=False,
                 dtype=torch.float32, storage=None):
        # storage (torch.bfloat16 / torch.float16 / None): emulate the HIP path's 16-bit STORAGE points (packed conv
        # weights, raw conv outputs without bias, activations, upsampled / pooled tensors, and the gradients of all of
        # these) inside an fp32/fp64 evaluation, so that reduced-precision runs can be checked against "the same
        # roundings, exact arithmetic" instead of only against the unrounded network
        self.storage = storage
        self.ncls = num_classes
        self.cin = input_channels
        self.ds = bool(deep_supervision)
        self.dtype = dtype
        self.training = True
        self.params = OrderedDict()
        self.buffers = OrderedDict()
        for k, v in state.items():
            t = _to_t(v, dtype).clone()
            if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
                self.buffers[k] = t
            else:
                self.params[k] = t.requires_grad_(True)

    # -- nn.Module-like surface used by the reference drivers -----------------
    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def parameters(self):
        return list(self.params.values())

    def state_dict(self):
        out = OrderedDict()
        for k, v in self.params.items():
            out[k] = v.detach()
        out.update(self.buffers)
        return out

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None

    # -- VGGBlock.forward, finished/archs1.py:23-32 ---------------------------
    def _conv_bn_relu(self, x, prefix, k):
        w = self.params["%sconv%d.weight" % (prefix, k)]
        b = self.params["%sconv%d.bias" % (prefix, k)]
        if self.storage is None:
            y = F.conv2d(x, w, b, padding=1)                              # archs1.py:18,20
        else:   # packed weights are rounded, the conv output is stored without its bias (the BatchNorm absorbs it)
            y = self._r(F.conv2d(x, _RoundST.apply(w, self.storage, False), None, padding=1)) + b.view(1, -1, 1, 1)
        g = self.params["%sbn%d.weight" % (prefix, k)]
        be = self.params["%sbn%d.bias" % (prefix, k)]
        rm = self.buffers["%sbn%d.running_mean" % (prefix, k)]
        rv = self.buffers["%sbn%d.running_var" % (prefix, k)]
        if self.training:
            self.buffers["%sbn%d.num_batches_tracked" % (prefix, k)] += 1
        y = F.batch_norm(y, rm, rv, g, be, self.training, BN_MOMENTUM, BN_EPS)  # archs1.py:19,21
        return self._r(F.relu(y))                                         # archs1.py:17

    def _r(self, t):
        return t if self.storage is None else _RoundST.apply(t, self.storage, True)

    def _block(self, x, i, j):
        p = "conv%d_%d." % (i, j)
        return self._conv_bn_relu(self._conv_bn_relu(x, p, 1), p, 2)

    # -- NestedUNet.forward, finished/archs1.py:113-143 -----------------------
    def __call__(self, inp):
        x = {}
        inp = inp.to(self.dtype)
        if self.storage is not None:
            inp = inp.to(self.storage).to(self.dtype)
        for (i, j) in _nodes():
            if j == 0:
                src = inp if i == 0 else self._r(F.max_pool2d(x[(i - 1, 0)], 2, 2))      # archs1.py:82
            else:
                up = self._r(F.interpolate(x[(i + 1, j - 1)], scale_factor=2, mode="bilinear",
                                           align_corners=True))                # archs1.py:83
                src = torch.cat([x[(i, k)] for k in range(j)] + [up], 1)       # archs1.py:116-131
            x[(i, j)] = self._block(src, i, j)
        self.features = x
        if self.ds:                                                          # archs1.py:133-138
            return [F.conv2d(x[(0, k)], self.params["final%d.weight" % k],
                             self.params["final%d.bias" % k]) for k in (1, 2, 3, 4)]
        return F.conv2d(x[(0, 4)], self.params["final.weight"], self.params["final.bias"])


def bce_dice_loss(logits, target):
    """losses.py:107-117."""
    bce = F.binary_cross_entropy_with_logits(logits, target)
    smooth = 1e-5
    n = target.size(0)
    p = torch.sigmoid(logits).reshape(n, -1)
    t = target.reshape(n, -1)
    inter = (p * t).sum(1)
    dice = (2.0 * inter + smooth) / (p.sum(1) + t.sum(1) + smooth)
    return 0.5 * bce + (1 - dice.sum() / n)


def bce_dice_loss_np(logits, target):
    """float64 numpy restatement of losses.py:107-117 (independent of torch)."""
    x = np.asarray(logits, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    bce = np.mean(np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x))))
    n = x.shape[0]
    p = (1.0 / (1.0 + np.exp(-x))).reshape(n, -1)
    tt = t.reshape(n, -1)
    dice = (2.0 * (p * tt).sum(1) + 1e-5) / (p.sum(1) + tt.sum(1) + 1e-5)
    return 0.5 * bce + 1.0 - dice.sum() / n


def iou_score(logits, target):
    """metrics.py:6-18: global (whole-batch) IoU of sigmoid(x)>0.5 vs t>0.5."""
    inter, union = iou_counts(logits, target)
    return (inter + 1e-5) / (union + 1e-5)


def iou_counts(logits, target):
    """The reference's own expression, in its own precision: `torch.sigmoid(output)` on the fp32 tensor, then `> 0.5`
    on the host (metrics.py:10-12). NOT `x > 0`: in fp32 the sigmoid of a logit in (0, ~9e-8) rounds to exactly 0.5."""
    x = logits.detach().cpu() if torch.is_tensor(logits) else torch.from_numpy(np.asarray(logits))
    t = target.detach().cpu().numpy() if torch.is_tensor(target) else np.asarray(target)
    o_ = torch.sigmoid(x.float()).numpy() > 0.5
    t_ = t > 0.5
    return int((o_ & t_).sum()), int((o_ | t_).sum())


def criterion_ds(outputs, target):
    """trains.py:118-127: mean of the per-head losses under deep supervision."""
    if isinstance(outputs, (list, tuple)):
        loss = 0
        for o in outputs:
            loss = loss + bce_dice_loss(o, target)
        return loss / len(outputs), outputs[-1]
    return bce_dice_loss(outputs, target), outputs


class AverageMeter:
    """utils.py:17-33."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def cosine_lr(base_lr, min_lr, t, t_max):
    """Closed form of CosineAnnealingLR as configured at trains.py:237-239."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1.0 + math.cos(math.pi * t / t_max))


class SGD:
    """torch.optim.SGD semantics as configured at trains.py:229-231
    (momentum buffer initialised to the first gradient; wd added to the grad)."""

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=False):
        self.params = list(params)
        self.lr, self.mom, self.wd, self.nesterov = lr, momentum, weight_decay, nesterov
        self.bufs = [None] * len(self.params)

    @torch.no_grad()
    def step(self):
        for k, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad + self.wd * p if self.wd else p.grad.clone()
            if self.mom:
                if self.bufs[k] is None:
                    self.bufs[k] = g.clone()
                else:
                    self.bufs[k].mul_(self.mom).add_(g)
                g = g + self.mom * self.bufs[k] if self.nesterov else self.bufs[k]
            p.add_(g, alpha=-self.lr)


def train_step(net, opt, inp, target):
    """One iteration of the loop body at trains.py:113-135. Returns (loss, iou)."""
    net.train()
    out = net(inp)
    loss, last = criterion_ds(out, target)
    iou = iou_score(last, target)
    net.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.item()), float(iou)


# ---------------------------------------------------------------------------
# independent numpy restatements of the small ops
# ---------------------------------------------------------------------------

def maxpool2x2_np(x):
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def upsample2x_bilinear_ac_np(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)."""
    n, c, h, w = x.shape
    ho, wo = 2 * h, 2 * w

    def taps(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = np.arange(n_out, dtype=np.float64) * scale
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        f = src - i0
        return i0, i1, f

    y0, y1, fy = taps(h, ho)
    x0, x1, fx = taps(w, wo)
    xd = x.astype(np.float64)
    top = xd[:, :, y0][:, :, :, x0] * (1 - fx) + xd[:, :, y0][:, :, :, x1] * fx
    bot = xd[:, :, y1][:, :, :, x0] * (1 - fx) + xd[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None]


# ---------------------------------------------------------------------------
# Lovasz hinge (reference losses.py:49-61 lovasz_grad, :64-96 lovasz_hinge / _flat, :120-129 LovaszHingeLoss:
# per_image=True, mean over images, channel dim squeezed)
# ---------------------------------------------------------------------------

def lovasz_hinge(logits, labels):
    """logits, labels: torch [N, H, W] (any float dtype). Differentiable restatement with torch ops only."""
    losses = []
    for lg, lb in zip(logits, labels):
        lg, lb = lg.reshape(-1), lb.reshape(-1)
        signs = 2.0 * lb - 1.0
        errors = 1.0 - lg * signs
        errors_sorted, perm = torch.sort(errors, dim=0, descending=True)
        gt_sorted = lb[perm]
        gts = gt_sorted.sum()
        intersection = gts - gt_sorted.cumsum(0)
        union = gts + (1.0 - gt_sorted).cumsum(0)
        jaccard = 1.0 - intersection / union
        if gt_sorted.numel() > 1:
            jaccard = torch.cat([jaccard[:1], jaccard[1:] - jaccard[:-1]])
        losses.append(torch.dot(torch.relu(errors_sorted), jaccard.detach()))
    return sum(losses) / len(losses)
