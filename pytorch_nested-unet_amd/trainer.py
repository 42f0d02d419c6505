"""Fused training step: the loop body of reference trains.py:113-135
(forward, BCEDiceLoss (mean over heads under deep supervision, :118-124), IoU on
the last head, backward, SGD(momentum, wd) :229-231,133) enqueued as ONE hipGraph
replay per step, with no host synchronisation: loss and IoU counts stay on the
device and are read back only when the caller asks (AverageMeter semantics of
utils.py:17-33 are kept by `epoch_stats`).

Data parallel (new capability): one process per GPU, local
BatchNorm statistics (plain nn.BatchNorm2d semantics, archs1.py:19,21), one
gradient all-reduce per step over RCCL, buckets in gradient-ready order.
"""
import math

import torch
import torch.distributed as dist

from . import _lib as L
from .parallel import allreduce_flat_


class TrainStep:
    def __init__(self, model, batch_shape, lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=False,
                 use_graph=True, process_group=None):
        self.model = model
        self.eng = model.engine()
        dev = self.eng.device
        n, cin, h, w = batch_shape
        self.n, self.h, self.w = n, h, w
        self.ncls = model.num_classes
        x0 = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.pl = model.plan_for(x0)
        self.heads = self.pl.heads
        self.x = x0
        self.t = torch.zeros((n, self.ncls, h, w), dtype=torch.float32, device=dev)
        self.logits = torch.empty((self.heads, n, self.ncls, h, w), dtype=torch.float32, device=dev)
        self.dlogits = torch.empty_like(self.logits)
        self.per = self.ncls * h * w
        self.ws_stride = (3 * n + 1 + 3) // 4 * 4          # 16-byte aligned rows (zeroed by nunet_zero_async)
        self.loss_ws = torch.empty((self.heads, self.ws_stride), dtype=torch.float32, device=dev)
        self.loss_heads = torch.zeros(self.heads, dtype=torch.float32, device=dev)
        self.gscale = torch.full((1,), 1.0 / self.heads, dtype=torch.float32, device=dev)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.mom = torch.zeros_like(self.eng.flat_params)
        self.momentum, self.wd, self.nesterov = momentum, weight_decay, nesterov
        # running sums for the epoch meters: sum of per-step loss, steps, IoU numer/denom per step
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        self.iou_counts = torch.zeros(2, dtype=torch.int64, device=dev)
        self.iou_sum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.steps = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        self.use_graph = use_graph
        self.g_fb = None
        self.g_opt = None
        for p, off in zip(self.eng.module_params, self.eng.param_off):
            p.grad = self.eng.flat_grads[off:off + p.numel()].view(p.shape)

    # -- pieces -------------------------------------------------------------------
    def _fwd_bwd(self):
        lib, eng, pl = L.lib(), self.eng, self.pl
        st = L.stream()
        L.check(lib.nunet_plan_forward(pl.handle, L.ptr(eng.flat_params), L.ptr(eng.bnbuf), L.ptr(eng.nbt),
                                       L.ptr(self.x), L.ptr(pl.arena), L.ptr(self.logits), 1, st), "plan_forward")
        plane = self.n * self.per * 4
        for k in range(self.heads):
            L.check(lib.nunet_bce_dice_fwd(L.ptr(self.logits, k * plane), L.ptr(self.t), self.n, self.per,
                                           L.ptr(self.loss_ws, k * self.ws_stride * 4), L.ptr(self.loss_heads, 4 * k), st),
                    "bce_dice_fwd")
        self.iou_counts.zero_()
        L.check(lib.nunet_iou_counts(L.ptr(self.logits, (self.heads - 1) * plane), L.ptr(self.t), self.n * self.per,
                                     L.ptr(self.iou_counts), st), "iou_counts")
        for k in range(self.heads):
            L.check(lib.nunet_bce_dice_bwd(L.ptr(self.logits, k * plane), L.ptr(self.t), self.n, self.per,
                                           L.ptr(self.loss_ws, k * self.ws_stride * 4), L.ptr(self.gscale),
                                           L.ptr(self.dlogits, k * plane), st), "bce_dice_bwd")
        L.check(lib.nunet_plan_backward(pl.handle, L.ptr(eng.flat_params), L.ptr(self.dlogits), L.ptr(pl.arena),
                                        L.ptr(eng.flat_grads), 0, st), "plan_backward")
        # meters (device side): loss = mean over heads; iou = (I+eps)/(U+eps) of this step
        self.loss_sum += self.loss_heads.mean()
        c = self.iou_counts.double()
        self.iou_sum += (c[0] + 1e-5) / (c[1] + 1e-5)
        pl.trained_forward = True

    def _opt(self):
        eng = self.eng
        L.check(L.lib().nunet_sgd_step(L.ptr(eng.flat_params), L.ptr(eng.flat_grads), L.ptr(self.mom),
                                       eng.flat_params.numel(), L.ptr(self.lr), self.momentum, self.wd,
                                       1 if self.nesterov else 0, 0, 1.0 / self.world, L.stream()), "sgd_step")

    def _allreduce(self):
        if self.world > 1:
            allreduce_flat_(self.eng.flat_grads, self.pg)

    def capture(self, inp, target):
        """Capture forward+loss+backward(+SGD) into hipGraphs. Needs one REAL batch for the
        eager warm-up (an all-zero batch gives degenerate BatchNorm variances). Parameters,
        BN running statistics, momentum and meters are snapshotted before and restored after,
        so warm-up and capture leave no trace in the training trajectory."""
        if not self.use_graph:
            return
        eng = self.eng
        self.x.copy_(inp)
        self.t.copy_(target)
        snap = [t.clone() for t in (eng.flat_params, eng.bnbuf, eng.nbt, self.mom, self.loss_sum, self.iou_sum)]
        steps0 = self.steps
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd()
                self._allreduce()
                self._opt()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb):
            self._fwd_bwd()
            if self.world == 1:
                self._opt()
        if self.world > 1:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                self._opt()
        torch.cuda.synchronize()
        with torch.no_grad():
            for dst, src in zip((eng.flat_params, eng.bnbuf, eng.nbt, self.mom, self.loss_sum, self.iou_sum), snap):
                dst.copy_(src)
        self.steps = steps0

    def step(self, inp=None, target=None):
        """One training iteration. `inp`/`target` are device tensors (copied into the
        static graph inputs); None re-uses what is already staged."""
        if inp is not None:
            self.x.copy_(inp, non_blocking=True)
            self.t.copy_(target, non_blocking=True)
        if self.g_fb is not None:
            self.g_fb.replay()
            if self.world > 1:
                self._allreduce()
                self.g_opt.replay()
        else:
            self._fwd_bwd()
            self._allreduce()
            self._opt()
        self.steps += 1

    def set_lr(self, lr):
        self.lr.fill_(lr)

    def reset_meters(self):
        self.loss_sum.zero_()
        self.iou_sum.zero_()
        self.steps = 0

    def epoch_stats(self):
        """(mean loss, mean IoU) over the steps since reset_meters(); one host sync.
        Equal-sized batches make this the sample-weighted AverageMeter of utils.py:29-33."""
        k = max(self.steps, 1)
        return float(self.loss_sum.item()) / k, float(self.iou_sum.item()) / k


def cosine_lr(base_lr, min_lr, epoch, t_max):
    """CosineAnnealingLR closed form as configured at reference trains.py:237-239."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1.0 + math.cos(math.pi * epoch / t_max))
