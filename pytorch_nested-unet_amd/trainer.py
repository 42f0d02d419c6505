"""Fused training step: the loop body of reference trains.py:113-135
(forward, BCEDiceLoss (mean over heads under deep supervision, :118-124), IoU on
the last head, backward, SGD(momentum, wd) :229-231,133) enqueued as ONE hipGraph
replay per step, with no host synchronisation: loss and IoU counts stay on the
device and are read back only when the caller asks (AverageMeter semantics of
utils.py:17-33 are kept by `epoch_stats`).

Data parallel (new capability): one process per GPU, one gradient
all-reduce per step over RCCL, buckets in gradient-ready order. Replica state:
  * parameters, momentum and BatchNorm buffers are broadcast from rank 0 when the
    TrainStep is built (what DistributedDataParallel does at construction), so ranks
    that initialised differently or loaded a checkpoint on rank 0 only start identical;
  * BatchNorm batch statistics stay LOCAL to a replica (plain nn.BatchNorm2d semantics,
    archs1.py:19,21: every replica normalises over its own 16 images, exactly like the
    single-GPU reference step); the running statistics therefore drift apart between
    ranks, and the policy is "rank 0's buffers are the model's": sync_bn_buffers()
    broadcasts them before validation / checkpointing (train.py does);
  * p.grad holds the rank-MEAN gradient in every step layout (as DDP leaves it).
"""
import math
import os

import torch
import torch.distributed as dist

from . import _lib as L
from .parallel import allreduce_flat_, broadcast_flat_


class TrainStep:
    def __init__(self, model, batch_shape, lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=False,
                 use_graph=True, process_group=None, keep_grads=True, fused_update=None, loss="BCEDiceLoss", input_u8=False,
                 schedule=None, segmented=None):
        """loss: 'BCEDiceLoss' (reference losses.py:103-117, the default of trains.py:58) or 'LovaszHingeLoss'
        (losses.py:120-129, the loss of the reference's published table README.md:102-108; one class only) - both run
        inside the step's graph and under data parallel.
        input_u8: the step's inputs are the DECODED uint8 batch (images [N,H,W,C], masks [N,H,W,K] in {0,255}) plus optional
        per-sample augmentation codes; Normalize, /255, the mask scaling, rot90 / flips and the layout change of the
        reference's sample pipeline (dataset.py:66-74, trains.py:258-266) run as the first two launches of the step's graph
        (step_u8). Only uint8 crosses PCIe and the per-step NCHW->NHWC launch of the float path is gone.
        How the captured step is executed (all forms are bit-identical, tests/test_net_gpu.py):
          schedule  'lanes' - every op on its block's lane; 'list' - lanes chosen by a list scheduler over the hazard graph with
                    measured per-op costs (nunet_plan_calibrate); 'wave' - one stream, grouped convolution launches
          segmented False - ONE hipGraph, the lanes as parallel branches (ROCm replays those node by node from the host);
                    'flags' - one single-stream graph per lane, cross-lane dependencies as device-side flags (csrc/graph.hip);
                    True - single-stream graph segments with events between graph launches
        Both None (default): single-process training times (False, 'lanes') against ('flags', 'list') on the captured step and
        keeps the faster (self.executor_choice); data-parallel training keeps (False, 'lanes'), whose graph can hold the RCCL
        exchange. NUNET_SCHEDULE / NUNET_SEGMENTED force a form (tools)."""
        self.model = model
        self.eng = model.engine()
        dev = self.eng.device
        n, cin, h, w = batch_shape
        self.n, self.h, self.w = n, h, w
        self.ncls = model.num_classes
        x0 = torch.zeros(batch_shape, dtype=torch.float32, device=dev)
        self.pl = model.plan_for(x0)
        env_sched, env_seg = os.environ.get("NUNET_SCHEDULE"), os.environ.get("NUNET_SEGMENTED")
        self.executor_auto = schedule is None and segmented is None and env_sched is None and env_seg is None
        self.executor_choice = None          # {(segmented, schedule): ms per step} when the form was chosen by timing
        self.schedule = schedule or env_sched or "lanes"
        if self.schedule not in ("lanes", "wave", "list"):
            raise L.NunetError("TrainStep: schedule %r is not 'lanes', 'wave' or 'list'" % (self.schedule,))
        self._set_schedule(self.schedule)
        if segmented is None:
            segmented = {"0": False, "1": True, "2": "flags", "flags": "flags"}.get(env_seg or "0", False)
        self.segmented = segmented
        self._calibrated = False
        self.dp_exec = (False, "lanes")      # executor of the data-parallel layout 1 pass (see _choose_layout)
        self.heads = self.pl.heads
        self.x = x0
        self.t = torch.zeros((n, self.ncls, h, w), dtype=torch.float32, device=dev)
        self.input_u8 = bool(input_u8)
        if self.input_u8:
            import numpy as np
            from .dataset import MEAN, STD
            self.x_u8 = torch.zeros((n, h, w, cin), dtype=torch.uint8, device=dev)
            self.t_u8 = torch.zeros((n, h, w, self.ncls), dtype=torch.uint8, device=dev)
            self.aug = torch.zeros(n, dtype=torch.int32, device=dev)
            self._mean = torch.tensor(np.resize(np.asarray(MEAN, np.float32), cin), device=dev)
            self._std = torch.tensor(np.resize(np.asarray(STD, np.float32), cin), device=dev)
        self.logits = torch.empty((self.heads, n, self.ncls, h, w), dtype=torch.float32, device=dev)
        self.dlogits = torch.empty_like(self.logits)
        self.per = self.ncls * h * w
        kinds = {"BCEDiceLoss": L.LOSS_BCE_DICE, "LovaszHingeLoss": L.LOSS_LOVASZ_HINGE}
        if loss not in kinds:
            raise L.NunetError("TrainStep: loss %r is not one of %s" % (loss, sorted(kinds)))
        if loss == "LovaszHingeLoss" and self.ncls != 1:
            raise L.NunetError("LovaszHingeLoss squeezes the class dimension (reference losses.py:126-127): num_classes must be 1")
        self.loss_name, self.loss_kind = loss, kinds[loss]
        from .metrics import iou_logit_threshold
        self.iou_thr = iou_logit_threshold()
        self.loss_ws = torch.empty((L.lib().nunet_loss_step_ws_bytes(n, self.per, self.heads, self.loss_kind) + 7) // 8, dtype=torch.float64, device=dev)
        self.loss_out = torch.zeros(self.heads + 1, dtype=torch.float32, device=dev)   # per head, then their mean
        # device-side epoch meters: [sum of step losses, sum of step IoUs, last intersection, last union]
        self.meters = torch.zeros(4, dtype=torch.float64, device=dev)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.mom = torch.zeros_like(self.eng.flat_params)
        self.momentum, self.wd, self.nesterov = momentum, weight_decay, nesterov
        self.steps = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        # NUNET_FORCE_DP=1: take the data-parallel code path (bucketed exchange, three graphs) with any world size, so that a
        # one-GPU box can rehearse it over RCCL with a single rank (bench.py initialises the process group)
        self.dp = self.world > 1 or (os.environ.get("NUNET_FORCE_DP") == "1" and dist.is_initialized())
        if self.dp:
            # RCCL's high-priority stream lives in this process: beside it lowest-priority lanes are served in time slices
            L.check(L.lib().nunet_plan_set_lane_priority(self.pl.handle, 0), "plan_set_lane_priority")
        # Data-parallel step layout (NUNET_DP_MODE). 1 (default): one lane-faithful graph for forward + loss + the whole
        # backward, ONE exchange of the complete gradient scratch, one graph for unpack + SGD. 0: backward cut in two phases
        # so that bucket 0's exchange runs beside phase 2. Measured on one MI355X (single-rank RCCL group,
        # NUNET_FORCE_DP=1 python bench.py): the cut costs 350-500 us per step (phase 1 is the anti-diagonal's dependency chain with nothing
        # beside it, 2.71 ms vs 2.18 ms) - more than the 36.7 MB exchange it hides is expected to take on >= 4 GPUs.
        # 3: the whole step is ONE graph that contains both exchanges as nodes: the backward pass is issued as phase 1 left open
        # (nunet_plan_backward_phase bit 3: no join), the first bucket's all-reduce goes to a side stream that waits for exactly
        # the kernels producing it, phase 2 continues on the open lanes beside it, the second bucket and the optimiser step
        # follow the join. Overlap without a fork / join barrier inside the pass and without any host call during the step
        # (RCCL only: a gloo exchange is a host round trip and cannot be captured).
        # NUNET_DP_MODE unset / "auto": with more than one rank, capture() times layouts 1 and 3 on the real exchange and
        # keeps the faster (the decision is all-reduced, so every rank takes the same one); one rank: layout 1.
        mode = os.environ.get("NUNET_DP_MODE", "auto")
        self.dp_auto = mode == "auto" and self.dp
        self.dp_mode = 1 if mode == "auto" else int(mode)
        if self.dp and self.dp_mode == 3 and dist.get_backend(process_group) == "gloo":
            self.dp_mode = 1             # a gloo exchange is a host round trip: it cannot sit inside the step's graph
        self.dp_choice = None        # {layout: ms per step} when the layout was chosen by measurement
        self.use_graph = use_graph
        # optimiser step layout: 0 = unpack, SGD, (next forward's) pack as three streaming launches; 2 (default) = unpack
        # with the SGD step as its epilogue (nunet_plan_sgd, 46 us against 59 us for the pair); 1 = one tile kernel that
        # also repacks the weights (nunet_plan_update). The flat OIHW gradients (p.grad views) are materialised with
        # keep_grads=True in the fused layouts; in every layout they hold the rank-MEAN gradient (as
        # DistributedDataParallel leaves p.grad).
        # 3 = the optimiser step INSIDE the backward pass (nunet_plan_set_inpass_update): every VGGBlock is stepped and repacked
        # as an op of the pass behind its weight gradients, beside the rest of the pass; single-process training only (a
        # data-parallel step exchanges the gradients before the update: it falls back to 2).
        self.fused_update = int(os.environ.get("NUNET_FUSED_UPDATE", "2")) if fused_update is None else int(fused_update)   
        if self.fused_update == 3 and self.dp:
            self.fused_update = 2
        self.keep_grads = keep_grads
        self._inpass_set = False
        self._packed = False          # the arena's packed weights match the fp32 parameters
        self.g_fb = None
        self.g_b2 = None
        self.g_opt = None
        self._buckets = self._grad_scratch() if self.dp else None
        self._b0_armed = False
        self._comm = torch.cuda.Stream() if self.dp else None
        if self.dp and self.dp_mode == 2:
            self._arm_bucket0(True)
        for p, off in zip(self.eng.module_params, self.eng.param_off):
            p.grad = self.eng.flat_grads[off:off + p.numel()].view(p.shape)
        if self.world > 1:
            self.broadcast_state()

    def _arm_bucket0(self, on):
        self._b0_armed = L.lib().nunet_plan_bucket0_enable(self.pl.handle, 1 if on else 0) == 1

    def broadcast_state(self, src=0):
        """Rank `src`'s parameters, momentum and BatchNorm buffers become every rank's (construction time; also after
        loading a checkpoint on one rank)."""
        eng = self.eng
        for t in (eng.flat_params, self.mom, eng.bnbuf):
            broadcast_flat_(t, src, self.pg)
        nbt = eng.nbt.to(torch.float64)          # (gloo / RCCL both take floating tensors; counts are exact in fp64)
        broadcast_flat_(nbt, src, self.pg)
        eng.nbt.copy_(nbt.to(torch.int64))
        self._packed = False

    def sync_bn_buffers(self, src=0):
        """BatchNorm running statistics policy under data parallel: rank `src`'s buffers are the model's. Call before
        validating or saving a checkpoint (batch statistics, and hence the buffers, are per replica during training)."""
        if self.world > 1:
            eng = self.eng
            broadcast_flat_(eng.bnbuf, src, self.pg)
            nbt = eng.nbt.to(torch.float64)
            broadcast_flat_(nbt, src, self.pg)
            eng.nbt.copy_(nbt.to(torch.int64))

    # -- pieces -------------------------------------------------------------------
    def _fwd_loss(self):
        lib, eng, pl = L.lib(), self.eng, self.pl
        st = L.stream()
        flags = 3 if (self.fused_update in (1, 3) and self._packed) else 1
        if self.input_u8:
            # the sample pipeline on the device: image -> the plan's padded NHWC tile, mask -> {0,1} fp32 NCHW target
            L.check(lib.nunet_plan_stage_u8(pl.handle, L.ptr(self.x_u8), L.ptr(self._mean), L.ptr(self._std), L.ptr(self.aug), 1.0 / 255.0,
                                            L.ptr(pl.arena), L.nbytes(pl.arena), st), "plan_stage_u8")
            L.check(lib.nunet_preprocess_u8(L.ptr(self.t_u8), self.n, self.h, self.w, self.ncls, None, None, L.ptr(self.aug), 1.0,
                                            L.ptr(self.t), st), "preprocess_u8 (masks)")
            flags |= 4
        L.check(lib.nunet_plan_forward(pl.handle, L.ptr(eng.flat_params), L.ptr(eng.bnbuf), L.ptr(eng.nbt),
                                       L.ptr(self.x), L.ptr(pl.arena), L.nbytes(pl.arena), L.ptr(self.logits), flags, st), "plan_forward")
        L.check(lib.nunet_loss_step(L.ptr(self.logits), L.ptr(self.t), self.n, self.per, self.heads, self.loss_kind, L.ptr(self.loss_ws), L.nbytes(self.loss_ws),
                                    L.ptr(self.dlogits), L.ptr(self.loss_out), L.ptr(self.meters), self.iou_thr, st), "loss_step")
        pl.trained_forward = True

    def _bwd(self, phases):
        eng, pl = self.eng, self.pl
        L.check(L.lib().nunet_plan_backward_phase(pl.handle, L.ptr(eng.flat_params), L.ptr(self.dlogits), L.ptr(pl.arena), L.nbytes(pl.arena),
                                                  L.ptr(eng.flat_grads), 0, phases, L.stream()), "plan_backward_phase")

    def _fwd_bwd(self):
        self._fwd_loss()
        self._bwd(3 if self.fused_update else 7)

    def _update(self):
        """scratch -> SGD -> repacked weights in one launch (replaces unpack + sgd + the next forward's repack)."""
        eng, pl = self.eng, self.pl
        if self.fused_update == 3:      # already done, block by block, inside the backward pass
            return
        if self.fused_update == 2:      # gradient scratch -> SGD in one launch; the next forward repacks
            L.check(L.lib().nunet_plan_sgd(pl.handle, L.ptr(eng.flat_params), L.ptr(self.mom), L.ptr(pl.arena), L.nbytes(pl.arena), L.ptr(self.lr),
                                           self.momentum, self.wd, 1 if self.nesterov else 0, 1.0 / self.world,
                                           L.ptr(eng.flat_grads) if self.keep_grads else None, L.stream()), "plan_sgd")
            return
        L.check(L.lib().nunet_plan_update(pl.handle, L.ptr(eng.flat_params), L.ptr(self.mom), L.ptr(pl.arena), L.nbytes(pl.arena), L.ptr(self.lr),
                                          self.momentum, self.wd, 1 if self.nesterov else 0, 1.0 / self.world,
                                          L.ptr(eng.flat_grads) if self.keep_grads else None, L.stream()), "plan_update")

    def sync_weights(self):
        """Repack the plan's 16-bit weights from the fp32 parameters: call after changing the parameters by anything
        other than step() (load_state_dict, a stock optimiser) when fused_update is on."""
        self._arm_inpass_update()
        if self.fused_update in (1, 3):
            L.check(L.lib().nunet_plan_repack(self.pl.handle, L.ptr(self.eng.flat_params), L.ptr(self.pl.arena), L.nbytes(self.pl.arena), L.stream()), "plan_repack")
            self._packed = True

    def _arm_inpass_update(self):
        """fused_update 3: hand the plan the optimiser's state so that its backward pass carries the step (off otherwise)."""
        on = self.fused_update == 3
        if on == self._inpass_set:
            return
        eng = self.eng
        if on:
            L.check(L.lib().nunet_plan_set_inpass_update(self.pl.handle, L.ptr(eng.flat_params), L.ptr(self.mom), L.ptr(self.lr), self.momentum,
                                                         self.wd, 1 if self.nesterov else 0, 1.0 / self.world,
                                                         L.ptr(eng.flat_grads) if self.keep_grads else None), "plan_set_inpass_update")
        else:
            L.check(L.lib().nunet_plan_set_inpass_update(self.pl.handle, None, None, None, 0.0, 0.0, 0, 1.0, None), "plan_set_inpass_update")
        self._inpass_set = on

    def _opt(self):
        if self.fused_update:
            return self._update()
        eng = self.eng
        if self.world > 1:
            eng.flat_grads.mul_(1.0 / self.world)      # p.grad = rank mean in every layout
        L.check(L.lib().nunet_sgd_step(L.ptr(eng.flat_params), L.ptr(eng.flat_grads), L.ptr(self.mom),
                                       eng.flat_params.numel(), L.ptr(self.lr), self.momentum, self.wd,
                                       1 if self.nesterov else 0, 0, 1.0, L.stream()), "sgd_step")

    def _grad_scratch(self):
        """The plan's native-layout gradient scratch as two fp32 views in gradient-ready order."""
        import ctypes as C
        off, b0, tot = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().nunet_plan_grad_scratch(self.pl.handle, C.byref(off), C.byref(b0), C.byref(tot)), "plan_grad_scratch")
        flat = self.pl.arena[off.value:off.value + 4 * tot.value].view(torch.float32)
        self._scratch = flat
        return flat[:b0.value], flat[b0.value:]

    def _exchange(self, t):
        """Sum a gradient bucket over ranks; asynchronous under RCCL so that it overlaps what follows."""
        if dist.get_backend(self.pg) == "gloo":
            allreduce_flat_(t, self.pg)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _dp_step(self, run1, run2, run3):
        """Data-parallel step: bucket 0 (heads + the last anti-diagonal = 75 % of the gradient bytes,
        complete after backward phase 1) is all-reduced while phase 2 runs; bucket 1 follows; unpack +
        SGD (grad_scale 1/world) run on the reduced scratch."""
        if self.dp_mode in (1, 2):
            run1()
            if run2 is not None:
                run2()
            if self.dp_mode == 2 and self._b0_armed and dist.get_backend(self.pg) != "gloo":
                # bucket 0 starts on the side stream as soon as the pass (still running) has recorded "bucket 0
                # complete"; bucket 1 follows the pass on the caller's stream
                b0, b1 = self._buckets
                cur = torch.cuda.current_stream()
                L.check(L.lib().nunet_plan_bucket0_wait(self.pl.handle, self._comm.cuda_stream), "plan_bucket0_wait")
                with torch.cuda.stream(self._comm):
                    h0 = dist.all_reduce(b0, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                h1 = dist.all_reduce(b1, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                with torch.cuda.stream(self._comm):
                    h0.wait()
                cur.wait_stream(self._comm)
                h1.wait()
            else:
                h = self._exchange(self._scratch)
                if h is not None:
                    h.wait()
            run3()
            return
        b0, b1 = self._buckets
        run1()
        h0 = self._exchange(b0)
        run2()
        h1 = self._exchange(b1)
        for h in (h0, h1):
            if h is not None:
                h.wait()
        run3()

    def _in_graph_exchange_step(self):
        """Layout 3: forward, loss, the backward pass with both gradient exchanges and the optimiser step as one stream of
        work - captured, one graph. Bucket 0 (heads + the last anti-diagonal, 75 % of the bytes) is all-reduced on a side stream
        ordered behind its producing kernels only, while phase 2 continues on the pass's open lanes."""
        lib, pl = L.lib(), self.pl
        b0, b1 = self._buckets
        self._fwd_loss()
        self._bwd(1 | 8)                                   # phase 1, pass left open: the current stream has waited for nothing
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()                         # (a stream this capture has not used: it joins the capture by its waits)
        side.wait_stream(cur)                              # single-lane issue keeps everything on `cur`; with lanes this is the fork point only
        L.check(lib.nunet_plan_bucket0_wait(pl.handle, side.cuda_stream), "plan_bucket0_wait")
        with torch.cuda.stream(side):
            h0 = dist.all_reduce(b0, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._bwd(2 | 16)                                  # phase 2 on the open lanes, beside the exchange; joins `cur`
        # the second bucket is ISSUED before `cur` waits for the first: the collective stream then never waits on an event that
        # descends from its own tail (the stream-capture pattern ROCm 7.2 crashes on, tests/test_capture_gpu.py); the collective
        # stream is in order, so waiting for the second exchange covers the first
        h1 = dist.all_reduce(b1, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        h1.wait()
        self._side, self._h0 = side, h0                    # (kept alive until the next step)
        if not self.fused_update:
            self._bwd(4)
        self._opt()

    def _eager_step(self):
        if self.dp and self.dp_mode == 3:
            return self._in_graph_exchange_step()
        if self.dp:
            one_pass = self.dp_mode in (1, 2)     # the bucket-0 event is recorded by a pass that runs both phases together
            self._dp_step((lambda: (self._fwd_loss(), self._bwd(3))) if one_pass else (lambda: (self._fwd_loss(), self._bwd(1))),
                          None if one_pass else (lambda: self._bwd(2)),
                          lambda: (None if self.fused_update else self._bwd(4), self._opt()))
        else:
            self._fwd_bwd()
            self._opt()

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
        if self.input_u8:          # (uint8 images, uint8 masks)
            self.x_u8.copy_(inp)
            self.t_u8.copy_(target)
            self.aug.zero_()
        else:
            self.x.copy_(inp)
            self.t.copy_(target)
        snap = [t.clone() for t in (eng.flat_params, eng.bnbuf, eng.nbt, self.mom, self.meters)]
        steps0 = self.steps
        self.sync_weights()            # from here on every step leaves the packed weights current
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.dp and self.dp_mode == 3 and dist.get_backend(self.pg) == "gloo":
            self.dp_mode = 1                                # a host-side exchange cannot be a graph node
        if not self.dp:
            # the whole step, recorded once on a side stream and replayed on the caller's: as ONE hipGraph (default), or
            # - segmented=True - as a program of single-stream graph segments over the plan's lanes with the cross-lane
            # dependencies as events between graph launches (csrc/graph.hip nunet_seg_*: explicit node -> queue placement on lanes
            # chosen by measurement; 2.33 vs 1.91 ms per step on MI355X: every segment launch costs 10-13 us on its lane)
            body = lambda: (self._fwd_bwd(), self._opt())
            if self.executor_auto:
                self._choose_executor(s, body)
            else:
                self.g_fb = self._build_executor(s, body, self.segmented, self.schedule)
        else:
            if self.dp_auto:
                self._choose_layout(s)
            if self.dp_mode in (1, 2):
                self._capture_one_pass(s)
            elif self.dp_mode == 3:
                self._capture_in_graph_exchange(s)
        if self.dp and self.dp_mode not in (1, 2, 3):
            self.g_fb = torch.cuda.CUDAGraph()          # forward + loss + backward phase 1
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):   # (the RCCL watchdog thread polls events meanwhile)
                self._fwd_loss()
                self._bwd(1)
            self.g_b2 = torch.cuda.CUDAGraph()          # backward phase 2
            with torch.cuda.graph(self.g_b2, capture_error_mode="thread_local"):   # (the RCCL watchdog thread polls events meanwhile)
                self._bwd(2)
            self.g_opt = torch.cuda.CUDAGraph()         # unpack + SGD
            with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):   # (the RCCL watchdog thread polls events meanwhile)
                if not self.fused_update:
                    self._bwd(4)
                self._opt()
        torch.cuda.synchronize()
        with torch.no_grad():
            for dst, src in zip((eng.flat_params, eng.bnbuf, eng.nbt, self.mom, self.meters), snap):
                dst.copy_(src)
        self.steps = steps0
        self.sync_weights()            # the restored parameters, repacked

    def _set_schedule(self, schedule):
        self.schedule = schedule
        L.check(L.lib().nunet_plan_set_schedule(self.pl.handle, {"lanes": 0, "wave": 1, "list": 2}[schedule]), "plan_set_schedule")

    def _build_executor(self, s, body, segmented, schedule):
        """The step body recorded in one of its executable forms (see __init__)."""
        self._set_schedule(schedule)
        self.segmented = segmented
        if schedule == "list" and not self._calibrated:
            # the list scheduler's per-op costs, measured: the step on ONE lane with a device timestamp behind every op
            L.check(L.lib().nunet_plan_calibrate(self.pl.handle, 1), "plan_calibrate")
            try:
                g = _NativeGraph(s, body)
                with torch.cuda.stream(s):
                    for _ in range(3):
                        g.replay()
                torch.cuda.synchronize()
                self._single_lane_ms = self._time_program(g, 5)     # (with the stamp kernels: an upper bound of the one-lane step)
            finally:
                L.check(L.lib().nunet_plan_calibrate(self.pl.handle, 0), "plan_calibrate")
            # (kept alive while the program below picks its lanes: with the calibration graph - and its launch stream - destroyed
            #  first, 4 of 20 processes came up with a flag program at 4.8 ms per step instead of 1.7; with it alive 0 of 12, like
            #  the capture-time choice, whose first candidate - a native graph - is alive at that point: 0 of 26. Which stream
            #  inherits the freed hardware-queue slot decides; ROCm offers no way to ask.)
            self._calib_graph = g
            self._calibrated = True
        if segmented == "flags" and getattr(self, "_single_lane_ms", None):
            # A flag program is checked against the one-lane step before it is used: which hardware queue a lane inherits is
            # ROCm's choice, and unchecked 1 process in 5 came up at 4.8 ms per step instead of 1.7 (every lane's kernels
            # serialised). Such a program is thrown away and recorded again on newly picked lanes.
            prog = None
            for attempt in range(3):
                prog = _SegProgram(s, body, flags=True)
                ms = self._time_program(prog, 8)
                if ms < 0.9 * self._single_lane_ms:
                    break
                print("[nunet] flag-synchronised program came up at %.2f ms per step (one lane: %.2f): picking new lanes (%d)"
                      % (ms, self._single_lane_ms, attempt + 1))
                if attempt < 2:
                    prog = None
                    L.check(L.lib().nunet_plan_reset_lanes(self.pl.handle), "plan_reset_lanes")
        else:
            prog = _SegProgram(s, body, flags=segmented == "flags") if segmented else _NativeGraph(s, body)
        self._calib_graph = None
        return prog

    def _time_program(self, g, reps):
        """ms per replay of a recorded program, issued the way step() issues it (batch copy + replay from the caller's stream)."""
        buf = self.x_u8 if self.input_u8 else self.x
        fresh = buf.clone()
        def one():
            buf.copy_(fresh, non_blocking=True)
            g.replay()
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            one()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def _choose_executor(self, s, body, reps=20):
        """Time the two executable forms of the captured step on this device and keep the faster: the multi-branch hipGraph
        with block lanes, and the flag-synchronised single-stream graphs with list-scheduled lanes. (The caller restores the
        training state.) A form that cannot be recorded or replayed here - e.g. no two streams on distinct hardware queues -
        drops out with a message."""
        forms = [(False, "lanes"), ("flags", "list")]
        progs, times = {}, {}
        for form in forms:
            try:
                g = self._build_executor(s, body, *form)
                # timed the way step() will run it: from the caller's stream, behind the copies of a fresh batch (a stream that
                # shares a hardware queue with one of the lanes shows here, not after the choice)
                torch.cuda.current_stream().wait_stream(s)
                buf = self.x_u8 if self.input_u8 else self.x
                fresh = buf.clone()
                def one():
                    buf.copy_(fresh, non_blocking=True)
                    g.replay()
                for _ in range(3):
                    one()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    one()
                e1.record()
                torch.cuda.synchronize()
                one()                            # (a cross-lane wait that timed out fails the NEXT launch)
                torch.cuda.synchronize()
                progs[form], times[form] = g, e0.elapsed_time(e1) / reps
            except Exception as e:
                if form == forms[0]:
                    raise
                print("[nunet] executor %s/%s is not available here: %s" % (form[0], form[1], e))
        best = min(times, key=times.get)
        self.executor_choice = times
        self.g_fb = progs.pop(best)
        progs.clear()
        self._set_schedule(best[1])
        self.segmented = best[0]

    def _capture_one_pass(self, s):
        """Layouts 1 and 2: forward + loss + the whole backward as one graph (layout 2: with the bucket-0 event recorded
        inside it), the exchange between, unpack + SGD as a second graph."""
        self._arm_bucket0(self.dp_mode == 2)
        body = lambda: (self._fwd_loss(), self._bwd(3))
        # layout 1 may run its pass as flag-synchronised list-scheduled lanes (self.dp_exec, chosen by _choose_layout); layout 2
        # records an event inside the pass's graph and keeps the single hipGraph
        ex = self.dp_exec if self.dp_mode == 1 else (False, "lanes")
        self.g_fb = self._build_executor(s, body, *ex)
        self.g_b2 = None
        self.g_opt = _NativeGraph(s, lambda: (None if self.fused_update else self._bwd(4), self._opt()))

    def _capture_in_graph_exchange(self, s):
        """Layout 3: the whole data-parallel step, exchanges included, as one graph."""
        self.g_fb = _NativeGraph(s, self._in_graph_exchange_step)
        self.g_b2 = self.g_opt = None

    def _choose_layout(self, s, reps=8):
        """Time layout 1 (one exchange after the pass) against layout 3 (both exchanges inside the step's graph, bucket 0
        beside phase 2 of the backward pass) on the real process group and keep the faster. The caller restores the
        training state. (Layout 2 - an event recorded inside the graph - stays selectable but is not a candidate: ROCm 7.2
        releases a waiter on such an event only when the graph ends, tests/test_dist_gpu.py.)"""
        times = []
        modes = [(1, (False, "lanes"))]
        if dist.get_backend(self.pg) != "gloo":
            # (the flag-synchronised lanes only where every rank has a device of its own: ranks sharing one GPU - the rehearsal
            #  tests - also share its hardware queues, and a polling kernel may then sit in front of the signal it waits for)
            # (one-rank rehearsal with RCCL initialised: flag lanes 1.82 ms, graph 1.96-1.98 ms per step - with the side lanes at
            #  DEFAULT stream priority; at the lowest priority, beside RCCL's high-priority stream, 3.4-4.4 ms. NUNET_DP_FLAGS=0
            #  takes the candidate out.)
            if self.world <= torch.cuda.device_count() and os.environ.get("NUNET_DP_FLAGS", "1") == "1":
                modes.append((1, ("flags", "list")))
            modes.append((3, (False, "lanes")))
        def all_agree(ok):
            """A candidate is timed only if EVERY rank recorded it: the timing loop holds collectives, and a rank that skipped it
            would leave the others waiting in them."""
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.eng.device)
            if dist.get_backend(self.pg) == "gloo":
                h = f.cpu(); dist.all_reduce(h, op=dist.ReduceOp.MIN, group=self.pg); f = h
            else:
                dist.all_reduce(f, op=dist.ReduceOp.MIN, group=self.pg)
            return bool(f.item() > 0.5)

        for mode, ex in modes:
            self.dp_mode = mode
            self.dp_exec = ex
            err = None
            try:
                if mode == 3:
                    self._capture_in_graph_exchange(s)
                    run = self.g_fb.replay
                else:
                    self._capture_one_pass(s)
                    run = lambda: self._dp_step(self.g_fb.replay, None, self.g_opt.replay)
            except Exception as e:       # e.g. a runtime that cannot capture the collectives, no two streams on distinct queues
                err = e
            if not all_agree(err is None):
                if (mode, ex) == modes[0]:
                    raise err if err is not None else L.NunetError("data-parallel layout 1 could not be captured on another rank")
                print("[nunet] data-parallel candidate (layout %d, %s/%s) could not be captured on every rank%s"
                      % (mode, ex[0], ex[1], ": %s" % err if err is not None else ""))
                times.append(float("inf"))
                continue
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        t = torch.tensor(times, dtype=torch.float64, device=self.eng.device)
        t = torch.nan_to_num(t, posinf=1e30)
        if dist.get_backend(self.pg) == "gloo":
            h = t.cpu(); dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.pg); t = h
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)     # the slowest rank decides, identically everywhere
        t = t.tolist()
        label = lambda m, ex: {1: "exchange_after_pass", 2: "bucket0_event_in_graph", 3: "exchange_inside_graph"}[m] + ("" if ex[0] is False else "/flags+list")
        self.dp_choice = {label(m, ex): v for (m, ex), v in zip(modes, t)}
        best = min(range(len(t)), key=lambda q: t[q])
        self.dp_mode, self.dp_exec = modes[best]
        self.g_fb = self.g_opt = None

    def step(self, inp=None, target=None):
        """One training iteration. `inp`/`target` are device tensors (copied into the
        static graph inputs); None re-uses what is already staged."""
        if inp is not None:
            if self.input_u8:
                raise L.NunetError("this TrainStep was built with input_u8=True: feed it with step_u8(images_u8, masks_u8, aug)")
            self.x.copy_(inp, non_blocking=True)
            self.t.copy_(target, non_blocking=True)
        self._run()

    def step_u8(self, images_u8, masks_u8, aug=None):
        """One training iteration from the decoded uint8 batch (device tensors; images [N,H,W,C], masks [N,H,W,K] with
        values {0,255}) and optional augmentation codes (dataset.draw_augmentation); needs input_u8=True."""
        if not self.input_u8:
            raise L.NunetError("step_u8 needs a TrainStep built with input_u8=True")
        self.x_u8.copy_(images_u8, non_blocking=True)
        self.t_u8.copy_(masks_u8, non_blocking=True)
        if aug is None:
            self.aug.zero_()
        else:
            if self.h != self.w and bool((aug & 1).any()):
                raise L.NunetError("rot90 by an odd count needs square images")
            self.aug.copy_(aug, non_blocking=True)
        self._run()

    def _run(self):
        if self.model._engine is not self.eng or not self.eng.intact():
            raise L.NunetError("the module's parameter arenas were re-homed (moved to another device / parameters replaced) "
                               "after this TrainStep was built: its graphs would update orphaned memory. Build a new TrainStep.")
        if self.g_fb is not None:
            if self.dp and self.dp_mode != 3:
                self._dp_step(self.g_fb.replay, self.g_b2.replay if self.g_b2 is not None else None, self.g_opt.replay)
            else:
                self.g_fb.replay()
        else:
            if self.fused_update in (1, 3) and not self._packed:
                self.sync_weights()
            self._eager_step()
        self.steps += 1

    def set_lr(self, lr):
        self.lr.fill_(lr)

    def reset_meters(self):
        self.meters.zero_()
        self.steps = 0

    def epoch_stats(self):
        """(mean loss, mean IoU) over the steps since reset_meters(); one host sync.
        Equal-sized batches make this the sample-weighted AverageMeter of utils.py:29-33."""
        k = max(self.steps, 1)
        m = self.meters.tolist()
        return m[0] / k, m[1] / k


class _SegProgram:
    """nunet_seg_* wrapper with the replay() surface of torch.cuda.CUDAGraph: the step body is run twice on `side_stream` - a dry
    pass that launches nothing and finds the cross-lane events, then the recording pass."""

    def __init__(self, side_stream, body, flags=False):
        import ctypes as C
        lib = L.lib()
        self.stream = side_stream            # the program replays on this stream: keep it alive
        self.handle = None
        self.flags = bool(flags)
        with torch.cuda.stream(side_stream):
            for mode in (1, 2 if flags else 0):      # NUNET_SEG_DRY, then NUNET_SEG_FLAGS / NUNET_SEG_RECORD
                L.check(lib.nunet_seg_begin(L.stream(), mode), "seg_begin")
                try:
                    body()
                finally:
                    h = C.c_void_p()
                    rc = lib.nunet_seg_end(L.stream(), C.byref(h))
                L.check(rc, "seg_end")
        self.handle = h

    def info(self):
        import ctypes as C
        v = [C.c_int32() for _ in range(4)]
        L.check(L.lib().nunet_seg_info(self.handle, *[C.byref(x) for x in v]), "seg_info")
        return dict(zip(("graph_launches", "event_records", "event_waits", "kernel_nodes"), (x.value for x in v)))

    def replay(self):
        L.check(L.lib().nunet_seg_launch(self.handle, L.stream()), "seg_launch")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.lib().nunet_seg_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _NativeGraph:
    """nunet_graph_* wrapper with the replay() surface of torch.cuda.CUDAGraph."""

    def __init__(self, side_stream, body):
        import ctypes as C
        lib = L.lib()
        with torch.cuda.stream(side_stream):
            L.check(lib.nunet_graph_begin(L.stream()), "graph_begin")
            try:
                body()
            finally:
                h = C.c_void_p()
                rc = lib.nunet_graph_end(L.stream(), C.byref(h))
            L.check(rc, "graph_end")
        self.handle = h

    def info(self):
        import ctypes as C
        v = [C.c_int32() for _ in range(5)]
        L.check(L.lib().nunet_graph_info(self.handle, *[C.byref(x) for x in v]), "graph_info")
        return dict(zip(("nodes", "edges_captured", "edges_final", "padding", "lanes"), (x.value for x in v)))

    def replay(self):
        L.check(L.lib().nunet_graph_launch(self.handle, L.stream()), "graph_launch")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.lib().nunet_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def cosine_lr(base_lr, min_lr, epoch, t_max):
    """CosineAnnealingLR closed form as configured at reference trains.py:237-239."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1.0 + math.cos(math.pi * epoch / t_max))
