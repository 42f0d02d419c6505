"""Engine: flat parameter/gradient arenas + per-shape plans over libnunet.

Owns what the reference leaves to torch.nn/autograd below the
`model(input)` / `loss.backward()` call sites (reference trains.py:126,132):
  * a flat fp32 parameter arena in reference parameters() order; the module's
    nn.Parameters are views into it (so stock optimisers/state_dict keep working)
  * a flat gradient arena (p.grad are views), BN running-stat arena
  * one nunet_plan + device workspace per (N, H, W)
"""
import ctypes as C

import torch

from . import _lib as L


class Plan:
    def __init__(self, n, h, w, cin, ncls, ds, dtype, unet, device):
        lib = L.lib()
        cfg = L.PlanCfg(n, h, w, cin, ncls, 1 if ds else 0, dtype, 1 if unet else 0)
        self.handle = lib.nunet_plan_create(C.byref(cfg))
        if not self.handle:
            raise L.NunetError("nunet_plan_create: " + lib.nunet_last_error().decode())
        self.n, self.h, self.w, self.ncls, self.dtype = n, h, w, ncls, dtype
        self.heads = lib.nunet_plan_num_heads(self.handle)
        self.nparams = lib.nunet_plan_param_count(self.handle)
        self.nbnbuf = lib.nunet_plan_bnbuf_count(self.handle)
        self.nbn = lib.nunet_plan_bn_layers(self.handle)
        self.arena_bytes = lib.nunet_plan_arena_bytes(self.handle)
        self.arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=device)
        self.trained_forward = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.lib().nunet_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def feature(self, i, j):
        """Block output x_{i,j} as an [N,H_i,W_i,C] NHWC view (tests/debug)."""
        pitch, ch = C.c_int32(), C.c_int32()
        off = L.lib().nunet_plan_feature(self.handle, i, j, C.byref(pitch), C.byref(ch))
        if off < 0:
            raise L.NunetError("no feature x%d_%d" % (i, j))
        es = 4 if self.dtype == L.F32 else 2
        hi, wi = self.h >> i, self.w >> i
        flat = self.arena[off:].view(L.TORCH_DTYPE[self.dtype])
        return flat.as_strided((self.n, hi, wi, ch.value), (hi * wi * pitch.value, wi * pitch.value, pitch.value, 1))


class Engine:
    """Flat arenas for one module instance on one device."""

    def __init__(self, module, device):
        self.device = device
        self.module_params = [p for p in module.parameters()]
        names = [k for k, _ in module.named_parameters()]
        self.param_names = names
        total = sum(p.numel() for p in self.module_params)
        self.flat_params = torch.empty(total, dtype=torch.float32, device=device)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=device)
        self.param_off = []
        off = 0
        with torch.no_grad():
            for p in self.module_params:
                if p.dtype != torch.float32:
                    raise L.NunetError("parameters must stay fp32 master weights (use the dtype= ctor argument "
                                       "for bf16/fp16 compute), got %s" % p.dtype)
                n = p.numel()
                self.flat_params[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[off:off + n].view(p.shape)
                self.param_off.append(off)
                off += n
        # BN buffers in state_dict order: running_mean, running_var (+ nbt)
        bns = [m for m in module.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        self.bns = bns
        nb = sum(2 * m.num_features for m in bns)
        self.bnbuf = torch.empty(nb, dtype=torch.float32, device=device)
        self.nbt = torch.zeros(len(bns), dtype=torch.int64, device=device)
        off = 0
        with torch.no_grad():
            for k, m in enumerate(bns):
                c = m.num_features
                self.bnbuf[off:off + c].copy_(m.running_mean)
                m.running_mean.data = self.bnbuf[off:off + c]
                self.bnbuf[off + c:off + 2 * c].copy_(m.running_var)
                m.running_var.data = self.bnbuf[off + c:off + 2 * c]
                self.nbt[k] = m.num_batches_tracked.to(device)
                m.num_batches_tracked.data = self.nbt[k]
                off += 2 * c
        self.plans = {}

    def intact(self):
        p0, pl = self.module_params[0], self.module_params[-1]
        base = self.flat_params.data_ptr()
        return (p0.data_ptr() == base and pl.data_ptr() == base + 4 * self.param_off[-1]
                and self.bns[0].running_mean.data_ptr() == self.bnbuf.data_ptr())

    def plan(self, n, h, w, cin, ncls, ds, dtype, unet):
        key = (n, h, w, dtype)
        pl = self.plans.get(key)
        if pl is None:
            pl = Plan(n, h, w, cin, ncls, ds, dtype, unet, self.device)
            if pl.nparams != self.flat_params.numel() or pl.nbnbuf != self.bnbuf.numel() or pl.nbn != len(self.bns):
                raise L.NunetError("plan/module parameter layout mismatch: %d vs %d params" %
                                   (pl.nparams, self.flat_params.numel()))
            self.plans[key] = pl
        return pl

    def forward(self, pl, inp, training):
        L.require_gpu_tensor(inp, torch.float32, "input")
        logits = torch.empty((pl.heads, pl.n, pl.ncls, pl.h, pl.w), dtype=torch.float32, device=self.device)
        L.check(L.lib().nunet_plan_forward(pl.handle, L.ptr(self.flat_params), L.ptr(self.bnbuf), L.ptr(self.nbt),
                                           L.ptr(inp), L.ptr(pl.arena), L.nbytes(pl.arena), L.ptr(logits), 1 if training else 0,
                                           L.stream()), "nunet_plan_forward")
        pl.trained_forward = bool(training)
        return logits

    def backward(self, pl, dlogits, accumulate):
        if not pl.trained_forward:
            raise L.NunetError("backward needs a training-mode forward (BatchNorm batch statistics) first")
        L.require_gpu_tensor(dlogits, torch.float32, "grad_output")
        L.check(L.lib().nunet_plan_backward(pl.handle, L.ptr(self.flat_params), L.ptr(dlogits), L.ptr(pl.arena), L.nbytes(pl.arena),
                                            L.ptr(self.flat_grads), 1 if accumulate else 0, L.stream()),
                "nunet_plan_backward")

    def attach_grads(self):
        """Decide assign/accumulate and point every p.grad at the flat arena.
        Returns `accumulate` for nunet_plan_backward."""
        ps = [p for p in self.module_params if p.requires_grad]
        gbase = self.flat_grads.data_ptr()
        first = self.module_params[0]
        ours = first.grad is not None and first.grad.data_ptr() == gbase
        if ours:
            return True
        foreign = [p for p in ps if p.grad is not None]
        if foreign:
            # grads produced elsewhere: fold them into the arena, then accumulate
            with torch.no_grad():
                self.flat_grads.zero_()
                for p, off in zip(self.module_params, self.param_off):
                    if p.grad is not None:
                        self.flat_grads[off:off + p.numel()].add_(p.grad.reshape(-1))
        for p, off in zip(self.module_params, self.param_off):
            if p.requires_grad:
                p.grad = self.flat_grads[off:off + p.numel()].view(p.shape)
        return bool(foreign)


class _NetFn(torch.autograd.Function):
    """Whole-network autograd node: forward = nunet_plan_forward, backward =
    nunet_plan_backward writing parameter gradients straight into p.grad."""

    @staticmethod
    def forward(ctx, anchor, inp, module, pl, training):
        ctx.module, ctx.pl = module, pl
        return module._engine.forward(pl, inp, training)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.module._engine
        acc = eng.attach_grads()
        eng.backward(ctx.pl, dlogits.contiguous(), acc)
        return None, None, None, None, None
