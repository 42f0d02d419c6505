"""Data-parallel helpers: one process per GPU, replicas with local
BatchNorm statistics, one gradient exchange per step.

The reference has no distributed path at all (`model = model.cuda()`, trains.py:223);
this is new capability behind the same loop body (trains.py:113-135)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend="nccl"):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def allreduce_flat_(flat, group=None):
    """Sum a flat gradient arena over ranks in place. RCCL handles device tensors; under
    gloo (CPU tests, single-GPU rehearsal) device tensors take a host round trip."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_flat_(flat, src=0, group=None):
    """Overwrite `flat` on every rank with rank `src`'s copy (initial state / checkpoint policy). Same host round
    trip as allreduce_flat_ under gloo."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)
    return flat


def shard_seed(base_seed, rank, step, world):
    """Seed of the synthetic shard rank `rank` consumes at `step`: disjoint across ranks."""
    return base_seed + step * world + rank


def grad_ready_order(deep_supervision=False, unet=False):
    """Block names in the order their gradients complete during backward (reverse of the
    forward execution order, heads first) — the bucket order for overlapping the
    all-reduce with the level-0/1 backward."""
    if unet:
        fwd = ["conv%d_0" % i for i in range(5)] + ["conv%d_%d" % (i, 4 - i) for i in (3, 2, 1, 0)]
        heads = ["final"]
    else:
        fwd = ["conv%d_%d" % (s - j, j) for s in range(5) for j in range(s + 1)]
        heads = ["final4", "final3", "final2", "final1"] if deep_supervision else ["final"]
    return heads + fwd[::-1]


def param_ranges(module):
    """{block name: (offset, numel)} of each top-level child's parameters in the flat
    arena (reference parameters() order)."""
    out, off = {}, 0
    for name, child in module.named_children():
        n = sum(p.numel() for p in child.parameters())
        if n:
            out[name] = (off, n)
        off += n
    return out


def sgd_reference_step_(p, g, mom, lr, momentum, weight_decay, grad_scale):
    """CPU restatement of nunet_sgd_step (torch.optim.SGD semantics, trains.py:229-231)
    used by the gloo tests."""
    gg = g * grad_scale + weight_decay * p
    mom.mul_(momentum).add_(gg)
    p.add_(mom, alpha=-lr)
    return p
