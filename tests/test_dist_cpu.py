"""N>1 host logic on CPU: world_size-2 gloo process group (127.0.0.1 rendezvous)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nunet_amd
from nunet_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                      # identical replicas, as bench.py builds them
    p = torch.randn(1000)
    mom = torch.zeros(1000)
    for step in range(3):
        g = torch.full((1000,), float(rank + 1)) * (step + 1)     # rank-dependent shard gradient
        parallel.allreduce_flat_(g)
        assert torch.allclose(g, torch.full((1000,), 3.0 * (step + 1)))          # 1 + 2
        parallel.sgd_reference_step_(p, g, mom, 1e-3, 0.9, 1e-4, 1.0 / world)
    # replicas stay bit-identical after the exchange + optimiser
    ref = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(ref, p)
    assert torch.equal(ref[0], ref[1])
    seeds = {parallel.shard_seed(1234, rr, s, world) for rr in range(world) for s in range(10)}
    assert len(seeds) == 10 * world           # disjoint shards
    q.put((rank, float(p.sum())))
    dist.destroy_process_group()


def test_gloo_world2_allreduce_and_sgd():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert abs(res[0] - res[1]) == 0.0


def test_grad_ready_order_and_ranges():
    m = nunet_amd.archs.NestedUNet(1, 3, True)
    order = parallel.grad_ready_order(True)
    assert order[:5] == ["final4", "final3", "final2", "final1", "conv0_4"] and order[-1] == "conv0_0"
    rng = parallel.param_ranges(m)
    assert set(order) == set(rng)
    total = sum(n for _, n in rng.values())
    assert total == sum(p.numel() for p in m.parameters()) == 9163428
    # ranges tile the flat arena in parameters() order
    off = 0
    for name, _ in m.named_children():
        if name in rng:
            assert rng[name][0] == off
            off += rng[name][1]
    # ~3/4 of the gradient bytes are complete after the first five blocks of backward
    first = sum(rng[k][1] for k in parallel.grad_ready_order(False)[1:6])
    m0 = nunet_amd.archs.NestedUNet(1, 3, False)
    assert abs(first / sum(p.numel() for p in m0.parameters()) - 0.756) < 0.005


def _run_bench(args, env_extra=None, timeout=180):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_self_launches_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must spawn its own ranks (the driver's first
    multi-GPU run may call it exactly like the 1-GPU line): the parent makes no GPU call, the children rendezvous on
    127.0.0.1, rank 0 alone prints ONE JSON line. Exercised up to the process group (gloo, CPU) by --dist-dry-run."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dist-dry-run"])
    assert rc == 0, err
    assert len(lines) == 1
    assert lines[0]["ranks_seen"] == 2 and lines[0]["world_size"] == 2 and lines[0]["n_gpus"] == 2


def test_bench_under_a_launcher_keeps_its_environment():
    """The documented launch (torch.distributed.run sets RANK / WORLD_SIZE): no second level of spawning."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-dry-run"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and '"ranks_seen": 2' in lines[0]


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    rc, lines, err = _run_bench(["--gpus", "2", "--dist-dry-run"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert rc != 0 and not lines and "does not match" in err
