#!/usr/bin/env python3
"""bench.py — images/sec of one NestedUNet training step on MI355X.

Workload (BASELINE.json configs[1]): NestedUNet(1, 3, deep_supervision=False), 96x96,
per-GPU batch 16, bf16 storage / fp32 accumulate, BCEDiceLoss, SGD(lr 1e-3, mom 0.9,
wd 1e-4). One "step" = the loop body of reference trains.py:113-135: forward, loss,
IoU counts, backward, (gradient all-reduce for N>1), SGD — nothing skipped.
Inputs are synthetic (pytorch_nested-unet_amd/synth.py) and resident in HBM before the
timed region; each step copies the next staged batch into the graph's static inputs.

  python bench.py --gpus 1 --steps 200 --warmup 50
  python bench.py --gpus N ...          (no launcher: spawns one child process per GPU itself, before any GPU call)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel class, live hipEvent timing vs the MFMA / HBM roof
  cpu_baseline the CPU oracle (oracle/nunet_oracle.py, kind "port") timed on this host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this image needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise);
# the launcher environment normally carries it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
TRAIN_GFLOP_PER_IMG_96 = 29.081


def train_gflop_per_img(hw):
    """fwd + dgrad + wgrad of the fully convolutional net: proportional to the pixel count (x7.111 at 256, x28.44 at 512)."""
    return TRAIN_GFLOP_PER_IMG_96 * (hw / 96.0) ** 2


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--deep-supervision", action="store_true")
    ap.add_argument("--num-classes", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: min(affinity, 16) = the GPU box's CPU share")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 leg (the reference's own arithmetic, reported beside the bf16 line)")
    ap.add_argument("--fp32-steps", type=int, default=30)
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="rendezvous only: every rank joins a gloo group on the CPU, all-reduces its rank and rank 0 prints the ranks it saw "
                         "(exercises the launcher / self-spawn path without a GPU)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher: one child process per GPU, started BEFORE this process has made any
    GPU call (it never makes one: a process that touched the GPU must not be replaced or forked into ranks). Children get
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exactly as torch.distributed.run would set them; rank 0's stdout (the JSON
    line) is this process's stdout. Returns the worst child exit code."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NUNET_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        try:
            p.wait()
        except KeyboardInterrupt:
            for q in procs:
                q.terminate()
            raise
        rc = rc or p.returncode
    if rc:          # a rank died: do not leave its peers waiting in a collective
        for q in procs:
            if q.poll() is None:
                q.terminate()
    return rc


def dist_dry_run(rank, world):
    """Rendezvous check on the CPU (gloo): what a SCALE run needs to work before any kernel runs."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = 1
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dist_dry_run": True, "backend": "gloo", "n_gpus": world, "ranks_seen": int(t.sum()),
                          "world_size": dist.get_world_size()}))
    dist.destroy_process_group()


def cpu_baseline(args, budget_s):
    """Reference CPU path: the oracle restatement (stock torch fp32 on the host cores), bounded samples of
    (a) the workload of this line (same batch / size) and (b) BASELINE.json configs[0] (batch 8, 96x96)."""
    import numpy as np
    from oracle import nunet_oracle as O
    import nunet_amd
    synth = nunet_amd.synth
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = args.cpu_threads or min(cores, 16)
    torch.set_num_threads(cores)

    def sample(batch, size, budget):
        st = synth.closed_form_state(args.num_classes, 3, args.deep_supervision, True)
        net = O.OracleNet(st, args.num_classes, 3, args.deep_supervision)
        opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        img, msk = synth.synth_batch(batch, size, size, 3, args.num_classes, seed=1234)
        x, t = torch.from_numpy(img), torch.from_numpy(msk)
        O.train_step(net, opt, x, t)                       # warm-up (allocator, oneDNN primitives)
        times = []
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end and len(times) < 20:
            t0 = time.perf_counter()
            O.train_step(net, opt, x, t)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        return batch / med, "%d steps of bs=%d %dx%d fp32 train step (oracle/nunet_oracle.py, torch CPU), median %.0f ms/step" % (
            len(times), batch, size, size, med * 1e3)

    v, smp = sample(args.batch, args.size, budget_s * 0.6)
    out = {"value": v, "unit": "images/sec", "cores": cores, "kind": "port", "sample": smp}
    if not (args.batch == 8 and args.size == 96):
        v8, smp8 = sample(8, 96, budget_s * 0.4)
        out["configs0_bs8_96"] = {"value": v8, "unit": "images/sec", "sample": smp8}
    return out


def _sha16(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = ["conv3x3.hip", "elementwise.hip", "plan.hip", "common.h"]


def kernel_source_hash():
    """Identity of the kernels a committed PMC file was collected on (profiles/*_pmc_traffic.json carries it)."""
    d = os.path.join(ROOT, "pytorch_nested-unet_amd", "csrc")
    return _sha16([os.path.join(d, f) for f in KERNEL_SOURCES])


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one (nothing in this process has touched the GPU, torch is not even imported yet)
        sys.exit(self_launch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.dist_dry_run:
        return dist_dry_run(rank, world)
    global torch
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_dp = os.environ.get("NUNET_FORCE_DP") == "1"      # rehearse the N>1 code path (RCCL, three graphs) with one rank
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nunet_amd
    from nunet_amd import _lib as L
    from nunet_amd.trainer import TrainStep
    synth = nunet_amd.synth

    n, hw = args.batch, args.size

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(dtype, steps, warmup):
        """Build the replica at `dtype`, capture the step, W untimed + K timed steps between barrier + synchronize pairs."""
        torch.manual_seed(0)                                # identical replicas on every rank
        model = nunet_amd.archs.NestedUNet(args.num_classes, 3, args.deep_supervision, dtype=dtype).to(dev)
        model.train()
        ts = TrainStep(model, (n, 3, hw, hw), lr=1e-3, momentum=0.9, weight_decay=1e-4, use_graph=not args.no_graph)
        # pre-stage a small pool of synthetic batches in HBM (rank-distinct shards)
        pool = []
        for k in range(4):
            img, msk = synth.synth_batch(n, hw, hw, 3, args.num_classes, seed=1234 + 100 * rank + k)
            pool.append((torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev)))
        ts.capture(*pool[0])
        for k in range(warmup):
            ts.step(*pool[k % len(pool)])
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ts.step(*pool[k % len(pool)])
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return ts, dt

    ts, dt = timed_run(args.dtype, args.steps, args.warmup)
    loss, iou = ts.epoch_stats()
    dp_info = None
    if world > 1 or force_dp:
        dp_info = {"backend": dist.get_backend(), "rccl_ranks_seen": dist.get_world_size(), "layout": ts.dp_mode,
                   "layout_chosen_by": "measurement at capture (slowest rank decides)" if ts.dp_choice else "NUNET_DP_MODE / default",
                   "layout_ms": dict(ts.dp_choice) if ts.dp_choice else None,
                   "exchange_mb": ts._scratch.numel() * 4 / 1e6}

    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt
    executor = {"segmented": ts.segmented, "schedule": ts.schedule,
                "chosen_by": "timing at capture" if ts.executor_choice else ("data-parallel layout timing (dp.layout_ms)" if ts.dp_choice else "argument / environment"),
                "ms": {"%s/%s" % k: round(v, 4) for k, v in ts.executor_choice.items()} if ts.executor_choice else None}

    roofline = None
    if not args.no_roofline and rank == 0 and world == 1:     # single-process leg: no collectives inside
        # Live per-kernel timing of the same step: eager, single-lane issue (kernels alone on the device), every launch
        # through hipExtLaunchKernelGGL with a start/stop event pair = the dispatch's own begin/end timestamps (the
        # figures rocprofv3 --kernel-trace reports; profiles/r03_summary.md holds that trace of the same command).
        L.check(L.lib().nunet_plan_set_multistream(ts.pl.handle, 0), "set_multistream")
        for _ in range(2):
            ts._fwd_bwd(); ts._opt()
        torch.cuda.synchronize()
        reps = max(3, min(10, args.steps))
        L.profile_begin()
        for _ in range(reps):
            ts._fwd_bwd()
            ts._opt()
        torch.cuda.synchronize()
        prof = L.profile_end()
        tot = sum(e["ms"] for e in prof)
        prof.sort(key=lambda e: -e["ms"])
        top = prof[0]
        avg_ms = top["ms"] / top["launches"]
        tflops = top["flops"] / top["launches"] / (avg_ms * 1e-3) / 1e12
        gbs = top["bytes"] / top["launches"] / (avg_ms * 1e-3) / 1e9
        # which roof bounds the dominant class: time at the MFMA peak vs time at the HBM peak of its ALGORITHMIC work
        mfma_bound = top["flops"] > 0 and (top["flops"] / (PEAK[args.dtype] * 1e12)) >= (top["bytes"] / (HBM_PEAK_GBS * 1e9))
        if mfma_bound:
            roofline = {"bound": "mfma", "achieved": tflops, "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                        "frac": tflops / PEAK[args.dtype], "traffic": None}
        else:
            roofline = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "traffic": None}
        roofline["achieved_tflops"] = tflops
        roofline["achieved_gbs"] = gbs
        roofline["algorithmic_bytes_per_launch"] = top["bytes"] / top["launches"]
        roofline["algorithmic_flops_per_launch"] = top["flops"] / top["launches"]
        # HBM traffic of that class from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs,
        # FETCH_SIZE x2 correction of MI355X_MICROARCH.md): attached only when the file was collected on THESE kernel
        # sources and this workload; otherwise null (a stale figure is worse than none)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")))
            same = pmc.get("kernel_source_sha16") == kernel_source_hash() and pmc.get("workload") == [args.dtype, n, hw]
            rec = pmc.get("classes", {}).get(top["name"])
            if same and rec:
                roofline["traffic"] = rec["hbm_bytes_per_launch_corrected"]
                roofline["traffic_source"] = "profiles/r03_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
            else:
                roofline["traffic_note"] = "profiles/r03_pmc_traffic.json was collected on different kernel sources / workload: not attached"
        except Exception:
            pass
        roofline["kernel"] = top["name"]
        roofline["avg_launch_us"] = avg_ms * 1e3
        roofline["launches_per_step"] = top["launches"] / reps
        roofline["share_of_kernel_time"] = top["ms"] / tot
        roofline["timing"] = "per-dispatch begin/end timestamps (hipExtLaunchKernelGGL start/stop events), single lane, eager"
        roofline["sum_kernel_us_per_step"] = tot * 1e3 / reps
        roofline["overlap_factor"] = (tot / reps) / ms_per_step     # summed kernel time / wall time of the multi-lane graph step
        roofline["classes"] = [{"name": e["name"], "launches_per_step": e["launches"] / reps,
                                "us_per_step": e["ms"] * 1e3 / reps,
                                "tflops": (e["flops"] / (e["ms"] * 1e-3) / 1e12) if e["ms"] > 0 else 0.0,
                                "gbs": (e["bytes"] / (e["ms"] * 1e-3) / 1e9) if e["ms"] > 0 else 0.0} for e in prof]
        step_tflops = train_gflop_per_img(hw) * n / 1e3 / (ms_per_step * 1e-3)
        roofline["whole_step_tflops"] = step_tflops
        roofline["whole_step_frac_of_mfma_peak"] = step_tflops / PEAK[args.dtype]

    # The same workload in the REFERENCE's own arithmetic (fp32 storage, exact-fp32 MFMA 32x32x2; reference
    # finished/archs1.py:14-32 has no autocast anywhere, and north_star's 1e-4 parity bound is an fp32 bound): a second
    # timed region after the headline one, its own replica and graph, >= 20 steps. Reported beside the line, never as `value`.
    fp32 = None
    if rank == 0 and world == 1 and not args.no_fp32 and args.dtype != "fp32":
        del ts
        torch.cuda.empty_cache()
        k32 = max(20, args.fp32_steps)
        ts32, dt32 = timed_run("fp32", k32, 5)
        ms32 = dt32 / k32 * 1e3
        tf32 = train_gflop_per_img(hw) * n / 1e3 / (ms32 * 1e-3)
        fp32 = {"value": n * k32 / dt32, "unit": "images/sec", "ms_per_step": ms32, "steps": k32, "warmup": 5,
                "whole_step_tflops": tf32, "frac_of_fp32_mfma_peak": tf32 / PEAK["fp32"],
                "executor": "%s/%s" % (ts32.segmented, ts32.schedule),
                "note": "same workload and step, fp32 storage and exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's precision"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "images/sec NestedUNet 96x96 bs=16 train" if hw == 96 and n == 16 else
                      "images/sec NestedUNet %dx%d bs=%d train" % (hw, hw, n),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "NestedUNet(num_classes=%d, in=3, deep_supervision=%s) %dx%d per-GPU batch %d, "
                                   "%s storage / fp32 accumulate, BCEDiceLoss, SGD(1e-3, 0.9, wd 1e-4), "
                                   "fwd+loss+iou+bwd%s+sgd per step (BASELINE.json configs[%d])"
                                   % (args.num_classes, args.deep_supervision, hw, hw, n, args.dtype,
                                      "+allreduce" if world > 1 else "", 2 if args.deep_supervision else 1),
                       "global_batch": world * n, "parallelism": "dp%d" % world, "hip_graph": not args.no_graph},
            "final_loss": loss, "final_iou": iou,
            "roofline": roofline, "cpu_baseline": cpu, "fp32": fp32,
            # what a SCALE record needs to be checkable: the ranks the collective backend really joined, the step layout the
            # data-parallel TrainStep chose and the timings it chose by (None with one rank: no exchange in the step)
            "dp": dp_info,
            # how the captured step is executed (TrainStep picks between the multi-branch hipGraph and the flag-synchronised
            # list-scheduled lanes by timing both at capture)
            "executor": executor,
        }
        print(json.dumps(out))
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
