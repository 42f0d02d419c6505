#!/usr/bin/env python3
"""Where the data-parallel step spends its time on ONE GPU (single-rank RCCL group, NUNET_FORCE_DP=1):
each of the three graphs alone, the exchanges alone, and the whole step, against the single-graph step.

  NUNET_FORCE_DP=1 python3 tools/dp_probe.py
"""
import os, sys, time, importlib
os.environ.setdefault("NUNET_FORCE_DP", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
archs = importlib.import_module("pytorch_nested-unet_amd.archs")
trainer = importlib.import_module("pytorch_nested-unet_amd.trainer")
synth = importlib.import_module("pytorch_nested-unet_amd.synth")


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.manual_seed(0)
    m = archs.NestedUNet(1, 3, False, dtype="bf16").cuda()
    x, t = synth.synth_batch(16, 96, 96, 3, 1, seed=1)
    x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    ts = trainer.TrainStep(m, (16, 3, 96, 96), lr=1e-3)
    ts.capture(x, t)
    b0, b1 = ts._buckets
    print("buckets MB: %.1f %.1f" % (b0.numel() * 4 / 1e6, b1.numel() * 4 / 1e6))
    print("whole DP step      %8.1f us" % timeit(lambda: ts.step()))
    for name in ("g_fb", "g_b2", "g_opt"):
        g = getattr(ts, name)
        if g is not None: print("%-18s %8.1f us" % (name + " alone", timeit(g.replay)))
    print("exchange b0        %8.1f us" % timeit(lambda: dist.all_reduce(b0)))
    print("exchange b1        %8.1f us" % timeit(lambda: dist.all_reduce(b1)))
    def no_comm():
        ts.g_fb.replay()
        if ts.g_b2 is not None: ts.g_b2.replay()
        ts.g_opt.replay()
    print("graphs, no exchange %7.1f us" % timeit(no_comm))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
