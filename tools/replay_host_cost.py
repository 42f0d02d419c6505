#!/usr/bin/env python3
"""Host cost of one hipGraph replay of the fused train step vs its GPU duration."""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("pytorch_nested-unet_amd")
archs = importlib.import_module("pytorch_nested-unet_amd.archs")
trainer = importlib.import_module("pytorch_nested-unet_amd.trainer")
synth = importlib.import_module("pytorch_nested-unet_amd.synth")
torch.manual_seed(0)
m = archs.NestedUNet(1, 3, False, dtype="bf16").cuda()
x, t = synth.synth_batch(16, 96, 96, 3, 1, seed=1)
x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
ts = trainer.TrainStep(m, (16, 3, 96, 96), lr=1e-3)
ts.capture(x, t)
for _ in range(20): ts.step()
torch.cuda.synchronize()
K = 100
host = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter(); ts.step(); host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print(f"host per replay: median {host[K//2]*1e3:.3f} ms  min {host[0]*1e3:.3f}  max {host[-1]*1e3:.3f}; loop {1e3*(t1-t0)/K:.3f} ms/step; with sync {1e3*(t2-t0)/K:.3f} ms/step")
# one isolated replay: GPU idle before, host cost and GPU span
for _ in range(3):
    torch.cuda.synchronize(); time.sleep(0.01)
    a = time.perf_counter(); ts.step(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print(f"isolated: host {1e3*(b-a):.3f} ms, until done {1e3*(c-a):.3f} ms")
