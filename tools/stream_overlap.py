"""Experiment: do independent conv launches on 2-4 streams overlap (eager and hipGraph)?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
DEV = "cuda:0"; dt = L.BF16; N = 16
def mk(H, cin, cout):
    s0 = torch.randn(N, H, H, cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(9 * cout * cin, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.zeros(N, H, H, cout, device=DEV, dtype=torch.bfloat16)
    d = L.ConvDesc(dt, N, H, H, L.ptr(s0), cin, cin, None, 0, 0, L.ptr(w), None, L.ptr(y), cout, cout, None, 0, 0, 0, 0, 0, None)
    return (s0, w, y, d)
jobs = {"L0": mk(96, 96, 32), "L1": mk(48, 192, 64), "L2": mk(24, 384, 128), "L3": mk(12, 768, 256)}
def run(name, reps=10):
    d = jobs[name][3]
    for _ in range(reps):
        L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
streams = [torch.cuda.Stream() for _ in range(4)]
def serial():
    for n in jobs: run(n)
def parallel():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for s, n in zip(streams, jobs):
        with torch.cuda.stream(s): run(n)
    for s in streams: cur.wait_stream(s)
for n in jobs:
    print(n, "alone x10: %.0f us" % timed(lambda: run(n)))
print("serial  eager: %.0f us" % timed(serial))
print("4-stream eager: %.0f us" % timed(parallel))
def cap(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
g1 = cap(serial); g2 = cap(parallel)
print("serial  graph: %.0f us" % timed(g1.replay))
print("4-branch graph: %.0f us" % timed(g2.replay))
