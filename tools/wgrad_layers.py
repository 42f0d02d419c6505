"""Per-block weight-gradient pair timings (both convs of a VGGBlock in one launch) on the NestedUNet 96x96 bs16 shapes,
hipGraph replay of R launches: us per launch incl. the same-stream node gap, TFLOP/s of the pair.
   NUNET_WG_T1 / NUNET_WG_T2: workgroup targets of the two problems (default: plan policy)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16
N = int(os.environ.get("NB", "16")); HW = int(os.environ.get("HW", "96")); R = 20
NBF = [32, 64, 128, 256, 512]
bf = torch.bfloat16
keep = []

def t(*shape):
    x = torch.randn(*shape, device="cuda").to(bf); keep.append(x); return x

def time_graph(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(R): fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (10 * R)

def desc(H, c0, c1, cout, target):
    s0 = t(N, H, H, c0); s1 = t(N, H, H, c1) if c1 else None; dy = t(N, H, H, cout)
    d = L.WgradDesc(dt, N, H, H, L.ptr(s0), c0, c0, L.ptr(s1), c1, c1, L.ptr(dy), cout, cout, None, 9 * cout * (c0 + c1), 0, target, 0, int(os.environ.get("ITEM_SHAPE", "0")))
    ks = L.lib().nunet_conv3x3_wgrad_slabs(C.byref(d))
    slabs = torch.empty(ks * 9 * cout * (c0 + c1), dtype=torch.float32, device="cuda"); keep.append(slabs)
    d.dw = L.ptr(slabs).value; d.max_slabs = ks; d.dw_floats = slabs.numel()
    keep.append(d)
    return d, ks

print("%-6s %8s %6s %6s %9s   (bf16 N=%d %dx%d)" % ("block", "us", "ks1", "ks2", "TFLOP/s", N, HW, HW))
tot = 0.0
for i in range(5):
    H = HW >> i; f = NBF[i]
    for j in range(5 - i):
        if j == 0:
            c0, c1 = (32 if i == 0 else NBF[i - 1]), 0
        else:
            c0, c1 = j * f, NBF[i + 1]
        cin1 = c0 + c1
        t1 = int(os.environ.get("NUNET_WG_T1", "0")) or (128 if cin1 < f else 256)
        t2 = int(os.environ.get("NUNET_WG_T2", "0")) or (128 if f < cin1 else 256)
        d1, k1 = desc(H, c0, c1, f, t1)
        d2, k2 = desc(H, f, 0, f, t2)
        us = time_graph(lambda: L.check(L.lib().nunet_conv3x3_wgrad_pair(C.byref(d1), C.byref(d2), L.stream())))
        fl = 2.0 * 9 * (cin1 + f) * f * N * H * H
        print("B%d%d    %8.1f %6d %6d %9.1f" % (i, j, us, k1, k2, fl / us / 1e6))
        tot += us
print("sum    %8.1f" % tot)
