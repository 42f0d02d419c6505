"""Do two INDEPENDENT conv launches overlap when they run side by side (two streams, eager)? If a pair takes ~max(a, b) the
kernels leave room for each other (latency-bound) and grouping independent problems into one launch would pay; ~a + b: not."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16; N = 16; bf = torch.bfloat16; keep = []

def t(*shape, scale=1.0):
    x = (torch.randn(*shape, device="cuda") * scale).to(bf); keep.append(x); return x

def mk(H, c0, c1, cout, lt=0):
    d = L.ConvDesc(); d.dtype = dt; d.N = N; d.H = H; d.W = H
    s0 = t(N, H, H, c0); d.src0 = L.ptr(s0).value; d.C0 = c0; d.P0 = c0
    if c1:
        s1 = t(N, H, H, c1); d.src1 = L.ptr(s1).value; d.C1 = c1; d.P1 = c1
    w = t(9 * cout * (c0 + c1), scale=0.05); d.wpack = L.ptr(w).value
    y = t(N, H, H, cout); d.dst0 = L.ptr(y).value; d.D0 = cout; d.Q0 = cout
    st = L.fx_zeros(cout, "cuda"); keep.append(st); d.stats = L.ptr(st).value
    if lt == 1:
        cin = c0
        g, b = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"); keep.extend([g, b])
        d.in_tf = 1; d.tf_gamma = L.ptr(g).value; d.tf_beta = L.ptr(b).value; d.tf_training = 1; d.tf_eps = 1e-5; d.tf_momentum = 0.1
        v = torch.zeros(2 * cin, dtype=torch.float64); v[cin:] = N * H * H
        fx = L.fx_encode(v, cin, "cuda"); keep.append(fx); d.tf_fx = L.ptr(fx).value
        mi = torch.ones(2 * cin, device="cuda"); keep.append(mi); d.tf_mean_invstd = L.ptr(mi).value
    keep.append(d)
    return d

def run(d, s):
    with torch.cuda.stream(s):
        L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))

def timed(descs, reps=30):
    ss = [torch.cuda.Stream() for _ in descs]
    for _ in range(3):
        for d, s in zip(descs, ss): run(d, s)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for s in ss: s.wait_event(e0)
        for d, s in zip(descs, ss): run(d, s)
        for s in ss: torch.cuda.current_stream().wait_stream(s)
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / reps

cases = {"B02.conv1 L0 128->32": mk(96, 96, 32, 32), "B21.conv1 L2 384->128": mk(24, 256, 128, 128), "B40.conv1 L4 256->512": mk(6, 256, 0, 512),
         "B11.conv1 L1 192->64": mk(48, 64, 128, 64), "B01.conv2 L0 32->32 bn": mk(96, 32, 0, 32, lt=1), "B20.conv2 L2 128->128 bn": mk(24, 128, 0, 128, lt=1)}
alone = {k: timed([d]) for k, d in cases.items()}
for k, v in alone.items(): print("%-28s alone %.1f us (incl. ~8 us of event / launch overhead)" % (k, v))
names = list(cases)
for a, b in ((0, 1), (0, 2), (1, 2), (0, 3), (4, 5), (0, 4), (1, 3)):
    v = timed([cases[names[a]], cases[names[b]]])
    print("%-24s + %-24s together %.1f us   (sum %.1f, max %.1f)" % (names[a], names[b], v, alone[names[a]] + alone[names[b]], max(alone[names[a]], alone[names[b]])))
v = timed([cases[names[0]], cases[names[1]], cases[names[2]]])
print("three (B02, B21, B40 conv1) together %.1f us (sum %.1f)" % (v, sum(alone[names[i]] for i in range(3))))
