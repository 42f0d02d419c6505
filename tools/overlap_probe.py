"""Do a latency-bound conv chain and weight-gradient launches overlap when they sit on different streams of one
hipGraph? A: 20 dependent L0 32->32 convs; B: 8 independent wgrad pairs (L0 192->32 + 32->32). Times A, B, A||B."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16; bf = torch.bfloat16
N, H = 16, int(os.environ.get("HW", "96"))
keep = []
def t(*s):
    x = torch.randn(*s, device="cuda").to(bf); keep.append(x); return x
def conv_desc(c0, cout):
    d = L.ConvDesc(); d.dtype = dt; d.N = N; d.H = H; d.W = H
    s0 = t(N, H, H, c0); d.src0 = L.ptr(s0).value; d.C0 = c0; d.P0 = c0
    w = t(9 * cout * c0); d.wpack = L.ptr(w).value
    y = t(N, H, H, cout); d.dst0 = L.ptr(y).value; d.D0 = cout; d.Q0 = cout
    st = L.fx_zeros(cout, "cuda"); keep.append(st); d.stats = L.ptr(st).value
    keep.append(d); return d
def wg_desc(c0, c1, cout, target):
    s0 = t(N, H, H, c0); s1 = t(N, H, H, c1) if c1 else None; dy = t(N, H, H, cout)
    d = L.WgradDesc(dt, N, H, H, L.ptr(s0), c0, c0, L.ptr(s1), c1, c1, L.ptr(dy), cout, cout, None, 9 * cout * (c0 + c1), 0, target)
    ks = L.lib().nunet_conv3x3_wgrad_slabs(C.byref(d))
    sl = torch.empty(ks * 9 * cout * (c0 + c1), dtype=torch.float32, device="cuda"); keep.append(sl)
    d.dw = L.ptr(sl).value; d.max_slabs = ks; keep.append(d); return d
cd = conv_desc(32, 32)
T1, T2 = int(os.environ.get("T1", "256")), int(os.environ.get("T2", "128"))
w1, w2 = wg_desc(128, 64, 32, T1), wg_desc(32, 0, 32, T2)
NA_, NB_ = int(os.environ.get('NA', '20')), int(os.environ.get('NBP', '8'))
def A():
    for _ in range(NA_): L.check(L.lib().nunet_conv3x3_fwd(C.byref(cd), L.stream()))
def B():
    for _ in range(NB_): L.check(L.lib().nunet_conv3x3_wgrad_pair(C.byref(w1), C.byref(w2), L.stream()))
def graph_of(fa, fb):
    s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
    for f in (fa, fb):
        if f: f()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if fa and fb:
                ev = torch.cuda.Event(); ev.record(s)
                s2.wait_event(ev)
                with torch.cuda.stream(s2):
                    fb()
                    ev2 = torch.cuda.Event(); ev2.record(s2)
                fa()
                s.wait_event(ev2)
                L.check(L.lib().nunet_debug_spin(0, 1, L.stream()))      # a node after the join, so that the join is a real edge
            elif fa: fa()
            else: fb()
    return g
def timeit(g):
    import time
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / 20
ta, tb = timeit(graph_of(A, None)), timeit(graph_of(None, B))
gab = graph_of(A, B)
sl = keep[[i for i, k in enumerate(keep) if isinstance(k, torch.Tensor) and k.dtype == torch.float32][0]]
sl.zero_(); torch.cuda.synchronize(); gab.replay(); torch.cuda.synchronize()
print("B ran inside the combined graph:", bool(sl.abs().sum() > 0))
tab = timeit(gab)
print("NA=%d NB=%d " % (NA_, NB_) + "T=%d,%d  A (convs) %.0f us   B (wgrad pairs) %.0f us   A||B %.0f us   sum %.0f   max %.0f" % (T1, T2, ta, tb, tab, ta + tb, max(ta, tb)))
