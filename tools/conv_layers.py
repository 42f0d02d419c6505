"""Per-layer conv3x3 timings on the NestedUNet 96x96 bs16 shapes: forward conv1 (plain, two sources), conv2 (BatchNorm
input transform), dgrad2 (BN-backward input transform + fused reduce epilogue), dgrad1 (BN-backward input transform,
split destination). Each case is captured into a hipGraph of R launches and replayed: us per launch incl. the ~3 us
same-stream node gap the real step also pays.   NUNET_CONV_SMALL=1 python tools/conv_layers.py"""
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

def t(*shape, scale=1.0):
    x = (torch.randn(*shape, device="cuda") * scale).to(bf); keep.append(x); return x

def f32(n, fill=None):
    x = torch.randn(n, device="cuda") if fill is None else torch.full((n,), fill, device="cuda"); keep.append(x); return x

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

def conv(H, c0, c1, cout, lt=0, bnr=False, d1=0, ws=None):
    d = L.ConvDesc()
    d.dtype = dt; d.N = N; d.H = H; d.W = H
    tl = int(os.environ.get("TILE", "0"))
    d.tile = tl if (tl not in (2, 4) or cout % 64 == 0) else 0
    s0 = t(N, H, H, c0); d.src0 = L.ptr(s0).value; d.C0 = c0; d.P0 = c0
    if c1:
        s1 = t(N, H, H, c1); d.src1 = L.ptr(s1).value; d.C1 = c1; d.P1 = c1
    w = t(9 * cout * (c0 + c1), scale=0.05); d.wpack = L.ptr(w).value
    y = t(N, H, H, cout - d1); d.dst0 = L.ptr(y).value; d.D0 = cout - d1; d.Q0 = cout - d1
    if d1:
        y1 = t(N, H, H, d1); d.dst1 = L.ptr(y1).value; d.D1 = d1; d.Q1 = d1
    if ws is not None:
        d.splitk_ws = L.ptr(ws).value; d.splitk_ws_floats = ws.numel()
    if lt == 0 or lt == 1:
        st = L.fx_zeros(cout, "cuda"); keep.append(st); d.stats = L.ptr(st).value
    if lt:
        cin = c0
        g, b = f32(cin, 1.0), f32(cin, 0.0)
        d.in_tf = lt; d.tf_gamma = L.ptr(g).value; d.tf_beta = L.ptr(b).value
        fx = L.fx_zeros(cin, "cuda"); keep.append(fx); d.tf_fx = L.ptr(fx).value
        mi = f32(2 * cin, 1.0); d.tf_mean_invstd = L.ptr(mi).value
        store = t(N, H, H, cin); d.tf_store = L.ptr(store).value; d.tf_ps = cin
        d.tf_eps = 1e-5; d.tf_momentum = 0.1
        if lt == 1:
            d.tf_training = 1
            # plausible sums: mean 0, E[x^2] = 1
            v = torch.zeros(2 * cin, dtype=torch.float64); v[cin:] = N * H * H
            fx2 = L.fx_encode(v, cin, "cuda"); keep.append(fx2); d.tf_fx = L.ptr(fx2).value
        else:
            yy = t(N, H, H, cin); d.tf_y = L.ptr(yy).value; d.tf_py = cin
    if bnr:
        y1 = t(N, H, H, cout); d.bn_y = L.ptr(y1).value; d.bn_py = cout
        mi2 = f32(2 * cout, 1.0); d.bn_mean_invstd = L.ptr(mi2).value
        g2, b2 = f32(cout, 1.0), f32(cout, 0.0); d.bn_gamma = L.ptr(g2).value; d.bn_beta = L.ptr(b2).value
        sm = L.fx_zeros(cout, "cuda"); keep.append(sm); d.bn_sums = L.ptr(sm).value
    keep.append(d)
    return time_graph(lambda: L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream())))

ONLY = os.environ.get("ONLY")   # "i,j,k": level, column, kind (0 conv1, 1 conv2, 2 dgrad2, 3 dgrad1) - one case, R*13 eager-ish launches (for PC sampling)
print("%-8s %10s %10s %10s %10s   (us per launch | TFLOP/s, bf16 N=%d %dx%d, TILE=%s)" % ("block", "conv1", "conv2", "dgrad2", "dgrad1", N, HW, HW, os.environ.get("TILE", "0")))
MAXLEV = int(os.environ.get("MAXLEV", "5"))
tot = [0.0] * 4
for i in range(min(5, MAXLEV)):
    H = HW >> i; f = NBF[i]
    ws = torch.zeros(8 * N * H * H * (5 * f), device="cuda") if i >= 3 else None
    for j in range(5 - i):
        if j == 0:
            c0, c1 = (32 if i == 0 else NBF[i - 1]), 0
        else:
            c0, c1 = j * f, NBF[i + 1]
        if ONLY:
            oi, oj, ok = (int(v) for v in ONLY.split(","))
            if (oi, oj) != (i, j): continue
            v = [lambda: conv(H, c0, c1, f, ws=ws), lambda: conv(H, f, 0, f, lt=1, ws=ws), lambda: conv(H, f, 0, f, lt=2, bnr=True, ws=ws),
                 lambda: conv(H, f, 0, c0 + c1, lt=2, d1=c1, ws=ws)][ok]()
            print("B%d%d kind %d: %.1f us" % (i, j, ok, v))
            if os.environ.get("KSTAMP"):
                # diagnostic library (tools/kstamp_build.sh): phase stamps of wave 0 of the first WGS workgroups, ONE eager launch
                import numpy as np
                WGS = 2048
                buf = torch.zeros(WGS * 32, dtype=torch.int64, device="cuda")
                fn = L.lib().nunet_kstamp_set; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
                assert fn(buf.data_ptr(), WGS) == 0
                d = [k for k in keep if isinstance(k, L.ConvDesc)][-1]
                torch.cuda.synchronize()
                L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream())); torch.cuda.synchronize()
                st = buf.cpu().numpy().reshape(WGS, 32)
                live = st[:, 0] > 0
                st = st[live].astype(np.float64)
                t0 = st[:, 0].min()
                # slots 30 / 31: chip-wide 100 MHz counter at workgroup entry / exit
                rt0, rt1 = st[:, 30], st[:, 31]
                base = rt0.min()
                print("workgroups stamped: %d | entry after first (us): p50 %.2f p90 %.2f max %.2f | exit (us): p10 %.2f p50 %.2f p90 %.2f max %.2f | life (us) p50 %.2f | s_memtime cycles per us: %.0f"
                      % (live.sum(), *(np.percentile(rt0 - base, q) / 100 for q in (50, 90, 100)), *(np.percentile(rt1 - base, q) / 100 for q in (10, 50, 90, 100)),
                         np.median(rt1 - rt0) / 100, np.median((st[:, :30].max(axis=1) - st[:, 0]) / np.maximum(rt1 - rt0, 1) * 100)))
                names = ["entry", "tables", "coef0", "loop"] + [x + str(n) for n in range(5) for x in "abcdeghijkf"]
                rel = st - t0
                print("%-8s %10s %10s %10s   %s" % ("stamp", "median", "p10", "p90", "median delta to previous"))
                prev = None
                for k in range(30):
                    col = rel[:, k][st[:, k] > 0]
                    if col.size == 0: continue
                    dl = "" if prev is None else "%.0f" % np.median((st[:, k] - st[:, prev])[(st[:, k] > 0) & (st[:, prev] > 0)])
                    print("%-8s %10.0f %10.0f %10.0f   %s   (n=%d)" % (names[k] if k < len(names) else str(k), np.median(col), np.percentile(col, 10), np.percentile(col, 90), dl, col.size))
                    prev = k
            continue
        a = conv(H, c0, c1, f, ws=ws)
        b = conv(H, f, 0, f, lt=1, ws=ws)
        c = conv(H, f, 0, f, lt=2, bnr=True, ws=ws)
        if i == 0 and j == 0:
            dd = 0.0
        else:
            dd = conv(H, f, 0, c0 + c1, lt=2, d1=c1, ws=ws)
        fl = lambda cin, cout, us: 2.0 * 9 * cin * cout * N * H * H / (us * 1e6) if us > 0 else 0.0
        print("B%d%d      %10.1f %10.1f %10.1f %10.1f   | %6.0f %6.0f %6.0f %6.0f" % (i, j, a, b, c, dd, fl(c0 + c1, f, a), fl(f, f, b), fl(f, f, c), fl(f, c0 + c1, dd)))
        keep.clear(); torch.cuda.empty_cache()
        for k, v in enumerate((a, b, c, dd)): tot[k] += v
if not ONLY: print("sum      %10.1f %10.1f %10.1f %10.1f   total %.1f" % (*tot, sum(tot)))
