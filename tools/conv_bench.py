"""Micro-benchmark of the conv kernels on the NestedUNet layer shapes (bf16, N=16, 96x96)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
DEV = "cuda:0"
dt = L.BF16
N = int(os.environ.get('NB', '16'))
shapes = [  # (name, H, C0, C1, Cout)
    ("L0 32->32", 96, 32, 0, 32), ("L0 96->32", 96, 32, 64, 32), ("L0 192->32", 96, 128, 64, 32),
    ("L0 dgrad 32->192", 96, 32, 0, 192), ("L0 dgrad 32->96", 96, 32, 0, 96),
    ("L1 64->64", 48, 64, 0, 64), ("L1 320->64", 48, 192, 128, 64), ("L1 dgrad 64->320", 48, 64, 0, 320),
    ("L2 128->128", 24, 128, 0, 128), ("L2 512->128", 24, 256, 256, 128),
    ("L3 256->256", 12, 256, 0, 256), ("L3 768->256", 12, 256, 512, 256),
    ("L4 256->512", 6, 256, 0, 512), ("L4 512->512", 6, 512, 0, 512),
]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if os.environ.get("ONLY"):
    shapes = [s_ for s_ in shapes if any(s_[0].startswith(o) for o in os.environ["ONLY"].split(","))]
reps = 30
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, H, c0, c1, cout in shapes:
    cin = c0 + c1
    s0 = torch.randn(N, H, H, c0 + 32, device=DEV).to(torch.bfloat16)
    s1 = torch.randn(N, H, H, max(c1, 16), device=DEV).to(torch.bfloat16)
    w = (torch.randn(9 * cout * cin, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.zeros(N, H, H, cout, device=DEV, dtype=torch.bfloat16)
    stats = torch.zeros(8 * 2 * cout, device=DEV)
    dy = torch.randn(N, H, H, cout, device=DEV).to(torch.bfloat16)
    dw = torch.zeros(9 * cout * cin, device=DEV)
    gf = 2 * 9 * cin * cout * N * H * H / 1e9
    mb = (N * H * H * (cin + cout) * 2 + 9 * cin * cout * 2) / 1e6
    if which == "fwd":
        d = L.ConvDesc(dt, N, H, H, L.ptr(s0), c0, c0 + 32, L.ptr(s1) if c1 else None, c1, max(c1, 16), L.ptr(w), None,
                       L.ptr(y), cout, cout, None, 0, 0, 0, 0, 0, L.ptr(stats))
        if os.environ.get("SK"):
            ws = torch.zeros(8 * N * H * H * cout + 256, device=DEV)
            d.splitk_ws = L.ptr(ws).value; d.splitk_ws_floats = ws.numel()
        fn = lambda: L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))
    else:
        d = L.WgradDesc(dt, N, H, H, L.ptr(s0), c0, c0 + 32, L.ptr(s1) if c1 else None, c1, max(c1, 16), L.ptr(dy), cout, cout, L.ptr(dw))
        fn = lambda: L.check(L.lib().nunet_conv3x3_wgrad(C.byref(d), L.stream()))
    us = timeit(fn)
    print("%-20s %6.2f GF %6.1f MB | %7.1f us %6.1f TF %6.0f GB/s" % (name, gf, mb, us, gf / us * 1e3 / 1e3 * 1e0 if False else gf / (us * 1e-6) / 1e3, mb / (us * 1e-6) / 1e3))
