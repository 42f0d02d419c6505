"""conv3x3 forward with / without the BatchNorm statistics epilogue (same shapes), for a rocprofv3 kernel trace."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16
N = 16
for name, H, cin, cout in [("L0", 96, 32, 32), ("L0b", 96, 192, 32), ("L1", 48, 64, 64), ("L2", 24, 128, 128)]:
    s0 = torch.randn(N, H, H, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * cout * cin, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.zeros(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(16 * cout, device="cuda")
    for st in (0, 1):
        d = L.ConvDesc(dt, N, H, H, L.ptr(s0), cin, cin, None, 0, 0, L.ptr(w), None, L.ptr(y), cout, cout, None, 0, 0, 0, 0, 0, L.ptr(stats) if st else None)
        for _ in range(6):
            L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))
        torch.cuda.synchronize()
print("done")
