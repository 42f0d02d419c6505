"""wgrad kernel durations on the NestedUNet layer shapes, for a rocprofv3 kernel trace."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16
N = 16
for name, H, c0, c1, cout in [("L0 32", 96, 32, 0, 32), ("L0 192", 96, 128, 64, 32), ("L1 64", 48, 64, 0, 64), ("L1 320", 48, 192, 128, 64),
                              ("L2 128", 24, 128, 0, 128), ("L3 256", 12, 256, 0, 256), ("L4 512", 6, 512, 0, 512)]:
    cin = c0 + c1
    s0 = torch.randn(N, H, H, c0, device="cuda").to(torch.bfloat16)
    s1 = torch.randn(N, H, H, max(c1, 16), device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, H, cout, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(8 * 9 * cout * cin, device="cuda")
    d = L.WgradDesc(dt, N, H, H, L.ptr(s0), c0, c0, L.ptr(s1) if c1 else None, c1, max(c1, 16), L.ptr(dy), cout, cout, L.ptr(dw))
    for _ in range(6):
        L.check(L.lib().nunet_conv3x3_wgrad(C.byref(d), L.stream()))
    torch.cuda.synchronize()
print("done")
