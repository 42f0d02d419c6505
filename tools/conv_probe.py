"""conv3x3 forward kernel durations on the NestedUNet layer shapes (bf16, N=16), for a rocprofv3 kernel trace."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16
N = int(os.environ.get("NB", "16"))
HW = int(os.environ.get("HW", "96"))
shapes = [("L0 32->32", 1, 32, 0, 32), ("L0 192->32", 1, 128, 64, 32), ("L1 64->64", 2, 64, 0, 64), ("L1 320->64", 2, 192, 128, 64),
          ("L2 128->128", 4, 128, 0, 128), ("L2 512->128", 4, 256, 256, 128), ("L3 256->256", 8, 256, 0, 256), ("L3 768->256", 8, 256, 512, 256),
          ("L4 512->512", 16, 512, 0, 512)]
for name, div, c0, c1, cout in shapes:
    H = HW // div
    s0 = torch.randn(N, H, H, c0, device="cuda").to(torch.bfloat16)
    s1 = torch.randn(N, H, H, max(c1, 16), device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * cout * (c0 + c1), device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.zeros(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(16 * cout, device="cuda")
    d = L.ConvDesc(dt, N, H, H, L.ptr(s0), c0, c0, L.ptr(s1) if c1 else None, c1, max(c1, 16), L.ptr(w), None, L.ptr(y), cout, cout, None, 0, 0, 0, 0, 0, L.ptr(stats))
    for _ in range(6):
        L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))
    torch.cuda.synchronize()
print("done")
