"""conv3x3 with / without the fused BN-backward reduce epilogue, same shapes, for a rocprofv3 kernel trace."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
dt = L.BF16
N = 16
for name, H, cin, cout in [("L0", 96, 32, 32), ("L1", 48, 64, 64), ("L2", 24, 128, 128), ("L3", 12, 256, 256)]:
    s0 = torch.randn(N, H, H, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * cout * cin, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.zeros(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    y1 = torch.randn(N, H, H, cout, device="cuda").to(torch.bfloat16)
    mi = torch.cat([torch.zeros(cout), torch.ones(cout)]).cuda()
    gamma, beta = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    sums = torch.zeros(8 * 2 * cout, device="cuda")
    for bnr in (0, 1):
        d = L.ConvDesc(dt, N, H, H, L.ptr(s0), cin, cin, None, 0, 0, L.ptr(w), None, L.ptr(y), cout, cout, None, 0, 0, 0, 0, 0, None)
        if bnr:
            d.bn_y = L.ptr(y1).value; d.bn_py = cout; d.bn_mean_invstd = L.ptr(mi).value
            d.bn_gamma = L.ptr(gamma).value; d.bn_beta = L.ptr(beta).value; d.bn_sums = L.ptr(sums).value
        for _ in range(6):
            L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()))
        torch.cuda.synchronize()
    dummy = torch.zeros(cout, device="cuda"); dyb = torch.zeros_like(y)
    b = L.BnBwdDesc(dt, N, H, H, cout, L.ptr(y), cout, L.ptr(y1), cout, L.ptr(mi), L.ptr(gamma), L.ptr(beta), L.ptr(sums), L.ptr(dummy), L.ptr(dummy), L.ptr(dummy), L.ptr(dyb), cout)
    for _ in range(6):
        L.check(L.lib().nunet_bn_relu_bwd_reduce(C.byref(b), L.stream()))
    torch.cuda.synchronize()
print("done")
