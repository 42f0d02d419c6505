import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import test_ops_gpu as T
L = T.L
dt = L.F16
n, h, w, c = 3, 8, 12, 64
g = torch.Generator().manual_seed(2)
bias = torch.randn(c, generator=g) * 0.3
ys = T.q(torch.randn(n, c, h, w, generator=g) * 0.7, dt)
gamma = 1 + 0.2 * torch.randn(c, generator=g); beta = 0.2 * torch.randn(c, generator=g)
rm = 0.1 * torch.randn(c, generator=g); rv = 0.5 + torch.rand(c, generator=g)
yb = T.nhwc(ys, dt); dd = ys.double()
stats = torch.cat([dd.sum((0, 2, 3)), (dd * dd).sum((0, 2, 3)), torch.zeros(14 * c, dtype=torch.float64)]).float().to("cuda")
a = torch.zeros((n, h, w, 160), dtype=T.tdt(dt), device="cuda")
pooled = torch.zeros((n, h // 2, w // 2, c), dtype=T.tdt(dt), device="cuda")
rmg, rvg = rm.clone().cuda(), rv.clone().cuda()
nbt = torch.tensor([4], dtype=torch.int64, device="cuda"); save = torch.zeros(2 * c, device="cuda")
bg, gg, beg = bias.cuda(), gamma.cuda(), beta.cuda()
d = L.BnFwdDesc(dt, n, h, w, c, L.ptr(yb), c, L.ptr(bg), L.ptr(stats), L.ptr(gg), L.ptr(beg), L.ptr(rmg), L.ptr(rvg), L.ptr(nbt), L.ptr(save), 1, 0.1, 1e-5,
                L.ptr(a, 32 * a.element_size()), 160, L.ptr(pooled), c)
L.check(L.lib().nunet_bn_relu_fwd(C.byref(d), L.stream()), "bn")
got = T.to_nchw(a, c, off=32)
P = T.to_nchw(pooled, c); R = F.max_pool2d(got, 2, 2)
bad = (P != R).nonzero()
print("mismatches", bad.shape[0], "of", P.numel())
for idx in bad[:12].tolist():
    nn, cc, yy, xx = idx
    print(idx, "pooled", float(P[nn, cc, yy, xx]), "ref", float(R[nn, cc, yy, xx]), "quad", got[nn, cc, 2*yy:2*yy+2, 2*xx:2*xx+2].flatten().tolist())
print("channels of mismatches:", sorted(set(bad[:, 1].tolist()))[:40])
