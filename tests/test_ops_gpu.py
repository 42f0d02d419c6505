"""Per-kernel parity: every C-ABI entry of libnunet.so against the CPU oracle ops
(stock torch fp32/fp64 on CPU) on seeded inputs. Runs on the MI355X only."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import nunet_amd  # noqa: E402
from nunet_amd import _lib as L  # noqa: E402

@pytest.fixture(autouse=True)
def _canaries(guard_bands):
    """every device buffer these tests allocate sits between guard bands that are checked after the test (conftest.py)"""
    yield


DEV = "cuda:0"
TOL = {L.F32: 2e-5, L.BF16: 1.2e-2, L.F16: 2e-3}   # relative to the output's max |value|
DT = {L.F32: "fp32", L.BF16: "bf16", L.F16: "fp16"}


def tdt(dt):
    return L.TORCH_DTYPE[dt]


def nhwc(x, dt, pitch=None, off=0):
    """NCHW fp32 CPU tensor -> NHWC `dt` GPU buffer with optional channel pitch/offset.
    Returns (buffer, view_of_the_channels)."""
    n, c, h, w = x.shape
    pitch = pitch or c
    buf = torch.zeros((n, h, w, pitch), dtype=tdt(dt), device=DEV)
    buf[..., off:off + c] = x.permute(0, 2, 3, 1).to(DEV).to(tdt(dt))
    return buf


def to_nchw(buf, c, off=0):
    return buf[..., off:off + c].float().permute(0, 3, 1, 2).cpu()


def q(x, dt):
    """round-trip through the storage dtype (what the kernel actually sees)"""
    return x.to(tdt(dt)).float()


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def pack(w, dt, cin_pad=None, want_wd=True):
    cout, cin = w.shape[:2]
    cin_pad = cin_pad or cin
    wg = w.contiguous().to(DEV)
    wf = torch.zeros(9 * cout * cin_pad, dtype=tdt(dt), device=DEV)
    wd = torch.zeros(9 * cout * cin, dtype=tdt(dt), device=DEV) if want_wd else None
    L.check(L.lib().nunet_pack_weights(L.ptr(wg), cout, cin, cin_pad, dt, L.ptr(wf), L.ptr(wd), L.stream()), "pack")
    return wf, wd


def conv_desc(dt, n, h, w, src0, c0, p0, wpack, dst0, d0, q0, src1=None, c1=0, p1=0, bias=None,
              dst1=None, d1=0, q1=0, slot_w=0, mask=0, acc1=0, stats=None):
    return L.ConvDesc(dt, n, h, w, L.ptr(src0), c0, p0, L.ptr(src1), c1, p1, L.ptr(wpack), L.ptr(bias),
                      L.ptr(dst0), d0, q0, L.ptr(dst1), d1, q1, slot_w, mask, acc1, L.ptr(stats))


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("shape", [
    (2, 20, 24, 32, 0, 32),     # BN=32 config, ragged tiles
    (1, 16, 16, 64, 64, 64),    # two sources, BN=64 config
    (5, 6, 6, 32, 0, 64),       # multi-image tiles
    (2, 8, 40, 96, 0, 32),      # three channel chunks from one source
    (1, 2, 2, 64, 32, 96),      # tiny spatial, Cout = 96 -> BN=32 config
    (3, 1, 1, 32, 0, 32),       # 1x1 images (level 4 of a 16x16 input)
    (16, 12, 12, 64, 32, 64),   # 12x12 images: stacked-rows tiling (level 3 of the 96x96 workload)
    (5, 12, 12, 32, 0, 32),     # stacked rows, BM=256 config, ragged last tile
    (7, 3, 5, 32, 0, 64),       # stacked rows on odd tiny images
])
def test_conv3x3_fwd(dt, shape):
    n, h, w, c0, c1, cout = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    cin = c0 + c1
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    # source 0 lives in a wider level buffer (pitch > C0), source 1 is dense
    s0 = nhwc(x[:, :c0], dt, pitch=c0 + 32, off=0)
    s1 = nhwc(x[:, c0:], dt) if c1 else None
    wf, _ = pack(wt, dt)
    y = torch.full((n, h, w, cout), 7.0, dtype=tdt(dt), device=DEV)
    stats = L.fx_zeros(cout, DEV)
    bg = b.to(DEV)
    d = conv_desc(dt, n, h, w, s0, c0, c0 + 32, wf, y, cout, cout, src1=s1, c1=c1, p1=c1, bias=bg, stats=stats)
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "conv")
    ref = F.conv2d(q(x, dt).double(), q(wt, dt).double(), b.double(), padding=1)
    got = to_nchw(y, cout)
    assert rel_err(got, ref) < TOL[dt], (DT[dt], shape)
    # BN partial sums are taken about the bias on the rounded outputs
    dd = got.double() - b.double().view(1, -1, 1, 1)
    s = L.fx_decode(stats, cout)
    m = n * h * w
    np.testing.assert_allclose(s[:cout].numpy() / m, dd.sum((0, 2, 3)).numpy() / m, atol=1e-4 * float(dd.abs().max()) + 1e-6)
    np.testing.assert_allclose(s[cout:].numpy() / m, (dd * dd).sum((0, 2, 3)).numpy() / m, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_conv3x3_dgrad_split_accumulate(dt):
    """dgrad = conv with the flipped/transposed pack; output split over two destinations,
    slot-wise accumulate mask on the first (the zero-copy concat gradient)."""
    n, h, w, cin, cout = 2, 12, 20, 96, 32      # forward conv 96 -> 32; dgrad yields 96 = 64 (2 slots) + 32
    g = torch.Generator().manual_seed(3)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    dy = torch.randn(n, cout, h, w, generator=g)
    prev = torch.randn(n, 64, h, w, generator=g)
    _, wd = pack(wt, dt)
    dyb = nhwc(dy, dt)
    gx = nhwc(prev, dt, pitch=160, off=0)        # level grad buffer, 5 slots of 32
    gup = torch.zeros((n, h, w, 32), dtype=tdt(dt), device=DEV)
    d = conv_desc(dt, n, h, w, dyb, cout, cout, wd, gx, 64, 160, dst1=gup, d1=32, q1=32, slot_w=32, mask=0b10)
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "dgrad")
    ref = F.conv_transpose2d(q(dy, dt).double(), q(wt, dt).double(), padding=1)
    got0 = to_nchw(gx, 64)
    exp0 = ref[:, :64].clone()
    exp0[:, 32:64] += q(prev, dt)[:, 32:64].double()     # slot 1 accumulates, slot 0 overwrites
    assert rel_err(got0, exp0) < TOL[dt]
    assert rel_err(to_nchw(gup, 32), ref[:, 64:]) < TOL[dt]
    assert float(gx[..., 64:].float().abs().max()) == 0.0      # other slots untouched


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("shape", [
    (2, 20, 24, 32, 0, 32, 32),
    (1, 16, 16, 64, 64, 64, 64),
    (5, 6, 6, 64, 32, 64, 32),      # Cin = 96: ci tile straddles the two sources
    (2, 8, 8, 32, 0, 32, 32),       # first-layer style: real Cin=3 padded to 32 handled by caller
    (3, 1, 1, 32, 0, 64, 32),
    (16, 12, 12, 64, 32, 64, 32),   # stacked-rows tiling
    (7, 3, 5, 32, 0, 32, 32),
])
def test_conv3x3_wgrad(dt, shape):
    n, h, w, c0, c1, cout, _ = shape
    g = torch.Generator().manual_seed(11)
    cin = c0 + c1
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    s0 = nhwc(x[:, :c0], dt, pitch=c0 + 64)
    s1 = nhwc(x[:, c0:], dt) if c1 else None
    dyb = nhwc(dy, dt)
    dw = wgrad_run(dt, n, h, w, s0, c0, c0 + 64, s1, c1, dyb, cout)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(q(x, dt).double(), wt, padding=1).backward(q(dy, dt).double())
    gout = torch.zeros(cout * cin * 9, dtype=torch.float32, device=DEV)
    L.check(L.lib().nunet_unpack_wgrad(L.ptr(dw), cout, cin, cin, L.ptr(gout), 0, L.stream()), "unpack")
    got = gout.view(cout, cin, 3, 3).cpu()
    assert rel_err(got, wt.grad) < (5e-5 if dt == L.F32 else 1e-3), (DT[dt], shape)   # inputs are exact in T; only fp32 accumulation differs



def wgrad_desc(dt, n, h, w, s0, c0, p0, s1, c1, dyb, cout, slabs, max_slabs=0, target=0, item_shape=0):
    return L.WgradDesc(dt, n, h, w, L.ptr(s0), c0, p0, L.ptr(s1), c1, c1, L.ptr(dyb), cout, cout, L.ptr(slabs),
                       9 * cout * (c0 + c1), max_slabs, target, 0 if slabs is None else slabs.numel(), item_shape)


def wgrad_run(dt, n, h, w, s0, c0, p0, s1, c1, dyb, cout, target=0, item_shape=0):
    """K-split slabs (plain stores, every slab fully overwritten: pre-filled with garbage) + the fixed-order reduce."""
    cin = c0 + c1
    probe = wgrad_desc(dt, n, h, w, s0, c0, p0, s1, c1, dyb, cout, None, 0, target, item_shape)
    ks = L.lib().nunet_conv3x3_wgrad_slabs(C.byref(probe))
    assert ks >= 1
    slabs = torch.full((ks * 9 * cout * cin,), 1e30, dtype=torch.float32, device=DEV)
    d = wgrad_desc(dt, n, h, w, s0, c0, p0, s1, c1, dyb, cout, slabs, ks, target, item_shape)
    L.check(L.lib().nunet_conv3x3_wgrad(C.byref(d), L.stream()), "wgrad")
    dw = torch.full((9 * cout * cin,), 3.0, dtype=torch.float32, device=DEV)
    L.check(L.lib().nunet_wgrad_reduce(L.ptr(slabs), 9 * cout * cin, ks, 9 * cout * cin, L.ptr(dw), 0, L.stream()), "wgrad_reduce")
    return dw


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("item_shape,shape", [
    (21, (2, 20, 24, 64, 32, 64)),      # 64 Cout x 32 Cin items, two sources, ragged pixel tiles
    (21, (16, 12, 12, 64, 0, 128)),     # stacked-rows tiling
    (21, (2, 8, 8, 32, 0, 32)),         # Cout not a multiple of 64: falls back to 32 x 32
    (12, (2, 20, 24, 64, 32, 32)),      # 32 x 64 items; Cin = 96: the second input tile is half empty
    (12, (3, 12, 20, 128, 0, 32)),
    (12, (7, 3, 5, 32, 0, 32)),         # fewer than 64 input channels: falls back
])
def test_conv3x3_wgrad_item_shapes(dt, item_shape, shape):
    """nunet_wgrad_desc.item_shape: wider work items (A x B planes of [pixels][32 channels] in LDS, one fragment feeding A or B MFMAs)
    compute the same gradient as the default 32 x 32 items - against torch and, per element, within fp32 summation-order noise of
    the default shape (the K-split differs, the arithmetic does not)."""
    n, h, w, c0, c1, cout = shape
    g = torch.Generator().manual_seed(31)
    cin = c0 + c1
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    s0 = nhwc(x[:, :c0], dt, pitch=c0 + 64)
    s1 = nhwc(x[:, c0:], dt) if c1 else None
    dyb = nhwc(dy, dt)
    dw = wgrad_run(dt, n, h, w, s0, c0, c0 + 64, s1, c1, dyb, cout, item_shape=item_shape)
    dw0 = wgrad_run(dt, n, h, w, s0, c0, c0 + 64, s1, c1, dyb, cout)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(q(x, dt).double(), wt, padding=1).backward(q(dy, dt).double())
    gout = torch.zeros(cout * cin * 9, dtype=torch.float32, device=DEV)
    L.check(L.lib().nunet_unpack_wgrad(L.ptr(dw), cout, cin, cin, L.ptr(gout), 0, L.stream()), "unpack")
    assert rel_err(gout.view(cout, cin, 3, 3).cpu(), wt.grad) < (5e-5 if dt == L.F32 else 1e-3), (DT[dt], shape)
    assert rel_err(dw.cpu(), dw0.cpu()) < 2e-5


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_conv3x3_wgrad_pair(dt):
    """Two independent weight-gradient problems in one launch equal two single launches
    (the two convolutions of a VGGBlock, reference archs1.py:18,20)."""
    n, h, w = 3, 12, 12
    g = torch.Generator().manual_seed(5)
    probs = [(64, 32, 64), (64, 0, 64)]        # (C0, C1, Cout): conv1 with a concat input, conv2 mid -> out
    descs, keep, refs, slabs_l, kss = [], [], [], [], []
    for c0, c1, cout in probs:
        cin = c0 + c1
        x = torch.randn(n, cin, h, w, generator=g)
        dy = torch.randn(n, cout, h, w, generator=g)
        s0 = nhwc(x[:, :c0], dt, pitch=c0 + 32)
        s1 = nhwc(x[:, c0:], dt) if c1 else None
        dyb = nhwc(dy, dt)
        keep += [s0, s1, dyb]
        refs.append(wgrad_run(dt, n, h, w, s0, c0, c0 + 32, s1, c1, dyb, cout, target=128))
        probe = wgrad_desc(dt, n, h, w, s0, c0, c0 + 32, s1, c1, dyb, cout, None, 0, 128)
        ks = L.lib().nunet_conv3x3_wgrad_slabs(C.byref(probe))
        slabs = torch.full((ks * 9 * cout * cin,), -7.0, dtype=torch.float32, device=DEV)
        slabs_l.append(slabs); kss.append(ks)
        descs.append(wgrad_desc(dt, n, h, w, s0, c0, c0 + 32, s1, c1, dyb, cout, slabs, ks, 128))
    L.check(L.lib().nunet_conv3x3_wgrad_pair(C.byref(descs[0]), C.byref(descs[1]), L.stream()), "wgrad_pair")
    for (c0, c1, cout), slabs, ks, ref in zip(probs, slabs_l, kss, refs):
        nw = 9 * cout * (c0 + c1)
        dw = torch.zeros(nw, dtype=torch.float32, device=DEV)
        L.check(L.lib().nunet_wgrad_reduce(L.ptr(slabs), nw, ks, nw, L.ptr(dw), 0, L.stream()), "wgrad_reduce")
        assert float(ref.abs().max()) > 0
        assert torch.equal(dw, ref)      # same slices, same summation order: bit-identical


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_pack_unpack_roundtrip(dt):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 3, 3, 3, generator=g)
    wf, _ = pack(w, dt, cin_pad=32, want_wd=False)
    wf = wf.float().view(9, 64, 32).cpu()
    ref = q(w, dt).permute(2, 3, 0, 1).reshape(9, 64, 3)
    assert torch.equal(wf[:, :, :3], ref) and float(wf[:, :, 3:].abs().max()) == 0
    w2 = torch.randn(32, 48, 3, 3, generator=g)
    _, wd = pack(w2, dt)
    wd = wd.float().view(9, 48, 32).cpu()
    assert torch.equal(wd, q(w2, dt).flip(2, 3).permute(2, 3, 1, 0).reshape(9, 48, 32))
    dw = torch.randn(9, 64, 32, generator=g)
    gg = torch.ones(64 * 3 * 9, dtype=torch.float32, device=DEV)
    dw_g = dw.to(DEV)
    L.check(L.lib().nunet_unpack_wgrad(L.ptr(dw_g), 64, 3, 32, L.ptr(gg), 1, L.stream()), "unpack")
    exp = dw[:, :, :3].permute(1, 2, 0).reshape(64, 3, 3, 3) + 1.0
    np.testing.assert_allclose(gg.view(64, 3, 3, 3).cpu().numpy(), exp.numpy(), rtol=1e-6)


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_bn_relu_fwd(dt, pool, training):
    """The conv output is stored without its bias; BN folds the bias in analytically."""
    n, h, w, c = 3, 8, 12, 64
    g = torch.Generator().manual_seed(2)
    bias = torch.randn(c, generator=g) * 0.3
    ys = q(torch.randn(n, c, h, w, generator=g) * 0.7, dt)          # stored tensor (acc, no bias)
    y = ys + bias.view(1, -1, 1, 1)                                  # what the reference BN sees
    gamma = 1 + 0.2 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    rm = 0.1 * torch.randn(c, generator=g)
    rv = 0.5 + torch.rand(c, generator=g)
    yb = nhwc(ys, dt)
    dd = ys.double()
    stats = L.fx_encode(torch.cat([dd.sum((0, 2, 3)), (dd * dd).sum((0, 2, 3))]), c, DEV)     # replica 0 holds everything
    a = torch.zeros((n, h, w, 160), dtype=tdt(dt), device=DEV)
    pooled = torch.zeros((n, h // 2, w // 2, c), dtype=tdt(dt), device=DEV) if pool else None
    rmg, rvg = rm.clone().to(DEV), rv.clone().to(DEV)
    nbt = torch.tensor([4], dtype=torch.int64, device=DEV)
    save = torch.zeros(2 * c, dtype=torch.float32, device=DEV)
    bias_g, gamma_g, beta_g = bias.to(DEV), gamma.to(DEV), beta.to(DEV)   # keep alive across the launch
    # the fused x2 upsample of the activation (second block role of the same launch) into a pitched buffer
    up = torch.full((n, 2 * h, 2 * w, c + 8), 5.0, dtype=tdt(dt), device=DEV)
    d = L.BnFwdDesc(dt, n, h, w, c, L.ptr(yb), c, L.ptr(bias_g), L.ptr(stats), L.ptr(gamma_g), L.ptr(beta_g),
                    L.ptr(rmg), L.ptr(rvg), L.ptr(nbt), L.ptr(save), 1 if training else 0, 0.1, 1e-5,
                    L.ptr(a, 32 * a.element_size()), 160, L.ptr(pooled), c, L.ptr(up), c + 8)
    L.check(L.lib().nunet_bn_relu_fwd(C.byref(d), L.stream()), "bn")
    # ... equals the stand-alone upsample of the STORED activation bit for bit (every tap is rounded to the storage type first)
    up_ref = torch.zeros((n, 2 * h, 2 * w, c), dtype=tdt(dt), device=DEV)
    L.check(L.lib().nunet_upsample2x_fwd(dt, n, h, w, c, L.ptr(a, 32 * a.element_size()), 160, L.ptr(up_ref), c, L.stream()), "up")
    assert torch.equal(up[..., :c], up_ref) and float((up[..., c:].float() - 5.0).abs().max()) == 0
    rm2, rv2 = rm.clone().double(), rv.clone().double()
    ref = F.relu(F.batch_norm(y.double(), rm2, rv2, gamma.double(), beta.double(), training, 0.1, 1e-5))
    got = to_nchw(a, c, off=32)
    assert rel_err(got, ref) < TOL[dt]
    assert float(a[..., :32].float().abs().max()) == 0 and float(a[..., 96:].float().abs().max()) == 0
    if pool:
        assert torch.equal(to_nchw(pooled, c), F.max_pool2d(got, 2, 2))
    if training:
        np.testing.assert_allclose(rmg.cpu().numpy(), rm2.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rvg.cpu().numpy(), rv2.numpy(), rtol=1e-5, atol=1e-6)
        assert int(nbt.item()) == 5
        mean = ys.double().mean((0, 2, 3))
        var = ys.double().var((0, 2, 3), unbiased=False)
        np.testing.assert_allclose(save[:c].cpu().numpy(), mean.numpy(), atol=1e-5)
        np.testing.assert_allclose(save[c:].cpu().numpy(), (1 / (var + 1e-5).sqrt()).numpy(), rtol=1e-4)
    else:
        assert torch.equal(rmg.cpu(), rm) and int(nbt.item()) == 4


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
def test_bn_relu_bwd(dt):
    n, h, w, c = 3, 8, 12, 64
    g = torch.Generator().manual_seed(4)
    y = q(torch.randn(n, c, h, w, generator=g), dt)
    da = q(torch.randn(n, c, h, w, generator=g), dt)
    gamma = 1 + 0.2 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    out = F.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5))
    out.backward(da.double())
    mean = y.double().mean((0, 2, 3))
    istd = 1 / (y.double().var((0, 2, 3), unbiased=False) + 1e-5).sqrt()
    mi = torch.cat([mean, istd]).float().to(DEV)
    yb = nhwc(y, dt)
    dab = nhwc(da, dt, pitch=160, off=64)
    sums = L.fx_zeros(c, DEV)
    dg = torch.full((c,), 9.0, dtype=torch.float32, device=DEV)      # assigned, not accumulated
    db = torch.full((c,), 9.0, dtype=torch.float32, device=DEV)
    dbias = torch.full((c,), 9.0, dtype=torch.float32, device=DEV)
    dyb = torch.zeros((n, h, w, c), dtype=tdt(dt), device=DEV)
    gamma_g, beta_g = gamma.to(DEV), beta.to(DEV)
    d = L.BnBwdDesc(dt, n, h, w, c, L.ptr(dab, 64 * dab.element_size()), 160, L.ptr(yb), c, L.ptr(mi),
                    L.ptr(gamma_g), L.ptr(beta_g), L.ptr(sums), L.ptr(dg), L.ptr(db), L.ptr(dbias),
                    L.ptr(dyb), c)
    L.check(L.lib().nunet_bn_relu_bwd_reduce(C.byref(d), L.stream()), "bn bwd reduce")
    L.check(L.lib().nunet_bn_relu_bwd_apply(C.byref(d), L.stream()), "bn bwd apply")
    assert rel_err(to_nchw(dyb, c), yd.grad) < TOL[dt]
    assert rel_err(dg.cpu(), gd.grad) < 1e-4 and rel_err(db.cpu(), bd.grad) < 1e-4
    # conv-bias gradient = sum of dy = 0 analytically (a bias in front of a BatchNorm): stored as exact zeros
    assert float(dbias.abs().max()) == 0.0


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("shape,sk", [((2, 20, 24, 32, 32), False), ((3, 12, 12, 64, 64), False), ((16, 6, 6, 512, 512), True)])
def test_conv3x3_fused_bn_bwd_reduce(dt, shape, sk):
    """The BatchNorm+ReLU backward reduce taken in the conv epilogue (plain and K-split finalize) equals
    nunet_bn_relu_bwd_reduce run on the tensor the conv stored."""
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    y1 = q(torch.randn(n, cout, h, w, generator=g), dt)                 # raw output of the BN's conv
    gamma = 1 + 0.2 * torch.randn(cout, generator=g)
    beta = 0.2 * torch.randn(cout, generator=g)
    mean = y1.double().mean((0, 2, 3))
    istd = 1 / (y1.double().var((0, 2, 3), unbiased=False) + 1e-5).sqrt()
    mi = torch.cat([mean, istd]).float().to(DEV)
    s0 = nhwc(x, dt)
    wf, _ = pack(wt, dt)
    yb = nhwc(y1, dt)
    out = torch.zeros((n, h, w, cout), dtype=tdt(dt), device=DEV)
    sums = L.fx_zeros(cout, DEV)
    gamma_g, beta_g = gamma.to(DEV), beta.to(DEV)
    d = conv_desc(dt, n, h, w, s0, cin, cin, wf, out, cout, cout)
    ws = None
    if sk:
        ws = torch.zeros(8 * n * h * w * cout + 256, dtype=torch.float32, device=DEV)
        d.splitk_ws = L.ptr(ws).value; d.splitk_ws_floats = ws.numel()
    d.bn_y = L.ptr(yb).value; d.bn_py = cout; d.bn_mean_invstd = L.ptr(mi).value
    d.bn_gamma = L.ptr(gamma_g).value; d.bn_beta = L.ptr(beta_g).value; d.bn_sums = L.ptr(sums).value
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "conv+bnr")
    ref_sums = torch.zeros_like(sums)
    dummy = torch.zeros(cout, dtype=torch.float32, device=DEV)
    dyb = torch.zeros_like(out)
    b = L.BnBwdDesc(dt, n, h, w, cout, L.ptr(out), cout, L.ptr(yb), cout, L.ptr(mi), L.ptr(gamma_g), L.ptr(beta_g),
                    L.ptr(ref_sums), L.ptr(dummy), L.ptr(dummy), L.ptr(dummy), L.ptr(dyb), cout)
    L.check(L.lib().nunet_bn_relu_bwd_reduce(C.byref(b), L.stream()), "reduce")
    torch.cuda.synchronize()
    ref = F.conv2d(q(x, dt).double(), q(wt, dt).double(), padding=1)
    assert rel_err(to_nchw(out, cout), ref) < TOL[dt]
    assert float(ref_sums.abs().max()) > 0
    tot = L.fx_decode(sums, cout)
    rtot = L.fx_decode(ref_sums, cout)
    scale = float(rtot.abs().max())
    assert float((tot - rtot).abs().max()) < 2e-4 * scale + 1e-5, (DT[dt], shape)


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_maxpool(dt):
    n, h, w, c = 2, 8, 12, 32
    g = torch.Generator().manual_seed(6)
    x = q(torch.randn(n, c, h, w, generator=g), dt)
    x[:, :, 0:2, 0:2] = 1.5            # ties: first maximum in scan order must win
    dy = q(torch.randn(n, c, h // 2, w // 2, generator=g), dt)
    prev = q(torch.randn(n, c, h, w, generator=g), dt)
    xb = nhwc(x, dt, pitch=96, off=32)
    yb = torch.zeros((n, h // 2, w // 2, c), dtype=tdt(dt), device=DEV)
    es = xb.element_size()
    L.check(L.lib().nunet_maxpool2x2_fwd(dt, n, h, w, c, L.ptr(xb, 32 * es), 96, L.ptr(yb), c, L.stream()), "pool")
    xd = x.double().requires_grad_(True)
    ref = F.max_pool2d(xd, 2, 2)
    assert torch.equal(to_nchw(yb, c).double(), ref.detach())
    ref.backward(dy.double())
    for acc in (0, 1):
        dxb = nhwc(prev, dt)
        dyb = nhwc(dy, dt)
        L.check(L.lib().nunet_maxpool2x2_bwd(dt, n, h, w, c, L.ptr(xb, 32 * es), 96, L.ptr(dyb), c,
                                             L.ptr(dxb), c, acc, L.stream()), "pool bwd")
        exp = xd.grad + (prev.double() if acc else 0)
        assert rel_err(to_nchw(dxb, c), exp) < TOL[dt]


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("hw", [(6, 10), (1, 1), (3, 2), (4, 4), (4, 7), (5, 9), (12, 6), (24, 4), (48, 48), (3, 8), (8, 3)])
def test_upsample(dt, hw):
    h, w = hw
    n, c = 2, 64
    g = torch.Generator().manual_seed(8)
    x = q(torch.randn(n, c, h, w, generator=g), dt)
    dy = q(torch.randn(n, c, 2 * h, 2 * w, generator=g), dt)
    prev = q(torch.randn(n, c, h, w, generator=g), dt)
    xb = nhwc(x, dt, pitch=128, off=64)
    es = xb.element_size()
    yb = torch.zeros((n, 2 * h, 2 * w, c), dtype=tdt(dt), device=DEV)
    L.check(L.lib().nunet_upsample2x_fwd(dt, n, h, w, c, L.ptr(xb, 64 * es), 128, L.ptr(yb), c, L.stream()), "up")
    xd = x.double().requires_grad_(True)
    ref = F.interpolate(xd, scale_factor=2, mode="bilinear", align_corners=True)
    assert rel_err(to_nchw(yb, c), ref.detach()) < TOL[dt]
    ref.backward(dy.double())
    for acc in (0, 1):
        dxb = nhwc(prev, dt, pitch=128, off=64)
        dyb = nhwc(dy, dt)
        L.check(L.lib().nunet_upsample2x_bwd(dt, n, h, w, c, L.ptr(dyb), c, L.ptr(dxb, 64 * es), 128, acc,
                                             L.stream()), "up bwd")
        exp = xd.grad + (prev.double() if acc else 0)
        assert rel_err(to_nchw(dxb, c, off=64), exp) < TOL[dt]


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("k", [1, 4])
def test_head(dt, k):
    n, h, w, c = 2, 12, 20, 32
    g = torch.Generator().manual_seed(9)
    x = q(torch.randn(n, c, h, w, generator=g), dt)
    wt = torch.randn(k, c, 1, 1, generator=g) * 0.2
    b = torch.randn(k, generator=g) * 0.1
    dl = torch.randn(n, k, h, w, generator=g)
    xb = nhwc(x, dt, pitch=160, off=128)
    es = xb.element_size()
    logits = torch.zeros((n, k, h, w), dtype=torch.float32, device=DEV)
    wt_g, b_g, dl_g = wt.to(DEV), b.to(DEV), dl.to(DEV)
    L.check(L.lib().nunet_head_fwd(dt, n, h, w, c, k, L.ptr(xb, 128 * es), 160, L.ptr(wt_g), L.ptr(b_g),
                                   L.ptr(logits), L.stream()), "head")
    xd = x.double().requires_grad_(True)
    wd_, bd = wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xd, wd_, bd)
    assert rel_err(logits.cpu(), ref.detach()) < 2e-5
    ref.backward(dl.double())
    dxb = torch.zeros((n, h, w, 160), dtype=tdt(dt), device=DEV)
    nslabs = 64
    slabs = torch.full((nslabs, k * c + k), 7.0, dtype=torch.float32, device=DEV)     # fully overwritten
    L.check(L.lib().nunet_head_bwd(dt, n, h, w, c, k, L.ptr(xb, 128 * es), 160, L.ptr(wt_g), L.ptr(dl_g),
                                   L.ptr(dxb, 128 * es), 160, 0, L.ptr(slabs), nslabs, L.stream()), "head bwd")
    assert rel_err(to_nchw(dxb, c, off=128), xd.grad) < TOL[dt]
    tot = slabs.sum(0).cpu()
    assert rel_err(tot[:k * c].view(k, c), wd_.grad.view(k, c)) < 1e-4
    assert rel_err(tot[k * c:], bd.grad) < 1e-4


def _bnr_setup(n, h, w, c, dt, g):
    """A BatchNorm whose backward reduce is fused into the kernel that completes its output gradient."""
    y = q(torch.randn(n, c, h, w, generator=g), dt)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(c, generator=g)).to(DEV)
    mean = y.double().mean((0, 2, 3))
    istd = 1 / (y.double().var((0, 2, 3), unbiased=False) + 1e-5).sqrt()
    mi = torch.cat([mean, istd]).float().to(DEV)
    yb = nhwc(y, dt)
    sums = L.fx_zeros(c, DEV)
    d = L.BnrDesc(L.ptr(yb), c, L.ptr(mi), L.ptr(gamma), L.ptr(beta), L.ptr(sums))
    return d, sums, (yb, mi, gamma, beta)


def _bnr_check(dt, n, h, w, c, da_ptr, pda, sums, keep):
    """The fused sums equal nunet_bn_relu_bwd_reduce run on the completed gradient tensor."""
    yb, mi, gamma, beta = keep
    ref = L.fx_zeros(c, DEV)
    dummy = torch.zeros(c, dtype=torch.float32, device=DEV)
    b = L.BnBwdDesc(dt, n, h, w, c, da_ptr, pda, L.ptr(yb), c, L.ptr(mi), L.ptr(gamma), L.ptr(beta), L.ptr(ref),
                    L.ptr(dummy), L.ptr(dummy), L.ptr(dummy), None, 0)
    L.check(L.lib().nunet_bn_relu_bwd_reduce(C.byref(b), L.stream()), "reduce")
    torch.cuda.synchronize()
    tot, rtot = L.fx_decode(sums, c), L.fx_decode(ref, c)
    assert float(rtot.abs().max()) > 0
    assert float((tot - rtot).abs().max()) < 2e-4 * float(rtot.abs().max()) + 1e-5


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("acc", [0, 1])
def test_fused_bn_reduce_in_head_bwd(dt, acc):
    """nunet_head_bwd_bnr: same gradients as the plain kernel, plus the BatchNorm-backward sums of the tensor it completes."""
    g = torch.Generator().manual_seed(41 + acc)
    n, h, w = 3, 12, 10
    es = 4 if dt == L.F32 else 2
    k, c = 2, 32
    xh = nhwc(q(torch.randn(n, c, h, w, generator=g), dt), dt, pitch=160, off=128)
    wt = (torch.randn(k, c, generator=g) * 0.2).to(DEV)
    dl = torch.randn(n, k, h, w, generator=g).to(DEV)
    prev32 = q(torch.randn(n, c, h, w, generator=g), dt)
    d, sums, keep = _bnr_setup(n, h, w, c, dt, g)
    a, b_ = nhwc(prev32, dt, pitch=160, off=128), nhwc(prev32, dt, pitch=160, off=128)
    s1 = torch.zeros((64, k * c + k), dtype=torch.float32, device=DEV); s2 = torch.zeros_like(s1)
    L.check(L.lib().nunet_head_bwd(dt, n, h, w, c, k, L.ptr(xh, 128 * es), 160, L.ptr(wt), L.ptr(dl), L.ptr(a, 128 * es), 160, acc,
                                   L.ptr(s1), 64, L.stream()), "head")
    L.check(L.lib().nunet_head_bwd_bnr(dt, n, h, w, c, k, L.ptr(xh, 128 * es), 160, L.ptr(wt), L.ptr(dl), L.ptr(b_, 128 * es), 160, acc,
                                       L.ptr(s2), 64, C.byref(d), L.stream()), "head bnr")
    assert torch.equal(a, b_) and torch.equal(s1, s2)
    _bnr_check(dt, n, h, w, c, L.ptr(b_, 128 * es), 160, sums, keep)


def test_bce_dice_and_iou_against_reference_goldens():
    from conftest import load_golden
    g = load_golden("small_ops")
    crit = nunet_amd.losses.BCEDiceLoss()
    for tag in ("k1", "k4"):
        x = torch.from_numpy(g["x_" + tag]).to(DEV).requires_grad_(True)
        t = torch.from_numpy(g["t_" + tag]).to(DEV)
        loss = crit(x, t)
        (loss * 1.0).backward()
        assert abs(float(loss) - float(g["loss_" + tag])) < 2e-6
        np.testing.assert_allclose(x.grad.cpu().numpy(), g["dx_" + tag], atol=2e-9, rtol=2e-4)
        assert abs(nunet_amd.metrics.iou_score(x, t) - float(g["iou_" + tag])) < 1e-12
    # upstream gradient scaling (the /4 of deep supervision, reference trains.py:120-123)
    x = torch.from_numpy(g["x_k1"]).to(DEV).requires_grad_(True)
    (crit(x, torch.from_numpy(g["t_k1"]).to(DEV)) / 4).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["dx_k1"] / 4, atol=1e-9, rtol=2e-4)


@pytest.mark.parametrize("nesterov", [False, True])
def test_sgd_matches_torch(nesterov):
    g = torch.Generator().manual_seed(1)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=nesterov)
    p = p0.clone().to(DEV)
    mom = torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV)
    for step in range(3):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        gr_g = gr.to(DEV)
        L.check(L.lib().nunet_sgd_step(L.ptr(p), L.ptr(gr_g), L.ptr(mom), n, L.ptr(lr), 0.9, 1e-4,
                                       1 if nesterov else 0, 1 if step == 0 else 0, 1.0, L.stream()), "sgd")
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_nchw_to_nhwc_pad():
    x = torch.randn(2, 3, 5, 7)
    y = torch.full((2, 5, 7, 32), 9.0, dtype=torch.bfloat16, device=DEV)
    x_g = x.to(DEV)
    L.check(L.lib().nunet_nchw_to_nhwc(L.ptr(x_g), 2, 3, 5, 7, L.BF16, L.ptr(y), 32, L.stream()), "layout")
    assert torch.equal(y[..., :3].float().cpu(), x.permute(0, 2, 3, 1).bfloat16().float())
    assert float(y[..., 3:].float().abs().max()) == 0


def test_argument_errors_are_loud():
    d = L.ConvDesc()
    assert L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()) == -1
    assert b"null" in L.lib().nunet_last_error()
    with pytest.raises(L.NunetError):
        nunet_amd.archs.NestedUNet(1)(torch.zeros(1, 3, 32, 32))          # CPU module: no fallback
    m = nunet_amd.archs.NestedUNet(1).to(DEV)
    with pytest.raises(L.NunetError):
        m(torch.zeros(1, 3, 40, 40, device=DEV))                          # not a multiple of 16


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_conv3x3_splitk_slabs(dt):
    """Grid-starved long-K shape: the K-split path (fp32 slabs + deterministic finalize) gives the same
    outputs / BN sums / accumulate semantics as the single-pass kernel, bit-identically run to run."""
    n, h, w, c0, c1, cout = 2, 12, 12, 256, 256, 128
    g = torch.Generator().manual_seed(21)
    cin = c0 + c1
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    prev = torch.randn(n, cout, h, w, generator=g)
    s0, s1 = nhwc(x[:, :c0], dt), nhwc(x[:, c0:], dt)
    wf, _ = pack(wt, dt)
    ws = torch.full((8 * n * h * w * cout,), 7.0, dtype=torch.float32, device=DEV)   # garbage: slabs are fully overwritten
    outs, sts = [], []
    for accum in (0, 1, 0):
        y = nhwc(prev, dt)
        stats = L.fx_zeros(cout, DEV)
        d = conv_desc(dt, n, h, w, s0, c0, c0, wf, y, cout, cout, src1=s1, c1=c1, p1=c1, stats=stats,
                      slot_w=cout, mask=accum)
        d.splitk_ws = L.ptr(ws).value
        d.splitk_ws_floats = ws.numel()
        L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "conv splitk")
        ref = F.conv2d(q(x, dt).double(), q(wt, dt).double(), None, padding=1)
        exp = ref + (q(prev, dt).double() if accum else 0)
        assert rel_err(to_nchw(y, cout), exp) < TOL[dt]
        if not accum:
            got = to_nchw(y, cout).double()
            m = n * h * w
            tot = L.fx_decode(stats, cout)
            np.testing.assert_allclose(tot[:cout].numpy() / m, got.sum((0, 2, 3)).numpy() / m, atol=1e-4)
            outs.append(y.clone()); sts.append(stats.clone())
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(sts[0].view(L.BN_SUM_REPLICAS, -1).sum(0), sts[1].view(L.BN_SUM_REPLICAS, -1).sum(0))   # integer sums: exact


def test_conv3x3_rejects_partial_channel_chunks():
    """C0 / C1 must be whole 64-byte channel chunks (the staging has no ragged path): loud error, no silent padding."""
    x = torch.zeros((1, 4, 4, 16), dtype=torch.bfloat16, device=DEV)
    wf = torch.zeros(9 * 32 * 16, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros((1, 4, 4, 32), dtype=torch.bfloat16, device=DEV)
    d = conv_desc(L.BF16, 1, 4, 4, x, 16, 16, wf, y, 32, 32)
    assert L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()) == -1
    assert b"multiples of 32" in L.lib().nunet_last_error()


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(2, 20, 24, 32, 32), (3, 12, 12, 64, 64), (16, 6, 6, 512, 512), (5, 12, 12, 32, 32)])
def test_conv3x3_input_transform_bn_relu(dt, training, shape):
    """NUNET_TF_BN_RELU: conv(relu(bn(y1))) with the BatchNorm applied on the way into LDS equals bn_relu_fwd followed
    by the plain conv (reference finished/archs1.py:23-28), bit for bit; the side-stored activation, the saved
    statistics and the running-statistics update equal bn_relu_fwd's."""
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(23)
    y1 = q(torch.randn(n, cin, h, w, generator=g) * 0.8 + 0.3, dt)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = (torch.randn(cin, generator=g) * 0.3).to(DEV)
    gamma = (1 + 0.2 * torch.randn(cin, generator=g)).to(DEV)
    gamma[1] = -gamma[1]                                           # a negative scale must work too
    beta = (0.2 * torch.randn(cin, generator=g)).to(DEV)
    rm0, rv0 = 0.1 * torch.randn(cin, generator=g), 0.5 + torch.rand(cin, generator=g)
    yb = nhwc(y1, dt)
    dd = y1.double()
    stats = L.fx_encode(torch.cat([dd.sum((0, 2, 3)), (dd * dd).sum((0, 2, 3))]), cin, DEV)
    wf, _ = pack(wt, dt)
    # reference path: stand-alone BN kernel, then the plain conv
    a_ref = torch.zeros((n, h, w, cin), dtype=tdt(dt), device=DEV)
    rm_a, rv_a = rm0.clone().to(DEV), rv0.clone().to(DEV)
    nbt_a = torch.tensor([2], dtype=torch.int64, device=DEV)
    save_a = torch.zeros(2 * cin, dtype=torch.float32, device=DEV)
    b = L.BnFwdDesc(dt, n, h, w, cin, L.ptr(yb), cin, L.ptr(bias), L.ptr(stats), L.ptr(gamma), L.ptr(beta),
                    L.ptr(rm_a), L.ptr(rv_a), L.ptr(nbt_a), L.ptr(save_a), 1 if training else 0, 0.1, 1e-5,
                    L.ptr(a_ref), cin, None, 0)
    L.check(L.lib().nunet_bn_relu_fwd(C.byref(b), L.stream()), "bn")
    out_ref = torch.zeros((n, h, w, cout), dtype=tdt(dt), device=DEV)
    d0 = conv_desc(dt, n, h, w, a_ref, cin, cin, wf, out_ref, cout, cout)
    ws = torch.zeros(8 * n * h * w * cout, dtype=torch.float32, device=DEV)
    d0.splitk_ws = L.ptr(ws).value; d0.splitk_ws_floats = ws.numel()
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d0), L.stream()), "conv")
    # fused path
    out = torch.zeros_like(out_ref)
    a_side = torch.full((n, h, w, cin), 5.0, dtype=tdt(dt), device=DEV)
    rm_b, rv_b = rm0.clone().to(DEV), rv0.clone().to(DEV)
    nbt_b = torch.tensor([2], dtype=torch.int64, device=DEV)
    save_b = torch.zeros(2 * cin, dtype=torch.float32, device=DEV)
    st2 = L.fx_zeros(cout, DEV)
    d = conv_desc(dt, n, h, w, yb, cin, cin, wf, out, cout, cout, stats=st2)
    d.splitk_ws = L.ptr(ws).value; d.splitk_ws_floats = ws.numel()
    d.in_tf = L.TF_BN_RELU; d.tf_training = 1 if training else 0
    d.tf_fx = L.ptr(stats).value; d.tf_gamma = L.ptr(gamma).value; d.tf_beta = L.ptr(beta).value; d.tf_conv_bias = L.ptr(bias).value
    d.tf_running_mean = L.ptr(rm_b).value; d.tf_running_var = L.ptr(rv_b).value; d.tf_nbt = L.ptr(nbt_b).value
    d.tf_mean_invstd = L.ptr(save_b).value; d.tf_momentum = 0.1; d.tf_eps = 1e-5
    d.tf_store = L.ptr(a_side).value; d.tf_ps = cin
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "conv+tf")
    torch.cuda.synchronize()
    assert float(a_ref.float().abs().max()) > 0
    assert torch.equal(a_side, a_ref)                       # every pixel stored exactly once, same rounding
    assert torch.equal(out, out_ref)
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b) and torch.equal(save_a, save_b) and torch.equal(nbt_a, nbt_b)
    assert int(nbt_b.item()) == (3 if training else 2)
    # and against torch: conv(relu(bn(y1 + bias)))
    yfull = (y1 + bias.cpu().view(1, -1, 1, 1)).double()
    a64 = F.relu(F.batch_norm(yfull, rm0.clone().double(), rv0.clone().double(), gamma.cpu().double(), beta.cpu().double(), training, 0.1, 1e-5))
    ref = F.conv2d(q(a64.float(), dt).double(), q(wt, dt).double(), padding=1)
    assert rel_err(to_nchw(out, cout), ref) < 4 * TOL[dt]
    m = n * h * w
    got = to_nchw(out, cout).double()
    tot = L.fx_decode(st2, cout)
    np.testing.assert_allclose(tot[:cout].numpy() / m, got.sum((0, 2, 3)).numpy() / m, atol=1e-4 * float(got.abs().max()) + 1e-6)


@pytest.mark.parametrize("dt", [L.F32, L.BF16, L.F16])
@pytest.mark.parametrize("shape", [(2, 20, 24, 32, 96), (3, 12, 12, 64, 64), (16, 6, 6, 512, 256), (5, 12, 12, 32, 32)])
def test_conv3x3_input_transform_bn_relu_bwd(dt, shape):
    """NUNET_TF_BN_RELU_BWD: dgrad with the BatchNorm+ReLU backward APPLY pass taken on the way into LDS equals
    nunet_bn_relu_bwd_apply followed by the plain dgrad, bit for bit; the side-stored dy (for the weight gradient)
    and d gamma / d beta / d bias equal the stand-alone kernel's."""
    n, h, w, c, cout = shape          # c: channels of the BatchNorm (= Cin of the dgrad conv)
    g = torch.Generator().manual_seed(29)
    y = q(torch.randn(n, c, h, w, generator=g), dt)
    da = q(torch.randn(n, c, h, w, generator=g), dt)
    wt = torch.randn(cout, c, 3, 3, generator=g) / (3 * c ** 0.5)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(c, generator=g)).to(DEV)
    mean = y.double().mean((0, 2, 3))
    istd = 1 / (y.double().var((0, 2, 3), unbiased=False) + 1e-5).sqrt()
    mi = torch.cat([mean, istd]).float().to(DEV)
    yb = nhwc(y, dt)
    dab = nhwc(da, dt, pitch=c + 64, off=32)               # the gradient lives in a slot of a wider level buffer
    da_ptr = L.ptr(dab, 32 * dab.element_size())
    sums = L.fx_zeros(c, DEV)
    vec = [torch.full((c,), 9.0, dtype=torch.float32, device=DEV) for _ in range(6)]
    dy_ref = torch.zeros((n, h, w, c), dtype=tdt(dt), device=DEV)
    b = L.BnBwdDesc(dt, n, h, w, c, da_ptr, c + 64, L.ptr(yb), c, L.ptr(mi), L.ptr(gamma), L.ptr(beta), L.ptr(sums),
                    L.ptr(vec[0]), L.ptr(vec[1]), L.ptr(vec[2]), L.ptr(dy_ref), c)
    L.check(L.lib().nunet_bn_relu_bwd_reduce(C.byref(b), L.stream()), "reduce")
    L.check(L.lib().nunet_bn_relu_bwd_apply(C.byref(b), L.stream()), "apply")
    wf, _ = pack(wt, dt)                                      # any packed [9][cout][c] weights do
    ws = torch.zeros(8 * n * h * w * cout, dtype=torch.float32, device=DEV)
    out_ref = torch.zeros((n, h, w, cout), dtype=tdt(dt), device=DEV)
    d0 = conv_desc(dt, n, h, w, dy_ref, c, c, wf, out_ref, cout, cout)
    d0.splitk_ws = L.ptr(ws).value; d0.splitk_ws_floats = ws.numel()
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d0), L.stream()), "dgrad")
    out = torch.zeros_like(out_ref)
    dy_side = torch.full((n, h, w, c), 5.0, dtype=tdt(dt), device=DEV)
    d = L.ConvDesc()
    d.dtype = dt; d.N = n; d.H = h; d.W = w
    d.src0 = da_ptr.value; d.C0 = c; d.P0 = c + 64
    d.wpack = L.ptr(wf).value; d.dst0 = L.ptr(out).value; d.D0 = cout; d.Q0 = cout
    d.splitk_ws = L.ptr(ws).value; d.splitk_ws_floats = ws.numel()
    d.in_tf = L.TF_BN_RELU_BWD; d.tf_y = L.ptr(yb).value; d.tf_py = c; d.tf_fx = L.ptr(sums).value
    d.tf_gamma = L.ptr(gamma).value; d.tf_beta = L.ptr(beta).value; d.tf_mean_invstd = L.ptr(mi).value
    d.tf_dgamma = L.ptr(vec[3]).value; d.tf_dbeta = L.ptr(vec[4]).value; d.tf_dbias = L.ptr(vec[5]).value
    d.tf_store = L.ptr(dy_side).value; d.tf_ps = c
    L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "dgrad+tf")
    torch.cuda.synchronize()
    assert float(dy_ref.float().abs().max()) > 0
    assert torch.equal(dy_side, dy_ref)
    assert torch.equal(out, out_ref)
    for k in range(3):
        assert torch.equal(vec[k], vec[3 + k])
    assert float(vec[5].abs().max()) == 0.0


def test_per_channel_sums_are_order_independent():
    """Fixed-point per-channel sums: the BatchNorm statistics a conv leaves are bit-identical from launch to launch
    (fp32 atomics would depend on the arrival order of 576 workgroups)."""
    n, h, w, cin, cout = 16, 96, 96, 32, 32
    g = torch.Generator().manual_seed(31)
    x = nhwc(torch.randn(n, cin, h, w, generator=g), L.BF16)
    wf, _ = pack(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), L.BF16)
    tots = []
    for _ in range(3):
        y = torch.zeros((n, h, w, cout), dtype=torch.bfloat16, device=DEV)
        st = L.fx_zeros(cout, DEV)
        d = conv_desc(L.BF16, n, h, w, x, cin, cin, wf, y, cout, cout, stats=st)
        L.check(L.lib().nunet_conv3x3_fwd(C.byref(d), L.stream()), "conv")
        tots.append(st.view(L.BN_SUM_REPLICAS, -1).sum(0).clone())
    assert torch.equal(tots[0], tots[1]) and torch.equal(tots[0], tots[2])
    assert float(L.fx_decode(st, cout)[cout:].min()) > 0


def test_lovasz_hinge_against_reference_goldens():
    """LovaszHingeLoss on device vs the reference's own outputs."""
    from conftest import load_golden
    g = load_golden("lovasz")
    crit = nunet_amd.losses.LovaszHingeLoss()
    for tag in ("a", "b", "c"):
        x = torch.from_numpy(g["x_" + tag]).to(DEV).requires_grad_(True)
        t = torch.from_numpy(g["t_" + tag]).to(DEV)
        loss = crit(x, t)
        (loss * 0.5).backward()
        assert abs(float(loss.detach()) - float(g["loss_" + tag])) < 2e-5 * max(1.0, float(g["loss_" + tag])), tag
        np.testing.assert_allclose(x.grad.cpu().numpy() * 2, g["dx_" + tag], atol=2e-7, rtol=2e-4, err_msg=tag)


@pytest.mark.parametrize("shape", [(2, 144, 160), (3, 256, 256), (1, 512, 512)])
def test_lovasz_hinge_large_images_match_oracle(shape):
    """Images above the in-LDS sort limit (16384 px): chunk sorts + global bitonic merge passes + chunked scan,
    against the oracle's restatement (itself pinned to the reference goldens in tests/test_oracle.py).
    144x160 pads 23040 keys to 32768; 256x256 and 512x512 are the BASELINE cfg4 / cfg5 image sizes."""
    from oracle import nunet_oracle as O
    n, h, w = shape
    g = torch.Generator().manual_seed(h)
    x = torch.randn(n, 1, h, w, generator=g) * 2
    t = (torch.rand(n, 1, h, w, generator=g) > 0.7).float()
    t[0, 0, : h // 2] = 0                                   # a large label-free region
    xd = x.to(DEV).requires_grad_(True)
    loss = nunet_amd.losses.LovaszHingeLoss()(xd, t.to(DEV))
    loss.backward()
    xo = x.double().requires_grad_(True)
    ref = O.lovasz_hinge(xo.squeeze(1), t.double().squeeze(1))
    ref.backward()
    assert abs(float(loss.detach()) - float(ref)) < 2e-5 * max(1.0, float(ref))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xo.grad.numpy(), atol=3e-7, rtol=2e-3)
    with pytest.raises(L.NunetError):
        big = torch.zeros(1, 1, 2048, 4096, device=DEV)                                         # 2^23 px: refused loudly
        nunet_amd.losses.LovaszHingeLoss()(big, big)


def test_device_input_pipeline_matches_host_pipeline(synth):
    """uint8 HWC -> Normalize -> /255 -> CHW (+rot90/flip) on the device equals the
    host pipeline of the reference's Dataset (dataset.py:66-74) as restated by synth.synth_images."""
    from nunet_amd import dataset as D
    n, h, w = 5, 32, 32
    rng = np.random.default_rng(1234)
    raw = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    ref = synth.synth_images(n, h, w, 3, seed=1234)                 # same stream -> same raw bytes
    out = D.preprocess_images(torch.from_numpy(raw).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-7, rtol=2e-5)
    codes = torch.tensor([0, 1, 2 | 4, 3 | 8, 4 | 8], dtype=torch.int32, device=DEV)
    aug = D.preprocess_images(torch.from_numpy(raw).to(DEV), codes).cpu().numpy()
    for i, cde in enumerate(codes.tolist()):
        e = np.rot90(ref[i], k=cde & 3, axes=(1, 2))
        if cde & 4:
            e = e[:, :, ::-1]
        if cde & 8:
            e = e[:, ::-1, :]
        np.testing.assert_allclose(aug[i], e, atol=2e-7, rtol=2e-5, err_msg=str(cde))
    m8 = (rng.random((n, h, w, 2)) > 0.5).astype(np.uint8) * 255
    mk = D.preprocess_masks(torch.from_numpy(m8).to(DEV), codes).cpu().numpy()
    assert set(np.unique(mk)) <= {0.0, 1.0}
    e = np.rot90((m8[1] / 255.0).transpose(2, 0, 1), k=1, axes=(1, 2))
    np.testing.assert_array_equal(mk[1], e.astype(np.float32))
    assert D.draw_augmentation(8).dtype == torch.int32


@pytest.mark.parametrize("shape", [(2, 1, 8, 16), (1, 4, 16, 16), (3, 2, 96, 96)])
def test_sigmoid_masks_u8(shape):
    """Mask export of the evaluation driver (reference val.py:100-105): (sigmoid(x) * 255).astype('uint8'), BYTE-EXACT
    against the host expression: random logits plus every truncation edge (the 255 thresholds and their fp32 neighbours).
    (Element counts are multiples of 16 so that the host's torch.sigmoid takes its vectorised path everywhere; its scalar
    tail can differ from it by an ulp.)"""
    from nunet_amd.metrics import sigmoid_masks_u8, sigmoid_u8_thresholds
    g = torch.Generator().manual_seed(9)
    x = torch.randn(*shape, generator=g) * 4
    thr = sigmoid_u8_thresholds(torch.device(DEV)).cpu()
    edges = torch.cat([thr, torch.nextafter(thr, torch.tensor(-1e9)), torch.nextafter(thr, torch.tensor(1e9)),
                       torch.tensor([-100.0, 100.0, 0.0, -0.0, 20.0])])
    flat = x.view(-1)
    k = min(edges.numel(), flat.numel())
    flat[:k] = edges[:k]
    got = sigmoid_masks_u8(x.to(DEV)).cpu()
    ref = torch.from_numpy((torch.sigmoid(x).numpy() * np.float32(255)).astype("uint8"))
    assert torch.equal(got, ref), int((got != ref).sum())
    assert int(ref.max()) == 255 and int(ref.min()) == 0


def test_undersized_workspaces_are_refused_without_a_launch():
    """Every C-ABI entry that takes a caller-owned workspace (or the plan arena) takes its size: a buffer smaller than the
    library's own *_bytes() answer comes back as NUNET_EINVAL with a message - not as a GPU memory fault (round 2's loss
    workspace fault went exactly through this gap). The refused calls must not have written anything either: the guard
    bands of this module's fixture sit right behind the short buffers."""
    lib = L.lib()
    n, per = 4, 96 * 96
    x = torch.randn(n, per, device=DEV)
    t = (torch.rand(n, per, device=DEV) > 0.5).float()
    loss = torch.zeros(1, device=DEV)
    # BCEDice
    need = lib.nunet_bce_dice_ws_bytes(n)
    short = torch.zeros(need // 4 - 1, device=DEV)
    assert lib.nunet_bce_dice_fwd(L.ptr(x), L.ptr(t), n, per, L.ptr(short), L.nbytes(short), L.ptr(loss), L.stream()) == -1
    assert b"nunet_bce_dice_ws_bytes" in lib.nunet_last_error()
    assert lib.nunet_bce_dice_bwd(L.ptr(x), L.ptr(t), n, per, L.ptr(short), L.nbytes(short), None, L.ptr(torch.zeros_like(x)), L.stream()) == -1
    ok = torch.zeros((need + 3) // 4, device=DEV)
    L.check(lib.nunet_bce_dice_fwd(L.ptr(x), L.ptr(t), n, per, L.ptr(ok), L.nbytes(ok), L.ptr(loss), L.stream()), "exact size is accepted")
    # fused loss step, both losses
    dl, lo = torch.zeros(2, n, per, device=DEV), torch.zeros(3, device=DEV)
    xx = torch.randn(2, n, per, device=DEV)
    for kind in (L.LOSS_BCE_DICE, L.LOSS_LOVASZ_HINGE):
        need = lib.nunet_loss_step_ws_bytes(n, per, 2, kind)
        assert need > 0
        short = torch.zeros(need // 8 - 1, dtype=torch.float64, device=DEV)
        rc = lib.nunet_loss_step(L.ptr(xx), L.ptr(t), n, per, 2, kind, L.ptr(short), L.nbytes(short), L.ptr(dl), L.ptr(lo), None, 0.0, L.stream())
        assert rc == -1 and b"nunet_loss_step_ws_bytes" in lib.nunet_last_error()
    # Lovasz hinge, small (in-LDS) and large (global sort) images
    for pp in (per, 256 * 256):
        need = lib.nunet_lovasz_ws_bytes(n, pp)
        short = torch.zeros(max(need - 8, 8), dtype=torch.uint8, device=DEV)
        xl, tl = torch.randn(n, pp, device=DEV), torch.zeros(n, pp, device=DEV)
        rc = lib.nunet_lovasz_hinge_fwd(L.ptr(xl), L.ptr(tl), n, pp, L.ptr(short), L.nbytes(short), L.ptr(torch.zeros_like(xl)), L.ptr(loss), L.stream())
        assert rc == -1 and b"nunet_lovasz_ws_bytes" in lib.nunet_last_error()
    # weight-gradient slabs
    s0 = torch.zeros(2, 12, 12, 32, dtype=torch.bfloat16, device=DEV)
    dyb = torch.zeros(2, 12, 12, 32, dtype=torch.bfloat16, device=DEV)
    probe = wgrad_desc(L.BF16, 2, 12, 12, s0, 32, 32, None, 0, dyb, 32, None, 0, 64)
    ks = lib.nunet_conv3x3_wgrad_slabs(C.byref(probe))
    assert ks > 1
    slabs = torch.zeros((ks - 1) * 9 * 32 * 32, device=DEV)           # one slab short
    d = wgrad_desc(L.BF16, 2, 12, 12, s0, 32, 32, None, 0, dyb, 32, slabs, 0, 64)
    assert lib.nunet_conv3x3_wgrad(C.byref(d), L.stream()) == -1 and b"slabs of this launch need" in lib.nunet_last_error()
    # the plan arena
    cfg = L.PlanCfg(2, 32, 32, 3, 1, 0, L.BF16, 0)
    p = lib.nunet_plan_create(C.byref(cfg))
    try:
        nb = lib.nunet_plan_arena_bytes(p)
        arena = torch.zeros(nb - 256, dtype=torch.uint8, device=DEV)
        params = torch.zeros(lib.nunet_plan_param_count(p), device=DEV)
        bnb = torch.zeros(lib.nunet_plan_bnbuf_count(p), device=DEV)
        logits = torch.zeros(1, 2, 1, 32, 32, device=DEV)
        inp = torch.zeros(2, 3, 32, 32, device=DEV)
        assert lib.nunet_plan_forward(p, L.ptr(params), L.ptr(bnb), None, L.ptr(inp), L.ptr(arena), L.nbytes(arena), L.ptr(logits), 1, L.stream()) == -1
        assert b"nunet_plan_arena_bytes" in lib.nunet_last_error()
        assert lib.nunet_plan_backward(p, L.ptr(params), L.ptr(logits), L.ptr(arena), L.nbytes(arena), L.ptr(params), 0, L.stream()) == -1
        lr = torch.zeros(1, device=DEV)
        assert lib.nunet_plan_sgd(p, L.ptr(params), L.ptr(bnb), L.ptr(arena), L.nbytes(arena), L.ptr(lr), 0.9, 1e-4, 0, 1.0, None, L.stream()) == -1
        assert lib.nunet_plan_update(p, L.ptr(params), L.ptr(bnb), L.ptr(arena), L.nbytes(arena), L.ptr(lr), 0.9, 1e-4, 0, 1.0, None, L.stream()) == -1
        assert lib.nunet_plan_repack(p, L.ptr(params), L.ptr(arena), L.nbytes(arena), L.stream()) == -1
    finally:
        lib.nunet_plan_destroy(p)


def test_iou_counts_at_the_sigmoid_threshold():
    """iou_score thresholds `sigmoid(x) > 0.5` in fp32 (reference metrics.py:10-12). In fp32 the sigmoid of a tiny positive
    logit rounds to exactly 0.5, so `x > 0` over-counts those pixels; the kernel compares against the smallest logit whose
    REFERENCE sigmoid exceeds 0.5 (metrics.iou_logit_threshold, bisection with the host's torch.sigmoid). Integer counts
    must equal the host expression's on every edge value."""
    thr = nunet_amd.metrics.iou_logit_threshold()
    assert 0.0 < thr < 1e-6
    f32 = np.float32
    below = float(np.nextafter(f32(thr), f32(0)))
    above = float(np.nextafter(f32(thr), f32(1)))
    edges = [0.0, -0.0, 1e-30, 1e-12, 1e-9, 2e-8, 5.9e-8, 6e-8, below, thr, above, 1e-7, 1.2e-7, 1e-6, -thr, -1e-9, 1.0, -1.0,
             float("inf"), float("-inf"), float("nan")]
    x = np.zeros(256, f32)
    x[:len(edges)] = np.array(edges, f32)
    x[len(edges):] = np.linspace(-3e-7, 3e-7, 256 - len(edges)).astype(f32)
    for tval in (1.0, 0.0):
        t = np.full(256, tval, f32)
        a = torch.sigmoid(torch.from_numpy(x)).numpy() > 0.5        # the reference expression (host fp32)
        b = t > 0.5
        cnt = nunet_amd.metrics.iou_counts(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV))
        assert cnt.tolist() == [int((a & b).sum()), int((a | b).sum())], (tval, cnt.tolist())
    assert not bool(torch.sigmoid(torch.from_numpy(x))[2] > 0.5) and x[2] > 0      # the case `x > 0` got wrong


@pytest.mark.parametrize("kind", ["BCEDiceLoss", "LovaszHingeLoss"])
@pytest.mark.parametrize("heads", [1, 4])
def test_loss_step_matches_the_standalone_losses(kind, heads):
    """nunet_loss_step (what the fused training step runs, trains.py:118-128): loss of every head, their mean, the gradient
    of the mean, IoU of the LAST head and the epoch meters - against the stand-alone loss entries (themselves pinned to the
    reference goldens above) and the host IoU expression, for both losses."""
    from oracle import nunet_oracle as O
    n, h, w = 3, 40, 56
    per = h * w
    g = torch.Generator().manual_seed(17 + heads)
    x = torch.randn(heads, n, 1, h, w, generator=g) * 1.5
    x[-1, 0, 0, 0, :8] = torch.tensor([0.0, 1e-9, 5e-8, 9e-8, 1e-7, -1e-9, 2e-7, 0.0])      # sigmoid-threshold edge values in the IoU head
    t = (torch.rand(n, 1, h, w, generator=g) > 0.6).float()
    lib = L.lib()
    k = L.LOSS_BCE_DICE if kind == "BCEDiceLoss" else L.LOSS_LOVASZ_HINGE
    xd, td = x.to(DEV).contiguous(), t.to(DEV).contiguous()
    ws = torch.zeros((lib.nunet_loss_step_ws_bytes(n, per, heads, k) + 7) // 8, dtype=torch.float64, device=DEV)
    dl = torch.full((heads, n, per), 9.0, device=DEV)
    lo = torch.zeros(heads + 1, device=DEV)
    meters = torch.zeros(4, dtype=torch.float64, device=DEV)
    meters[0] = 2.0
    thr = nunet_amd.metrics.iou_logit_threshold()
    for _ in range(2):        # twice: the workspace carries nothing over from one call to the next
        L.check(lib.nunet_loss_step(L.ptr(xd), L.ptr(td), n, per, heads, k, L.ptr(ws), L.nbytes(ws), L.ptr(dl), L.ptr(lo), L.ptr(meters), thr, L.stream()), "loss_step")
    crit = getattr(nunet_amd.losses, kind)()
    ref_l, ref_g = [], []
    for q in range(heads):
        xq = xd[q].clone().requires_grad_(True)
        lq = crit(xq, td)
        (lq / heads).backward()
        ref_l.append(float(lq.detach()))
        ref_g.append(xq.grad.reshape(n, per))
    got_l = lo.tolist()
    for q in range(heads):
        assert abs(got_l[q] - ref_l[q]) < 2e-6 * max(1.0, abs(ref_l[q]))
        np.testing.assert_allclose(dl[q].cpu().numpy(), ref_g[q].cpu().numpy(), atol=1e-9, rtol=2e-5)
    assert abs(got_l[heads] - sum(ref_l) / heads) < 2e-6 * max(1.0, abs(sum(ref_l) / heads))
    a = torch.sigmoid(x[-1].reshape(-1)).numpy() > 0.5
    b = t.reshape(-1).numpy() > 0.5
    inter, union = int((a & b).sum()), int((a | b).sum())
    m = meters.tolist()
    assert m[2] == inter and m[3] == union
    assert abs(m[1] - 2 * (inter + 1e-5) / (union + 1e-5)) < 1e-12
    assert abs(m[0] - 2.0 - 2 * got_l[heads]) < 1e-6
    assert abs(O.iou_score(x[-1], t) - (inter + 1e-5) / (union + 1e-5)) < 1e-12     # the oracle agrees with the host expression
