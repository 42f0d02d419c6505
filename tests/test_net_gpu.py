"""Whole-path parity on the MI355X: NestedUNet / UNet forward, loss, IoU, gradients,
BN running statistics and short SGD trajectories against (a) golden vectors captured
from the imported reference and (b) the CPU oracle in fp64 on the same seeded inputs.

Tolerances. north_star asks for logits within 1e-4 (fp32) of the reference CPU
forward; that is asserted directly on the goldens, for every case but one: d_n2_16x16_k1,
whose level-4 BatchNorm normalises over two 1x1 "images" (it divides by a near-zero
std, the reference's OWN fp32 logits are further than 1e-4 from an fp64 evaluation),
is bounded by 4x that measured conditioning error. Gradients of this network are
ill-conditioned in fp32 (the reference's own fp32 gradients sit ~4e-3 relative from
an fp64 evaluation), so gradient parity is asserted against the fp64
oracle with the bound max(4 x reference-fp32 error, 1e-3) per tensor. Every reduction of
the HIP path is order-independent or fixed-order (bit-reproducible runs): no tolerance
here allows for summation-order noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import nunet_amd  # noqa: E402
from nunet_amd import _lib as L  # noqa: E402
from conftest import GOLDEN_CASES, load_golden  # noqa: E402
from oracle import nunet_oracle as O  # noqa: E402

DEV = "cuda:0"


def build(cfg, synth, dtype="fp32", unet=False):
    n, h, w, cin, ncls, ds, train, fresh = cfg
    cls = nunet_amd.archs.UNet if unet else nunet_amd.archs.NestedUNet
    m = cls(ncls, cin, ds, dtype=dtype)
    st = synth.closed_form_state(ncls, cin, ds, fresh)
    if unet:   # same hash generator, the U-Net's own shapes (the state tests/golden/make_golden.py run_unet() loads)
        st = synth.closed_form_state_unet(ncls, cin)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.to(DEV)
    img, msk = synth.synth_batch(n, h, w, cin, ncls, seed=1234)
    return m, st, torch.from_numpy(img), torch.from_numpy(msk)


def run_step(m, x, t, ds):
    crit = nunet_amd.losses.BCEDiceLoss()
    out = m(x.to(DEV))
    tg = t.to(DEV)
    if ds:
        loss = 0
        for o in out:
            loss = loss + crit(o, tg)
        loss = loss / len(out)
        last = out[-1]
    else:
        loss = crit(out, tg)
        last = out
    iou = nunet_amd.metrics.iou_score(last, tg)
    m.zero_grad()
    loss.backward()
    return out, loss, iou


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_fp32_matches_reference_goldens(name, synth):
    cfg = GOLDEN_CASES[name]
    n, h, w, cin, ncls, ds, train, fresh = cfg
    g = load_golden(name)
    m, st, x, t = build(cfg, synth)
    if not train:
        m.eval()
        with torch.no_grad():
            o = m(x.to(DEV))
        scale = max(1.0, float(np.abs(g["logits0"]).max()))
        assert float(np.abs(o.cpu().numpy() - g["logits0"]).max()) < 1e-4 * scale
        loss = nunet_amd.losses.BCEDiceLoss()(o, t.to(DEV))
        assert abs(float(loss) - float(g["loss"])) < 1e-4 * max(1.0, float(g["loss"]))
        assert abs(nunet_amd.metrics.iou_score(o, t.to(DEV)) - float(g["iou"])) < 5e-3
        return
    m.train()
    out, loss, iou = run_step(m, x, t, ds)
    outs = out if ds else [out]
    # north_star: fp32 logits within 1e-4 of the reference - asserted as such for every case. The ONE exception is named:
    # d_n2_16x16_k1, where BatchNorm over two 1x1 "images" at level 4 divides by a tiny std and the reference's OWN fp32
    # logits are further than 1e-4 from an fp64 evaluation; its bound is 4x that conditioning error.
    with torch.no_grad():
        l64 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float64)(x.double())
        l32 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float32)(x)
    l64, l32 = (l64 if ds else [l64]), (l32 if ds else [l32])
    for k, o in enumerate(outs):
        ref = g["logits%d" % k]
        scale = max(1.0, float(np.abs(ref).max()))
        cond = float((l32[k].double() - l64[k]).abs().max()) / scale
        bound = max(1e-4, 4 * cond) if name == "d_n2_16x16_k1" else 1e-4
        assert float(np.abs(o.detach().cpu().numpy() - ref).max()) < bound * scale, (name, k, cond)
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    # IoU is a hard threshold on logits: allow the flip of a handful of near-zero logits
    assert abs(iou - float(g["iou"])) < 5e-3
    # BN running statistics after one step
    sd = m.state_dict()
    for k, nm in enumerate(str(s) for s in g["bn_names"]):
        assert abs(float(sd[nm].double().sum()) - g["bn_sum"][k]) < 2e-4 * (1 + abs(g["bn_sum"][k])), nm
    for key in g.files:
        if key.startswith("bn/"):
            np.testing.assert_allclose(sd[key[3:]].cpu().numpy(), g[key], rtol=2e-4, atol=2e-6, err_msg=key)
    # gradients vs the fp64 oracle, bounded by the reference's own fp32 error
    o64 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float64)
    l64, _ = O.criterion_ds(o64(x.double()), t.double())
    l64.backward()
    o32 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float32)
    l32, _ = O.criterion_ds(o32(x), t)
    l32.backward()
    names = [str(s) for s in g["grad_names"]]
    worst = 0.0
    for k, (nm, p) in enumerate(m.named_parameters()):
        assert nm == names[k]
        g64 = o64.params[nm].grad
        g32 = o32.params[nm].grad.double()
        mine = p.grad.detach().cpu().double()
        nrm = float(g64.norm())
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            # analytically zero (BatchNorm follows); only rounding noise, must stay tiny
            assert float(mine.abs().max()) < 1e-4, nm
            continue
        err_ref = float((g32 - g64).norm()) / (nrm + 1e-30)
        err_mine = float((mine - g64).norm()) / (nrm + 1e-30)
        worst = max(worst, err_mine)
        assert err_mine < max(4 * err_ref, 1e-3), (nm, err_mine, err_ref)
        # the golden (reference fp32) gradients agree too, within the reference's own conditioning: L2 norm, the 64
        # strided samples of every tensor, and the full small tensors
        tol = max(4 * err_ref, 1e-3)
        assert abs(float(mine.norm()) - g["grad_l2"][k]) < tol * g["grad_l2"][k] + 1e-7, nm
        flat = mine.reshape(-1)
        stride = max(1, flat.numel() // 64)
        smp = flat[::stride][:64].numpy()
        gs = g["grad_sample"][k][:smp.size].astype(np.float64)
        assert np.linalg.norm(smp - gs) < 3 * tol * max(np.linalg.norm(gs), g["grad_l2"][k] * (smp.size / flat.numel()) ** 0.5) + 1e-7, nm
        if "grad/" + nm in g.files:
            gfull = g["grad/" + nm].astype(np.float64)
            assert np.linalg.norm(mine.numpy() - gfull) < tol * np.linalg.norm(gfull) + 1e-7, nm
    print(name, "worst grad rel err vs fp64 oracle:", worst)


@pytest.mark.parametrize("dtype,tol", [("bf16", 6e-2), ("fp16", 1e-2)])
def test_reduced_precision_forward_and_grads(dtype, tol, synth):
    cfg = GOLDEN_CASES["a_n2_32x32_k1"]
    g = load_golden("a_n2_32x32_k1")
    m, st, x, t = build(cfg, synth, dtype=dtype)
    m.train()
    out, loss, iou = run_step(m, x, t, False)
    ref = g["logits0"]
    lerr = float(np.abs(out.detach().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    print(dtype, "logit rel err", lerr, "loss", float(loss.detach()), float(g["loss"]))
    assert lerr < tol
    assert abs(float(loss.detach()) - float(g["loss"])) < tol
    # Gradients: the yardstick is the fp64 oracle with the SAME storage points rounded to the 16-bit type (packed weights,
    # raw conv outputs, activations, pooled / upsampled tensors and their gradients; exact arithmetic in between) - what
    # 16-bit storage alone does to this ill-conditioned gradient. Per tensor the HIP path may be at most 1.5x (+2 %) as far
    # from the exact gradient as that emulation, the medians must agree within 20 %, and the two rounded evaluations
    # must be closer to each other than to the exact one (the bound used for BASELINE configs[1] below).
    sdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    o64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64)
    O.bce_dice_loss(o64(x.double()), t.double()).backward()
    e64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64, storage=sdt)
    le = O.bce_dice_loss(e64(x.double()), t.double())
    le.backward()
    assert abs(float(loss.detach()) - float(le)) < (2e-3 if dtype == "bf16" else 3e-4)
    e_hip, e_emu, e_mut = {}, {}, {}
    for nm, p in m.named_parameters():
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            continue
        g64, ge, mine = o64.params[nm].grad, e64.params[nm].grad, p.grad.cpu().double()
        nrm = float(g64.norm()) + 1e-30
        e_hip[nm] = float((mine - g64).norm()) / nrm
        e_emu[nm] = float((ge - g64).norm()) / nrm
        e_mut[nm] = float((mine - ge).norm()) / nrm
    med = lambda d: float(np.median(list(d.values())))
    print(dtype, "grad rel-L2 max / median: HIP vs exact %.3f / %.3f, %s-storage oracle vs exact %.3f / %.3f, HIP vs that oracle %.3f / %.3f"
          % (max(e_hip.values()), med(e_hip), dtype, max(e_emu.values()), med(e_emu), max(e_mut.values()), med(e_mut)))
    for nm in e_hip:
        assert e_hip[nm] <= 1.5 * e_emu[nm] + 0.02, (nm, e_hip[nm], e_emu[nm])
    assert med(e_hip) <= 1.2 * med(e_emu) + 0.01
    assert med(e_mut) < med(e_hip)


def test_features_match_oracle(synth):
    """Every block output x_{i,j} (NHWC level-buffer slots) vs the oracle's features."""
    cfg = GOLDEN_CASES["a_n2_32x32_k1"]
    m, st, x, t = build(cfg, synth)
    m.train()
    with torch.no_grad():
        m(x.to(DEV))
    pl = m.plan_for(x.to(DEV))
    o64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64)
    o64(x.double())
    for (i, j), f in o64.features.items():
        got = pl.feature(i, j).float().permute(0, 3, 1, 2).cpu().double()
        err = float((got - f.detach()).abs().max() / f.detach().abs().max())
        assert err < 2e-4, ((i, j), err)


def test_unet_matches_oracle(synth):
    """Plain U-Net (reference finished/archs1.py:35-71) shares every kernel."""
    import torch.nn.functional as F
    cfg = (2, 32, 32, 3, 1, False, True, True)
    m, st, x, t = build(cfg, synth, unet=True)
    m.train()
    out, loss, iou = run_step(m, x, t, False)
    # oracle: the nested oracle's block primitive on the U-Net wiring
    net = O.OracleNet(st, 1, 3, False, dtype=torch.float64)
    xs = {}
    inp = x.double()
    for i in range(5):
        xs[i] = net._block(inp if i == 0 else F.max_pool2d(xs[i - 1], 2, 2), i, 0)
    d = xs[4]
    for i in (3, 2, 1, 0):
        up = F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=True)
        d = net._block(torch.cat([xs[i], up], 1), i, 4 - i)
    ref = F.conv2d(d, net.params["final.weight"], net.params["final.bias"])
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 1e-4
    # the reference's own U-Net (finished/archs1.py:35-71) on the same weights and inputs: golden logits, loss, IoU,
    # BN running statistics and gradients
    g = load_golden("h_unet_n2_32x32_k1")
    assert float(np.abs(out.detach().cpu().numpy() - g["logits0"]).max()) < 1e-4 * max(1.0, float(np.abs(g["logits0"]).max()))
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    assert abs(iou - float(g["iou"])) < 5e-3
    sd = m.state_dict()
    for k, nm in enumerate(str(s_) for s_ in g["bn_names"]):
        assert abs(float(sd[nm].double().sum()) - g["bn_sum"][k]) < 2e-4 * (1 + abs(g["bn_sum"][k])), nm
    for k, (nm, p) in enumerate(m.named_parameters()):
        assert nm == str(g["grad_names"][k])
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            continue
        assert abs(float(p.grad.double().norm()) - g["grad_l2"][k]) < 2e-2 * g["grad_l2"][k] + 1e-7, nm
        if "grad/" + nm in g.files:
            gfull = g["grad/" + nm].astype(np.float64)
            assert np.linalg.norm(p.grad.cpu().double().numpy() - gfull) < 2e-2 * np.linalg.norm(gfull) + 1e-7, nm
    l64 = O.bce_dice_loss(ref, t.double())
    l64.backward()
    assert abs(float(loss) - float(l64)) < 2e-5
    for nm, p in m.named_parameters():
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            continue
        g64 = net.params[nm].grad
        assert float((p.grad.cpu().double() - g64).norm() / g64.norm()) < 2e-2, nm


def test_trajectory_against_reference(synth):
    """8 SGD steps + cosine schedule (reference trains.py:113-135,229-239,323-324).
    The loop is chaotic at the 1e-3 level (fp32 reference vs fp64 evaluation of the
    same loop differ by up to 1e-2 in loss after 8 steps), hence the band."""
    g = load_golden("trajectory_n4_32x32")
    m = nunet_amd.archs.NestedUNet(1, 3, False)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.closed_form_state(1, 3, False, True).items()})
    m = m.to(DEV)
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3, momentum=0.9,
                          nesterov=False, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=4, eta_min=1e-5)
    crit = nunet_amd.losses.BCEDiceLoss()
    meter = nunet_amd.utils.AverageMeter()
    step = 0
    for ep in range(4):
        m.train()
        for _ in range(2):
            img, msk = synth.synth_batch(4, 32, 32, 3, 1, seed=1234 + step)
            x, t = torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV)
            out = m(x)
            loss = crit(out, t)
            iou = nunet_amd.metrics.iou_score(out, t)
            opt.zero_grad()
            loss.backward()
            opt.step()
            meter.update(loss.item(), 4)
            assert abs(opt.param_groups[0]["lr"] - g["lr"][step]) < 1e-12
            assert abs(loss.item() - g["loss"][step]) < (1e-4 if step < 2 else 2e-2), (step, loss.item(), g["loss"][step])
            assert abs(iou - g["iou"][step]) < (5e-3 if step < 2 else 5e-2)
            step += 1
        sched.step()
    m.eval()
    img, msk = synth.synth_batch(4, 32, 32, 3, 1, seed=99)
    with torch.no_grad():
        o = m(torch.from_numpy(img).to(DEV))
    vloss = float(crit(o, torch.from_numpy(msk).to(DEV)))
    assert abs(vloss - float(g["val_loss"])) < 3e-2
    # state_dict round trip into a fresh module (reference trains.py:345 / val.py:58)
    m2 = nunet_amd.archs.NestedUNet(1, 3, False)
    m2.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    m2 = m2.to(DEV).eval()
    with torch.no_grad():
        o2 = m2(torch.from_numpy(img).to(DEV))
    assert torch.equal(o, o2)


def test_grad_accumulation_and_zero_grad_semantics(synth):
    cfg = GOLDEN_CASES["a_n2_32x32_k1"]
    m, st, x, t = build(cfg, synth)
    m.train()
    run_step(m, x, t, False)
    g1 = [p.grad.clone() for p in m.parameters()]
    crit = nunet_amd.losses.BCEDiceLoss()
    # second backward without zero_grad accumulates (autograd semantics)
    crit(m(x.to(DEV)), t.to(DEV)).backward()
    named = [(k, p) for k, p in m.named_parameters() if not (k.endswith("conv1.bias") or k.endswith("conv2.bias"))]
    g1 = {k: a for (k, _), a in zip(m.named_parameters(), g1)}

    def close(a, b):      # semantics check (x1 vs x2); the second forward sees running-stat-independent batch statistics: same gradient
        return float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-10

    for k, p in named:      # conv biases before BN have pure-noise gradients: skipped
        assert close(p.grad, 2 * g1[k]), k
    m.zero_grad(set_to_none=False)
    crit(m(x.to(DEV)), t.to(DEV)).backward()
    for k, p in named:
        assert close(p.grad, g1[k]), k

def test_fused_train_step_graph_matches_eager_and_oracle(synth):
    """TrainStep (trainer.py): the hipGraph-captured step must follow the same trajectory
    as the eager step and as the CPU oracle loop (reference trains.py:113-135)."""
    from nunet_amd.trainer import TrainStep
    n, hw = 4, 32
    st = synth.closed_form_state(1, 3, False, True)
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=1234 + k) for k in range(3)]
    net = O.OracleNet(st, 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    ref = [O.train_step(net, opt, torch.from_numpy(b[0]), torch.from_numpy(b[1])) for b in batches]
    for graph in (False, True):
        m = nunet_amd.archs.NestedUNet(1, 3, False)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), use_graph=graph)
        if graph:
            ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
        for k, b in enumerate(batches):
            ts.reset_meters()
            ts.step(torch.from_numpy(b[0]).to(DEV), torch.from_numpy(b[1]).to(DEV))
            loss, iou = ts.epoch_stats()
            assert abs(loss - ref[k][0]) < (1e-4 if k == 0 else 3e-3), (graph, k, loss, ref[k][0])
            assert abs(iou - ref[k][1]) < 2e-2, (graph, k, iou, ref[k][1])
        assert bool(torch.isfinite(ts.eng.flat_params).all())
        # parameters after 3 steps agree with the oracle's (momentum + weight decay path)
        w = m.conv0_4.conv2.weight.detach().cpu()
        rw = net.params["conv0_4.conv2.weight"].detach()
        assert float((w - rw).abs().max()) < 2e-5


@pytest.mark.parametrize("dtype,iou_band", [("fp32", 0.02), ("bf16", 0.03)])
def test_training_log_follows_reference(dtype, iou_band, synth):
    """'val IoU vs ref' (BASELINE.json metric, second half) in its offline form: the REFERENCE
    model/loss/metric trained on the seeded learnable blob set (tests/golden/make_golden.py trainlog: SGD 1e-2,
    momentum 0.9, wd 1e-4, cosine, 30 epochs of 512 images, bs 16, 96x96, validation on 128 held-out images,
    reference trains.py:106-188,321,331-339) vs the same loop on the HIP path (same init seed, same shuffle stream)."""
    from nunet_amd.trainer import TrainStep, cosine_lr
    from nunet_amd.metrics import iou_counts, iou_from_counts
    g = load_golden("train_log_blobs")
    ref = g["log"]
    epochs, train_size, val_size, bs, hw, lr = (int(v) if k < 5 else float(v) for k, v in enumerate(g["config"]))
    torch.manual_seed(41)
    m = nunet_amd.archs.NestedUNet(1, 3, False, dtype=dtype).to(DEV).train()
    img, msk = synth.synth_blob_pairs(train_size, hw, hw, seed=1000)
    vimg, vmsk = synth.synth_blob_pairs(val_size, hw, hw, seed=2000)
    x, t = torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV)
    vx, vt = torch.from_numpy(vimg).to(DEV), torch.from_numpy(vmsk).to(DEV)
    ts = TrainStep(m, (bs, 3, hw, hw), lr=lr, momentum=0.9, weight_decay=1e-4)
    ts.capture(x[:bs], t[:bs])
    crit = nunet_amd.losses.BCEDiceLoss()
    gen = torch.Generator().manual_seed(41)
    rows = []
    for ep in range(epochs):
        perm = torch.randperm(train_size, generator=gen).to(DEV)
        ts.set_lr(cosine_lr(lr, 1e-5, ep, epochs))
        ts.reset_meters()
        m.train()
        for k in range(train_size // bs):
            idx = perm[k * bs:(k + 1) * bs]
            ts.step(x[idx], t[idx])
        tl, ti = ts.epoch_stats()
        m.eval()
        vl = vi = 0.0
        with torch.no_grad():
            for k in range(0, val_size, bs):
                o = m(vx[k:k + bs].contiguous())
                vl += float(crit(o, vt[k:k + bs].contiguous())) * bs
                vi += iou_from_counts(iou_counts(o.contiguous(), vt[k:k + bs].contiguous())) * bs
        rows.append((tl, ti, vl / val_size, vi / val_size))
        print("epoch", ep, dtype, "hip", rows[-1], "ref", tuple(ref[ep][2:]))
    rows = np.array(rows)
    assert abs(ref[0][1] - lr) < 1e-12 and np.all(np.isfinite(rows))
    # the metric itself: validation IoU, mean of the last five epochs, against the reference's
    ref_iou, hip_iou = float(ref[-5:, 5].mean()), float(rows[-5:, 3].mean())
    assert ref_iou > 0.6, "fixture: the reference itself did not learn the task"
    assert abs(hip_iou - ref_iou) <= iou_band, (hip_iou, ref_iou)
    # per epoch once both have converged (the early epochs, where eval-mode BN lags fast-moving weights, swing in the
    # reference too): second half of the schedule
    half = epochs // 2
    assert np.all(np.abs(rows[half:, 3] - ref[half:, 5]) <= 2 * iou_band), (rows[half:, 3], ref[half:, 5])
    assert np.all(np.abs(rows[half:, 2] - ref[half:, 4]) <= 0.05), (rows[half:, 2], ref[half:, 4])            # val loss
    # training side: first epoch (no divergence yet) tightly, every epoch in a band, and it learns
    assert abs(rows[0, 0] - ref[0, 2]) < 0.03 and abs(rows[0, 1] - ref[0, 3]) < 0.05, (rows[0], ref[0])
    assert np.all(np.abs(rows[:, 0] - ref[:, 2]) < 0.08), (rows[:, 0], ref[:, 2])
    assert np.all(np.abs(rows[half:, 1] - ref[half:, 3]) < 2 * iou_band), (rows[half:, 1], ref[half:, 3])
    assert rows[-1, 0] < 0.5 * rows[0, 0]


@pytest.mark.parametrize("dtype,tol", [("fp32", 5e-5), ("bf16", 1e-3)])
def test_cfg1_bs8_96_training_follows_oracle(dtype, tol, synth):
    """BASELINE.json configs[0] (batch 8, 96x96 - the reference's own CPU-runnable case, trains.py defaults: BCEDiceLoss,
    SGD lr 1e-3 / momentum 0.9 / wd 1e-4) through the hipGraph TrainStep against the fp32 CPU oracle running the same
    loop (reference trains.py:113-135,229-231): per-step loss over 12 steps from the reference's default initialisation."""
    from nunet_amd.trainer import TrainStep
    n, hw, steps = 8, 96, 12
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, False).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=4321 + k) for k in range(3)]
    m = nunet_amd.archs.NestedUNet(1, 3, False, dtype=dtype)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ts = TrainStep(m, (n, 3, hw, hw))
    ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
    hip = []
    for k in range(steps):
        img, msk = batches[k % 3]
        ts.reset_meters()
        ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        hip.append(ts.epoch_stats()[0])
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    net = O.OracleNet({k: v.numpy() for k, v in sd.items()}, 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    ref = [O.train_step(net, opt, torch.from_numpy(batches[k % 3][0]), torch.from_numpy(batches[k % 3][1]))[0] for k in range(steps)]
    print("hip", " ".join("%.5f" % v for v in hip)); print("ref", " ".join("%.5f" % v for v in ref))
    assert np.all(np.isfinite(hip))
    assert np.max(np.abs(np.array(hip) - np.array(ref))) < tol, (hip, ref)


def test_cfg2_bf16_bs16_96_training_follows_oracle(synth):
    """BASELINE.json configs[1] itself (bf16 storage, batch 16, 96x96, BCEDiceLoss, SGD defaults) through the
    hipGraph TrainStep, against the fp32 CPU oracle running the same loop (reference trains.py:113-135,229-231):
    per-step loss over 20 steps, and every parameter gradient of the first step against the fp64 oracle."""
    from nunet_amd.trainer import TrainStep
    n, hw, steps = 16, 96, 20
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, False).state_dict().items()}   # default init (reference init)
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=1234 + k) for k in range(4)]
    m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ts = TrainStep(m, (n, 3, hw, hw), keep_grads=True)
    ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
    hip, grads0 = [], None
    for k in range(steps):
        img, msk = batches[k % 4]
        ts.reset_meters()
        ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        hip.append(ts.epoch_stats()[0])
        if k == 0:
            grads0 = {nm: p.grad.detach().cpu().double().clone() for nm, p in m.named_parameters()}
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    st = {k: v.numpy() for k, v in sd.items()}
    net = O.OracleNet(st, 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    ref = [O.train_step(net, opt, torch.from_numpy(batches[k % 4][0]), torch.from_numpy(batches[k % 4][1]))[0] for k in range(steps)]
    print("hip", " ".join("%.4f" % v for v in hip)); print("ref", " ".join("%.4f" % v for v in ref))
    assert np.all(np.isfinite(hip))
    assert np.max(np.abs(np.array(hip) - np.array(ref))) < 1e-3, (hip, ref)
    assert abs(hip[0] - ref[0]) < 5e-4
    # Gradients of step 0. Against the UNROUNDED fp64 network 16-bit storage leaves large per-tensor errors at
    # initialisation: the loss gradient is nearly orthogonal to the features (d gamma = sum dz * xhat is a correlation
    # that almost cancels), so the fp32 reference itself is ~1e-5 off and bf16's 2^16 x coarser rounding reaches tens of
    # percent - a property of the storage format, not of the kernels. The kernel statement is therefore made against the
    # fp64 oracle with the same KIND of 16-bit storage points emulated (oracle storage=bfloat16: packed weights, raw conv
    # outputs, activations, upsampled / pooled tensors and all their gradients rounded, exact arithmetic in between):
    # per tensor, the HIP path's distance from the exact gradient may not exceed 1.5x the distance bf16 storage alone
    # causes (+2 %), and the two rounded evaluations must be closer to each other than to the exact one.
    x0, t0 = torch.from_numpy(batches[0][0]).double(), torch.from_numpy(batches[0][1]).double()
    o64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64)
    O.bce_dice_loss(o64(x0), t0).backward()
    e64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64, storage=torch.bfloat16)
    le = O.bce_dice_loss(e64(x0), t0)
    le.backward()
    assert abs(hip[0] - float(le)) < 2e-4
    e_hip, e_emu, e_mut = {}, {}, {}
    for nm, gmine in grads0.items():
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            assert float(gmine.abs().max()) == 0.0, nm          # exact zeros: bias in front of a BatchNorm
            continue
        ge, g64 = e64.params[nm].grad, o64.params[nm].grad
        nrm = float(g64.norm()) + 1e-30
        e_hip[nm] = float((gmine - g64).norm()) / nrm
        e_emu[nm] = float((ge - g64).norm()) / nrm
        e_mut[nm] = float((gmine - ge).norm()) / nrm
    med = lambda d: float(np.median(list(d.values())))
    print("bf16 bs16 grad rel-L2 (max / median): HIP vs exact %.3f / %.3f, bf16-storage oracle vs exact %.3f / %.3f, HIP vs bf16-storage oracle %.3f / %.3f"
          % (max(e_hip.values()), med(e_hip), max(e_emu.values()), med(e_emu), max(e_mut.values()), med(e_mut)))
    for nm in e_hip:
        assert e_hip[nm] <= 1.5 * e_emu[nm] + 0.02, (nm, e_hip[nm], e_emu[nm])
    assert med(e_hip) <= 1.2 * med(e_emu) + 0.01
    assert med(e_mut) < med(e_hip)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_training_step_is_bit_reproducible(dtype, synth):
    """SURVEY.md §5.2 determinism check: two independent runs of three bs16 96x96 training steps (hipGraph, multi-lane)
    end with bit-identical parameters, BatchNorm buffers, momentum, loss and gradients. Every reduction has a fixed
    order (weight-gradient slabs, split-K slabs, head slabs, loss slabs) or is order-independent (fixed-point
    per-channel sums)."""
    from nunet_amd.trainer import TrainStep
    n, hw = 16, 96
    torch.manual_seed(3)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, True).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=77 + k) for k in range(3)]
    outs = []
    for run in range(2):
        m = nunet_amd.archs.NestedUNet(1, 3, True, dtype=dtype)       # deep supervision: four heads write GX[0] slots
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, keep_grads=True)
        ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
        for img, msk in batches:
            ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (ts.eng.flat_params, ts.eng.bnbuf, ts.eng.nbt, ts.mom, ts.eng.flat_grads, ts.loss_out, ts.meters)])
        del ts, m
    for a, b, nm in zip(outs[0], outs[1], ("params", "bn buffers", "nbt", "momentum", "grads", "loss", "meters")):
        assert torch.equal(a, b), nm


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", [1, 2])
def test_fused_update_equals_unpack_sgd_pack(dtype, mode, synth):
    """nunet_plan_update (mode 1: scratch -> SGD -> repacked weights in one launch) and nunet_plan_sgd (mode 2: scratch ->
    SGD) against the launches they replace (unpack into the OIHW gradient arena + nunet_sgd_step), applied to the SAME
    gradient scratch from the SAME state: parameters, momentum and gradients agree to rounding (momentum, weight decay,
    nesterov on); after mode 1 the next forward with the repack skipped gives the logits of a forward that repacks."""
    from nunet_amd.trainer import TrainStep
    cfg = GOLDEN_CASES["a_n2_32x32_k1"]
    m, st, x, t = build(cfg, synth, dtype=dtype)
    m.train()
    ts = TrainStep(m, tuple(x.shape), lr=5e-2, momentum=0.9, weight_decay=1e-3, nesterov=True, use_graph=False, fused_update=0, keep_grads=True)
    ts.x.copy_(x.to(DEV)); ts.t.copy_(t.to(DEV))
    ts.mom.normal_(0, 1e-3)                        # a non-trivial momentum buffer
    ts._fwd_loss(); ts._bwd(3)                     # gradient scratch complete, not yet unpacked
    eng = ts.eng
    p0, m0 = eng.flat_params.clone(), ts.mom.clone()
    ts._bwd(4); ts._opt()                          # reference: unpack + nunet_sgd_step
    torch.cuda.synchronize()
    pa, ma, ga = eng.flat_params.clone(), ts.mom.clone(), eng.flat_grads.clone()
    eng.flat_params.copy_(p0); ts.mom.copy_(m0); eng.flat_grads.zero_()
    ts.fused_update = mode
    ts._opt()                                      # nunet_plan_update / nunet_plan_sgd on the same scratch
    torch.cuda.synchronize()
    pb, mb, gb = eng.flat_params.clone(), ts.mom.clone(), eng.flat_grads.clone()
    assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())     # (the head slabs are summed in a different order)
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())
    assert float((ma - mb).abs().max()) <= 1e-5 * float(ma.abs().max()) + 1e-9
    if mode == 1:
        ts._packed = True
        ts._fwd_loss()                             # repack skipped: uses the weights plan_update packed
        l1 = ts.logits.clone()
        ts._packed = False
        ts._fwd_loss()                             # repacks from the fp32 parameters
        assert torch.equal(l1, ts.logits)          # same packed weights, deterministic reductions: bit-identical


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("ds", [False, True])
def test_optimiser_step_inside_the_backward_pass(dtype, ds, synth):
    """fused_update=3 (nunet_plan_set_inpass_update): every VGGBlock - and the heads - stepped and repacked as an op of the backward
    pass, behind its weight gradients. It is the arithmetic of nunet_plan_update (mode 1), launched in slices: after three captured
    steps parameters, momentum, p.grad, BatchNorm buffers and losses are BIT-identical to mode 1's, under every executor."""
    from nunet_amd.trainer import TrainStep
    n, hw = 16, 96
    torch.manual_seed(21)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, ds).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=500 + k) for k in range(3)]
    outs = []
    for mode, seg, sched in ((1, False, "lanes"), (3, False, "lanes"), (3, "flags", "list")):
        m = nunet_amd.archs.NestedUNet(1, 3, ds, dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True, fused_update=mode, segmented=seg, schedule=sched)
        ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
        for img, msk in batches:
            ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (ts.eng.flat_params, ts.mom, ts.eng.flat_grads, ts.eng.bnbuf, ts.loss_out)])
        del ts, m
    for other in outs[1:]:
        for a, b, nm in zip(outs[0], other, ("params", "momentum", "grads", "bn buffers", "loss")):
            assert torch.equal(a, b), nm


@pytest.mark.parametrize("ds", [False, True])
@pytest.mark.parametrize("graph", [False, True])
def test_lovasz_hinge_inside_the_fused_step(ds, graph, synth):
    """LovaszHingeLoss (reference losses.py:120-129, the loss of its published table README.md:102-108) as a TrainStep loss:
    inside the step's graph, under deep supervision the mean over heads (trains.py:118-123). One step against the
    generic autograd path (module forward -> nunet_amd.losses.LovaszHingeLoss -> backward -> the same SGD step)."""
    from nunet_amd.trainer import TrainStep
    n, hw = 4, 32
    torch.manual_seed(5)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, ds).state_dict().items()}
    img, msk = synth.synth_blob_pairs(n, hw, hw, seed=31)
    x, t = torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV)
    # generic path
    m0 = nunet_amd.archs.NestedUNet(1, 3, ds)
    m0.load_state_dict(sd)
    m0 = m0.to(DEV).train()
    crit = nunet_amd.losses.LovaszHingeLoss()
    out = m0(x)
    outs = out if ds else [out]
    losses = [crit(o, t) for o in outs]
    loss = sum(losses) / len(losses)
    loss.backward()
    g0 = {k: p.grad.detach().clone() for k, p in m0.named_parameters()}
    iou0 = nunet_amd.metrics.iou_score(outs[-1], t)
    opt = torch.optim.SGD(m0.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    opt.step()
    # fused step
    m1 = nunet_amd.archs.NestedUNet(1, 3, ds)
    m1.load_state_dict(sd)
    m1 = m1.to(DEV).train()
    ts = TrainStep(m1, (n, 3, hw, hw), lr=1e-2, momentum=0.9, weight_decay=1e-4, loss="LovaszHingeLoss", use_graph=graph)
    if graph:
        ts.capture(x, t)
    ts.reset_meters()
    ts.step(x, t)
    tl, ti = ts.epoch_stats()
    lo = ts.loss_out.tolist()
    for k, lk in enumerate(losses):
        assert abs(lo[k] - float(lk)) < 2e-6 * max(1.0, abs(float(lk))), (k, lo[k], float(lk))
    assert abs(tl - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    assert abs(ti - iou0) < 1e-12
    for k, p in m1.named_parameters():
        ref = g0[k]
        assert float((p.grad - ref).norm()) <= 1e-5 * float(ref.norm()) + 1e-9, k
    for (k, p), q in zip(m1.named_parameters(), m0.parameters()):
        assert float((p - q).abs().max()) < 1e-6, k
    with pytest.raises(L.NunetError):
        TrainStep(nunet_amd.archs.NestedUNet(2, 3, False).to(DEV), (n, 3, hw, hw), loss="LovaszHingeLoss")    # one class only, as the reference


def test_lovasz_training_log_follows_reference(synth):
    """The REFERENCE trained with LovaszHingeLoss (tests/golden/make_golden.py trainlog_lovasz: 10 epochs of 256 blob
    images, bs 16, 96x96, SGD 1e-2 / 0.9 / 1e-4, cosine; validation on 64 held-out images) vs the same loop through
    the fused HIP step in fp32: the loss the reference's only published accuracy table was made with (README.md:102-108)."""
    from nunet_amd.trainer import TrainStep, cosine_lr
    from nunet_amd.metrics import iou_counts, iou_from_counts
    g = load_golden("train_log_blobs_lovasz")
    ref = g["log"]
    epochs, train_size, val_size, bs, hw, lr = (int(v) if k < 5 else float(v) for k, v in enumerate(g["config"]))
    torch.manual_seed(41)
    m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="fp32").to(DEV).train()
    img, msk = synth.synth_blob_pairs(train_size, hw, hw, seed=1000)
    vimg, vmsk = synth.synth_blob_pairs(val_size, hw, hw, seed=2000)
    x, t = torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV)
    vx, vt = torch.from_numpy(vimg).to(DEV), torch.from_numpy(vmsk).to(DEV)
    ts = TrainStep(m, (bs, 3, hw, hw), lr=lr, momentum=0.9, weight_decay=1e-4, loss="LovaszHingeLoss")
    ts.capture(x[:bs], t[:bs])
    crit = nunet_amd.losses.LovaszHingeLoss()
    gen = torch.Generator().manual_seed(41)
    rows = []
    for ep in range(epochs):
        perm = torch.randperm(train_size, generator=gen).to(DEV)
        ts.set_lr(cosine_lr(lr, 1e-5, ep, epochs))
        ts.reset_meters()
        m.train()
        for k in range(train_size // bs):
            idx = perm[k * bs:(k + 1) * bs]
            ts.step(x[idx], t[idx])
        tl, ti = ts.epoch_stats()
        m.eval()
        vl = vi = 0.0
        with torch.no_grad():
            for k in range(0, val_size, bs):
                o = m(vx[k:k + bs].contiguous())
                vl += float(crit(o, vt[k:k + bs].contiguous())) * bs
                vi += iou_from_counts(iou_counts(o.contiguous(), vt[k:k + bs].contiguous())) * bs
        rows.append((tl, ti, vl / val_size, vi / val_size))
        print("epoch", ep, "hip", rows[-1], "ref", tuple(ref[ep][2:]))
    rows = np.array(rows)
    assert np.all(np.isfinite(rows))
    assert float(ref[-1, 5]) > 0.6, "fixture: the reference itself did not learn the task"
    # first epoch (nothing has diverged yet) tightly; afterwards bands; the final validation IoU within 0.02
    assert abs(rows[0, 0] - ref[0, 2]) < 0.03 and abs(rows[0, 1] - ref[0, 3]) < 0.05, (rows[0], ref[0])
    assert np.all(np.abs(rows[:, 0] - ref[:, 2]) < 0.08), (rows[:, 0], ref[:, 2])
    half = epochs // 2
    assert np.all(np.abs(rows[half:, 3] - ref[half:, 5]) <= 0.04), (rows[half:, 3], ref[half:, 5])
    assert abs(float(rows[-3:, 3].mean()) - float(ref[-3:, 5].mean())) <= 0.02
    assert np.all(np.abs(rows[half:, 2] - ref[half:, 4]) <= 0.06), (rows[half:, 2], ref[half:, 4])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_step_from_decoded_uint8_batches(dtype, synth):
    """TrainStep(input_u8=True): the reference's sample pipeline (dataset.py:66-74 Normalize, /255, mask / 255; trains.py:258-259
    rot90 / flips) runs on the device as the head of the step's graph, writing the plan's padded NHWC image directly.
    Its steps must equal - bit for bit: the staged image has the float path's arithmetic and rounding - steps of the float
    TrainStep fed with nunet_amd.dataset.preprocess_images / preprocess_masks of the same uint8 batch and codes."""
    from nunet_amd.trainer import TrainStep
    from nunet_amd import dataset as D
    n, hw = 4, 32
    torch.manual_seed(9)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, False).state_dict().items()}
    raw, m8 = synth.synth_blob_pairs_u8(3 * n, hw, hw, seed=77)
    raw, m8 = torch.from_numpy(raw).to(DEV), torch.from_numpy(m8).to(DEV)
    codes = [None, torch.tensor([1, 2 | 4, 3 | 8, 12], dtype=torch.int32, device=DEV), D.draw_augmentation(n, torch.Generator().manual_seed(2), DEV)]
    res = []
    for u8 in (False, True):
        m = nunet_amd.archs.NestedUNet(1, 3, False, dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, input_u8=u8)
        if u8:
            ts.capture(raw[:n], m8[:n])
        else:
            ts.capture(D.preprocess_images(raw[:n]), D.preprocess_masks(m8[:n]))
        losses = []
        for k in range(3):
            xb, tb = raw[k * n:(k + 1) * n], m8[k * n:(k + 1) * n]
            ts.reset_meters()
            if u8:
                ts.step_u8(xb, tb, codes[k])
            else:
                ts.step(D.preprocess_images(xb, codes[k]), D.preprocess_masks(tb, codes[k]))
            losses.append(ts.epoch_stats())
        torch.cuda.synchronize()
        res.append((losses, ts.eng.flat_params.clone(), ts.eng.bnbuf.clone()))
        if u8:
            with pytest.raises(L.NunetError):
                ts.step(D.preprocess_images(raw[:n]), D.preprocess_masks(m8[:n]))     # float tensors into a uint8 step: loud
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    # and the pipeline is the reference's: against the host expression on the first batch
    img, msk = synth.synth_blob_pairs(3 * n, hw, hw, seed=77)
    np.testing.assert_allclose(D.preprocess_images(raw[:n]).cpu().numpy(), img[:n], atol=2e-7, rtol=2e-5)
    assert np.array_equal(D.preprocess_masks(m8[:n]).cpu().numpy(), msk[:n])


def test_segmented_step_program_equals_the_single_graph(synth):
    """The step recorded as a program of single-stream graph segments over the plan's lanes (TrainStep(segmented=True): nunet_seg_*, cross-lane
    dependencies as event records / waits between graph launches on streams chosen by measurement) computes exactly what the
    one multi-branch hipGraph computes: bit-identical parameters, momentum and BatchNorm buffers after three steps. (A dropped
    dependency would show as a race here: the program keeps only the event records some other lane waits on.)"""
    import os
    from nunet_amd.trainer import TrainStep, _SegProgram
    n, hw = 16, 96
    torch.manual_seed(11)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, True).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=300 + k) for k in range(3)]
    outs = []
    for seg, sched in ((False, "lanes"), (True, "lanes"), ("flags", "lanes"), ("flags", "list"), (False, "list")):
        m = nunet_amd.archs.NestedUNet(1, 3, True, dtype="bf16")
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, segmented=seg, schedule=sched)
        ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
        assert isinstance(ts.g_fb, _SegProgram) == bool(seg)
        if seg:
            info = ts.g_fb.info()
            print("segmented program (%s, %s):" % (seg, sched), info)
            assert info["event_waits"] >= info["event_records"] > 0 and info["kernel_nodes"] > 100
            # flag-synchronised lanes: ONE single-stream graph per lane; event mode: a graph per segment between two cuts
            assert (1 <= info["graph_launches"] <= 4) if seg == "flags" else info["graph_launches"] > 4
        for rep in range(8 if seg == "flags" else 1):       # (the flag program replays the same three batches: a race would not repeat)
            if rep:
                ts.eng.flat_params.copy_(p0); ts.mom.copy_(m0); ts.eng.bnbuf.copy_(b0)
            else:
                p0, m0, b0 = ts.eng.flat_params.clone(), ts.mom.clone(), ts.eng.bnbuf.clone()
            for img, msk in batches:
                ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
            torch.cuda.synchronize()
            if rep:
                assert all(torch.equal(a, b) for a, b in zip(first, (ts.eng.flat_params, ts.mom, ts.eng.bnbuf, ts.loss_out))), rep
            else:
                first = [t.clone() for t in (ts.eng.flat_params, ts.mom, ts.eng.bnbuf, ts.loss_out)]
        torch.cuda.synchronize()
        outs.append(first)
        del ts, m
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_flag_program_survives_a_lane_reset_and_the_capture_time_choice_is_recorded(synth):
    """nunet_plan_reset_lanes: the next recording picks new side-lane streams (the old ones stay alive, a program recorded on them
    stays valid); a program recorded after the reset computes the same bits. And the default TrainStep (no executor pinned) times
    both executors at capture and says which it kept."""
    from nunet_amd.trainer import TrainStep, _SegProgram
    n, hw = 16, 96
    torch.manual_seed(13)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, False).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=340 + k) for k in range(2)]

    def run(ts):
        for img, msk in batches:
            ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        torch.cuda.synchronize()
        return [t.clone() for t in (ts.eng.flat_params, ts.mom, ts.eng.bnbuf, ts.loss_out)]

    outs = []
    for reset in (False, True):
        m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="bf16")
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, segmented="flags", schedule="list")
        x0, t0 = torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV)
        ts.capture(x0, t0)
        if reset:
            old = ts.g_fb
            L.check(L.lib().nunet_plan_reset_lanes(ts.pl.handle), "plan_reset_lanes")
            ts.capture(x0, t0)                     # a second recording: new lanes
            assert isinstance(ts.g_fb, _SegProgram) and ts.g_fb is not old
        outs.append(run(ts))
        del ts, m
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2)
    ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
    assert ts.executor_choice and set(ts.executor_choice) == {(False, "lanes"), ("flags", "list")}
    assert (ts.segmented, ts.schedule) == min(ts.executor_choice, key=ts.executor_choice.get)
    for a, b in zip(outs[0], run(ts)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_single_stream_schedule_with_grouped_convs_equals_the_lane_schedule(dtype, synth):
    """TrainStep(schedule='wave') / nunet_plan_set_schedule: the pass emitted on ONE stream in dependency order by a critical-path list scheduler, every ready 3x3
    convolution of the same kernel variant riding in the launch of the one it picked (nunet_conv3x3_group: up to 4 problems per
    launch, workgroups dealt round-robin). A grouped problem computes exactly what its own launch computes, so the step is
    bit-identical to the multi-lane schedule's - parameters, momentum, BatchNorm buffers, losses - with deep supervision on."""
    import os
    from nunet_amd.trainer import TrainStep
    n, hw = 16, 96
    torch.manual_seed(13)
    sd = {k: v.clone() for k, v in nunet_amd.archs.NestedUNet(1, 3, True).state_dict().items()}
    batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=400 + k) for k in range(3)]
    outs = []
    for sched in ("lanes", "wave"):
        m = nunet_amd.archs.NestedUNet(1, 3, True, dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        ts = TrainStep(m, (n, 3, hw, hw), lr=1e-2, schedule=sched)
        ts.capture(torch.from_numpy(batches[0][0]).to(DEV), torch.from_numpy(batches[0][1]).to(DEV))
        if sched == "wave":
            info = ts.g_fb.info()
            assert info["lanes"] == 1, info                        # one stream: ROCm's batch-submit path, no parallel branches
        for img, msk in batches:
            ts.step(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV))
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (ts.eng.flat_params, ts.mom, ts.eng.bnbuf, ts.loss_out)])
        del ts, m
    for a, b in zip(*outs):
        assert torch.equal(a, b)
