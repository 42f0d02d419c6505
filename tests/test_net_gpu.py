"""Whole-path parity on the MI355X: NestedUNet / UNet forward, loss, IoU, gradients,
BN running statistics and short SGD trajectories against (a) golden vectors captured
from the imported reference and (b) the CPU oracle in fp64 on the same seeded inputs.

Tolerances. north_star asks for logits within 1e-4 (fp32) of the reference CPU
forward; that is asserted directly on the goldens. Gradients of this network are
ill-conditioned in fp32 (the reference's own fp32 gradients sit ~4e-3 relative from
an fp64 evaluation), so gradient parity is asserted against the fp64
oracle with the bound max(4 x reference-fp32 error, 2e-3) per tensor (atomic summation order
adds run-to-run noise of the same size as the reference's own fp32 error in the 16x16 case)."""
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
    # north_star: fp32 logits within 1e-4 of the reference. Where the reference's OWN fp32 logits are further than that
    # from an fp64 evaluation (BatchNorm over two 1x1 "images" at level 4 of the 16x16 case divides by a tiny std) the
    # bound is 4x that conditioning error instead. (Every per-channel sum is order-independent fixed point and every
    # other reduction has a fixed tree, so a run is bit-reproducible: no allowance for summation-order noise.)
    with torch.no_grad():
        l64 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float64)(x.double())
        l32 = O.OracleNet(st, ncls, cin, ds, dtype=torch.float32)(x)
    l64, l32 = (l64 if ds else [l64]), (l32 if ds else [l32])
    for k, o in enumerate(outs):
        ref = g["logits%d" % k]
        scale = max(1.0, float(np.abs(ref).max()))
        cond = float((l32[k].double() - l64[k]).abs().max()) / scale
        assert float(np.abs(o.detach().cpu().numpy() - ref).max()) < max(1e-4, 4 * cond) * scale, (name, k, cond)
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
    o64 = O.OracleNet(st, 1, 3, False, dtype=torch.float64)
    O.bce_dice_loss(o64(x.double()), t.double()).backward()
    errs = []
    for nm, p in m.named_parameters():
        if nm.endswith("conv1.bias") or nm.endswith("conv2.bias"):
            continue
        g64 = o64.params[nm].grad
        errs.append(float((p.grad.cpu().double() - g64).norm() / (g64.norm() + 1e-30)))
    print(dtype, "grad rel err max/median", max(errs), float(np.median(errs)))
    # 16-bit storage of pre-BN tensors on an ill-conditioned gradient (the fp32 reference
    # itself is ~5e-3 off here; 16-bit rounding is 2^13..2^16 x coarser): direction-level
    # agreement of every gradient tensor with the fp64 oracle, as with torch autocast
    assert max(errs) < (0.6 if dtype == "bf16" else 0.25), max(errs)
    assert float(np.median(errs)) < (0.35 if dtype == "bf16" else 0.15)


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
