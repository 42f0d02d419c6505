"""Pin the oracle (oracle/nunet_oracle.py) against golden vectors captured from
the imported reference (tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle import nunet_oracle as O


def _summ(t):
    a = t.detach().double().reshape(-1).numpy()
    stride = max(1, a.size // 64)
    s = np.zeros(64, np.float32)
    v = a[::stride][:64]
    s[:v.size] = v
    return a.sum(), np.sqrt((a * a).sum()), s


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_oracle_matches_reference_goldens(name, synth):
    n, h, w, cin, ncls, ds, train, fresh = GOLDEN_CASES[name]
    g = load_golden(name)
    net = O.OracleNet(synth.closed_form_state(ncls, cin, ds, fresh), ncls, cin, ds)
    img, msk = synth.synth_batch(n, h, w, cin, ncls, seed=1234)
    x, t = torch.from_numpy(img), torch.from_numpy(msk)
    if not train:
        net.eval()
        with torch.no_grad():
            o = net(x)
        np.testing.assert_allclose(o.numpy(), g["logits0"], atol=1e-5, rtol=1e-5)
        assert abs(float(O.bce_dice_loss(o, t)) - float(g["loss"])) < 1e-6
        assert abs(O.iou_score(o, t) - float(g["iou"])) < 1e-12
        return
    out = net(x)
    loss, last = O.criterion_ds(out, t)
    outs = out if ds else [out]
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g["logits%d" % k], atol=1e-5, rtol=1e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    assert abs(O.iou_score(last, t) - float(g["iou"])) < 1e-12
    loss.backward()
    names = [str(s) for s in g["grad_names"]]
    assert names == list(net.params.keys())
    for k, nm in enumerate(names):
        s, l2, smp = _summ(net.params[nm].grad)
        assert abs(l2 - g["grad_l2"][k]) <= 1e-4 * g["grad_l2"][k] + 1e-9, nm
        np.testing.assert_allclose(smp, g["grad_sample"][k], atol=1e-6 + 1e-4 * g["grad_amax"][k], err_msg=nm)
    bn_names = [str(s) for s in g["bn_names"]]
    for k, nm in enumerate(bn_names):
        assert abs(float(net.buffers[nm].double().sum()) - g["bn_sum"][k]) < 1e-4 * (1 + abs(g["bn_sum"][k])), nm


def test_oracle_trajectory(synth):
    g = load_golden("trajectory_n4_32x32")
    net = O.OracleNet(synth.closed_form_state(1, 3, False, True), 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    step = 0
    for ep in range(4):
        opt.lr = O.cosine_lr(1e-3, 1e-5, ep, 4)
        for _ in range(2):
            img, msk = synth.synth_batch(4, 32, 32, 3, 1, seed=1234 + step)
            loss, iou = O.train_step(net, opt, torch.from_numpy(img), torch.from_numpy(msk))
            assert abs(opt.lr - g["lr"][step]) < 1e-12
            assert abs(loss - g["loss"][step]) < 2e-4, (step, loss, g["loss"][step])
            assert abs(iou - g["iou"][step]) < 2e-3
            step += 1
    net.eval()
    img, msk = synth.synth_batch(4, 32, 32, 3, 1, seed=99)
    with torch.no_grad():
        o = net(torch.from_numpy(img))
    assert abs(float(O.bce_dice_loss(o, torch.from_numpy(msk))) - float(g["val_loss"])) < 1e-3


def test_small_ops_goldens():
    g = load_golden("small_ops")
    for tag in ("k1", "k4"):
        x = torch.from_numpy(g["x_" + tag]).requires_grad_(True)
        t = torch.from_numpy(g["t_" + tag])
        loss = O.bce_dice_loss(x, t)
        loss.backward()
        assert abs(float(loss) - float(g["loss_" + tag])) < 1e-6
        assert abs(O.bce_dice_loss_np(g["x_" + tag], g["t_" + tag]) - float(g["loss_" + tag])) < 1e-6
        np.testing.assert_allclose(x.grad.numpy(), g["dx_" + tag], atol=1e-8, rtol=1e-5)
        assert abs(O.iou_score(x, t) - float(g["iou_" + tag])) < 1e-12
        i, u = O.iou_counts(g["x_" + tag], g["t_" + tag])
        assert abs((i + 1e-5) / (u + 1e-5) - float(g["iou_" + tag])) < 1e-12
    m = O.AverageMeter()
    for v, k in ((0.5, 4), (0.25, 2), (1.0, 1)):
        m.update(v, k)
    assert m.avg == float(g["meter_avg"])


def test_numpy_restatements_match_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 6, 10)).astype(np.float32)
    xt = torch.from_numpy(x)
    np.testing.assert_array_equal(O.maxpool2x2_np(x), torch.nn.functional.max_pool2d(xt, 2, 2).numpy())
    up = torch.nn.functional.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=True).numpy()
    np.testing.assert_allclose(O.upsample2x_bilinear_ac_np(x), up, atol=5e-6)


def test_oracle_lovasz_hinge_matches_reference_goldens():
    """The Lovasz-hinge restatement (oracle, used as the checker for image sizes the fixtures do not hold) against
    the reference's own outputs (tests/golden/lovasz.npz, written by make_golden.py from losses.py)."""
    import torch
    from conftest import load_golden
    from oracle import nunet_oracle as O
    g = load_golden("lovasz")
    for tag in ("a", "b", "c"):
        x = torch.from_numpy(g["x_" + tag]).double().requires_grad_(True)
        t = torch.from_numpy(g["t_" + tag]).double()
        loss = O.lovasz_hinge(x.squeeze(1) if x.dim() == 4 else x, t.squeeze(1) if t.dim() == 4 else t)
        loss.backward()
        assert abs(float(loss) - float(g["loss_" + tag])) < 2e-6 * max(1.0, float(g["loss_" + tag])), tag
        np.testing.assert_allclose(x.grad.numpy(), g["dx_" + tag], atol=1e-7, rtol=1e-4, err_msg=tag)
