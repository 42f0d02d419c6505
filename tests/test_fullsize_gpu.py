"""BASELINE.json's full-size configurations on the MI355X.

The CPU oracle finishes a forward (+backward) of ONE or two images at these sizes in seconds, so the
parity statements are: fp32 logits of a small batch against the oracle at the north_star tolerance
(1e-4), reduced-precision logits and training loss against the oracle within the storage type's
rounding, and - at the full batch size - properties that do not need an oracle pass over the whole
batch: in eval mode (running statistics) an image's logits do not depend on the rest of the batch, two
runs are bit-identical, and a training step leaves finite parameters with a smaller loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import nunet_amd  # noqa: E402
from oracle import nunet_oracle as O  # noqa: E402
from test_net_gpu import build, run_step, DEV  # noqa: E402


def oracle_logits(st, x, ncls, ds, train):
    o = O.OracleNet(st, ncls, 3, ds, dtype=torch.float32)
    o.training = train
    with torch.no_grad():
        out = o(x.float())
    return out


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_cfg4_256x256_fp32_logits_match_oracle(synth):
    """cfg4 geometry (256x256): fp32 logits of two images against the CPU oracle, train-mode BN, 1e-4."""
    cfg = (2, 256, 256, 3, 1, False, True, True)
    m, st, x, t = build(cfg, synth)
    m.train()
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    ref = oracle_logits(st, x, 1, False, True)
    assert got.shape == (2, 1, 256, 256)
    assert rel(got, ref) < 1e-4


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
def test_cfg4_256x256_bs32_batch_independence(dtype, tol, synth):
    """cfg4 at the full per-GPU batch (32 x 256 x 256): in eval mode (running statistics) the logits of an image do
    not depend on the rest of the batch and repeat bit-exactly. Tiling and K-split grouping differ between the two
    batch shapes, so fp32 agrees to summation-order noise and bf16 to its rounding amplified through 30 layers."""
    cfg = (32, 256, 256, 3, 1, False, False, False)     # non-trivial running statistics
    m, st, x, t = build(cfg, synth, dtype=dtype)
    m.eval()
    xd = x.to(DEV)
    with torch.no_grad():
        full = m(xd).clone()
        again = m(xd).clone()
        part = m(xd[4:8].contiguous()).clone()
    assert torch.equal(full, again)
    assert rel(full[4:8].cpu(), part.cpu()) < tol


def test_cfg4_256x256_bs32_bf16_training_steps(synth):
    """cfg4, bf16 storage, full per-GPU batch (32 x 256 x 256): three fused hipGraph training steps against the fp32 CPU
    oracle running the same loop (reference trains.py:113-135; SGD lr 1e-2, momentum 0.9, wd 1e-4): per-step loss."""
    cfg = (32, 256, 256, 3, 1, False, True, True)
    m, st, x, t = build(cfg, synth, dtype="bf16")
    from nunet_amd.trainer import TrainStep
    m.train()
    xd, td = x.to(DEV), t.to(DEV)
    ts = TrainStep(m, (32, 3, 256, 256), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    ts.capture(xd, td)
    hip = []
    for _ in range(3):
        ts.reset_meters(); ts.step(xd, td); hip.append(ts.epoch_stats()[0])
    import os
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    net = O.OracleNet(st, 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    ref = [O.train_step(net, opt, x, t)[0] for _ in range(3)]
    print("cfg4 hip", hip, "oracle", ref)
    assert np.all(np.isfinite(hip)) and np.max(np.abs(np.array(hip) - np.array(ref))) < 5e-3, (hip, ref)
    assert hip[2] < hip[0] and ref[2] < ref[0]
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_cfg5_512x512_4class_fp16_matches_oracle(synth):
    """cfg5 geometry: 4 classes, 512x512, ONE image per GPU (BatchNorm over a single image), fp16 storage:
    train-mode logits and BCE-Dice loss against the fp32 CPU oracle within fp16 rounding."""
    cfg = (1, 512, 512, 3, 4, False, True, True)
    m, st, x, t = build(cfg, synth, dtype="fp16")
    m.train()
    out, loss, iou = run_step(m, x, t, False)
    ref = oracle_logits(st, x, 4, False, True)
    assert out.shape == (1, 4, 512, 512)
    assert rel(out.detach().cpu(), ref) < 1e-2
    ref_loss = float(O.bce_dice_loss(ref, t.float()))
    assert abs(float(loss.detach()) - ref_loss) < 5e-3
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_cfg3_deep_supervision_bs16_loss_is_mean_of_heads(synth):
    """cfg3: deep supervision at the bench batch size: four outputs, loss = mean of the four BCE-Dice terms
    (reference trains.py:118-124), fp32 logits of every head against the oracle at 1e-4."""
    cfg = (16, 96, 96, 3, 1, True, True, True)
    m, st, x, t = build(cfg, synth)
    m.train()
    out, loss, iou = run_step(m, x, t, True)
    assert isinstance(out, (list, tuple)) and len(out) == 4
    o = O.OracleNet(st, 1, 3, True, dtype=torch.float32)
    with torch.no_grad():
        ref = o(x.float())
    for a, b in zip(out, ref):
        assert rel(a.detach().cpu(), b) < 1e-4
    ref_loss = sum(float(O.bce_dice_loss(b, t.float())) for b in ref) / 4
    assert abs(float(loss.detach()) - ref_loss) < 2e-5


@pytest.mark.parametrize("shape", [(3, 48, 80), (4, 16, 32), (5, 32, 64), (2, 112, 16), (7, 96, 96), (8, 96, 96), (16, 192, 192)])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_shape_sweep_logits_and_loss_match_oracle(shape, dtype, tol, synth):
    """Non-square and odd-batch geometries: every pyramid level picks its own tiling (regular, multi-image, stacked
    rows, K-split), so a sweep over shapes exercises the combinations the fixed goldens do not. Train-mode logits and
    BCE-Dice loss against the fp32 CPU oracle. (8, 96, 96) is BASELINE.json configs[0]'s geometry (batch 8, 96x96)."""
    n, h, w = shape
    if dtype != "fp32" and n * (h // 16) * (w // 16) < 64:
        pytest.skip("BatchNorm over < 64 values per channel at level 4: 16-bit rounding of near-equal values is amplified by 1/std")
    cfg = (n, h, w, 3, 1, False, True, True)
    m, st, x, t = build(cfg, synth, dtype=dtype)
    m.train()
    out, loss, iou = run_step(m, x, t, False)
    ref = oracle_logits(st, x, 1, False, True)
    assert rel(out.detach().cpu(), ref) < tol, (shape, dtype)
    ref_loss = float(O.bce_dice_loss(ref, t.float()))
    assert abs(float(loss.detach()) - ref_loss) < (2e-5 if dtype == "fp32" else 2e-2)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
