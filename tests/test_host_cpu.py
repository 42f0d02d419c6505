"""Host-side logic of the drivers, on the CPU: learning-rate schedules against torch's own schedulers (what the
reference instantiates at trains.py:237-246), the IoU logit threshold, the decoded uint8 dataset."""
import importlib.util
import os

import numpy as np
import torch

import nunet_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train_module():
    spec = importlib.util.spec_from_file_location("nunet_train_driver", os.path.join(ROOT, "train.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plateau_schedule_matches_torch():
    tr = _train_module()
    rng = np.random.default_rng(3)
    for factor, patience, min_lr in ((0.1, 2, 1e-5), (0.5, 0, 1e-4), (0.1, 3, 5e-3)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1e-2)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=factor, patience=patience, min_lr=min_lr)
        mine = tr.PlateauLR(1e-2, factor, patience, min_lr)
        v = 1.0
        for ep in range(40):
            v = v * (0.9 if ep < 6 else 1.0) + float(rng.normal()) * 1e-3 * (ep > 6)     # improves, then stalls with noise
            ref.step(v)
            assert abs(mine.step(v) - opt.param_groups[0]["lr"]) < 1e-15, (factor, patience, ep)


def test_multistep_and_cosine_schedules_match_torch():
    tr = _train_module()
    from nunet_amd.trainer import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1, 2, 7], gamma=2 / 3)
    for ep in range(12):
        assert abs(tr.multistep_lr(1e-3, [1, 2, 7], 2 / 3, ep) - opt.param_groups[0]["lr"]) < 1e-15
        opt.step(); ref.step()
    opt = torch.optim.SGD([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=30, eta_min=1e-5)
    for ep in range(30):
        assert abs(cosine_lr(1e-3, 1e-5, ep, 30) - opt.param_groups[0]["lr"]) < 1e-12
        opt.step(); ref.step()


def test_iou_logit_threshold_is_where_the_reference_sigmoid_crosses_one_half():
    thr = nunet_amd.metrics.iou_logit_threshold()
    x = np.zeros(256, np.float32)
    x[0] = thr
    x[1] = np.nextafter(np.float32(thr), np.float32(0))
    s = torch.sigmoid(torch.from_numpy(x)).numpy()
    assert s[0] > 0.5 and not (s[1] > 0.5) and x[1] > 0        # a positive logit that the reference counts as background


def test_decoded_uint8_set_is_the_float_set_before_the_sample_pipeline():
    s = nunet_amd.synth
    raw, m8 = s.synth_blob_pairs_u8(6, 32, 48, seed=1000)
    img, msk = s.synth_blob_pairs(6, 32, 48, seed=1000)
    assert raw.dtype == np.uint8 and raw.shape == (6, 32, 48, 3) and m8.shape == (6, 32, 48, 1) and set(np.unique(m8)) <= {0, 255}
    mean, std = np.asarray(nunet_amd.dataset.MEAN), np.asarray(nunet_amd.dataset.STD)
    ref = (((raw.astype(np.float64) / 255.0 - mean) / std) / 255.0).transpose(0, 3, 1, 2).astype(np.float32)   # dataset.py:66-74
    assert np.array_equal(ref, img)
    assert np.array_equal((m8 / 255.0).transpose(0, 3, 1, 2).astype(np.float32), msk)
