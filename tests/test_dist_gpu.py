"""Data-parallel rehearsal on ONE MI355X: two ranks share the GPU, gradients are exchanged with
gloo (RCCL cannot put two ranks on one device). Checks the DP step (trainer.TrainStep, world=2):
replicas stay identical and the update equals SGD on the rank-averaged gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, use_graph, dp_mode, q):
    os.environ.update(NUNET_DP_MODE=str(dp_mode), RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import nunet_amd
    from nunet_amd.trainer import TrainStep
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    synth = nunet_amd.synth
    st = synth.closed_form_state(1, 3, False, True)
    m = nunet_amd.archs.NestedUNet(1, 3, False)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().train()
    ts = TrainStep(m, (2, 3, 32, 32), lr=1e-2, momentum=0.9, weight_decay=1e-4, use_graph=use_graph)
    assert ts.world == 2
    img, msk = synth.synth_batch(2, 32, 32, 3, 1, seed=500 + rank)
    x, t = torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda()
    if use_graph:
        ts.capture(x, t)
    ts.step(x, t)
    torch.cuda.synchronize()
    w = m.conv0_4.conv2.weight.detach().cpu().clone()
    g = m.conv0_4.conv2.weight.grad.detach().cpu().clone()     # summed over ranks by the all-reduce
    ts.step(x, t)
    torch.cuda.synchronize()
    q.put((rank, w.numpy(), g.numpy(), m.conv3_1.conv1.weight.detach().cpu().numpy(), ts.epoch_stats()))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,dp_mode", [(False, 1), (True, 1), (True, 0)])
def test_two_rank_data_parallel_step(use_graph, dp_mode, synth):
    import nunet_amd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, dp_mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # replicas identical after one and after two steps
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][1], res[1][1])
    # single-process evaluation of each shard's gradient -> SGD on their mean reproduces the DP update
    st = synth.closed_form_state(1, 3, False, True)
    grads = []
    for rank in range(2):
        m = nunet_amd.archs.NestedUNet(1, 3, False)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        m = m.cuda().train()
        img, msk = synth.synth_batch(2, 32, 32, 3, 1, seed=500 + rank)
        loss = nunet_amd.losses.BCEDiceLoss()(m(torch.from_numpy(img).cuda()), torch.from_numpy(msk).cuda())
        loss.backward()
        grads.append(m.conv0_4.conv2.weight.grad.detach().cpu().numpy())
    gsum = grads[0] + grads[1]
    assert np.abs(res[0][1] - gsum).max() <= 2e-2 * np.abs(gsum).max()
    w0 = st["conv0_4.conv2.weight"]
    expect = w0 - 1e-2 * (gsum / 2 + 1e-4 * w0)
    assert np.abs(res[0][0] - expect).max() <= 2e-2 * 1e-2 * np.abs(gsum).max() + 1e-7
    assert np.isfinite(res[0][3][0]) and abs(res[0][3][0] - res[1][3][0]) < 0.5
