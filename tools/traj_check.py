"""Diagnostic: per-step loss of TrainStep (HIP) vs the CPU oracle on the bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nunet_amd
from nunet_amd.trainer import TrainStep
from oracle import nunet_oracle as O
synth = nunet_amd.synth
n, hw, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cpu_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.manual_seed(0)
ref_model = nunet_amd.archs.NestedUNet(1, 3, False)
sd = {k: v.clone() for k, v in ref_model.state_dict().items()}
batches = [synth.synth_batch(n, hw, hw, 3, 1, seed=1234 + k) for k in range(4)]
res = {}
for dt in ("fp32", "bf16"):
    for graph in (False, True):
        m = nunet_amd.archs.NestedUNet(1, 3, False, dtype=dt)
        m.load_state_dict(sd); m = m.cuda().train()
        ts = TrainStep(m, (n, 3, hw, hw), use_graph=graph)
        if graph:
            ts.capture(torch.from_numpy(batches[0][0]).cuda(), torch.from_numpy(batches[0][1]).cuda())
        losses = []
        for k in range(steps):
            img, msk = batches[k % 4]
            ts.reset_meters()
            ts.step(torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda())
            l, i = ts.epoch_stats()
            losses.append(l)
        res[(dt, graph)] = losses
        print(dt, "graph" if graph else "eager", " ".join("%.4f" % l for l in losses))
        print("   params finite:", bool(torch.isfinite(ts.eng.flat_params).all()), "bn finite:", bool(torch.isfinite(ts.eng.bnbuf).all()))
if cpu_steps:
    torch.set_num_threads(16)
    net = O.OracleNet({k: v.numpy() for k, v in sd.items()}, 1, 3, False)
    opt = O.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    out = []
    for k in range(cpu_steps):
        img, msk = batches[k % 4]
        l, i = O.train_step(net, opt, torch.from_numpy(img), torch.from_numpy(msk))
        out.append(l)
    print("oracle cpu", " ".join("%.4f" % l for l in out))
