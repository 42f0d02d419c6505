"""Diagnostic: compare hipGraph replay with eager execution piece by piece."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nunet_amd
from nunet_amd import _lib as L
from nunet_amd.trainer import TrainStep
synth = nunet_amd.synth
n, hw, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.manual_seed(0)
m = nunet_amd.archs.NestedUNet(1, 3, False, dtype=dt).cuda().train()
ts = TrainStep(m, (n, 3, hw, hw), use_graph=False)
img, msk = synth.synth_batch(n, hw, hw, 3, 1, seed=1234)
ts.x.copy_(torch.from_numpy(img)); ts.t.copy_(torch.from_numpy(msk))
lib, eng, pl = L.lib(), ts.eng, ts.pl

def fwd():
    L.check(lib.nunet_plan_forward(pl.handle, L.ptr(eng.flat_params), L.ptr(eng.bnbuf), L.ptr(eng.nbt),
                                   L.ptr(ts.x), L.ptr(pl.arena), L.ptr(ts.logits), 1, L.stream()), "fwd")
def bwd():
    L.check(lib.nunet_plan_backward(pl.handle, L.ptr(eng.flat_params), L.ptr(ts.dlogits), L.ptr(pl.arena),
                                    L.ptr(eng.flat_grads), 0, L.stream()), "bwd")
ts.dlogits.copy_(torch.randn_like(ts.dlogits) * 1e-3)
fwd(); bwd(); torch.cuda.synchronize()
ref_logits = ts.logits.clone(); ref_grads = eng.flat_grads.clone()
fwd(); bwd(); torch.cuda.synchronize()
print("eager repeatability: logits", float((ts.logits - ref_logits).abs().max()), "grads rel", float((eng.flat_grads - ref_grads).norm() / ref_grads.norm()))

def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g

g1 = capture(fwd)
for r in range(3):
    ts.logits.zero_(); g1.replay(); torch.cuda.synchronize()
    print("graph fwd replay", r, "logits err", float((ts.logits - ref_logits).abs().max()), "finite", bool(torch.isfinite(ts.logits).all()))
g2 = capture(bwd)
for r in range(3):
    eng.flat_grads.zero_(); g1.replay(); g2.replay(); torch.cuda.synchronize()
    print("graph bwd replay", r, "grads rel err", float((eng.flat_grads - ref_grads).norm() / ref_grads.norm()), "finite", bool(torch.isfinite(eng.flat_grads).all()))
g3 = capture(lambda: (fwd(), bwd()))
for r in range(3):
    eng.flat_grads.zero_(); g3.replay(); torch.cuda.synchronize()
    print("graph fwd+bwd replay", r, "grads rel err", float((eng.flat_grads - ref_grads).norm() / ref_grads.norm()), "logits err", float((ts.logits - ref_logits).abs().max()))
