import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, faulthandler
faulthandler.enable()
import nunet_amd
from nunet_amd import _lib as L
from nunet_amd.trainer import TrainStep
what = sys.argv[1]
m = nunet_amd.archs.NestedUNet(1, 3, False).cuda().train()
ts = TrainStep(m, (2, 3, 32, 32), use_graph=False)
lib, eng, pl = L.lib(), ts.eng, ts.pl
if os.environ.get("TORCH_LANES"):
    import ctypes as C
    lanes = [torch.cuda.Stream() for _ in range(int(os.environ["TORCH_LANES"]))]
    arr = (C.c_void_p * len(lanes))(*[l.cuda_stream for l in lanes])
    L.check(lib.nunet_plan_set_lanes(pl.handle, arr, len(lanes)))
ts.x.normal_(); ts.t.bernoulli_(0.3); ts.dlogits.normal_()
def fwd():
    L.check(lib.nunet_plan_forward(pl.handle, L.ptr(eng.flat_params), L.ptr(eng.bnbuf), L.ptr(eng.nbt), L.ptr(ts.x), L.ptr(pl.arena), L.ptr(ts.logits), 1, L.stream()))
def bwd():
    L.check(lib.nunet_plan_backward(pl.handle, L.ptr(eng.flat_params), L.ptr(ts.dlogits), L.ptr(pl.arena), L.ptr(eng.flat_grads), 0, L.stream()))
fn = {"fwd": fwd, "bwd": bwd, "both": lambda: (fwd(), bwd())}[what]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd(); bwd(); fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
print("capture ok", flush=True)
g.replay(); torch.cuda.synchronize()
print("replay ok", flush=True)
