"""Single-rank RCCL rehearsal of the data-parallel step on the flag-synchronised lanes (NUNET_DP_FLAGS=1 NUNET_DP_MODE=1): ms per step
and, with NUNET_STAMPS=1, the device timeline of one replay.   NUNET_STAMPS=1 python tools/dp_flags_probe.py [flags|graph]"""
import sys, os, time, ctypes as C
mode = sys.argv[1] if len(sys.argv) > 1 else "flags"
os.environ.update(NUNET_FORCE_DP="1", NUNET_DP_MODE="1", NUNET_DP_FLAGS="1" if mode == "flags" else "0", RANK="0", WORLD_SIZE="1",
                  MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import nunet_amd
from nunet_amd import _lib as L
from nunet_amd.trainer import TrainStep
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(0)
m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="bf16").cuda().train()
x, t = nunet_amd.synth.synth_batch(16, 96, 96, 3, 1, seed=1)
x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
ts = TrainStep(m, (16, 3, 96, 96), lr=1e-3)
if mode == "flags":
    ts.dp_auto = False; ts.dp_exec = ("flags", "list")
ts.capture(x, t)
for _ in range(20): ts.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): ts.step(x, t)
torch.cuda.synchronize()
print("mode %s: dp_mode %s exec %s: %.3f ms per step" % (mode, ts.dp_mode, ts.dp_exec, (time.perf_counter() - t0) * 10))
if os.environ.get("NUNET_STAMPS") == "1":
    def read(ps):
        buf = (C.c_uint64 * 512)(); n = C.c_int32(); lab = C.create_string_buffer(32768)
        L.check(L.lib().nunet_plan_stamps_read(ts.pl.handle, ps, buf, 512, C.byref(n), lab, 32768), "stamps_read")
        return [(buf[k], l) for k, l in zip(range(n.value), lab.value.decode().splitlines())]
    ev = read(0) + read(1)
    t0 = min(e[0] for e in ev)
    last = {}
    for tm, lab in sorted(((tk - t0) / 100.0, lab) for tk, lab in ev):
        lane = lab.split()[0]; d = tm - last.get(lane, 0.0); last[lane] = tm
        print("%9.1f %8.1f  %s" % (tm, d, lab))
dist.destroy_process_group()
