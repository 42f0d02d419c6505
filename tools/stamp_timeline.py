#!/usr/bin/env python3
"""Real timeline of one UNPROFILED hipGraph replay of the fused train step (NUNET_STAMPS=1 device
timestamps after every scheduled op; a profiler's per-dispatch overhead would reshape the schedule).

  NUNET_STAMPS=1 python3 tools/stamp_timeline.py [--lanes]
"""
import sys, os, ctypes as C, importlib
os.environ.setdefault("NUNET_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
L = importlib.import_module("pytorch_nested-unet_amd._lib")
archs = importlib.import_module("pytorch_nested-unet_amd.archs")
trainer = importlib.import_module("pytorch_nested-unet_amd.trainer")
synth = importlib.import_module("pytorch_nested-unet_amd.synth")


def read(plan, ps):
    buf = (C.c_uint64 * 512)(); n = C.c_int32(); lab = C.create_string_buffer(32768)
    L.check(L.lib().nunet_plan_stamps_read(plan, ps, buf, 512, C.byref(n), lab, 32768), "stamps_read")
    return [(buf[k], l) for k, l in zip(range(n.value), lab.value.decode().splitlines())]


def main():
    torch.manual_seed(0)
    m = archs.NestedUNet(1, 3, False, dtype="bf16").cuda()
    x, t = synth.synth_batch(16, 96, 96, 3, 1, seed=1)
    x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    ts = trainer.TrainStep(m, (16, 3, 96, 96), lr=1e-3, use_graph="--eager" not in sys.argv)
    ts.capture(x, t)
    for _ in range(30): ts.step()
    torch.cuda.synchronize()
    ev = read(ts.pl.handle, 0) + read(ts.pl.handle, 1)
    t0 = min(e[0] for e in ev)
    ev = sorted(((tk - t0) / 100.0, lab) for tk, lab in ev)      # 100 MHz -> us
    last = {}
    print("# end_us  (since previous stamp on the lane)  lane op")
    for tm, lab in ev:
        lane = lab.split()[0]
        d = tm - last.get(lane, 0.0)
        last[lane] = tm
        print(f"{tm:9.1f} {d:8.1f}  {lab}")
    print(f"# span {ev[-1][0]:.1f} us, {len(ev)} stamps")


if __name__ == "__main__":
    main()
