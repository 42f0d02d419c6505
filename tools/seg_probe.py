"""Segmented step: program statistics and host issue time per replay vs GPU time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nunet_amd
from nunet_amd.trainer import TrainStep
synth = nunet_amd.synth
torch.manual_seed(0)
m = nunet_amd.archs.NestedUNet(1, 3, False, dtype="bf16").cuda().train()
ts = TrainStep(m, (16, 3, 96, 96))
img, msk = synth.synth_batch(16, 96, 96, 3, 1, seed=1)
x, t = torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda()
ts.capture(x, t)
print(type(ts.g_fb).__name__, ts.g_fb.info())
for _ in range(20): ts.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): ts.step(x, t)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("host issue %.3f ms per step, total %.3f ms per step" % (th / 50 * 1e3, tt / 50 * 1e3))
