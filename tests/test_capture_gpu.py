"""Stream-capture regression (ROCm 7.2): a capturing stream that waits on an event DESCENDING from its own tail node
crashes hipStreamEndCapture (a host segfault, round-1 record gpurun_out/ms_test.log). The plan's lane scheduler never
produces that pattern: while capturing, a lane with cross-lane waits continues on a never-used stream
(csrc/plan.hip Sched::begin_v). These tests run in a CHILD process, so that a scheduler change which re-introduces the
pattern fails a test instead of taking the pytest process down."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SAFE_PATTERN = r"""
import torch
x = [torch.zeros(1 << 14, device="cuda") for _ in range(3)]
S = [torch.cuda.Stream() for _ in range(6)]
def ev(s):
    e = torch.cuda.Event(); e.record(s); return e
def op(s, k):
    with torch.cuda.stream(s): x[k].add_(1)
def body():   # two lanes ping-ponging; every continuation that has a cross-lane wait runs on a FRESH stream
    cur = torch.cuda.current_stream()
    x[2].add_(1)
    ef = ev(cur)
    s1, s2 = S[0], S[1]
    s1.wait_event(ef); op(s1, 0); e1 = ev(s1)
    s2.wait_event(e1); op(s2, 1); e2 = ev(s2)
    s3 = S[2]; s3.wait_event(e1); s3.wait_event(e2); op(s3, 0); e3 = ev(s3)
    s4 = S[3]; s4.wait_event(e2); s4.wait_event(e3); op(s4, 1); e4 = ev(s4)
    s5 = S[4]; s5.wait_event(e3); s5.wait_event(e4); op(s5, 0); op(s5, 0); e5 = ev(s5)
    cur.wait_event(e5); cur.wait_event(e4)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): body()
g.replay(); torch.cuda.synchronize()
assert x[0][0].item() == 8 and x[1][0].item() == 4, (x[0][0].item(), x[1][0].item())
print("CAPTURE-OK")
"""

PLAN_CAPTURE = r"""
import sys
sys.path.insert(0, %r)
import torch
import nunet_amd
from nunet_amd.trainer import TrainStep
synth = nunet_amd.synth
for cls, ds in ((nunet_amd.archs.NestedUNet, True), (nunet_amd.archs.NestedUNet, False), (nunet_amd.archs.UNet, False)):
    torch.manual_seed(0)
    m = cls(1, 3, ds, dtype="bf16").cuda().train()
    img, msk = synth.synth_batch(4, 32, 32, 3, 1, seed=5)
    x, t = torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda()
    ts = TrainStep(m, (4, 3, 32, 32))
    ts.capture(x, t)                      # multi-lane plan captured into one hipGraph
    for _ in range(3):
        ts.step(x, t)
    torch.cuda.synchronize()
    assert torch.isfinite(ts.eng.flat_params).all()
print("CAPTURE-OK")
"""


def _run(code):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE-OK" in r.stdout, "child exited %s\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_fresh_stream_continuation_pattern_captures():
    """The workaround pattern itself (lane continuation on fresh streams) captures, instantiates and replays."""
    _run(SAFE_PATTERN)


def test_multi_lane_plan_capture_does_not_crash_the_process():
    """Forward + backward + SGD of NestedUNet (with and without deep supervision) and UNet, multi-lane, captured and replayed."""
    _run(PLAN_CAPTURE % ROOT)
