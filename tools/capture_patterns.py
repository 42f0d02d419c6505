"""Which stream/event pattern inside a capture crashes hipStreamEndCapture? (pure torch)"""
import sys, torch, faulthandler
faulthandler.enable()
pat = sys.argv[1]
x = [torch.zeros(1 << 16, device="cuda") for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    cur = torch.cuda.current_stream()
    e_fork = torch.cuda.Event(); e_fork.record(cur)
    s1.wait_event(e_fork)
    with torch.cuda.stream(s1):
        x[0].add_(1)
        if pat == "dangling":        # recorded, never waited on
            e = torch.cuda.Event(); e.record(s1)
            x[0].add_(1)
        if pat == "selfwait":        # a stream waits on its own event
            e = torch.cuda.Event(); e.record(s1); s1.wait_event(e)
            x[0].add_(1)
        if pat == "many":
            for _ in range(150):
                e = torch.cuda.Event(); e.record(s1); x[0].add_(1)
        if pat == "reuse":           # same event object recorded twice in the capture
            e = torch.cuda.Event(); e.record(s1); x[0].add_(1); e.record(s1); x[0].add_(1)
        if pat == "cross":           # second lane forked from the first lane's event, both joined
            e = torch.cuda.Event(); e.record(s1)
            s2.wait_event(e)
            with torch.cuda.stream(s2):
                x[1].add_(1)
            e2 = torch.cuda.Event(); e2.record(s2); cur.wait_event(e2)
        if pat == "latefork":        # lane forked from the ORIGINAL fork event after other work was recorded
            pass
    if pat == "latefork":
        x[2].add_(1)
        s2.wait_event(e_fork)
        with torch.cuda.stream(s2):
            x[1].add_(1)
        e2 = torch.cuda.Event(); e2.record(s2); cur.wait_event(e2)
    e_join = torch.cuda.Event(); e_join.record(s1); cur.wait_event(e_join)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): body()
g.replay(); torch.cuda.synchronize()
print(pat, "ok", x[0][0].item())
