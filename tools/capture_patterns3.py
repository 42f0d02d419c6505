import sys, torch, faulthandler
faulthandler.enable()
pat = sys.argv[1]
x = [torch.zeros(1 << 16, device="cuda") for _ in range(6)]
S = [torch.cuda.Stream() for _ in range(8)]
def ev(s):
    e = torch.cuda.Event(); e.record(s); return e
def op(s, k):
    with torch.cuda.stream(s): x[k].add_(1)
def body():
    cur = torch.cuda.current_stream()
    x[5].add_(1)
    ef = ev(cur)
    s1, s2 = S[0], S[1]
    s1.wait_event(ef); op(s1, 0); e1 = ev(s1)
    s2.wait_event(e1); op(s2, 1); e2 = ev(s2)
    if pat == "pp1":            # s1 waits on a descendant of its own tail
        s1.wait_event(e2); op(s1, 0)
        cur.wait_event(ev(s1)); cur.wait_event(ev(s2))
    if pat == "pp1fix":         # continue lane 1 on a FRESH stream that waits on both
        s3 = S[2]
        s3.wait_event(e1); s3.wait_event(e2); op(s3, 0)
        cur.wait_event(ev(s3))
    if pat == "pp2fix":         # longer ping-pong entirely with fresh streams
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
print(pat, "ok", x[0][0].item(), x[1][0].item())
