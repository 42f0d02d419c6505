"""How much latency does ROCm's graph executor add per DEPENDENT kernel node? A chain of N small kernels (each ~3 us) is captured
(a) from one stream, (b) with a second stream forked once at the start (one long side kernel beside the chain), (c) with a side
branch forked at every k-th chain node and joined m nodes later (the shape of the plan's lanes). us per chain node, replayed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

N = 120
x = torch.zeros(1 << 16, device="cuda")
side_t = [torch.zeros(1 << 16, device="cuda") for _ in range(64)]
big = torch.zeros(1 << 24, device="cuda")


def chain_only():
    for _ in range(N):
        x.add_(1)


def one_side():
    cur = torch.cuda.current_stream()
    s = torch.cuda.Stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        for _ in range(8):
            big.add_(1)
    for _ in range(N):
        x.add_(1)
    cur.wait_stream(s)


def forks(k, m):
    def body():
        cur = torch.cuda.current_stream()
        pend = []
        for i in range(N):
            x.add_(1)
            if i % k == 0:
                s = torch.cuda.Stream()
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    side_t[(i // k) % 64].add_(1)
                    side_t[(i // k) % 64].add_(1)
                pend.append((i + m, s))
            for j, s in list(pend):
                if j == i:
                    cur.wait_stream(s)
                    pend.remove((j, s))
        for _, s in pend:
            cur.wait_stream(s)
    return body


def timeit(name, body):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-44s %.2f us per chain node" % (name, e0.elapsed_time(e1) * 1e3 / 20 / N))


def one_side_light():
    cur = torch.cuda.current_stream()
    s = torch.cuda.Stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        for _ in range(N):
            side_t[0].add_(1)
    for _ in range(N):
        x.add_(1)
    cur.wait_stream(s)


def fork_only(k):
    def body():
        cur = torch.cuda.current_stream()
        ss = []
        for i in range(N):
            x.add_(1)
            if i % k == 0:
                s = torch.cuda.Stream()
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    side_t[(i // k) % 64].add_(1)
                ss.append(s)
        for s in ss:
            cur.wait_stream(s)
    return body


def join_only(k):
    def body():
        cur = torch.cuda.current_stream()
        ss = []
        for j in range(N // k):           # all side kernels forked at the start
            s = torch.cuda.Stream()
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                side_t[j % 64].add_(1)
            ss.append(s)
        for i in range(N):
            if i % k == 0:
                cur.wait_stream(ss[i // k])   # the chain consumes one side result every k nodes
            x.add_(1)
    return body


if os.environ.get("PROBE_FORKS", "1") == "1":
    timeit("chain alone (one stream)", chain_only)
    timeit("chain + one side stream of light kernels", one_side_light)
    timeit("fork every 4 nodes (joined at the end only)", fork_only(4))
    timeit("join every 4 nodes (forked at the start)", join_only(4))
    timeit("side branch every 4 nodes, joined 2 later", forks(4, 2))
    timeit("side branch every 2 nodes, joined 3 later", forks(2, 3))


# ---- segmented execution: every (lane, segment) is its own SINGLE-stream graph (ROCm's fast path: pre-built AQL packets);
# cross-lane dependencies are events between graph launches on two real streams
def segmented(nseg, seglen, side_len):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    cap = torch.cuda.Stream()

    def cap_graph(fn):
        with torch.cuda.stream(cap):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                fn()
        return g
    chain_g = [cap_graph(lambda: [x.add_(1) for _ in range(seglen)]) for _ in range(nseg)]
    side_g = [cap_graph(lambda j=j: [side_t[j % 64].add_(1) for _ in range(side_len)]) for j in range(nseg)]
    ev_c = [torch.cuda.Event() for _ in range(nseg)]
    ev_s = [torch.cuda.Event() for _ in range(nseg)]

    def step():
        for i in range(nseg):
            with torch.cuda.stream(sa):
                if i >= 2:
                    sa.wait_event(ev_s[i - 2])        # the chain consumes the side result of two segments ago
                chain_g[i].replay()
                ev_c[i].record(sa)
            with torch.cuda.stream(sb):
                sb.wait_event(ev_c[i])                # the side segment needs chain segment i
                side_g[i].replay()
                ev_s[i].record(sb)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("segmented: %d segments x %d chain nodes (+%d side nodes each): %.2f us per chain node (host issue %.2f us per chain node)"
          % (nseg, seglen, side_len, tt * 1e6 / reps / (nseg * seglen), th * 1e6 / reps / (nseg * seglen)))


segmented(12, 10, 10)
segmented(24, 5, 5)
segmented(40, 3, 3)
segmented(60, 2, 2)
