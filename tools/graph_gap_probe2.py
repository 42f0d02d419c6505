"""Is the multi-branch tax of ROCm's graph executor paid by EVERY node, or only around cross-stream edges?"""
import sys, os, time
import torch
N = 120
x = torch.zeros(1 << 16, device="cuda"); y = torch.zeros(1 << 16, device="cuda")
def timeit(name, body):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st): body()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("%-58s %.2f us per chain node (host %.2f)" % (name, tt * 1e6 / 20 / N, th * 1e6 / 20 / N))
def chain(): [x.add_(1) for _ in range(N)]
def one_side_node(at):
    def body():
        cur = torch.cuda.current_stream(); s = torch.cuda.Stream()
        for i in range(N):
            x.add_(1)
            if i == at:
                s.wait_stream(cur)
                with torch.cuda.stream(s): y.add_(1)
        cur.wait_stream(s)
    return body
def side_chain(m):
    def body():
        cur = torch.cuda.current_stream(); s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(m): y.add_(1)
        for _ in range(N): x.add_(1)
        cur.wait_stream(s)
    return body
timeit("chain alone", chain)
timeit("chain + ONE side node forked at node 0, joined at the end", one_side_node(0))
timeit("chain + ONE side node forked at node 60, joined at the end", one_side_node(60))
for m in (5, 20, 60, 120):
    timeit("chain + independent side chain of %d nodes" % m, side_chain(m))
