"""GPU-side cost per DEPENDENT node of ROCm's graph executor with kernels long enough (10 us spin, one workgroup) that the host is
never the limit: us per chain node beyond the spin itself."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nunet_amd
from nunet_amd import _lib as L
N, US = 100, 10
def spin(): L.check(L.lib().nunet_debug_spin(US, 1, L.stream()), "spin")
def timeit(name, body):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st): body()
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("%-64s %.2f us per chain node over the %d us spin" % (name, tt * 1e6 / 5 / N - US, US))
def chain(): [spin() for _ in range(N)]
def side_chain(m):
    def body():
        cur = torch.cuda.current_stream(); s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(m): spin()
        for _ in range(N): spin()
        cur.wait_stream(s)
    return body
def forks(k, m):
    def body():
        cur = torch.cuda.current_stream(); pend = []
        for i in range(N):
            spin()
            if i % k == 0:
                s = torch.cuda.Stream(); s.wait_stream(cur)
                with torch.cuda.stream(s): spin()
                pend.append((i + m, s))
            for j, s in list(pend):
                if j == i: cur.wait_stream(s); pend.remove((j, s))
        for _, s in pend: cur.wait_stream(s)
    return body
def fork_only(k):
    def body():
        cur = torch.cuda.current_stream(); ss = []
        for i in range(N):
            spin()
            if i % k == 0:
                s = torch.cuda.Stream(); s.wait_stream(cur)
                with torch.cuda.stream(s): spin()
                ss.append(s)
        for s in ss: cur.wait_stream(s)
    return body
timeit("chain alone", chain)
timeit("chain + independent side chain of 100 spins", side_chain(100))
timeit("chain forking a 1-spin side branch every 4 nodes, joined at the end", fork_only(4))
timeit("chain forking every 4 nodes, each branch joined 2 nodes later", forks(4, 2))
timeit("chain forking every 2 nodes, each branch joined 3 nodes later", forks(2, 3))


def fork_chain_first(k):
    """the same forks, but the chain's NEXT node is captured before the side branch that shares its parent: the chain child is
    edge 0 of the parent's edge list, the side child edge 1"""
    def body():
        cur = torch.cuda.current_stream(); ss = []
        ev = None
        for i in range(N):
            spin()
            if ev is not None:                   # side branch of the PREVIOUS node, captured after this chain node
                s = torch.cuda.Stream(); s.wait_event(ev)
                with torch.cuda.stream(s): spin()
                ss.append(s); ev = None
            if i % k == 0:
                ev = torch.cuda.Event(); ev.record(cur)
        for s in ss: cur.wait_stream(s)
    return body
def forks_chain_first(k, m):
    def body():
        cur = torch.cuda.current_stream(); pend = []; ev = None
        for i in range(N):
            for j, s in list(pend):
                if j == i: cur.wait_stream(s); pend.remove((j, s))
            spin()
            if ev is not None:
                s = torch.cuda.Stream(); s.wait_event(ev)
                with torch.cuda.stream(s): spin()
                pend.append((i + m, s)); ev = None
            if i % k == 0:
                ev = torch.cuda.Event(); ev.record(cur)
        for _, s in pend: cur.wait_stream(s)
    return body
timeit("forks every 4, chain child captured FIRST, joined at the end", fork_chain_first(4))
timeit("forks every 4, chain child first, each branch joined 2 nodes later", forks_chain_first(4, 2))
timeit("forks every 2, chain child first, each branch joined 3 nodes later", forks_chain_first(2, 3))
