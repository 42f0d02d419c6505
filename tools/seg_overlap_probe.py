"""Do launches on two different streams run side by side - eagerly, and as single-stream graphs? The kernel is the library's
spin kernel (one workgroup busy-waits N us): two of them overlap perfectly whenever the hardware queues let them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nunet_amd
from nunet_amd import _lib as L
US, N = 30, 40
def spin():
    L.check(L.lib().nunet_debug_spin(US, 1, L.stream()), "spin")
def cap(n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        spin()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): spin()
    return g
torch.zeros(1, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
def eager(k):
    def f():
        for _ in range(N):
            for s in streams[:k]:
                with torch.cuda.stream(s): spin()
    return f
for k in (1, 2, 4):
    print("eager, %d stream(s) x %d spins of %d us: %.0f us  (ideal %d)" % (k, N, US, timed(eager(k)), N * US))
graphs = [cap(N) for _ in range(4)]
def graphed(k):
    def f():
        for g, s in zip(graphs[:k], streams[:k]):
            with torch.cuda.stream(s): g.replay()
    return f
for k in (1, 2, 4):
    print("one single-stream graph per stream, %d stream(s): %.0f us  (ideal %d)" % (k, timed(graphed(k)), N * US))

# ---- which streams share a hardware queue? pairwise overlap among 8 torch streams (pool order) and 3 priority classes
def pair(sa, sb, n=10):
    def f():
        for _ in range(n):
            with torch.cuda.stream(sa): spin()
            with torch.cuda.stream(sb): spin()
    return timed(f, 3) / (n * US)
cand = [("hi%d" % i, torch.cuda.Stream(priority=-1)) for i in range(3)] + [("n%d" % i, torch.cuda.Stream(priority=0)) for i in range(5)]
print("pairwise time / ideal (1.0 = overlap, 2.0 = same hardware queue):")
print("      " + " ".join("%5s" % n for n, _ in cand))
for i, (ni, si) in enumerate(cand):
    print("%5s " % ni + " ".join("%5.2f" % (pair(si, sj) if j > i else 0.0) for j, (nj, sj) in enumerate(cand)))
def multi(ss, n=10):
    def f():
        for _ in range(n):
            for s in ss:
                with torch.cuda.stream(s): spin()
    return timed(f, 3) / (n * US)
print("hi0 + n0 + n1 + n2 together: %.2f x ideal" % multi([cand[0][1], cand[3][1], cand[4][1], cand[5][1]]))
print("hi0 + hi1 + n0 + n1 together: %.2f x ideal" % multi([cand[0][1], cand[1][1], cand[3][1], cand[4][1]]))
