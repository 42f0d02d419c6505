#!/usr/bin/env python3
"""How does ROCm place the nodes of a captured multi-stream hipGraph on hardware queues, and against
what does it resolve a cross-queue dependency?  Synthetic fork/join patterns of spin kernels
(nunet_debug_spin: the workgroup count is the op's tag), captured with torch streams/events.

  rocprofv3 --kernel-trace --output-format csv -d out -o run -- python3 tools/graph_sched_probe.py run
  python3 tools/graph_sched_probe.py report out/run_kernel_trace.csv
"""
import sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

US = int(os.environ.get("PROBE_US", "20"))


def run():
    import torch, importlib
    L = importlib.import_module("pytorch_nested-unet_amd._lib")
    lib = L.lib()
    S = [torch.cuda.Stream() for _ in range(64)]
    nxt = [0]

    def fresh():
        nxt[0] += 1
        return S[nxt[0] - 1]

    def spin(s, tag, us=US):
        L.check(lib.nunet_debug_spin(us, tag, s.cuda_stream), "spin")

    def ev(s):
        e = torch.cuda.Event(); e.record(s); return e

    def pat_A(cur):      # side op forked after c2 but ISSUED after the whole chain
        e2 = None
        for k in range(10):
            spin(cur, 100 + k)
            if k == 2: e2 = ev(cur)
        s1 = fresh(); s1.wait_event(e2); spin(s1, 200); cur.wait_event(ev(s1))

    def pat_B(cur):      # same graph, side op issued right after c2
        for k in range(3): spin(cur, 100 + k)
        e2 = ev(cur)
        s1 = fresh(); s1.wait_event(e2); spin(s1, 200)
        for k in range(3, 10): spin(cur, 100 + k)
        cur.wait_event(ev(s1))

    def pat_C(cur):      # two side ops (after c2, after c5), both issued late
        e = {}
        for k in range(10):
            spin(cur, 100 + k)
            if k in (2, 5): e[k] = ev(cur)
        s1 = fresh(); s1.wait_event(e[2]); spin(s1, 200)
        s2 = fresh(); s2.wait_event(e[5]); spin(s2, 201)
        cur.wait_event(ev(s1)); cur.wait_event(ev(s2))

    def pat_D(cur):      # side chain x1>x2>x3 forked after c2 (issued early), joined back before c6
        for k in range(3): spin(cur, 100 + k)
        e2 = ev(cur)
        s1 = fresh(); s1.wait_event(e2)
        for k in range(3): spin(s1, 200 + k)
        e3 = ev(s1)
        for k in range(3, 6): spin(cur, 100 + k)
        cur.wait_event(e3)
        for k in range(6, 10): spin(cur, 100 + k)

    def pat_E(cur):      # as D, but the side chain issued after c5 (just before the join)
        for k in range(3): spin(cur, 100 + k)
        e2 = ev(cur)
        for k in range(3, 6): spin(cur, 100 + k)
        s1 = fresh(); s1.wait_event(e2)
        for k in range(3): spin(s1, 200 + k)
        e3 = ev(s1)
        cur.wait_event(e3)
        for k in range(6, 10): spin(cur, 100 + k)

    def pat_F(cur):      # 6 leaves forked after c0..c5 (issued as they become ready), chain continues: > 4 queues?
        leaves = []
        for k in range(10):
            spin(cur, 100 + k)
            if k < 6:
                e = ev(cur); s = fresh(); s.wait_event(e); spin(s, 200 + k, 60); leaves.append(ev(s))
        for e in leaves: cur.wait_event(e)

    def pat_G(cur):      # chain forked to its OWN stream first (chain not on the capture stream), leaves as F
        e0 = ev(cur); c = fresh(); c.wait_event(e0)
        leaves = []
        for k in range(10):
            spin(c, 100 + k)
            if k < 6:
                e = ev(c); s = fresh(); s.wait_event(e); spin(s, 200 + k, 60); leaves.append(ev(s))
        cur.wait_event(ev(c))
        for e in leaves: cur.wait_event(e)

    def pat_H(cur):      # two lanes ping-pong: a0 > b0 > a1 > b1 ... continuing each lane on fresh streams (Sched style)
        a = cur; b = None; ea = None; eb = None
        for k in range(5):
            spin(a, 100 + k); ea = ev(a)
            nb = fresh()
            if eb is not None: nb.wait_event(eb)
            nb.wait_event(ea); b = nb
            spin(b, 200 + k); eb = ev(b)
            if k < 4:
                na = fresh(); na.wait_event(ea); na.wait_event(eb); a = na
        cur.wait_event(ea); cur.wait_event(eb)

    pats = [pat_A, pat_B, pat_C, pat_D, pat_E, pat_F, pat_G, pat_H]
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for pi, pat in enumerate(pats):
        if only and pat.__name__[-1] not in only: continue
        nxt[0] = 0
        warm = torch.cuda.Stream(); warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm): pat(warm)
        torch.cuda.current_stream().wait_stream(warm); torch.cuda.synchronize()
        nxt[0] = 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pat(torch.cuda.current_stream())
        torch.cuda.synchronize()
        for rep in range(3):
            spin(torch.cuda.current_stream(), 4000 + pi, 5)      # marker
            g.replay()
            torch.cuda.synchronize()
    print("done")


def report(path):
    rows = [r for r in csv.DictReader(open(path)) if "debug_spin" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups, cur = [], None
    for r in rows:
        tag = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        if tag >= 4000:
            cur = [tag - 4000, []]; groups.append(cur)
        elif cur is not None:
            cur[1].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], tag))
    seen = {}
    for pi, ops in groups:
        seen[pi] = seen.get(pi, 0) + 1
        if seen[pi] != 3 or not ops: continue        # third replay of each pattern
        t0 = ops[0][0]
        print(f"--- pattern {'ABCDEFGH'[pi]}")
        for s, e, q, tag in ops:
            print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} q{q} tag {tag}")


if __name__ == "__main__":
    if sys.argv[1] == "run": run()
    else: report(sys.argv[2])
