#!/usr/bin/env python3
"""gpurun_out/prof_final/ (tools/make_profiles.sh) -> profiles/rNN_* : kernel-stats tables, PMC traffic json."""
import csv, json, os, sys, shutil, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(R, "profiles")


def stats_table(path, steps):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | launches/step | avg us | us/step | % |", "|---|---|---|---|---|"]
    for r in rows[:20]:
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        out.append("| `%s` | %.1f | %.1f | %.0f | %s |" % (r["Name"][:64], calls / steps, float(r["AverageNs"]) / 1e3, tot / steps / 1e3, r["Percentage"]))
    return "\n".join(out)


def pmc(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter: continue
        a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def main():
    steps = 27   # 20 timed + 5 warm-up + 2 capture warm-up passes
    shutil.copy(os.path.join(src, "multi", "run_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "single", "run_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats_single_lane.csv"))
    for n in ("multi", "single"):
        line = open(os.path.join(src, n + ".json")).read().strip().splitlines()[-1]
        open(os.path.join(P, tag + ("_bench_under_rocprof.json" if n == "multi" else "_bench_under_rocprof_single_lane.json")), "w").write(line + "\n")
    f = pmc(os.path.join(src, "pmc_fetch", "run_counter_collection.csv"), "FETCH_SIZE")
    w = pmc(os.path.join(src, "pmc_write", "run_counter_collection.csv"), "WRITE_SIZE")
    traffic = {}
    for k in f:
        if k not in w: continue
        fa, wa = f[k][1] / f[k][0], w[k][1] / w[k][0]
        traffic[k] = {"launches": f[k][0], "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa,
                      "hbm_bytes_per_launch_corrected": (2 * fa + wa) * 1024}
    json.dump(traffic, open(os.path.join(P, tag + "_pmc_traffic.json"), "w"), indent=1)
    print("## multi-lane\n" + stats_table(os.path.join(P, tag + "_bench_kernel_stats.csv"), steps))
    print("\n## single lane\n" + stats_table(os.path.join(P, tag + "_bench_kernel_stats_single_lane.csv"), steps))
    for n in ("multi", "single"):
        d = json.loads(open(os.path.join(src, n + ".json")).read().strip().splitlines()[-1])
        print(n, "bench under rocprof:", d["value"], d["ms_per_step"])


if __name__ == "__main__":
    main()
