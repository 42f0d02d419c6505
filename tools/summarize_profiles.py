#!/usr/bin/env python3
"""gpurun_out/prof_final/ (tools/make_profiles.sh) -> profiles/rNN_*: kernel-stats csv + tables, PMC traffic per bench
kernel class (what bench.py attaches as roofline.traffic), MFMA / LDS / instruction-mix counters per kernel."""
import collections, csv, hashlib, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 96
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 16
src = os.path.join(R, "gpurun_out", sys.argv[4] if len(sys.argv) > 4 else "prof_final")
P = os.path.join(R, "profiles")
STEPS = 27          # 20 timed + 5 warm-up + 2 capture warm-up passes
PMC_STEPS = 8 + 0   # eager: 6 timed + 2 warm-up (no capture warm-up with --no-graph)


def conv_class(name):
    """bench.py's class of a conv3x3_kernel instantiation: template args <T, WM, WN, SM, SN, SK, BNR, LT>."""
    m = re.search(r"conv3x3_kernelI\w+?Li(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
    if m:
        wm, wn, sm, sn = (int(v) for v in m.groups())
    else:   # demangled form: conv3x3_kernel<T, 4, 1, 2, 1, ...> is printed with placeholders by rocprof; fall back on the tail
        return None
    bm, bn = 32 * sm * wm, 32 * sn * wn
    return "conv3x3_fwd_dgrad<BM%d,BN%d>" % (bm, bn)


def kclass(name):
    if "conv3x3_kernel" in name: return conv_class(name)
    if "wgrad" in name: return "conv3x3_wgrad"
    for k, v in (("bn_relu_fwd", "bn_relu_fwd(+pool)"), ("bn_relu_bwd", "bn_relu_bwd"), ("upsample_fwd", "upsample2x_fwd"),
                 ("upsample_bwd", "upsample2x_bwd"), ("maxpool", "maxpool2x2"), ("head_", "head_1x1"), ("pack_kernel", "pack_weights"),
                 ("reduce_kernel", "wgrad_slab_reduce/unpack"), ("unpack_sgd", "sgd_step"), ("splitk_finalize", "splitk_finalize"),
                 ("loss_step", "bce_dice+iou"), ("nchw_to_nhwc", "layout"), ("zero_kernel", "zero")):
        if k in name: return v
    return None


def stats_table(path, steps):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | launches/step | avg us | us/step | % |", "|---|---|---|---|---|"]
    tot = 0.0
    for r in rows:
        tot += float(r["TotalDurationNs"]) / steps / 1e3
    for r in rows[:24]:
        calls, t = int(r["Calls"]), float(r["TotalDurationNs"])
        out.append("| `%s` | %.1f | %.1f | %.0f | %s |" % (r["Name"][:72], calls / steps, float(r["AverageNs"]) / 1e3, t / steps / 1e3, r["Percentage"]))
    return "\n".join(out), tot


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def sha16(paths):
    h = hashlib.sha256()
    for p in paths: h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def main():
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(src, "multi_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "single_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats_single_lane.csv"))
    lines = {}
    for n in ("multi", "single"):
        line = [ln for ln in open(os.path.join(src, n + ".json")).read().strip().splitlines() if ln.startswith("{")][-1]
        lines[n] = json.loads(line)
        open(os.path.join(P, tag + ("_bench_under_rocprof.json" if n == "multi" else "_bench_under_rocprof_single_lane.json")), "w").write(line + "\n")
    # ---- HBM traffic per kernel and per bench class: (2 x FETCH_SIZE + WRITE_SIZE) KB, per launch
    f, w = pmc(os.path.join(src, "pmc1.csv")), pmc(os.path.join(src, "pmc2.csv"))
    kernels, classes = {}, collections.defaultdict(lambda: [0, 0.0])
    for k in f:
        if k not in w or "FETCH_SIZE" not in f[k] or "WRITE_SIZE" not in w[k]: continue
        n = f[k]["FETCH_SIZE"][0]
        fa, wa = f[k]["FETCH_SIZE"][1] / n, w[k]["WRITE_SIZE"][1] / w[k]["WRITE_SIZE"][0]
        b = (2 * fa + wa) * 1024
        kernels[k] = {"launches": n, "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa, "hbm_bytes_per_launch_corrected": b}
        c = kclass(k)
        if c: classes[c][0] += n; classes[c][1] += b * n
    d = os.path.join(R, "pytorch_nested-unet_amd", "csrc")
    out = {"kernel_source_sha16": sha16([os.path.join(d, x) for x in ("conv3x3.hip", "elementwise.hip", "plan.hip", "common.h")]),
           "workload": ["bf16", BATCH, SIZE],
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager single-lane bench); "
                     "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md)",
           "classes": {c: {"launches": v[0], "hbm_bytes_per_launch_corrected": v[1] / v[0]} for c, v in classes.items()},
           "kernels": kernels}
    # the wgrad pair launch covers bench.py's two wgrad classes
    if "conv3x3_wgrad" in out["classes"]:
        for c in ("conv3x3_wgrad(Cout=32)", "conv3x3_wgrad(Cout>=64)"):
            out["classes"][c] = dict(out["classes"]["conv3x3_wgrad"], note="average over all weight-gradient pair launches")
    json.dump(out, open(os.path.join(P, tag + "_pmc_traffic.json"), "w"), indent=1)
    # ---- MFMA / LDS / instruction mix
    a, b = pmc(os.path.join(src, "pmc3.csv")), pmc(os.path.join(src, "pmc4.csv"))
    rows = []
    for k in a:
        g = lambda d_, c: (d_[k][c][1] / d_[k][c][0]) if k in d_ and c in d_[k] else 0.0
        n = list(a[k].values())[0][0]
        gui = g(a, "GRBM_GUI_ACTIVE") / 8.0                      # summed over the 8 XCDs
        mfma_busy = g(a, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0     # summed over 256 CUs x 4 SIMDs
        rows.append({"kernel": k, "class": kclass(k), "launches": n, "kernel_cycles": gui,
                     "mfma_busy_frac": mfma_busy / gui if gui else 0.0,
                     "insts_valu": g(b, "SQ_INSTS_VALU"), "insts_salu": g(b, "SQ_INSTS_SALU"), "insts_lds": g(b, "SQ_INSTS_LDS"), "insts_mfma": g(b, "SQ_INSTS_MFMA"),
                     "lds_active_frac": (g(b, "SQ_LDS_IDX_ACTIVE") / 256.0) / gui if gui else 0.0,
                     "lds_bank_conflict_frac_of_active": g(b, "SQ_LDS_BANK_CONFLICT") / g(b, "SQ_LDS_IDX_ACTIVE") if g(b, "SQ_LDS_IDX_ACTIVE") else 0.0,
                     "wave_cycles": g(a, "SQ_WAVE_CYCLES"), "wait_inst_lds": g(b, "SQ_WAIT_INST_LDS")})
    rows.sort(key=lambda r: -r["kernel_cycles"] * r["launches"])
    json.dump({"method": "rocprofv3 --kernel-trace --pmc, two passes (cycles; instruction mix), eager single-lane bench; per-launch averages. "
                         "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)", "kernels": rows},
              open(os.path.join(P, tag + "_pmc_mfma_lds.json"), "w"), indent=1)
    t1, tot1 = stats_table(os.path.join(P, tag + "_bench_kernel_stats.csv"), STEPS)
    t2, tot2 = stats_table(os.path.join(P, tag + "_bench_kernel_stats_single_lane.csv"), STEPS)
    print("## multi-lane (bench under rocprof: %.0f img/s, %.3f ms/step; summed kernel time %.0f us/step)\n" % (lines["multi"]["value"], lines["multi"]["ms_per_step"], tot1) + t1)
    print("\n## single lane (bench under rocprof: %.0f img/s, %.3f ms/step; summed kernel time %.0f us/step)\n" % (lines["single"]["value"], lines["single"]["ms_per_step"], tot2) + t2)
    print("\n## HBM traffic per bench class (bytes per launch)")
    for c, v in sorted(out["classes"].items()): print("| %s | %d | %.2f MB |" % (c, v["launches"], v["hbm_bytes_per_launch_corrected"] / 1e6))
    print("\n## MFMA / LDS / instruction mix (top kernels)")
    print("| kernel | launches | MFMA busy | LDS active | bank-conflict share of LDS-active | VALU : SALU : LDS : MFMA instructions |")
    for r in rows[:14]:
        print("| `%s` | %d | %.1f %% | %.1f %% | %.1f %% | %.0f : %.0f : %.0f : %.0f |" % (r["kernel"][:60], r["launches"], 100 * r["mfma_busy_frac"], 100 * r["lds_active_frac"],
              100 * r["lds_bank_conflict_frac_of_active"], r["insts_valu"], r["insts_salu"], r["insts_lds"], r["insts_mfma"]))


if __name__ == "__main__":
    main()
