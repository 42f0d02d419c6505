#!/usr/bin/env python3
"""Per-queue timeline of ONE training step from a rocprofv3 --kernel-trace CSV.

usage: tools/timeline.py <kernel_trace.csv> [step_index_from_end=3]
A step is delimited by the weight repack kernel (`pack_kernel`), the first launch of plan_forward.
Prints start(us) dur(us) queue class workgroups, then per-class totals and the union-busy time.
"""
import csv, sys, collections

SHORT = [("pack_kernel", "pack"), ("unpack_kernel", "unpack"), ("nchw_to_nhwc", "layout"), ("splitk_finalize", "skfin"),
         ("conv3x3_kernel", "conv"), ("wgrad_kernel", "wgrad"), ("bn_relu_fwd", "bnF"), ("bn_relu_bwd_kernel<", "bnB"),
         ("upsample_fwd", "upF"), ("upsample_bwd", "upB"), ("maxpool_bwd", "poolB"), ("maxpool_fwd", "poolF"),
         ("head_fwd", "headF"), ("head_bwd", "headB"), ("loss_step", "loss"), ("iou", "iou"), ("sgd_kernel", "sgd"),
         ("zero_kernel", "zero"), ("bce_dice", "loss")]


def short(name):
    for k, v in SHORT:
        if k in name:
            if v == "bnB":
                return "bnBr" if "true" in name.split("bn_relu_bwd_kernel<")[1][:24] or ", 0" in name else "bnB"
            return v
    return "other"


def main():
    if sys.argv[1].endswith(".db"):
        import sqlite3
        cur = sqlite3.connect(sys.argv[1]).cursor()
        rows = [dict(Kernel_Name=r[0], Start_Timestamp=r[1], End_Timestamp=r[2], Queue_Id=r[3], Grid_Size_X=r[4], Grid_Size_Y=r[5],
                     Grid_Size_Z=r[6], Workgroup_Size_X=r[7], Workgroup_Size_Y=r[8], Workgroup_Size_Z=r[9])
                for r in cur.execute("select name,start,end,queue_id,grid_x,grid_y,grid_z,workgroup_x,workgroup_y,workgroup_z from kernels")]
    else:
        rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"] and "unpack" not in r["Kernel_Name"]]
    a, b = starts[-back - 1], starts[-back]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    tot = collections.Counter(); cnt = collections.Counter()
    ev = []
    for r in step:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        wg = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        c = short(r["Kernel_Name"])
        print(f"{s:9.1f} {e - s:7.1f} q{r['Queue_Id']} {c:7s} wg={wg}")
        tot[c] += e - s; cnt[c] += 1
        ev.append((s, e))
    ev.sort()
    busy, cur_s, cur_e = 0.0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(e for _, e in ev)
    print(f"# span {span:.1f} us  union-busy {busy:.1f} us  sum {sum(tot.values()):.1f} us  launches {len(step)}")
    for c, v in tot.most_common():
        print(f"# {c:7s} {cnt[c]:4d} launches {v:8.1f} us")


if __name__ == "__main__":
    main()
