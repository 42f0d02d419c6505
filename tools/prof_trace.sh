#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace (per-dispatch start/end) of the multi-lane bench -> gpurun_out/prof_<tag>/trace.csv
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-trace}; shift || true
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline $@"
rocprofv3 --kernel-trace --output-format csv -d $O/multi -o run -- python3 $B > $O/multi.json 2> $O/multi.err
F=$(find $O/multi -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$O/trace_tail.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-330:]          # the last two steps
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,end_us,dur_us,queue,kernel\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write("%.1f,%.1f,%.1f,%s,%s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", ""), r["Kernel_Name"][:60].replace(",", ";")))
PY
rm -rf $O/multi
tail -1 $O/multi.json | cut -c1-150
