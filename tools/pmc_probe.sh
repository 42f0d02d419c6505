#!/bin/bash
# Run on the GPU box: rocprofv3 PMC passes (counters only, with --kernel-trace) of the eager single-lane bench.
# usage: tools/pmc_probe.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph"
i=0
for C in "$@"; do
  i=$((i+1))
  NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o run -- python3 $B > $O/p$i.json 2> $O/p$i.err
  F=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$O/pass$i.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,launches,avg\n")
    for k, d in acc.items():
        for c, (n, v) in d.items():
            f.write('"%s",%s,%d,%.3f\n' % (k.replace('"', "'"), c, n, v / n))
PY
  rm -rf $O/p$i
  echo "pass $i done: $C"
done
