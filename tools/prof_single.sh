#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace of the bench, single lane (kernels alone) and multi lane.
# usage: tools/prof_single.sh <tag> [extra bench args]
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-probe}; shift || true
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $@"
NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -o run -- python3 $B > $O/single.json 2> $O/single.err
cp $O/single/*/run_kernel_stats.csv $O/single_kernel_stats.csv 2>/dev/null || cp $(find $O/single -name "*kernel_stats.csv" | head -1) $O/single_kernel_stats.csv
rm -rf $O/single
echo "single-lane done"
