#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/make_profiles.sh [SIZE BATCH OUTDIR]): every rocprofv3 pass behind profiles/rNN_*.
# Outputs under gpurun_out/OUTDIR (default prof_final, workload 96 16); tools/summarize_profiles.py turns them into the tracked files.
# Each rocprofv3 invocation profiles `python3 bench.py` directly (no wrapper between -- and the program); counter
# passes carry only --kernel-trace beside --pmc.
set -e
SIZE=${1:-96}; BATCH=${2:-16}; OUT=${3:-prof_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$OUT
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the executor is pinned to the single hipGraph: the capture-time choice would otherwise run its trial steps under the profiler
# (extra launches in the per-step counts), and the flag-synchronised lanes' polling kernels spin for as long as the profiler
# serialises the queues (9.6 ms per step under --kernel-trace). Kernel durations do not depend on the executor.
export NUNET_SEGMENTED=0 NUNET_SCHEDULE=lanes
W="--size $SIZE --batch $BATCH --no-cpu-baseline --no-roofline --no-fp32"
B="$R/bench.py --steps 20 --warmup 5 $W"
keep() {  # keep() <dir> <pattern> <dest>: copy the one csv we need, drop the rest
  F=$(find $1 -name "$2" | head -1); cp "$F" "$3"; rm -rf $1
}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi -o run -- python3 $B > $O/multi.json 2> $O/multi.err
keep $O/multi "*kernel_stats.csv" $O/multi_kernel_stats.csv
echo "multi-lane done"
NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -o run -- python3 $B > $O/single.json 2> $O/single.err
keep $O/single "*kernel_stats.csv" $O/single_kernel_stats.csv
echo "single-lane done"
# counter passes: eager (no hipGraph), single lane, 8 steps
P="$R/bench.py --steps 6 --warmup 2 --no-graph $W"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -o run -- python3 $P > $O/pmc$i.json 2> $O/pmc$i.err
  keep $O/pmc$i "*counter_collection.csv" $O/pmc$i.csv
  echo "pmc pass $i done: $C"
done
ls -la $O
