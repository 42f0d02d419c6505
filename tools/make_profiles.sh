#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 passes behind profiles/rNN_*. Outputs under gpurun_out/prof_final/.
# Each rocprofv3 invocation profiles `python3 bench.py` directly (no wrapper between -- and the program).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi -o run -- python3 $B > $O/multi.json 2> $O/multi.err
echo "multi-lane done"
NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -o run -- python3 $B > $O/single.json 2> $O/single.err
echo "single-lane done"
NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $B --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "pmc fetch done"
NUNET_MULTISTREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $B --no-graph > $O/pmc_write.json 2> $O/pmc_write.err
echo "pmc write done"
ls $O/*
