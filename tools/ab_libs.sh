#!/bin/bash
# A/B of libnunet builds inside ONE gpurun call: tools/ab_libs.sh <out tag> <lib or "-"> ... ; per lib: bench default executor and flag-synchronised lanes
set -e
O=gpurun_out/ab_$1.txt; shift
: > $O
for lib in "$@"; do
  [ "$lib" = "-" ] && export NUNET_LIB_PATH= || export NUNET_LIB_PATH=$lib
  for seg in 2; do
    echo "=== lib: $lib segmented=$seg" >> $O
    NUNET_SCHEDULE=list NUNET_SEGMENTED=$seg timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --no-roofline --steps 200 2>&1 | grep '^{' | cut -c60-200 >> $O
  done
done
cat $O
