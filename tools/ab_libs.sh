#!/bin/bash
# A/B of libnunet builds inside ONE gpurun call: tools/ab_libs.sh <out tag> <lib or "-"> ... ; per lib the default bench (executor chosen at capture)
set -e
O=gpurun_out/ab_$1.txt; shift
: > $O
for lib in "$@"; do
  [ "$lib" = "-" ] && export NUNET_LIB_PATH= || export NUNET_LIB_PATH=$lib
  echo "=== lib: $lib" >> $O
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --no-roofline --steps 200 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['executor']['ms'])" >> $O
done
cat $O
