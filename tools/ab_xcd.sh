#!/bin/bash
# A/B of two libnunet builds inside ONE gpurun call: tools/ab_xcd.sh <variant .so> -> per-layer tables and the bench line for both
set -e
V=$1
O=gpurun_out/ab_$(basename $V .so).txt
: > $O
for lib in "" "$V"; do
  export NUNET_LIB_PATH=$lib
  echo "=== lib: ${lib:-default}" >> $O
  timeout -k 10 200 python tools/conv_layers.py >> $O 2>&1
  timeout -k 10 200 python tools/wgrad_layers.py >> $O 2>&1
  HW=256 NB=32 MAXLEV=3 timeout -k 10 200 python tools/conv_layers.py >> $O 2>&1
  HW=256 NB=32 timeout -k 10 200 python tools/wgrad_layers.py >> $O 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --steps 200 2>&1 | grep '^{' >> $O
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --size 256 --batch 32 --steps 20 2>&1 | grep '^{' >> $O
done
grep -v amdgpu.ids $O | grep "^sum\|^===\|^{" | cut -c1-260
