#!/bin/bash
set -e
O=gpurun_out/ab_wg.txt
: > $O
for t in "128 256" "256 512" "384 768" "512 1024" "768 1536"; do
  set -- $t
  echo "=== T1=$1 T2=$2" >> $O
  NUNET_WG_T1=$1 NUNET_WG_T2=$2 HW=256 NB=32 timeout -k 10 200 python tools/wgrad_layers.py >> $O 2>&1
done
for lib in "" tools/_diag/libnunet_wg2.so tools/_diag/libnunet_wg3.so; do
  export NUNET_LIB_PATH=$lib
  echo "=== lib: ${lib:-default}" >> $O
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --steps 200 2>&1 | grep '^{' | cut -c1-200 >> $O
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32 --size 256 --batch 32 --steps 20 2>&1 | grep '^{' | cut -c1-200 >> $O
done
grep -v amdgpu.ids $O
