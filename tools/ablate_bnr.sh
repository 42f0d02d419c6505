#!/bin/bash
# kernel durations of the BNR conv epilogue with parts disabled (NUNET_ABL bits 16 y loads, 32 compute, 64 shuffles, 128 atomics)
cd /tmp && export TMPDIR=/tmp
for m in 0 16 32 64 128 240; do
  lib=$GRAFT_REPO_ROOT/pytorch_nested-unet_amd/libnunet_abl$m.so
  [ $m = 0 ] && lib=$GRAFT_REPO_ROOT/pytorch_nested-unet_amd/libnunet.so
  NUNET_LIB_PATH=$lib rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bnrp_$m -o run -- python3 $GRAFT_REPO_ROOT/tools/bnr_probe.py > /dev/null 2>&1
done
