#!/bin/bash
# conv_bench over the ablation builds (see tools/build_ablations.sh)
export ONLY="${ONLY:-L0 32->32,L0 192->32,L1 320->64,L2 128->128,L3 256->256,L3 768->256,L4 512->512}"
for m in 0 1 2 3 4 8 15; do
  lib=pytorch_nested-unet_amd/libnunet_abl$m.so
  [ $m = 0 ] && lib=pytorch_nested-unet_amd/libnunet.so
  [ -f $lib ] || continue
  echo "== ABL $m"
  NUNET_LIB_PATH=$PWD/$lib python tools/conv_bench.py fwd 2>/dev/null
done
