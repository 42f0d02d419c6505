#!/bin/bash
# Run on the GPU box after tools/ablate_build.sh <bits...>: us per launch of conv cases with parts of the kernel compiled out.
#   CASES="0,0,1 1,1,0" BUILDS="1 2 4 8 16 31" bash tools/ablate_run.sh      (case = level,column,kind of tools/conv_layers.py)
CASES=${CASES:-"0,0,1 0,4,0 1,1,0 2,2,0 4,0,1"}
BUILDS=${BUILDS:-"1 2 4 8 16 31"}
for c in $CASES; do
  line="$c: full $(ONLY=$c python tools/conv_layers.py 2>/dev/null | grep kind | awk '{print $4}')"
  for b in $BUILDS; do
    v=$(NUNET_LIB_PATH=$GRAFT_REPO_ROOT/tools/_diag/libnunet_abl$b.so ONLY=$c timeout -k 10 100 python tools/conv_layers.py 2>/dev/null | grep kind | awk '{print $4}')
    line="$line | abl$b $v"
  done
  echo "$line"
done
