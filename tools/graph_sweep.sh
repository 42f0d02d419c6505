#!/bin/bash
# bench under the scheduling toggles: NUNET_LISTSCHED x NUNET_GRAPH_REWRITE x NUNET_GRAPH_OWN_STREAM
for own in 0 1; do for ls in 0 1; do for rw in 0 1; do
  r=$(NUNET_GRAPH_OWN_STREAM=$own NUNET_GRAPH_REWRITE=$rw NUNET_LISTSCHED=$ls python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "own_stream $own listsched $ls rewrite $rw: $r"
done; done; done
