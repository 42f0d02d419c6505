#!/bin/bash
# starting points for the edge-order autotuner
run() { r=$(env "$@" NUNET_GRAPH_VERBOSE=1 NUNET_GRAPH_TUNE=${IT:-400} python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>gpurun_out/tm.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"); echo "$* : $r | $(grep -E 'tuned' gpurun_out/tm.err | tail -1)"; }
run X=1
run NUNET_LANE_MAP=0123441234
run NUNET_LANE_MAP=0123411234
run NUNET_LANE_MAP=0123431234
run NUNET_LISTSCHED=1
