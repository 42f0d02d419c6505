#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
python tools/wgrad_stamps.py 2>/dev/null | tail -14
NUNET_WG_GLDS=0 python tools/wgrad_layers.py 2>/dev/null > gpurun_out/r2_wg8.log; cat gpurun_out/r2_wg8.log
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
run NUNET_WG_GLDS=0
