#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/conv_layers.py 2>/dev/null > gpurun_out/r2_cl4.log; cat gpurun_out/r2_cl4.log
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
run A=1
