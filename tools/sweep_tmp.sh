#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/conv_layers.py > gpurun_out/r2_cl2.log 2>&1
NUNET_SK_MAXITEMS=60 python tools/conv_layers.py > gpurun_out/r2_cl3.log 2>&1
paste gpurun_out/r2_cl2.log gpurun_out/r2_cl3.log
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
run A=1
run NUNET_SK_MAXITEMS=60
run NUNET_CONV_SMALL=0
run NUNET_CONV_SMALL=1
