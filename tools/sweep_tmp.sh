#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))"; }
run A=1
run NUNET_FUSE_BNR=0
run NUNET_SK_MAXITEMS=100
run NUNET_SK_MAXITEMS=100 NUNET_FUSE_BNR=0
run A=1
run NUNET_WG_DEFER=1
