#!/bin/bash
# usage: CFG="--size 256 --batch 32" tools/env_sweep_cfg.sh "VAR=a" "VAR=b" ...
for e in "$@"; do
  r=$(env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $CFG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$CFG | $e : $r"
done
