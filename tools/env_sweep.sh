#!/bin/bash
# usage: tools/env_sweep.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument = one environment, bench img/s printed)
for e in "$@"; do
  r=$(env $e python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$e : $r"
done
