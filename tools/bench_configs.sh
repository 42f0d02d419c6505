#!/bin/bash
# the other BASELINE configurations (parity cases; not bench lines): images/s, ms/step
run() { r=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"); echo "$* : $r"; }
run
run --deep-supervision
run --size 256 --batch 32
run --size 256 --batch 4
run --size 512 --batch 1 --num-classes 4 --dtype fp16
run --size 512 --batch 8 --num-classes 4 --dtype fp16
run --dtype fp32
run --dtype fp16
