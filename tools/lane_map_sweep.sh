for m in 0123401234 0123411234 0123421234 0123431234 0123441234 0123412341; do
  for q in 4 6; do
    r=$(DEBUG_HIP_FORCE_GRAPH_QUEUES=$q NUNET_LISTSCHED=0 NUNET_LANE_MAP=$m python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "map $m queues $q: $r"
  done
done
