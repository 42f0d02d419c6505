b() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-fp32 2>gpurun_out/r3_seg_err.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4), d['final_loss'])" || tail -15 gpurun_out/r3_seg_err.txt; }
NUNET_SEGMENTED=0 b "one hipGraph"
b "segmented"
NUNET_SEGMENTED=0 b "one hipGraph again"
b "segmented again"
