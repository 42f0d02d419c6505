#!/bin/bash
# sweep of the fragment prefetch distance of the conv sweep (libnunet_pd<N>.so built with -DNUNET_SWEEP_PD=N)
for pd in 1 2 3 4; do
  lib=$GRAFT_REPO_ROOT/pytorch_nested-unet_amd/libnunet_pd$pd.so
  [ $pd = 1 ] && lib=$GRAFT_REPO_ROOT/pytorch_nested-unet_amd/libnunet.so
  r=$(cd $GRAFT_REPO_ROOT && NUNET_LIB_PATH=$lib python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "PD $pd : $r"
  (cd /tmp && TMPDIR=/tmp NUNET_LIB_PATH=$lib rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cp_pd$pd -o run -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py > /dev/null 2>&1)
done
