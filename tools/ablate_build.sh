#!/bin/bash
# Diagnostic builds of libnunet with parts of the conv kernel left out (-DNUNET_ABLATE=bits, see conv3x3.hip):
#   tools/ablate_build.sh 1 2 4 8 16   -> tools/_diag/libnunet_abl<bits>.so   (git-ignored; they travel to the GPU box)
# then: NUNET_LIB_PATH=tools/_diag/libnunet_abl4.so ONLY=1,1,0 python tools/conv_layers.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
S=pytorch_nested-unet_amd/csrc
for f in elementwise plan prof lovasz graph; do [ -f $S/$f.o ] || make -C $S $f.o; done
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNUNET_ABLATE=$b -Wno-unused-variable -c $S/conv3x3.hip -o tools/_diag/conv_abl$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/libnunet_abl$b.so tools/_diag/conv_abl$b.o $S/elementwise.o $S/plan.o $S/prof.o $S/lovasz.o $S/graph.o
  rm tools/_diag/conv_abl$b.o
done
ls -la tools/_diag
