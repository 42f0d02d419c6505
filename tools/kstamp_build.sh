#!/bin/bash
# Diagnostic build of libnunet with in-kernel phase stamps (-DNUNET_KSTAMP) -> tools/_diag/libnunet_kstamp.so
# (git-ignored; travels to the GPU box). Use: NUNET_LIB_PATH=tools/_diag/libnunet_kstamp.so KSTAMP=1 ONLY=0,0,1 python tools/conv_layers.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
S=pytorch_nested-unet_amd/csrc
for f in conv3x3 elementwise plan prof lovasz graph; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNUNET_KSTAMP -Wno-unused-variable -c $S/$f.hip -o tools/_diag/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_diag/libnunet_kstamp.so tools/_diag/*.o
rm tools/_diag/*.o
ls -la tools/_diag
