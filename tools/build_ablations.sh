#!/bin/bash
# Diagnostic builds of the library with parts of the conv kernel disabled (NUNET_ABL bit mask:
# 1 = no global loads after the first, 2 = no LDS staging writes after the first, 4 = no MFMA sweep,
# 8 = no epilogue stores). Results are wrong by construction: timing only (tools/conv_bench.py with
# NUNET_LIB_PATH=pytorch_nested-unet_amd/libnunet_abl<mask>.so).
set -e
cd "$(dirname "$0")/../pytorch_nested-unet_amd/csrc"
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNUNET_ABL=$m -Wno-unused-variable -shared -o ../libnunet_abl$m.so conv3x3.hip elementwise.hip plan.hip prof.hip lovasz.hip graph.hip
done
