#!/bin/bash
# knock-out builds of the 256x256 GEMM kernel (diagnostics only): lib_g256_d<bits>.so next to the product library
set -e
cd "$(dirname "$0")/../../sdxl-training-improvements_amd"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DG256_DIAG=$d -c csrc/gemm256.hip -o build/gemm256_d$d.o
  objs=$(ls build/*.hip.o | grep -v gemm256.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../profiles/tools/lib_g256_d$d.so $objs build/gemm256_d$d.o
done
