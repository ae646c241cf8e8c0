#!/bin/bash
# knock-out builds of the co-resident 256-row GEMM kernel (diagnostics only): lib_cr_d<bits>.so next to this script
#   bits: 1 no MFMA, 2 no main-loop DMA, 4 no LDS fragment reads, 8 every workgroup stages tile (0, 0)
set -e
cd "$(dirname "$0")/../../sdxl-training-improvements_amd"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSDXL_CR_DIAG=$d -c csrc/gemm_cr256.hip -o build/gemm_cr256_d$d.o
  objs=$(ls build/*.hip.o | grep -v gemm_cr256.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../profiles/tools/lib_cr_d$d.so $objs build/gemm_cr256_d$d.o
done
