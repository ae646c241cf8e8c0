#!/bin/bash
# knock-out builds of the co-resident 256-row GEMM kernel (diagnostics only): lib_cr_d<bits>.so next to this script
#   bits: 1 no MFMA, 2 no main-loop DMA, 4 no LDS fragment reads, 8 every workgroup stages tile (0, 0)
set -e
cd "$(dirname "$0")/../../sdxl-training-improvements_amd"
#   a bit pattern >= 256 applies (bits & 255) to gemm.hip's kernels as well (-DSDXL_GEMM_DIAG)
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSDXL_CR_DIAG=$((d & 255)) -c csrc/gemm_cr256.hip -o build/gemm_cr256_d$d.o
  objs=$(ls build/*.hip.o | grep -v gemm_cr256.hip.o)
  if [ $d -ge 256 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSDXL_GEMM_DIAG=$((d & 255)) -c csrc/gemm.hip -o build/gemm_d$d.o
    objs="$(echo $objs | tr ' ' '\n' | grep -v 'build/gemm.hip.o' | tr '\n' ' ') build/gemm_d$d.o"
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../profiles/tools/lib_cr_d$d.so $objs build/gemm_cr256_d$d.o
done
