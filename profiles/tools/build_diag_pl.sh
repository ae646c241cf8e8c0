#!/bin/bash
# knock-out builds of the software-pipelined kernel (gemm_pl.hip, diagnostics only): profiles/tools/lib_pl_d<bits>.so
#   bits: 1 no MFMA, 2 no main-loop DMA, 4 no LDS fragment reads, 8 every workgroup stages tile (0, 0)
set -e
cd "$(dirname "$0")/../../sdxl-training-improvements_amd"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DSDXL_PL_DIAG=$d -c csrc/gemm_pl.hip -o build/gemm_pl_d$d.o
  objs=$(ls build/*.hip.o | grep -v gemm_pl.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=csrc/exports.map -o ../profiles/tools/lib_pl_d$d.so $objs build/gemm_pl_d$d.o
done
