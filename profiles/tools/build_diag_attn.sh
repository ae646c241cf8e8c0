#!/bin/bash
# knock-out builds of the attention forward kernel (diagnostics only): lib_attn_d<bits>.so next to the product library
set -e
cd "$(dirname "$0")/../../sdxl-training-improvements_amd"
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATTN_DIAG=$d -c csrc/attention.hip -o build/attention_d$d.o
  objs=$(ls build/*.hip.o | grep -v attention.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../profiles/tools/lib_attn_d$d.so $objs build/attention_d$d.o
done
