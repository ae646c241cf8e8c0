#!/bin/bash
# attention backward A/B on one box, interleaved; diagnostics build.  knob 35: 0 = attn_bwd_fused_kernel (attention.hip, shipped), 2 = attention_bwd_pl.hip
cd $GRAFT_REPO_ROOT
export SDXL_DIAG=1
for i in 1 2; do
  for k in ${@:-"35=0" "35=2"}; do
    echo "== knob $k"; SDXL_KNOBS=$k python profiles/tools/attn_bench.py --self-only --iters 30 2>&1 | grep "attn bwd"
  done
done
