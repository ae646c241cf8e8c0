#!/bin/bash
# attention forward A/B on one box, interleaved; diagnostics build.  knob 33: 0 = shipped policy, 1 = tiled kernel (attention.hip), 2 = pipelined 4-wave form
cd $GRAFT_REPO_ROOT
export SDXL_DIAG=1
for i in 1 2; do
  for k in ${@:-"33=1" "33=2" "33=0"}; do
    echo "== knob $k"; SDXL_KNOBS=$k python profiles/tools/attn_bench.py --self-only --iters 30 2>&1 | grep "attn fwd"
  done
done
