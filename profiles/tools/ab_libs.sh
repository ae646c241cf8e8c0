#!/bin/bash
# same-box A/B of two builds of the product library: bash profiles/tools/ab_libs.sh <rounds> ab_libs/old.so ab_libs/new.so ...  (ms per step, median)
R=$1; shift
P=sdxl-training-improvements_amd/libsdxlstep.so
cp $P /tmp/keep.so
B="python bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0 --steps 15 --warmup 4"
for i in $(seq 1 $R); do
  for l in "$@"; do
    cp $l $P
    echo -n "[$l] "; $B 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_time']['median_ms'])"
  done
done
cp /tmp/keep.so $P
