#!/bin/bash
# same-box A/B of knob settings: bash profiles/tools/ab.sh <rounds> "<knobs A>" "<knobs B>" ...   (each a space-separated list of id=value, "-" = none)
R=$1; shift
export SDXL_DIAG=1
B="python bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0 --steps 15 --warmup 4"
for i in $(seq 1 $R); do
  for k in "$@"; do
    a=""; if [ "$k" != "-" ]; then for kv in $k; do a="$a --knob $kv"; done; fi
    echo -n "[$k] "; $B $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_time']['median_ms'])"
  done
done
