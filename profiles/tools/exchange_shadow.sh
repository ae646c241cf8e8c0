#!/bin/bash
# price of the gradient exchange's device kernels on ONE GPU (bench.py --exchange-shadow CH:LDS_KB:GBPS), two alternations
B="python bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0 --steps 12 --warmup 3"
for i in 1 2; do
for s in none 16:64:100000 8:64 16:64 32:64 64:64 16:16 16:64:150; do
  echo -n "shadow $s: "
  if [ $s = none ]; then $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  else $B --exchange-shadow $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; fi
done; done
