B="python bench.py --no-cpu-baseline --no-optimizer --profile-steps 0 --steps 15 --warmup 4"
for i in 1 2 3; do
  for k in "16=1" "16=0" "16=2" "16=3"; do
    echo -n "knob $k: "; $B --knob $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_time']['median_ms'])"
  done
done
