cd /tmp && export TMPDIR=/tmp
for m in product diag_unfused; do
  echo "== $m"
  if [ $m = product ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$m -o t -- python $GRAFT_REPO_ROOT/profiles/tools/attn_bench.py --self-only --iters 10 2>&1 | grep -E "attn (fwd|bwd)"
  else
    SDXL_DIAG=1 SDXL_KNOB20=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$m -o t -- python $GRAFT_REPO_ROOT/profiles/tools/attn_bench.py --self-only --iters 10 2>&1 | grep -E "attn (fwd|bwd)"
  fi
  python - <<PY
import csv,glob
f=glob.glob('/tmp/at_$m/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attn' in r['Name']: print('   ', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg')
PY
done
