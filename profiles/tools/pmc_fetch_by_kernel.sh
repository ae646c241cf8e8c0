# HBM-side fetch bytes (FETCH_SIZE x 2 KiB, gfx950 correction) of the last step, per kernel and grid: bash profiles/tools/pmc_fetch_by_kernel.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python $R/bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0 --steps 1 --warmup 1 2>&1 | tail -1 | cut -c1-100
python - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True)[0])))
start = max(int(r['Dispatch_Id']) for r in rows if 'loss_prepare' in r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r['Dispatch_Id']) < start or r['Counter_Name'] != 'FETCH_SIZE': continue
    k = (r['Kernel_Name'][:64], r.get('Grid_Size', r.get('Grid_Size_X', '')), r.get('LDS_Block_Size', ''))
    agg[k][0] += 1; agg[k][1] += float(r['Counter_Value']) * 2 * 1024
tot = sum(v[1] for v in agg.values())
print("total fetch GB %.1f" % (tot / 1e9))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%7.2f GB %4d x %7.1f MB  %s grid %s" % (v[1] / 1e9, v[0], v[1] / v[0] / 1e6, k[0], k[1]))
PY
