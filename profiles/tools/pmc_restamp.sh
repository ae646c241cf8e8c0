# PMC passes only (restamp): bash profiles/tools/pmc_restamp.sh <commit>
export SDXL_MEASURE_COMMIT=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_restamp; rm -rf $O && mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0"
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$d -o p -- $B --steps 1 --warmup 1 2>&1 | tail -1 | cut -c1-100
done
cd $O/pmc && python - <<'PY'
import csv, glob, collections, json, os
def fam(n):
    if 'gemm_kernel' in n or 'gemm256_kernel' in n or 'gemm_sk_kernel' in n or 'conv_wgrad3_kernel' in n or 'cr256_kernel' in n or 'wgrad256_kernel' in n or 'pl_kernel' in n: return 'gemm'
    if 'attn_' in n: return 'attention'
    if 'splitk' in n: return 'splitk_reduce'
    if n.startswith('void at::') or 'at::native' in n or 'repack' in n or 'elementwise_kernel' in n or 'distribution' in n: return None
    return 'norm_elementwise_loss'
out = {"workload": "ddpm_b4_1024", "commit": os.environ.get("SDXL_MEASURE_COMMIT", "unknown"), "note": "last step of `bench.py --steps 1 --warmup 1`; FETCH_SIZE / WRITE_SIZE are in KiB "
       "(hbm_bytes = FETCH_SIZE x 2 x 1024 [gfx950 correction] + WRITE_SIZE x 1024); kernels are serialised by the collection"}
raw = {}
for d in ['SQ_VALU_MFMA_BUSY_CYCLES', 'FETCH_SIZE', 'WRITE_SIZE']:
    rows = list(csv.DictReader(open(glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0])))
    start = max(int(r['Dispatch_Id']) for r in rows if 'loss_prepare' in r['Kernel_Name'])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if int(r['Dispatch_Id']) < start: continue
        k = fam(r['Kernel_Name'])
        if k is None: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        agg[k]['launches_' + d] += 1
    raw[d] = agg
for k in raw['FETCH_SIZE']:
    f, w, s = raw['FETCH_SIZE'][k], raw['WRITE_SIZE'][k], raw['SQ_VALU_MFMA_BUSY_CYCLES'][k]
    out[k] = {"launches": int(f['launches_FETCH_SIZE']), "FETCH_SIZE_KiB": f['FETCH_SIZE'], "WRITE_SIZE_KiB": w['WRITE_SIZE'],
              "hbm_bytes_per_step": f['FETCH_SIZE'] * 2 * 1024 + w['WRITE_SIZE'] * 1024,
              "SQ_VALU_MFMA_BUSY_CYCLES": s['SQ_VALU_MFMA_BUSY_CYCLES'], "SQ_BUSY_CYCLES": s['SQ_BUSY_CYCLES'],
              "GRBM_GUI_ACTIVE": s['GRBM_GUI_ACTIVE'],
              "mfma_util": s['SQ_VALU_MFMA_BUSY_CYCLES'] / (s['GRBM_GUI_ACTIVE'] / 8 * 1024) if s['GRBM_GUI_ACTIVE'] else None}
json.dump(out, open('../pmc_step_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
cp $O/pmc_step_summary.json $R/gpurun_out/pmc_step_summary_restamp.json; rm -rf $O/pmc
