#!/bin/bash
# PMC passes over the attention kernels (separate --pmc runs, kernel-trace only): SQ occupancy / stall / pipe counters
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_attn}
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $R/profiles/tools/attn_bench.py --self-only --iters 2 > /dev/null 2>&1
done
cd $OUT && python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('p*/**/*counter_collection.csv',recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'attn' not in n: continue
        key=n.split('(')[0]+' grid'+r['Grid_Size_X']+'x'+r.get('Grid_Size_Y','')
        agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id'],key) not in seen and r['Counter_Name'] in ('SQ_WAVE_CYCLES','SQ_ACTIVE_INST_VALU','SQ_INSTS_VALU'):
            seen.add((r['Dispatch_Id'],key))
for k,v in sorted(agg.items()):
    print(k)
    wc=v.get('SQ_WAVE_CYCLES',1)
    for c in sorted(v): print(f"   {c:28s} {v[c]:16.0f}  {v[c]/wc:8.3f} of wave-cycles")
PY
find $OUT -name "*.csv" -size +2M -delete
