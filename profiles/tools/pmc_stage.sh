#!/bin/bash
# PMC counters of the staging arms of tools/stage_rate2 (review item 2c): separate --pmc passes, --kernel-trace only.   bash profiles/tools/pmc_stage.sh <outdir>
O=${1:-gpurun_out/r05j/pmc_stage}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$O
for arm in "0 hot" "4 hot" "9 hot" "0 cold" "4 cold"; do
  tag=$(echo $arm | tr ' ' '_')
  i=0
  for c in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs/$tag/$i -o p -- $R/profiles/tools/stage_rate2 $arm > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
out = open("$R/$O/summary.txt", "w")
for tag in ["0_hot", "4_hot", "9_hot", "0_cold", "4_cold"]:
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"/tmp/pmcs/{tag}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stage_kernel' not in r['Kernel_Name']: continue
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    line = f"{tag:8s} " + "  ".join(f"{k}={agg[k]/max(n[k],1):.4g}" for k in sorted(agg))
    print(line); out.write(line + "\n")
PY
