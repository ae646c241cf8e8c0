#!/bin/bash
# PMC passes over the co-resident 256-row GEMM kernel (separate --pmc runs, kernel-trace only) on TN 10240 x 1280 x 4096, lockstep (mode 128)
# and phased (140) loop: SQ stall / issue / pipe counters per wave-cycle.   bash profiles/tools/pmc_cr256.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_cr256}
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/cr_one.py <<'PY'
import ctypes as C, sys, torch
sys.path.insert(0, sys.argv[1])
import sdxl_amd
from sdxl_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
M, N, K = 10240, 1280, 4096
a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
o = torch.empty(M, N, device=dev, dtype=torch.float32)
for mode in (52, 128, 140):
    lib.check(L.sdxl_set_gemm_mode(mode))
    for _ in range(3):
        lib.check(L.sdxl_op_gemm(2, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None, None, 0, 1, None))
    torch.cuda.synchronize()
PY
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -o p -- python /tmp/cr_one.py $R > /dev/null 2>&1
done
cd $OUT && python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
for f in glob.glob('p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'cr256' not in n and 'gemm_kernel' not in n: continue
        import re
        m=re.search(r'(cr256_kernel<[^>]*>|gemm_kernel<[^>]*>)', n)
        key=m.group(1) if m else n[:60]
        agg[key][r['Counter_Name']]+=float(r['Counter_Value']); nd[key].add(r['Dispatch_Id'])
for k,v in sorted(agg.items()):
    print(k, f"({len(nd[k])} launches)")
    wc=v.get('SQ_WAVE_CYCLES',1)
    for c in sorted(v): print(f"   {c:28s} {v[c]:16.0f}  {v[c]/wc:8.3f} of wave-cycles")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
