# kernel durations of the fused dgrad + LayerNorm-backward launch against the plain dgrad and the separate LayerNorm pass (op level)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lnt -o t -- python $GRAFT_REPO_ROOT/profiles/tools/dbg_ln_epilogue.py 2>&1 | tail -5; find /tmp/lnt -name "*.csv" | head
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/lnt/**/*kernel_trace.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm_kernel' in n or 'ln_' in n or 'col_reduce' in n:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4} vgpr {r.get('VGPR_Count','?'):>4} scr {r.get('Scratch_Size','?')}  {n[:90]}")
PY
