cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --profile-steps 0 2>&1 | tail -1 | cut -c1-200
ls -la $GRAFT_REPO_ROOT/gpurun_out/trace/*
cd $GRAFT_REPO_ROOT/gpurun_out/trace && python - <<'PY'
import csv,glob
f=glob.glob('**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
# keep last 40% of rows by time to bound size: write compact file
out=open('compact.csv','w')
for r in rows:
    out.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r['Queue_Id']},{r.get('Stream_Id','')},{r['Kernel_Name'][:60].replace(',',';')}\n")
out.close()
PY
gzip -f compact.csv; rm -f $(ls | grep -v compact); ls -la
