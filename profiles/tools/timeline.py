"""Stream busy / gap analysis of one training step from a rocprofv3 --kernel-trace (raw *_kernel_trace.csv or the
compact .csv.gz written by trace.sh): python profiles/tools/timeline.py <trace>"""
import gzip,collections,sys,csv
rows=[]
path=sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/trace/compact.csv.gz'
if path.endswith('.gz'):
    for line in gzip.open(path,'rt'):
        s,e,q,st,name=line.rstrip('\n').split(',',4)
        rows.append((int(s),int(e),q,name))
else:
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),str(r['Queue_Id']),r['Kernel_Name'][:60].replace(',',';')))
mainq=collections.Counter(r[2] for r in rows if 'loss_prepare' in r[3]).most_common(1)[0][0]   # the caller's stream
rows=[(a,b,'2' if q==mainq else 'x',n) for a,b,q,n in rows]
rows.sort()
lp=[i for i,r in enumerate(rows) if 'loss_prepare' in r[3]]
lb=[i for i,r in enumerate(rows) if 'loss_bwd' in r[3]]
i0=lp[-1]; ib=lb[-1]
step=rows[i0:]
t0=step[0][0]; tb=rows[ib][0]; t1=max(r[1] for r in step)
print(f"step span {(t1-t0)/1e6:.2f} ms ; forward {(tb-t0)/1e6:.2f} ; backward {(t1-tb)/1e6:.2f}")
def busy(rs):
    # union length
    iv=sorted((r[0],r[1]) for r in rs); tot=0; cs,ce=None,None
    for s,e in iv:
        if cs is None: cs,ce=s,e
        elif s<=ce: ce=max(ce,e)
        else: tot+=ce-cs; cs,ce=s,e
    if cs is not None: tot+=ce-cs
    return tot
for phase,(a,b) in {'fwd':(t0,tb),'bwd':(tb,t1)}.items():
    ph=[r for r in step if r[0]>=a and r[0]<b]
    main=[r for r in ph if r[2]=='2']; side=[r for r in ph if r[2]!='2']
    print(phase, f"main: n={len(main)} sum={sum(r[1]-r[0] for r in main)/1e6:.2f} busy(union)={busy(main)/1e6:.2f} | side: n={len(side)} sum={sum(r[1]-r[0] for r in side)/1e6:.2f} busy={busy(side)/1e6:.2f} | any busy={busy(ph)/1e6:.2f} of {(b-a)/1e6:.2f}")
    # gaps on main
    gaps=[]; prev=None
    for r in sorted(main):
        if prev is not None and r[0]>prev: gaps.append(r[0]-prev)
        prev=max(prev,r[1]) if prev else r[1]
    import statistics
    if gaps: print("   main gaps: n=%d total=%.2f ms median=%.2f us mean=%.2f us"%(len(gaps),sum(gaps)/1e6,statistics.median(gaps)/1e3,sum(gaps)/len(gaps)/1e3))
    agg=collections.Counter(); cnt=collections.Counter()
    for r in ph:
        k=('S:' if r[2]!='2' else 'M:')+r[3][:44]; agg[k]+=r[1]-r[0]; cnt[k]+=1
    for k,v in agg.most_common(16): print(f"   {v/1e6:7.2f} ms {cnt[k]:5d}  {k}")
