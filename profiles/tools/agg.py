import csv,collections,sys
rows=[r for r in csv.reader(open(sys.argv[1]))]
agg=collections.OrderedDict()
for form,taps,M,N,K,sk,ms in rows:
    k=(int(form),int(taps),int(M),int(N),int(K),int(sk))
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(ms)
tot=sum(a[1] for a in agg.values())
print("total ms",tot)
names={0:'NT',1:'NN',2:'TN'}
byform=collections.Counter()
lim=float(sys.argv[2]) if len(sys.argv)>2 else 1.5
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    form,taps,M,N,K,sk=k
    fl=2.0*M*N*K*taps*a[0]
    byform[(names[form],taps)]+=a[1]
    if a[1]>lim: print(f"{names[form]} taps={taps} M={M:6d} N={N:6d} K={K:6d} sk={sk:2d} n={a[0]:3d} ms={a[1]:7.3f} ({100*a[1]/tot:4.1f}%) TF/s={fl/a[1]/1e9:7.1f} us/each={1e3*a[1]/a[0]:.1f}")
print({k:round(v,2) for k,v in byform.items()})
