import collections, csv, sys, glob
f=glob.glob('/tmp/tr/**/*kernel_trace.csv',recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), str(r['Queue_Id']), r['Kernel_Name'][:60].replace(',', ';')))
mainq = collections.Counter(r[2] for r in rows if 'loss_prepare' in r[3]).most_common(1)[0][0]
rows.sort()
i0 = [i for i, r in enumerate(rows) if 'loss_prepare' in r[3]][-1]
ib = [i for i, r in enumerate(rows) if 'loss_bwd' in r[3]][-1]
# before loss_prepare: what precedes (H2D etc.)
step = rows[i0-6:ib+1]
main = [r for r in step if r[2] == mainq]
side = [r for r in step if r[2] != mainq]
gaps = []
prev=None
for r in main:
    if prev and r[0] > prev[1]: gaps.append((r[0]-prev[1], prev, r))
    if not prev or r[1] > prev[1]: prev = r
gaps.sort(reverse=True)
print("forward: %d kernels, %d gaps, %.2f ms" % (len(main), len(gaps), sum(g[0] for g in gaps)/1e6))
for g,a,b in gaps[:14]:
    s=[x[3][:30] for x in side if x[0] < b[0] and x[1] > a[1]]
    print("%8.1f us after %-44s before %-44s | side: %s" % (g/1e3, a[3][:44], b[3][:44], '; '.join(s[:2])))
print("first kernels of the step:")
for r in rows[i0-3:i0+8]: print("  %s q%s %.1f us  start +%.1f" % (r[3][:50], r[2], (r[1]-r[0])/1e3, (r[0]-rows[i0][0])/1e3))
