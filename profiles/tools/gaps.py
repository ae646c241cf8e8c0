"""The largest idle gaps of the caller's stream in the backward of the last traced step, with the kernels around them and what the
side stream was running meanwhile: python profiles/tools/gaps.py <kernel_trace.csv> [n]"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), str(r['Queue_Id']), r['Kernel_Name'][:70].replace(',', ';')))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mainq = collections.Counter(r[2] for r in rows if 'loss_prepare' in r[3]).most_common(1)[0][0]
rows.sort()
i0 = [i for i, r in enumerate(rows) if 'loss_prepare' in r[3]][-1]
ib = [i for i, r in enumerate(rows) if 'loss_bwd' in r[3]][-1]
step = rows[ib:]
main = [r for r in step if r[2] == mainq]
side = [r for r in step if r[2] != mainq]
gaps = []
for a, b in zip(main, main[1:]):
    if b[0] > a[1]:
        gaps.append((b[0] - a[1], a, b))
tot = sum(g[0] for g in gaps)
print(f"backward: {len(main)} main-stream kernels, {len(gaps)} gaps, {tot / 1e6:.2f} ms idle")
hist = collections.Counter()
for g, _, _ in gaps:
    hist['<3us' if g < 3000 else '3-6us' if g < 6000 else '6-12us' if g < 12000 else '12-30us' if g < 30000 else '30-100us' if g < 100000 else '>100us'] += g
for k in ['<3us', '3-6us', '6-12us', '12-30us', '30-100us', '>100us']:
    print(f"  gaps {k:9s}: {hist[k] / 1e6:6.2f} ms")
# by (kernel before -> kernel after)
pair = collections.Counter(); pc = collections.Counter()
for g, a, b in gaps:
    k = a[3][:34] + ' -> ' + b[3][:34]
    pair[k] += g; pc[k] += 1
print("by transition:")
for k, v in pair.most_common(14):
    print(f"  {v / 1e6:6.2f} ms {pc[k]:4d}x avg {v / pc[k] / 1e3:6.1f} us  {k}")
gaps.sort(reverse=True)
print("largest:")
for g, a, b in gaps[:N]:
    s = [x[3][:40] for x in side if x[0] < b[0] and x[1] > a[1]]
    print(f"  {g / 1e3:7.1f} us after {a[3][:40]:40s} before {b[3][:40]:40s} | side: {'; '.join(s[:3])}")
