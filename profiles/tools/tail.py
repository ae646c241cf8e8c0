"""End of the backward: when does each stream finish, and what runs in the last milliseconds?  python profiles/tools/tail.py <kernel_trace.csv> [ms]"""
import collections, csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), str(r['Queue_Id']), r['Kernel_Name'][:50].replace(',', ';')) for r in csv.DictReader(open(sys.argv[1]))]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
mainq = collections.Counter(r[2] for r in rows if 'loss_prepare' in r[3]).most_common(1)[0][0]
rows.sort()
ib = [i for i, r in enumerate(rows) if 'loss_bwd' in r[3]][-1]
bw = rows[ib:]
t1 = max(r[1] for r in bw)
main = [r for r in bw if r[2] == mainq]; side = [r for r in bw if r[2] != mainq]
print(f"backward {(t1 - bw[0][0]) / 1e6:.2f} ms; main stream's last kernel ends {(t1 - max(r[1] for r in main)) / 1e3:.1f} us before the end, side stream's {(t1 - max(r[1] for r in side)) / 1e3:.1f} us")
for ms in range(int(win), 0, -1):
    a, b = t1 - ms * 1e6, t1 - (ms - 1) * 1e6
    def busy(rs):
        return sum(max(0, min(r[1], b) - max(r[0], a)) for r in rs) / 1e6
    top = collections.Counter()
    for r in bw:
        o = max(0, min(r[1], b) - max(r[0], a))
        if o > 0: top[('M:' if r[2] == mainq else 'S:') + r[3][:40]] += o
    print(f"  t_end-{ms:2d}ms .. -{ms - 1:2d}ms: main busy {busy(main):.2f}  side busy {busy(side):.2f}   " + " | ".join(f"{k} {v / 1e6:.2f}" for k, v in top.most_common(3)))
