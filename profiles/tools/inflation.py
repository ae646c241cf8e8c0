"""Which main-stream kernels of the backward pay for the co-running side stream, and beside what?
    python profiles/tools/inflation.py <overlapped kernel_trace.csv> <serialized kernel_trace.csv>
Both traces = the last step of `bench.py --steps 2 --warmup 1` (the second with SDXL_NO_SIDE_STREAM=1).  Main-stream launches of
the backward are matched between the two by (kernel, grid) in launch order; per kernel type: launches, serialized ms, overlapped
ms, ratio, and the side-stream kernel types that overlapped them most (share of the overlapped time)."""
import collections, csv, sys


def load(path):
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), str(r['Queue_Id']),
             r['Kernel_Name'][:52].replace(',', ';'), (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])) for r in csv.DictReader(open(path))]
    mainq = collections.Counter(r[2] for r in rows if 'loss_prepare' in r[3]).most_common(1)[0][0]
    rows.sort()
    ib = [i for i, r in enumerate(rows) if 'loss_bwd' in r[3]][-1]
    bw = rows[ib:]
    return [r for r in bw if r[2] == mainq], [r for r in bw if r[2] != mainq]


ov_main, ov_side = load(sys.argv[1])
se_main, _ = load(sys.argv[2])
# serialized durations per (name, grid) in order -- the serialized trace holds main AND (former) side kernels on one queue
ser = collections.defaultdict(list)
for r in se_main:
    ser[(r[3], r[4])].append(r[1] - r[0])
idx = collections.Counter()
agg = collections.OrderedDict()
side_sorted = sorted(ov_side)
for r in ov_main:
    k = (r[3], r[4])
    lst = ser.get(k)
    if not lst:
        continue
    s = lst[min(idx[k], len(lst) - 1)]
    idx[k] += 1
    a = agg.setdefault(r[3][:46], [0, 0.0, 0.0, collections.Counter()])
    a[0] += 1
    a[1] += s
    a[2] += r[1] - r[0]
    for q in side_sorted:
        if q[1] <= r[0]:
            continue
        if q[0] >= r[1]:
            break
        a[3][q[3][:34]] += min(q[1], r[1]) - max(q[0], r[0])
print(f"{'main-stream kernel (backward)':48s} {'n':>5s} {'serial ms':>10s} {'overl. ms':>10s} {'ratio':>6s}   beside (share of its overlapped time)")
for k, (n, s, o, c) in sorted(agg.items(), key=lambda kv: -(kv[1][2] - kv[1][1])):
    if o < 0.2e6:
        continue
    top = ", ".join(f"{kk} {100 * v / o:.0f}%" for kk, v in c.most_common(2))
    print(f"{k:48s} {n:5d} {s / 1e6:10.2f} {o / 1e6:10.2f} {o / s:6.2f}   {top}")
