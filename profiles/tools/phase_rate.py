"""MFMA rate over the backward / forward of one step, window by window: joins a rocprofv3 kernel trace with the launch log the
library writes under SDXL_LAUNCH_LOG (one line per GEMM / attention launch, in host launch order = Dispatch_Id order).
  python profiles/tools/phase_rate.py <kernel_trace.csv> <launch_log> [window_ms]
A kernel's flops are spread evenly over its duration; a window's rate = the flops that fall into it / its length."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
log = [l.strip().split(',') for l in open(sys.argv[2]) if l.strip()]
win = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
def family(n):
    if 'splitk' in n: return None
    if 'gemm_kernel' in n or 'gemm256_kernel' in n or 'gemm_sk_kernel' in n or 'conv_wgrad3_kernel' in n or 'wgrad256_kernel' in n: return 'G'
    if 'attn_fwd' in n: return 'A0'
    if 'attn_bwd_dq' in n: return 'A1'
    if 'attn_bwd_dkv' in n: return 'A2'
    return None
rows.sort(key=lambda r: int(r['Dispatch_Id']))
li = 0
ker = []        # (start, end, queue, name, flops, label)
for r in rows:
    f = family(r['Kernel_Name'])
    fl, lab = 0.0, r['Kernel_Name'][:28]
    if f is not None:
        while li < len(log) and not ((log[li][0] == 'G') == (f == 'G') and (f == 'G' or log[li][1] == f[1])): li += 1    # (defensive: kinds must alternate as logged)
        if li >= len(log): raise SystemExit("launch log shorter than the trace")
        rec = log[li]; li += 1
        if f == 'G':
            form, taps, M, N, K, sk, grp = map(int, rec[1:8])
            fl = 2.0 * M * N * K * taps * grp
            lab = f"{['NT','NN','TN'][form]}{taps if taps > 1 else ''} {M}x{N}x{K}" + (f" g{grp}" if grp > 1 else "")      # (taps 4 / 16: GemmP::up2)
        else:
            kind, B, H, Nq, Nk = map(int, rec[1:6])
            fl = 4.0 * B * H * Nq * Nk * 64 * {0: 1.0, 1: 1.5, 2: 2.0}[kind]      # executed: 2 / 3 / 4 products of Nq x Nk x 64
            lab = f"attn{kind} {Nq}x{Nk}"
    ker.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'], fl, lab))
mainq = collections.Counter(k[2] for k in ker if 'loss_prepare' in k[3]).most_common(1)[0][0]
ker.sort()
i0 = [i for i, k in enumerate(ker) if 'loss_prepare' in k[3]][-1]
step = ker[i0:]
t0 = step[0][0]; t1 = max(k[1] for k in step)
tb = [k for k in step if 'loss_bwd' in k[3]][-1][0]
print(f"step {(t1 - t0) / 1e6:.2f} ms, forward {(tb - t0) / 1e6:.2f}, backward {(t1 - tb) / 1e6:.2f}; executed {sum(k[4] for k in step) / 1e12:.2f} TFLOP "
      f"= {sum(k[4] for k in step) / (t1 - t0) / 1e3:.0f} TFLOP/s over the step")
w = int(win * 1e6)
a = t0
while a < t1:
    b = min(a + w, t1)
    fl = 0.0; top = collections.Counter(); mb = sb = 0
    for k in step:
        o = min(k[1], b) - max(k[0], a)
        if o <= 0: continue
        if k[2] == mainq: mb += o
        else: sb += o
        if k[4] > 0: fl += k[4] * o / max(1, k[1] - k[0])
        top[('M:' if k[2] == mainq else 'S:') + k[5]] += o
    ph = 'F' if a < tb else 'B'
    print(f"{ph} {(a - t0) / 1e6:6.1f} ms: {fl / (b - a) / 1e3:6.0f} TFLOP/s  main {mb / (b - a):.2f} side {sb / (b - a):.2f}  " + " | ".join(f"{k} {v / (b - a):.2f}" for k, v in top.most_common(4)))
    a = b

# per-problem table of the step as it ran (both streams overlapped): launches, total / mean duration, executed TFLOP/s
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k in step:
    if k[4] <= 0: continue
    key = ('F ' if k[0] < tb else 'B ') + ('M:' if k[2] == mainq else 'S:') + k[5]
    a = agg[key]; a[0] += 1; a[1] += k[1] - k[0]; a[2] += k[4]
print("\nproblem (phase stream form MxNxK)                   n    total ms   mean us   TFLOP/s")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{key:48s} {a[0]:4d} {a[1] / 1e6:9.2f} {a[1] / a[0] / 1e3:9.1f} {a[2] / a[1] / 1e3:8.0f}")
