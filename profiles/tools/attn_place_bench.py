"""Self-attention forward / backward with CONTROLLED buffer placement: every tensor is carved out of one arena at offsets that are multiples of 2 MiB
plus a per-tensor stagger (k * STAGGER bytes), so that two kernels (SDXL_KNOBS=...) are compared on identical addresses and the dependence of the
timings on the placement itself (channel aliasing of per-sample offsets: profiles/r06k_attn_stride_alias.txt) is visible instead of hidden in allocator luck.
    python profiles/tools/attn_place_bench.py [--iters N]        (SDXL_DIAG=1 SDXL_KNOBS=35=2,... as attn_bench.py)"""
import sys, os, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd
from sdxl_amd import lib
L = lib.load(); dev = torch.device('cuda:0')
for kv in os.environ.get('SDXL_KNOBS', '').split(','):
    if kv: lib.check(L.sdxl_set_knob(int(kv.split('=')[0]), int(kv.split('=')[1])))
ITERS = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
arena = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
def bench(fn, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
torch.manual_seed(0)
av = arena.view(torch.bfloat16)
for i in range(0, av.numel(), 1 << 26):
    av[i:i + (1 << 26)] = torch.randn(min(1 << 26, av.numel() - i), device=dev).to(torch.bfloat16)      # N(0, 1) everywhere (LSE / Delta are overwritten by the kernels)
for (B, H, N) in [(4, 10, 4096), (4, 20, 1024)]:
    Cc = H * 64
    sizes = {'qkv': B * N * 3 * Cc * 2, 'dqkv': B * N * 3 * Cc * 2, 'o': B * N * Cc * 2, 'do': B * N * Cc * 2, 'lse': B * H * N * 4, 'delta': B * H * N * 4}
    for stagger in (0, 4096 + 256, 65536 + 4096 + 256, 1 << 20):
        off, p = 0, {}
        for i, (k, sz) in enumerate(sizes.items()):
            p[k] = base + off + i * stagger
            off += (sz + i * stagger + (1 << 21) - 1) & ~((1 << 21) - 1)
        q, k_, v = p['qkv'], p['qkv'] + 2 * Cc, p['qkv'] + 4 * Cc
        dq, dk, dv = p['dqkv'], p['dqkv'] + 2 * Cc, p['dqkv'] + 4 * Cc
        cp = C.c_void_p
        lse_t = torch.empty(0)      # LSE must be valid for the backward: one forward first
        fwd = lambda: L.sdxl_op_attention_fwd(cp(q), cp(k_), cp(v), cp(p['o']), cp(p['lse']), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc, None)
        bwd = lambda: L.sdxl_op_attention_bwd(cp(q), cp(k_), cp(v), cp(p['o']), cp(p['do']), cp(p['lse']), cp(p['delta']), cp(dq), cp(dk), cp(dv), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc, None)
        fl = 4.0 * B * H * N * N * 64
        tf = bench(fwd); tb = bench(bwd)
        print(f"B{B} H{H} N{N:5d} stagger {stagger:8d}: fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF/s | bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF/s", flush=True)
