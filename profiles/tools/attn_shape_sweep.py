"""Self-attention forward / backward over a list of shapes (B, H, N): python profiles/tools/attn_shape_sweep.py [--iters N]   (SDXL_DIAG=1 SDXL_KNOBS=35=2,... as attn_bench.py)"""
import sys, os, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd
from sdxl_amd import lib
L = lib.load(); dev = torch.device('cuda:0')
for kv in os.environ.get('SDXL_KNOBS', '').split(','):
    if kv: lib.check(L.sdxl_set_knob(int(kv.split('=')[0]), int(kv.split('=')[1])))
ptr = lambda t: C.c_void_p(t.data_ptr())
ITERS = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
def bench(fn, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, N) in [(4, 10, 4096), (4, 20, 1024), (4, 20, 1280), (4, 16, 1280), (4, 10, 4032), (4, 20, 1008), (4, 10, 5120), (2, 10, 8192)]:
    Cc = H * 64
    qkv = torch.randn(B, N, 3 * Cc, device=dev).to(torch.bfloat16); q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
    dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:]
    # (O / dO in tensors of Cc + 64 columns: separately allocated [B][N][Cc] buffers put the per-sample offsets of QKV and O on the same channels --
    #  profiles/r06k_attn_stride_alias.txt; the step's arena does not: its level-1 kernels run at the padded rate)
    ob = torch.empty(B, N, Cc + 64, dtype=torch.bfloat16, device=dev); o = ob[..., :Cc]; lse = torch.empty(B * H, N, dtype=torch.float32, device=dev)
    dob = torch.randn(B, N, Cc + 64, device=dev).to(torch.bfloat16); do = dob[..., :Cc]; delta = torch.empty_like(lse)
    fl = 4.0 * B * H * N * N * 64
    tf = bench(lambda: L.sdxl_op_attention_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc + 64, None))
    tb = bench(lambda: L.sdxl_op_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc + 64, None))
    print(f"B{B} H{H} N{N:5d}: fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF/s | bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF/s", flush=True)
