"""Does a weight prefetch into L2 DURING THE PREVIOUS KERNEL make the one-round GEMMs of the step start warm?  (diagnostics build)
A level-2 self-attention forward (the kernel before the out-projection) followed by NT 4096 x 1280 x 1280 on pl_kernel with a DIFFERENT weight
every iteration (128 copies: 420 MB in rotation, none of it in L2 / MALL), with and without sdxl_op_pl_prefetch_b on a second stream issued
before the attention launch.  Run under rocprofv3 --kernel-trace for pl_kernel's duration:
    SDXL_DIAG=1 python profiles/tools/l2_prefetch_bench.py <mode: 0 none | 1 prefetch | 2 same weight every time (warm)> [parts] [K]"""
import ctypes as C, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd  # noqa
from sdxl_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
mode = int(sys.argv[1]); parts = int(sys.argv[2]) if len(sys.argv) > 2 else 16; K = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
p = lambda t: C.c_void_p(t.data_ptr())
r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
B, H, N = 4, 20, 1024; Cc = H * 64; M = B * N
qkv = r(B, N, 3 * Cc); o = torch.empty(B, N, Cc, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, N, device=dev)
NW = 128 if K <= 1280 else 48
W = [r(1280, K) for _ in range(NW)]
x = r(M, K) if K != Cc else None
y = torch.empty(M, 1280, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()
def it(i):
    w = W[0 if mode == 2 else i % NW]
    if mode == 1: lib.check(L.sdxl_op_pl_prefetch_b(0, p(w), M, 1280, K, K, parts, C.c_void_p(side.cuda_stream)))
    lib.check(L.sdxl_op_attention_fwd(p(qkv), p(qkv[..., Cc:]), p(qkv[..., 2 * Cc:]), p(o), p(lse), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc, None))
    a = o.view(M, Cc) if x is None else x
    lib.check(L.sdxl_op_gemm_ld(0, p(a), p(w), p(y), M, 1280, K, K, K, 1280, None, None, 0, 7, None))
for i in range(8): it(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 2 * NW
e0.record()
for i in range(n): it(i)
e1.record(); torch.cuda.synchronize()
print("mode %d parts %d K %d: %.1f us per (attention + GEMM)" % (mode, parts, K, 1e3 * e0.elapsed_time(e1) / n))
