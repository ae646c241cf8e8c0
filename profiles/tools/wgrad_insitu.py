"""Weight gradients (TN, sdxl_op_gemm: the plan's routing) hot vs cold: 20 launches on one operand set against a rotation over > 1.2 GB of sets
(nothing in L2 / MALL at launch: as in the step, where dY was just written and X dates from the forward).  Run under rocprofv3 for kernel names.
    python profiles/tools/wgrad_insitu.py [gemm mode: 140 = the co-resident 256-row kernel's phased loop (configuration 35), as the plan routes the level-2 weight gradients]"""
import ctypes as C, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd  # noqa
from sdxl_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
if len(sys.argv) > 1: lib.check(L.sdxl_set_gemm_mode(int(sys.argv[1])))
def timed(fns, iters):
    for f in fns[:3]: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in ((10240, 1280, 4096), (3840, 1280, 4096), (1280, 5120, 4096), (5120, 640, 16384)):
    def make():
        a, b = r(K, M), r(K, N); c = torch.zeros(M, N, device=dev)
        return (lambda: L.sdxl_op_gemm(2, p(a), p(b), p(c), M, N, K, None, None, 0, 0, None)), (a, b, c)
    f, keep = make()
    hot = timed([f], 20)
    nset = max(2, int(1.3e9 / (2.0 * K * (M + N) + 4.0 * M * N)))
    sets = [make() for _ in range(nset)]
    cold = timed([s[0] for s in sets], 4 * nset)
    fl = 2.0 * M * N * K
    print(f"TN {M}x{N}x{K}: hot {hot:7.1f} us ({fl / hot / 1e6:6.0f} TF/s)   cold ({nset} sets) {cold:7.1f} us ({fl / cold / 1e6:6.0f} TF/s)", flush=True)
    del sets
