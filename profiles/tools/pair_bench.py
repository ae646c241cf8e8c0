#!/usr/bin/env python3
"""Two-stream GEMM pair microbenchmark: a level-2 (4096-token) transformer layer's dgrad (NN, stream 1) and weight gradient (TN,
stream 2) launched side by side, as the backward of the step runs them -- per pair the time of `iters` concurrent launches under
each kernel configuration (sdxl_set_gemm_mode), next to each launch alone.

    python profiles/tools/pair_bench.py [--modes 1,124,128] [--iters 20]
"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

dev = torch.device("cuda:0")
if "--lib" in sys.argv:       # diagnostics: a knock-out library (profiles/tools/build_diag_cr.sh), same C ABI
    lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
L = lib.load()
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
# mode pairs "dgrad:wgrad" (sdxl_set_gemm_mode values: 1 policy, 4*31 = 124 (256x160), 4*32 = 128 (256x128), 52 = cfg 13)
MODES = [tuple(int(v) for v in x.split(":")) if ":" in x else (int(x), int(x)) for x in arg("--modes", "1,124,128,1:128,1:124").split(",")]
ITERS = int(arg("--iters", "20"))
r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)

# (name, dgrad NN (M, N, K), wgrad TN (M, N, K, splitk per mode family: [policy, cr]))
PAIRS = [("FF1  dU.W1 | dU^T.X ", (4096, 1280, 10240), (10240, 1280, 4096)),
         ("FF2  dY.W2 | dY^T.G ", (4096, 5120, 1280), (1280, 5120, 4096)),
         ("QKV  dQKV.W | dQKV^T.X", (4096, 1280, 3840), (3840, 1280, 4096)),
         ("OUT  dY.Wo | dY^T.A  ", (4096, 1280, 1280), (1280, 1280, 4096))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn_many = [fn] * iters
    for f in fn_many:
        f()
    s1.synchronize(); s2.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'pair':24s} {'mode':>7s} {'dgrad us':>9s} {'wgrad us':>9s} {'sum':>8s} {'pair us':>8s} {'pair TF/s':>9s}")
for name, (M, N, K), (Mw, Nw, Kw) in PAIRS:
    a, w = r(M, K), r(K, N)
    dx = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dy, x = r(Kw, Mw), r(Kw, Nw)
    dw = torch.zeros(Mw, Nw, device=dev, dtype=torch.float32)
    fl = 2.0 * M * N * K + 2.0 * Mw * Nw * Kw
    grp = Mw == 1280 and Nw == 1280        # the 1280 x 1280 weight gradients go out three at a time (GemmP::group)
    if grp:
        dys, xs, dws = [r(Kw, Mw) for _ in range(3)], [r(Kw, Nw) for _ in range(3)], [torch.zeros(Mw, Nw, device=dev, dtype=torch.float32) for _ in range(3)]
        arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
        ga = (3, arr(dys), arr(xs), arr(dws), None, Mw, Nw, Kw, 0, C.c_void_p(s2.cuda_stream))
        fl = 2.0 * M * N * K * 3 + 2.0 * Mw * Nw * Kw * 3
    for m1, m2 in MODES:
        mode = f"{m1}:{m2}"
        sk = 1 if m2 != 1 else 0          # policy: the plan's split-K; forced kernels: whole reductions
        a1 = (1, a.data_ptr(), w.data_ptr(), dx.data_ptr(), M, N, K, None, None, 0, 1, C.c_void_p(s1.cuda_stream))
        a2 = (2, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Mw, Nw, Kw, None, None, 0, sk, C.c_void_p(s2.cuda_stream))
        def f1():
            L.sdxl_set_gemm_mode(m1)
            rc = L.sdxl_op_gemm(*a1)
            if grp:
                L.sdxl_op_gemm(*a1); L.sdxl_op_gemm(*a1)
            return rc
        def f2():
            L.sdxl_set_gemm_mode(m2)
            return L.sdxl_op_wgrad_group(*ga) if grp else L.sdxl_op_gemm(*a2)
        lib.check(f1()); lib.check(f2())
        torch.cuda.synchronize()

        def alone(f, s):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(ITERS):
                f()
            e1.record(s)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / ITERS * 1e3

        t1, t2 = alone(f1, s1), alone(f2, s2)
        # the pair: both streams fed alternately, wall time from a common start to both streams' ends
        torch.cuda.synchronize()
        st0 = torch.cuda.Event(enable_timing=True); en1 = torch.cuda.Event(enable_timing=True); en2 = torch.cuda.Event(enable_timing=True)
        st0.record(); s1.wait_event(st0); s2.wait_event(st0)
        for _ in range(ITERS):
            f1(); f2()
        en1.record(s1); en2.record(s2)
        torch.cuda.synchronize()
        tp = max(st0.elapsed_time(en1), st0.elapsed_time(en2)) / ITERS * 1e3
        print(f"{name:24s} {mode:>7s} {t1:9.1f} {t2:9.1f} {t1 + t2:8.1f} {tp:8.1f} {fl / tp / 1e6:9.1f}", flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
