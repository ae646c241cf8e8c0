#!/usr/bin/env python3
"""GEMM microbenchmark on the model's shapes: sdxl_op_gemm (128-row kernel vs the 256 x 256 / 8-phase kernel) next to
torch.matmul (hipBLASLt) as a yardstick, 20 back-to-back launches on random data, plus a correctness check of each.

    python profiles/tools/gemm_bench.py [--quick]
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
if "--lib" in sys.argv:       # diagnostics: a library built with -DG256_DIAG=<bits> (knock-outs), same C ABI
    lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
L = lib.load()


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
FORMS = {"NT": 0, "NN": 1, "TN": 2}
SHAPES = [("NT", 4096, 1280, 1280, 1), ("NT", 4096, 3840, 1280, 1), ("NT", 4096, 10240, 1280, 1), ("NT", 4096, 1280, 5120, 1),
          ("NT", 16384, 5120, 640, 1), ("NT", 16384, 1920, 640, 1), ("NT", 8192, 8192, 8192, 1), ("NT", 4096, 4096, 4096, 1),
          ("NN", 4096, 1280, 1280, 1), ("NN", 4096, 3840, 1280, 1), ("NN", 4096, 5120, 1280, 1), ("NN", 4096, 1280, 10240, 1),
          ("NN", 4096, 1280, 3840, 1), ("NN", 16384, 2560, 640, 1), ("NN", 8192, 8192, 8192, 1),
          ("TN", 1280, 1280, 4096, 3), ("TN", 1280, 1280, 4096, 8), ("TN", 10240, 1280, 4096, 1), ("TN", 3840, 1280, 4096, 3),
          ("TN", 1280, 5120, 4096, 2), ("TN", 5120, 640, 16384, 1), ("TN", 2560, 2560, 16384, 2), ("TN", 8192, 8192, 8192, 1)]
if "--quick" in sys.argv:
    SHAPES = SHAPES[:4] + SHAPES[8:10] + SHAPES[15:18]

print(f"{'shape':34s} {'hipBLASLt':>10s} {'k128':>10s} {'k128/23':>10s} {'k256':>10s}   (TFLOP/s; k128/23 = split-K wave groups; err = max|d|/max|ref| of the last one vs hipBLASLt)")
for form, M, N, K, sk in SHAPES:
    fl = 2.0 * M * N * K
    if form == "NT":
        a, b = r(M, K), r(N, K)
        ref_fn = lambda o: torch.matmul(a, b.t(), out=o)
    elif form == "NN":
        a, b = r(M, K), r(K, N)
        ref_fn = lambda o: torch.matmul(a, b, out=o)
    else:
        a, b = r(K, M), r(K, N)
        ref_fn = lambda o: torch.matmul(a.t(), b, out=o)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_blas = bench(lambda: ref_fn(ob))
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if form == "TN" else torch.bfloat16)
    res = {}
    args = (FORMS[form], a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, None, None, 0, sk, st())   # built once: the
    fn = lambda: L.sdxl_op_gemm(*args)                                                                    # launch loop must not be host-bound
    for mode in (0, 92, 2):
        lib.check(L.sdxl_set_gemm_mode(mode))
        lib.check(fn())
        res[mode] = bench(fn)
    ref = ob.float()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"{form} {M}x{N}x{K} sk{sk:<3d}".ljust(34) + f" {fl / t_blas / 1e9:10.1f} {fl / res[0] / 1e9:10.1f} {fl / res[92] / 1e9:10.1f} {fl / res[2] / 1e9:10.1f}   err {err:.2e}",
          flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
