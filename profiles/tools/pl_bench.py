#!/usr/bin/env python3
"""The forward's one-round problems under the lockstep configurations (3: 8 waves 4-deep, 23: split-K wave groups, 13: 4 waves 2-deep, 1: 128 x 128)
and the pipelined one-wave-per-SIMD loop (5: 128 x 160, 6: 128 x 128; gemm.hip PL): us per launch (20 back to back), TFLOP/s, and the slope per
64-deep K-step between two reduction lengths (fixed cost removed).   python profiles/tools/pl_bench.py [--lib <knock-out build>] [--tag name]"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

if "--lib" in sys.argv:
    lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "product"
cfgs = [int(c) for c in sys.argv[sys.argv.index("--cfgs") + 1].split(",")] if "--cfgs" in sys.argv else [3, 23, 13, 5, 7]
L = lib.load()
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, iters=20, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


SHAPES = [("NT", 4096, 1280, (1280, 5120)), ("NN", 4096, 1280, (1280, 5120)), ("NN", 4096, 1280, (5120, 10240)), ("NT", 16384, 640, (640, 2560))]
print(f"# {tag}: us per launch | TFLOP/s at the two reduction lengths, slope = us per 64-deep K-step")
for form, M, N, Ks in SHAPES:
    for cfg in cfgs:
        lib.check(L.sdxl_set_gemm_mode(4 * cfg))
        us = []
        for K in Ks:
            a = r(M, K)
            b = r(N, K) if form == "NT" else r(K, N)
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            bias, res = (r(N), r(M, N)) if "--epilogue" in sys.argv else (None, None)      # (--epilogue: + bias + residual, as the forward launches them)
            args = (0 if form == "NT" else 1, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, bias.data_ptr() if bias is not None else None,
                    res.data_ptr() if res is not None else None, 0, 1, st())
            fn = lambda: L.sdxl_op_gemm(*args)
            lib.check(fn())
            us.append(bench(fn))
        slope = (us[1] - us[0]) / ((Ks[1] - Ks[0]) / 64)
        tf = [2.0 * M * N * K / u / 1e6 for K, u in zip(Ks, us)]
        print(f"{tag:10s} cfg {cfg:2d} {form} {M}x{N}x{Ks[0]:5d}: {us[0]:7.1f} us {tf[0]:7.1f} TF | x{Ks[1]:5d}: {us[1]:7.1f} us {tf[1]:7.1f} TF | slope {slope:5.3f} us / K-step", flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
