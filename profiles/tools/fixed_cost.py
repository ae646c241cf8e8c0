#!/usr/bin/env python3
"""Fixed cost of a one-round launch: NT / NN 4096 x 1280 x K for K = 64 .. 1280 (policy kernel), back to back on one stream; the intercept
of time over K-steps is what a launch pays besides its main loop (dispatch + first tile from HBM + epilogue + end-of-kernel write-back)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

L = lib.load()
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
M, N = 4096, 1280
for form in (0, 1):
    for K in (64, 128, 256, 512, 1280, 2560):
        a = r(M, K)
        b = r(N, K) if form == 0 else r(K, N)
        bias, res = r(N), r(M, N)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for variant in ("plain", "bias+resid"):
            bp = bias.data_ptr() if (variant != "plain" and form == 0) else None
            rp = res.data_ptr() if variant != "plain" else None
            fn = lambda: lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, bp, rp, 0, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            print(f"{'NT NN'.split()[form]} {M}x{N}x{K:5d} {variant:10s} {us:7.1f} us per launch ({K // 64} K-steps)", flush=True)
