#!/usr/bin/env python3
"""Leading-dimension sensitivity of the GEMM kernels: the same problem with K (= the row stride of the K-contiguous operands) or M / N
(= the row stride of the N-contiguous ones) padded by 32 / 64 elements.  python profiles/tools/ld_sweep.py [--modes 1,52,128]"""
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
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
MODES = [int(x) for x in arg("--modes", "1,52,128").split(",")]
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
SH = [(1, 4096, 1280, 10240), (1, 4096, 1280, 10304), (1, 4096, 1280, 5120), (1, 4096, 1280, 5184), (1, 4096, 1280, 3840), (1, 4096, 1280, 1280), (1, 4096, 1280, 1344),
      (0, 4096, 1280, 5120), (0, 4096, 1280, 5184), (0, 4096, 1280, 1280), (0, 4096, 1280, 1344),
      (2, 10240, 1280, 4096), (2, 10304, 1280, 4096), (2, 10240, 1344, 4096), (2, 1280, 5120, 4096), (2, 1280, 5184, 4096), (2, 1344, 5184, 4096),
      (1, 4096, 5120, 1280), (1, 4096, 5184, 1280), (2, 3840, 1280, 4096), (2, 3904, 1344, 4096)]
for form, M, N, K in SH:
    a = r(M, K) if form != 2 else r(K, M)
    b = r(N, K) if form == 0 else r(K, N)
    o = torch.empty(M, N, device=dev, dtype=torch.float32 if form == 2 else torch.bfloat16)
    row = f"{'NT NN TN'.split()[form]} {M}x{N}x{K}".ljust(24)
    for mode in MODES:
        lib.check(L.sdxl_set_gemm_mode(mode))
        fn = lambda: lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None, None, 0, 0 if (form == 2 and mode == 1) else 1,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row += f"  mode {mode:3d}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
    print(row, flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
