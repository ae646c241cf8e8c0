#!/usr/bin/env python3
"""The forward's linear shapes (NT, exclusive CUs) under the policy kernel, the 256 x 256 kernel and the co-resident 256-row tile
(lockstep / phased / 160 columns): python profiles/tools/fwd_cfg_sweep.py [--modes 1,2,128,140,124]"""
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
MODES = [int(x) for x in arg("--modes", "1,2,128,140,124").split(",")]
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
SH = [(0, 4096, 3840, 1280), (0, 4096, 10240, 1280), (0, 4096, 1280, 1280), (0, 4096, 1280, 5120),
      (0, 16384, 1920, 640), (0, 16384, 5120, 640), (0, 16384, 640, 640), (0, 16384, 640, 2560),
      (1, 4096, 1280, 3840), (1, 4096, 1280, 1280), (1, 4096, 5120, 1280), (1, 16384, 640, 1920), (1, 16384, 2560, 640)]
for form, M, N, K in SH:
    a = r(M, K)
    b = r(N, K) if form == 0 else r(K, N)
    bias = r(N)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = f"{'NT NN TN'.split()[form]} {M}x{N}x{K}".ljust(24)
    for mode in MODES:
        lib.check(L.sdxl_set_gemm_mode(mode))
        fn = lambda: lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, bias.data_ptr() if form == 0 else None, None, 0, 1,
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
