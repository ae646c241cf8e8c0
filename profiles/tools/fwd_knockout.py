#!/usr/bin/env python3
"""Component knock-outs of the forward's one-round kernels (gemm.hip built with -DSDXL_GEMM_DIAG: profiles/tools/build_diag_cr.sh 256+bits;
bits 1 no MFMA, 2 no main-loop DMA, 4 no fragment reads): NT 4096 x 1280 x K under configuration 3 (8 waves, 4-deep ring), 23 (the same
tile as two staggered split-K groups) and 13 (4 waves, 2-deep).   python profiles/tools/fwd_knockout.py --lib profiles/tools/lib_cr_d261.so --tag dma_only"""
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
L = lib.load()
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
for cfg in (3, 23, 13, 43):
    lib.check(L.sdxl_set_gemm_mode(4 * cfg))
    for form, M, N, K in ((0, 4096, 1280, 5120), (0, 4096, 1280, 1280)):
        a, b = r(M, K), r(N, K)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None, None, 0, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        steps = K // 64
        print(f"{tag:12s} cfg {cfg:2d} NT {M}x{N}x{K}: {us:7.1f} us  ({steps} K-steps of 36.9 KB: {us / steps:5.3f} us each incl. fixed cost)", flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
