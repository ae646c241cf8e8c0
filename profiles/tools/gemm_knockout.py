#!/usr/bin/env python3
"""Time the 256 x 256 GEMM kernel of one library build (product or a -DG256_DIAG knock-out) on a few shapes.
    python profiles/tools/gemm_knockout.py [--lib path/to/lib.so]"""
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
L = lib.load()
dev = torch.device("cuda:0")
lib.check(L.sdxl_set_gemm_mode(2))
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
out = []
for form, M, N, K in [(0, 8192, 8192, 8192), (0, 4096, 3840, 1280), (0, 4096, 10240, 1280), (0, 16384, 5120, 640), (1, 4096, 3840, 1280),
                      (2, 10240, 1280, 4096)]:
    a = r(M, K) if form != 2 else r(K, M)
    b = r(N, K) if form == 0 else r(K, N)
    o = torch.empty(M, N, device=dev, dtype=torch.float32 if form == 2 else torch.bfloat16)
    fn = lambda: lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None, None, 0, 1,
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
    out.append(f"{'NT NN TN'.split()[form]} {M}x{N}x{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF")
print(" | ".join(out))
