#!/usr/bin/env python3
"""Time the co-resident 256-row GEMM kernel of one library build (product or a -DSDXL_CR_DIAG knock-out) on the level-2 shapes.
    python profiles/tools/cr_knockout.py [--lib path/to/lib.so] [--tag text]
Prints us per launch, us per 32-deep K-step of a CU's share, and the staged KB/us per CU."""
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
SHAPES = [(1, 4096, 1280, 10240), (2, 10240, 1280, 4096), (1, 4096, 5120, 1280), (0, 4096, 1280, 5120), (1, 8192, 8192, 4096)]
if '--shapes' in sys.argv:
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in sys.argv[sys.argv.index('--shapes') + 1].split(',')]
for cfg, bn in ((31, 160), (32, 128)):
    lib.check(L.sdxl_set_gemm_mode(4 * cfg))
    for form, M, N, K in SHAPES:
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
        wgs = ((M + 255) // 256) * ((N + bn - 1) // bn)
        steps = wgs * (K // 32)
        per_cu = steps / min(wgs, 256)            # K-steps of the busiest CU's share (one workgroup per CU up to 256, then shared)
        if wgs > 256:
            per_cu = steps / 256
        kb = (256 + bn) * 32 * 2 / 1024
        print(f"{tag:10s} cr{bn} {'NT NN TN'.split()[form]} {M}x{N}x{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF  wgs {wgs:4d}  "
              f"{us / per_cu:6.3f} us/K-step/CU  {kb * per_cu / us:6.1f} KB/us/CU", flush=True)
lib.check(L.sdxl_set_gemm_mode(1))
