#!/usr/bin/env python3
"""time = F + T*k fit: sdxl_op_gemm at fixed M x N over a sweep of K (both kernels), to separate the fixed cost of a
launch (prologue, epilogue, launch gap) from the per-K-tile cost.   python profiles/tools/gemm_ksweep.py [form M N]"""
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
args = [int(a) for a in sys.argv[1:] if a.isdigit()]
form, M, N = args[:3] if len(args) >= 3 else (0, 4096, 3840)
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
print(f"form {form} M {M} N {N}: us per launch (20 back-to-back)")
for mode in (0, 92, 2):
    lib.check(L.sdxl_set_gemm_mode(mode))
    row = []
    for K in (64, 128, 256, 512, 1024, 1280, 2560, 5120):
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
        row.append(f"K={K}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
    print(" kernel " + {0: 'k128', 92: 'k128/23', 2: 'k256'}[mode] + ": " + "  ".join(row))
lib.check(L.sdxl_set_gemm_mode(1))
