#!/usr/bin/env python3
"""Per-section s_memtime stamps of the 256x256 GEMM main loop (library built with -DG256_DIAG=16 or 24):
    python profiles/tools/g256_stamps.py --lib profiles/tools/lib_g256_d16.so [form M N K]"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
L = lib.load()
args = [a for a in sys.argv[1:] if a.lstrip("-").isdigit()]
form, M, N, K = (int(x) for x in args[:4]) if len(args) >= 4 else (0, 8192, 8192, 8192)
dev = torch.device("cuda:0")
lib.check(L.sdxl_set_gemm_mode(2))
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
a = r(M, K) if form != 2 else r(K, M)
b = r(N, K) if form == 0 else r(K, N)
o = torch.empty(M, N, device=dev, dtype=torch.float32 if form == 2 else torch.bfloat16)
for _ in range(3):
    lib.check(L.sdxl_op_gemm(form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None, None, 0, 1, None))
buf = (C.c_ulonglong * 256)()
L.sdxl_debug_g256_stamps.argtypes = [C.c_void_p]
lib.check(L.sdxl_debug_g256_stamps(buf))
import numpy as np
s = np.array(buf[:], dtype=np.int64).reshape(2, 4, 4, 8)      # [half][tile][phase][event]
t0 = s[0, 0, 0, 0]
names = ["reads done", "vmcnt+barrier X", "MFMA section", "barrier Y"]
print(f"form {form} {M}x{N}x{K}: cycles per section (s_memtime ticks), K-tiles 8..11 averaged; start offsets relative to wave 0, tile 8")
for h in range(2):
    print(f" wave {4 * h}: phase-start offsets of tile 8: {[int(s[h, 0, p, 0] - t0) for p in range(2)]}  K-tile period {int(s[h, 1, 0, 0] - s[h, 0, 0, 0])} {int(s[h, 2, 0, 0] - s[h, 1, 0, 0])} {int(s[h, 3, 0, 0] - s[h, 2, 0, 0])}")
    for p in range(2):
        d = [(s[h, :, p, e + 1] - s[h, :, p, e]).mean() for e in range(4)]
        print(f"   phase {'AB'[p]}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, d)))
