"""LayerNorm backward microbenchmark (HBM roofline check): python profiles/tools/ln_bench.py
algorithmic bytes per row = 4 x C x 2 (x, dy, addend in; dx out) + 8 (stats)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

L = lib.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, Cc in ((4096, 1280), (16384, 640), (65536, 320)):
    x = torch.randn(M, Cc, device=dev).bfloat16()
    dy = torch.randn(M, Cc, device=dev).bfloat16()
    dx = torch.randn(M, Cc, device=dev).bfloat16()
    g = torch.randn(Cc, device=dev).bfloat16()
    stats = torch.stack([x.float().mean(1), x.float().var(1, unbiased=False).add(1e-5).rsqrt()], 1).contiguous()
    dg = torch.zeros(Cc, device=dev)
    db = torch.zeros(Cc, device=dev)
    y = torch.empty_like(x)
    for name, fn in (("fwd", lambda: L.sdxl_op_layernorm_fwd(p(x), p(y), p(g), p(g), p(stats), M, Cc, 1e-5, st)),
                     ("bwd dx only", lambda: L.sdxl_op_layernorm_bwd(p(x), p(dy), p(g), p(stats), p(dx), None, None, M, Cc, 1, st)),
                     ("bwd dx+params", lambda: L.sdxl_op_layernorm_bwd(p(x), p(dy), p(g), p(stats), p(dx), p(dg), p(db), M, Cc, 1, st))):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        byt = M * Cc * 2 * (2 if name == "fwd" else 4)
        print(f"M={M:6d} C={Cc:5d} {name:14s} {us:7.1f} us  {byt / us / 1e6:6.2f} TB/s")
