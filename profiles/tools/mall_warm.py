#!/usr/bin/env python3
"""Does it help a cold-operand GEMM if its operands were read once (into the memory-side cache) just before?  NT 4096 x 1280 x 5120 on gemm_pl.hip (cfg 7):
operand sets rotate over 1.6 GB; variants: nothing / the NEXT launch's weights read by a streaming kernel before each launch / weights + activations."""
import ctypes as C, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa
from sdxl_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 4096, 1280, 5120
nset = 21
sets = []
for _ in range(nset):
    a, b, bias, res, o = r(M, K), r(N, K), r(N), r(M, N), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    sets.append((a, b, bias, res, o))
def gemm(i, cfg):
    a, b, bias, res, o = sets[i % nset]
    lib.check(L.sdxl_op_gemm_ld(0, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, K, K, N, bias.data_ptr(), res.data_ptr(), N, cfg, st()))
def touch(t):
    return t.view(torch.int32).sum()      # one streaming read
def run(mode, cfg, iters=4 * nset):
    for i in range(3): gemm(i, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        if mode >= 1: touch(sets[(i + 1) % nset][1])
        if mode >= 2: touch(sets[(i + 1) % nset][0])
        gemm(i, cfg)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / iters * 1e3
    # the touch kernels alone
    e0.record()
    for i in range(iters):
        if mode >= 1: touch(sets[(i + 1) % nset][1])
        if mode >= 2: touch(sets[(i + 1) % nset][0])
    e1.record(); torch.cuda.synchronize()
    return tot, e0.elapsed_time(e1) / iters * 1e3
for cfg in (7, 23):
    for mode, name in ((0, "cold"), (1, "next weights read first"), (2, "next weights + activations read first")):
        tot, t = run(mode, cfg)
        print(f"cfg {cfg:2d} {name:40s}: {tot:6.1f} us per iteration, touches alone {t:5.1f} -> GEMM {tot - t:6.1f} us", flush=True)
