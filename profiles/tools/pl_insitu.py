#!/usr/bin/env python3
"""Why is a one-round GEMM slower inside the step than in a 20-launch loop?  NT 4096 x 1280 x 5120 (+ bias + residual: the FF2 forward) and
NN 4096 x 1280 x 10240 under configurations 23 (lockstep, 8 waves), 7 (gemm_pl.hip) and 8 (the same with its L2 prefetch wave), varying ONE condition at a time:
  loop      20 back-to-back launches on one operand set (what pl_bench.py measures)
  sustained 3000 back-to-back launches (~150 ms: the package reaches its power cap, as in the step)
  cold      the operand set rotates over > 1.5 GB of buffers (nothing is in the L2 / memory-side cache when a launch starts)
  padded    lda = K + 64 (the plan pads the feed-forward hidden tensors)
  mixed     each launch is followed by an unrelated memory-bound kernel over 84 MB (as LayerNorm / GEGLU traffic between GEMMs does)
python profiles/tools/pl_insitu.py"""
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
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fns, iters):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def make(form, M, N, K, pad, cfg):
    lda = K + pad
    a = r(M, lda)
    b = r(N, K) if form == 0 else r(K, N)
    bias, res, o = r(N), r(M, N), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    args = (form, a.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, lda, K if form == 0 else N, N, bias.data_ptr(), res.data_ptr(), N, cfg, st())
    keep = (a, b, bias, res, o)
    return (lambda: L.sdxl_op_gemm_ld(*args)), keep


big = torch.empty(42 * 1024 * 1024, device=dev, dtype=torch.bfloat16)
for form, M, N, K in ((0, 4096, 1280, 5120), (1, 4096, 1280, 10240), (0, 4096, 1280, 1280)):
    name = f"{'NT' if form == 0 else 'NN'} {M}x{N}x{K}"
    for cfg in (23, 7, 8):
        f, keep = make(form, M, N, K, 0, cfg)
        lib.check(f())
        loop = timed([f], 20)
        sus = timed([f], 3000)
        nset = max(2, int(1.6e9 / (2.0 * (M * K + N * K + 2 * M * N))))
        sets = [make(form, M, N, K, 0, cfg) for _ in range(nset)]
        cold = timed([s[0] for s in sets], 4 * nset)
        del sets
        fp, keepp = make(form, M, N, K, 64, cfg)
        padded = timed([fp], 20)
        both = timed([f, lambda: big.add_(1.0)], 40) * 2 - timed([lambda: big.add_(1.0)], 20)
        print(f"{name} cfg {cfg:2d}: loop {loop:6.1f} us | sustained {sus:6.1f} | cold operands ({nset} sets) {cold:6.1f} | lda + 64 {padded:6.1f} | behind an 84 MB streaming kernel {both:6.1f}", flush=True)
