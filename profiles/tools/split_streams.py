#!/usr/bin/env python3
"""Does splitting a dependent chain of small GEMMs by batch halves onto two streams raise throughput?
chain of L launches of NT [M x N x K] on one stream  vs  two chains of [M/2 x N x K] on two streams (same total work)."""
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


def run(form, M, N, K, nsplit, chain=24, reps=5):
    a = r(M, K) if form != 2 else r(K, M)
    b = r(N, K) if form == 0 else r(K, N)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    Ms = M // nsplit
    args = []
    for i, s in enumerate(streams):
        args.append((form, a.data_ptr() + i * Ms * K * 2, b.data_ptr(), o.data_ptr() + i * Ms * N * 2, Ms, N, K, None, None, 0, 1, C.c_void_p(s.cuda_stream)))

    def go():
        for _ in range(chain):
            for ar in args:
                L.sdxl_op_gemm(*ar)

    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0.record()
        for s in streams:
            s.wait_event(e0)
        go()
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            torch.cuda.current_stream().wait_event(e)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    us = best / chain * 1e3
    return us, 2.0 * M * N * K / us / 1e6


for form, M, N, K in [(0, 4096, 1280, 1280), (0, 4096, 1280, 5120), (1, 4096, 1280, 1280), (1, 4096, 1280, 10240), (0, 4096, 3840, 1280), (0, 4096, 10240, 1280),
                      (1, 4096, 5120, 1280), (0, 16384, 640, 640)]:
    row = []
    for ns in (1, 2, 4):
        us, tf = run(form, M, N, K, ns)
        row.append(f"{ns} stream(s): {us:7.1f} us/layer {tf:7.1f} TF")
    print(f"{'NT NN TN'.split()[form]} {M}x{N}x{K}: " + " | ".join(row), flush=True)
