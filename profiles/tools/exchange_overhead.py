"""What the per-segment gradient exchange costs the step besides the collective itself (one GPU, no communication):
(a) no per-segment callback (N = 1 path), (b) casts on the caller's stream + per-segment joins (round-1 N > 1 path),
(c) casts on the engine's side stream (join mode 2), (d) bf16 straight from the wgrad epilogues + small-range casts on the side stream.  python profiles/tools/exchange_overhead.py"""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib, synth  # noqa: E402
from sdxl_amd import unet as NU  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["ddpm_b4_1024"]
net = NU.NativeUNet(NU.make_config(), device=0)
synth.load_synthetic(net, seed=0)
net.plan(wl["B"], wl["H"], wl["W"], 77)
b = bench.make_batch(wl, 0, dev)
comm = torch.empty(net.param_elems, dtype=torch.bfloat16, device=dev)


def cast(k, off, n):
    lib.check(net.L.sdxl_grads_to_bf16(net.h, off, n, C.c_void_p(comm[off:off + n].data_ptr()), 0.125,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def cast_small(k, off, n):
    net.cast_small(off, n, comm[off:off + n], 0.125)


def run(mode, steps=10, warm=3):
    net.set_grad_emit(comm if mode == "d" else None, 0.125)
    def step():
        net.zero_grads()
        net.forward_loss(wl["method"], b["lat"], b["noise"], b["sigma_or_t"], b["timestep"], b["ehs"], b["pooled"], b["tid"])
        if mode == "a":
            net.backward(1.0, True)
        elif mode == "d":
            net.backward(1.0, True, on_segment=cast_small, segment_stream=True)
        else:
            net.backward(1.0, True, on_segment=cast, segment_stream=(mode == "c"))
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for m in ("a", "b", "c", "d", "a", "b", "c", "d"):
    print(m, f"{run(m):.2f} ms/step")
