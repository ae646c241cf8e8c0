"""GroupNorm (+SiLU) forward / backward on the model's shapes at B = 4 (sdxl_op_groupnorm_fwd / _bwd through the C ABI); run under
rocprofv3 --kernel-trace --stats for the per-kernel split: python profiles/tools/gn_bench.py [--cold]
--cold: every launch works on a different set of tensors, > 600 MB in rotation (nothing it reads is left in L2 / MALL: the backward of the
step reads activations the forward stored tens of milliseconds earlier)."""
import ctypes as C, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd  # noqa
from sdxl_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
B, G = 4, 32
# (HW, C, count in the UNet): resnet norm1 / norm2 of the three levels, the decoder's concatenated inputs, conv_norm_out
SHAPES = [(16384, 320, 9), (4096, 320, 1), (4096, 640, 9), (1024, 640, 1), (1024, 1280, 15), (1024, 2560, 2), (1024, 1920, 1),
          (4096, 1920, 1), (4096, 1280, 2), (4096, 960, 1), (16384, 960, 1), (16384, 640, 2)]
tot = {"fwd": 0.0, "bwd": 0.0}
COLD = '--cold' in sys.argv
for HW, Cc, cnt in SHAPES:
    nset = max(1, -(-600_000_000 // (B * HW * Cc * 2 * 3))) if COLD else 1
    sets = []
    for _ in range(nset):
        x = torch.randn(B, HW, Cc, device=dev).bfloat16()
        sets.append((x, torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)))
    g = torch.ones(Cc, device=dev).bfloat16(); b = torch.zeros(Cc, device=dev).bfloat16()
    stats = torch.empty(B * G * 2, device=dev); ws = torch.empty(256 * B * Cc * 2 + 256 * B * G * 2 + B * Cc * 5, device=dev)
    dg = torch.zeros(Cc, device=dev); db = torch.zeros(Cc, device=dev)
    L.sdxl_op_groupnorm_fwd(p(sets[0][0]), p(sets[0][2]), p(g), p(b), p(stats), p(ws), B, HW, Cc, G, 1e-5, 1, None)
    def fwd(i):
        x, dy, y, dx = sets[i % nset]
        return L.sdxl_op_groupnorm_fwd(p(x), p(y), p(g), p(b), p(stats), p(ws), B, HW, Cc, G, 1e-5, 1, None)
    def bwd(i):
        x, dy, y, dx = sets[i % nset]
        return L.sdxl_op_groupnorm_bwd(p(x), p(dy), p(g), p(b), p(stats), p(dx), p(dg), p(db), p(ws), B, HW, Cc, G, 1, 0, None)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for i in range(3): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(30, 2 * nset)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        byt = B * HW * Cc * 2 * (3 if name == "fwd" else 5)      # fwd: x twice + y; bwd: x, dy twice + dx
        tot[name] += us * cnt
        print(f"HW={HW:6d} C={Cc:5d} x{cnt:2d} {name} {us:7.1f} us  {byt / us / 1e6:5.2f} TB/s of the passes' bytes", flush=True)
    del sets
print("%s, per step (counts above): fwd %.2f ms, bwd %.2f ms" % ("cold" if COLD else "warm", tot["fwd"] / 1e3, tot["bwd"] / 1e3))
