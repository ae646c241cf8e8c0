"""Per-phase s_memtime sums of the attention forward tile loop (library built with tools/build_diag_attn.sh 128):
python profiles/tools/attn_stamps.py --lib profiles/tools/lib_attn_d128.so [B H N]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

lib.LIB_PATH = Path(sys.argv[sys.argv.index("--lib") + 1]).resolve()
L = lib.load()
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B, H, N = nums[:3] if len(nums) >= 3 else (4, 10, 4096)
dev = torch.device("cuda:0")
Cc = H * 64
qkv = torch.randn(B, N, 3 * Cc, device=dev).bfloat16()
o = torch.empty(B, N, Cc, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * H, N, dtype=torch.float32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
for _ in range(3):
    lib.check(L.sdxl_op_attention_fwd(p(q), p(k), p(v), p(o), p(lse), B, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc, None))
buf = (C.c_ulonglong * 96)()
L.sdxl_debug_attn_stamps.argtypes = [C.c_void_p]
lib.check(L.sdxl_debug_attn_stamps(buf))
names = ["wait own DMA (vmcnt 0)", "barrier", "K reads + S MFMAs + V^T reads issued", "max / exp / sums / cvt issued", "(unused)", "PV MFMAs issued"]
ntiles = N // 64
tot = [0.0] * 6
for wg in range(4):
    for w in range(4):
        for i in range(6):
            tot[i] += buf[(wg * 4 + w) * 6 + i] / 16.0
rt = tot[4]
tot[4] = 0.0
s = sum(tot)
print(f"whole tile loop: {s:.0f} s_memtime ticks in {rt * 10:.0f} ns (s_memrealtime, 100 MHz) -> {s / (rt * 10):.2f} ticks per ns")
print(f"B{B} H{H} N{N}: per wave and tile (mean over 16 waves, s_memtime ticks = 100 MHz? -> relative shares), total {s / ntiles:.1f} ticks/tile")
for i in range(6):
    if i != 4:
        print(f"  {names[i]:42s} {tot[i] / ntiles:8.1f}  {100 * tot[i] / s:5.1f} %")
