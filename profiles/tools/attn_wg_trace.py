"""Start / end time and placement of every workgroup of the attention forward (library built with tools/build_diag_attn.sh 256):
python profiles/tools/attn_wg_trace.py --lib profiles/tools/lib_attn_d256.so [B H N]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
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
buf = (C.c_ulonglong * (4096 * 3))()
L.sdxl_debug_attn_wg.argtypes = [C.c_void_p]
lib.check(L.sdxl_debug_attn_wg(buf))
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 3)
nwg = (N // 128) * B * H
a = a[:nwg]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
base = t0.min()
st = (t0 - base) * 0.01; en = (t1 - base) * 0.01          # us
print(f"{nwg} workgroups; kernel span {en.max():.1f} us; workgroup duration mean {np.mean(en - st):.1f} us, min {np.min(en - st):.1f}, max {np.max(en - st):.1f}")
xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
hw = a[:, 2].astype(np.int64) & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
print("distinct (xcc, se, cu):", len(set(key.tolist())), "; workgroups per XCC:", np.bincount(xcc).tolist())
for tt in np.linspace(0, en.max(), 9)[1:-1]:
    print(f"  running at {tt:6.1f} us: {int(((st <= tt) & (en > tt)).sum())}")
cnt = np.bincount(np.unique(key, return_inverse=True)[1])
print("workgroups per CU: min", cnt.min(), "max", cnt.max(), "mean", cnt.mean())
