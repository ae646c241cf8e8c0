#!/usr/bin/env python3
"""Would the forward gain from running two half batches on two streams (out of phase)?  A level-2 transformer block's forward
(LN, QKV, self-attention, out-proj, LN, q-proj, cross-attention, out-proj, LN, GEGLU FF1, FF2; distinct weights per block so that they
come from HBM as in the step) through the op-level C ABI: NB blocks at M = 4096 on one stream against 2 x M = 2048 on two streams,
the second started half a block late.   python profiles/tools/fwd_halves_bench.py [--blocks 8]"""
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
NB = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 8
Cc, H, N, B = 1280, 20, 1024, 4
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

W = []
for _ in range(NB):
    W.append(dict(g1=r(Cc) + 1, b1=r(Cc), wqkv=r(3 * Cc, Cc, sc=Cc ** -0.5), wo1=r(Cc, Cc, sc=Cc ** -0.5), bo1=r(Cc),
                  g2=r(Cc) + 1, b2=r(Cc), wq=r(Cc, Cc, sc=Cc ** -0.5), wo2=r(Cc, Cc, sc=Cc ** -0.5), bo2=r(Cc),
                  g3=r(Cc) + 1, b3=r(Cc), wf1=r(8 * Cc, Cc, sc=Cc ** -0.5), bf1=r(8 * Cc), wf2=r(Cc, 4 * Cc, sc=(4 * Cc) ** -0.5), bf2=r(Cc)))


class Half:
    def __init__(self, Bh):
        self.Bh, self.M = Bh, Bh * N
        M = self.M
        self.x = r(M, Cc)
        self.ln = torch.empty(M, Cc, device=dev, dtype=torch.bfloat16)
        self.st = torch.empty(M * 2, device=dev, dtype=torch.float32)
        self.qkv = torch.empty(M, 3 * Cc, device=dev, dtype=torch.bfloat16)
        self.a = torch.empty(M, Cc, device=dev, dtype=torch.bfloat16)
        self.lse = torch.empty(Bh * H * N, device=dev, dtype=torch.float32)
        self.x1 = torch.empty(M, Cc, device=dev, dtype=torch.bfloat16)
        self.q = torch.empty(M, Cc, device=dev, dtype=torch.bfloat16)
        self.kv = r(Bh * 77, 2 * Cc)
        self.u = torch.empty(M, 8 * Cc, device=dev, dtype=torch.bfloat16)
        self.g = torch.empty(M, 4 * Cc, device=dev, dtype=torch.bfloat16)

    def block(self, w, st):
        M, Bh, s = self.M, self.Bh, C.c_void_p(st.cuda_stream)
        L.sdxl_op_layernorm_fwd(P(self.x), P(self.ln), P(w["g1"]), P(w["b1"]), P(self.st), M, Cc, 1e-5, s)
        L.sdxl_op_gemm(0, P(self.ln), P(w["wqkv"]), P(self.qkv), M, 3 * Cc, Cc, None, None, 0, 1, s)
        q, k, v = self.qkv[:, :Cc], self.qkv[:, Cc:2 * Cc], self.qkv[:, 2 * Cc:]
        L.sdxl_op_attention_fwd(P(q), P(k), P(v), P(self.a), P(self.lse), Bh, H, N, N, 3 * Cc, 3 * Cc, 3 * Cc, Cc, s)
        L.sdxl_op_gemm(0, P(self.a), P(w["wo1"]), P(self.x1), M, Cc, Cc, P(w["bo1"]), P(self.x), 0, 1, s)
        L.sdxl_op_layernorm_fwd(P(self.x1), P(self.ln), P(w["g2"]), P(w["b2"]), P(self.st), M, Cc, 1e-5, s)
        L.sdxl_op_gemm(0, P(self.ln), P(w["wq"]), P(self.q), M, Cc, Cc, None, None, 0, 1, s)
        kk, vv = self.kv[:, :Cc], self.kv[:, Cc:]
        L.sdxl_op_attention_fwd(P(self.q), P(kk), P(vv), P(self.a), P(self.lse), Bh, H, N, 77, Cc, 2 * Cc, 2 * Cc, Cc, s)
        L.sdxl_op_gemm(0, P(self.a), P(w["wo2"]), P(self.x), M, Cc, Cc, P(w["bo2"]), P(self.x1), 0, 1, s)
        L.sdxl_op_layernorm_fwd(P(self.x), P(self.ln), P(w["g3"]), P(w["b3"]), P(self.st), M, Cc, 1e-5, s)
        L.sdxl_op_ff_geglu_fwd(P(self.ln), P(w["wf1"]), P(w["bf1"]), P(self.u), P(self.g), M, Cc, 4 * Cc, 64, s)
        L.sdxl_op_gemm(0, P(self.g), P(w["wf2"]), P(self.x1), M, Cc, 4 * Cc, P(w["bf2"]), P(self.x), 0, 1, s)


def run(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full = Half(B)
t_full = run(lambda: [full.block(w, s0) for w in W])
ha, hb = Half(B // 2), Half(B // 2)


def two_streams(offset):
    s1.wait_stream(s0); s2.wait_stream(s0)
    if offset:                         # half B starts half a block late: one extra block of work on s2 first would shift everything; use a
        ha.block(W[0], s1)             # simple stagger: s1 runs one block ahead
        for i in range(NB - 1):
            ha.block(W[i + 1], s1)
            hb.block(W[i], s2)
        hb.block(W[NB - 1], s2)
    else:
        for w in W:
            ha.block(w, s1)
            hb.block(w, s2)
    s0.wait_stream(s1); s0.wait_stream(s2)


t_two = run(lambda: two_streams(False))
t_stag = run(lambda: two_streams(True))
t_half_seq = run(lambda: [(ha.block(w, s0), hb.block(w, s0)) for w in W])
print(f"{NB} level-2 transformer blocks forward: full batch one stream {t_full:.3f} ms ({t_full / NB * 1e3:.0f} us per block) | two half batches, "
      f"two streams in phase {t_two:.3f} | staggered by one block {t_stag:.3f} | two halves on ONE stream {t_half_seq:.3f}")
