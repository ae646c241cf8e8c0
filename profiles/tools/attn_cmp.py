"""Two settings of the attention kernels against each other on the model's shapes (diagnostics build): forward O / LSE and backward dQ / dK / dV
with knob <id> = 0 and = <value>:  SDXL_DIAG=1 python profiles/tools/attn_cmp.py <id> <value>"""
import sys, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd
from sdxl_amd import lib
L = lib.load(); dev = torch.device('cuda:0')
KID, KV = int(sys.argv[1]), int(sys.argv[2])
ptr = lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
def run(B, H, N, Nk, selfa, knob):
    lib.check(L.sdxl_set_knob(KID, knob))
    torch.manual_seed(1)
    Cc = H * 64
    if selfa:
        qkv = r(B, N, 3 * Cc); q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]; ld = (3 * Cc,) * 3
        dqkv = torch.zeros_like(qkv); dq, dk, dv = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:]
    else:
        q = r(B, N, Cc); kv = r(B, Nk, 2 * Cc); k, v = kv[..., :Cc], kv[..., Cc:]; ld = (Cc, 2 * Cc, 2 * Cc)
        dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); dk, dv = dkv[..., :Cc], dkv[..., Cc:]
    o = torch.zeros(B, N, Cc, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B * H, N, dtype=torch.float32, device=dev)
    do = r(B, N, Cc); delta = torch.zeros_like(lse)
    lib.check(L.sdxl_op_attention_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, H, N, Nk, ld[0], ld[1], ld[2], Cc, None))
    lib.check(L.sdxl_op_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), B, H, N, Nk, ld[0], ld[1], ld[2], Cc, None))
    torch.cuda.synchronize()
    return [t.float().clone() for t in (o, lse, dq, dk, dv)]
for shp in [(2, 10, 4096, 4096, True), (4, 20, 1024, 1024, True), (2, 10, 4032, 4032, True), (1, 20, 1008, 1008, True), (2, 10, 4096, 77, False), (4, 20, 1024, 77, False), (1, 5, 200, 333, False), (1, 2, 64, 64, True)]:
    a = run(*shp, 0); b = run(*shp, KV)
    d = [((x - y).norm() / (x.norm() + 1e-30)).item() for x, y in zip(a, b)]
    print(shp, "rel diff O %.2e LSE %.2e dQ %.2e dK %.2e dV %.2e" % tuple(d), "finite", all(torch.isfinite(t).all().item() for t in b))
