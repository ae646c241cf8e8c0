"""Attention microbenchmark on the model's shapes (sdxl_op_attention_fwd / _bwd through the C ABI): python profiles/tools/attn_bench.py [--self-only] [--iters N]"""
import sys, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sdxl_amd
from sdxl_amd import lib
if '--lib' in sys.argv:      # knock-out builds (tools/build_diag_attn.sh)
    lib.LIB_PATH = Path(sys.argv[sys.argv.index('--lib') + 1]).resolve()
L = lib.load(); dev = torch.device('cuda:0')
import os
if os.environ.get('SDXL_KNOB20') == '1': lib.check(L.sdxl_set_knob(20, 1))      # diagnostics build: the two-launch self-attention backward
for kv in os.environ.get('SDXL_KNOBS', '').split(','):      # diagnostics build: SDXL_KNOBS=32=1,20=1
    if kv: lib.check(L.sdxl_set_knob(int(kv.split('=')[0]), int(kv.split('=')[1])))
ptr = lambda t: C.c_void_p(t.data_ptr())
r = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
ITERS = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
def bench(fn, flops, name, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s", flush=True)
SHAPES = [(4,10,4096,4096,True),(4,20,1024,1024,True),(4,10,4096,77,False),(4,20,1024,77,False)]
if '--self-only' in sys.argv: SHAPES = SHAPES[:2]
for (B,H,N,Nk,selfa) in SHAPES:
    Cc = H*64
    if selfa:
        qkv = r(B,N,3*Cc); q,k,v = qkv[...,:Cc], qkv[...,Cc:2*Cc], qkv[...,2*Cc:]; ld=(3*Cc,)*3
        dqkv = torch.empty_like(qkv); dq,dk,dv = dqkv[...,:Cc], dqkv[...,Cc:2*Cc], dqkv[...,2*Cc:]
    else:
        q = r(B,N,Cc); kv = r(B,Nk,2*Cc); k,v = kv[...,:Cc], kv[...,Cc:]; ld=(Cc,2*Cc,2*Cc)
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk,dv = dkv[...,:Cc], dkv[...,Cc:]
    o = torch.empty(B,N,Cc,dtype=torch.bfloat16,device=dev); lse = torch.empty(B*H,N,dtype=torch.float32,device=dev)
    do = r(B,N,Cc); delta = torch.empty_like(lse)
    fl = 4.0*B*H*N*Nk*64
    bench(lambda: L.sdxl_op_attention_fwd(ptr(q),ptr(k),ptr(v),ptr(o),ptr(lse),B,H,N,Nk,ld[0],ld[1],ld[2],Cc,None), fl, f"attn fwd B{B} H{H} {N}x{Nk}")
    bench(lambda: L.sdxl_op_attention_bwd(ptr(q),ptr(k),ptr(v),ptr(o),ptr(do),ptr(lse),ptr(delta),ptr(dq),ptr(dk),ptr(dv),B,H,N,Nk,ld[0],ld[1],ld[2],Cc,None), 2.5*fl, f"attn bwd")
