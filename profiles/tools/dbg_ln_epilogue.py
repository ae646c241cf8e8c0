"""Op-level runs of the dgrad whose epilogue runs a LayerNorm backward (sdxl_op_linear_dgrad_ln_bwd) beside the plain dgrad + the separate
LayerNorm backward, for a kernel trace (profiles/tools/ln_epilogue_trace.sh); prints the parity of the fused result as it goes."""
import sys, ctypes as C, torch, importlib
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
lib = importlib.import_module('sdxl-training-improvements_amd.lib')
L = lib.load()
dev = torch.device('cuda:0')
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)
def ptr(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def fp(t): return C.cast(t.data_ptr(), C.POINTER(C.c_float))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, Cc, Kr, acc) in [(4096, 1280, 1280, 1), (4096, 1280, 3840, 1), (4096, 1280, 10240, 1)]:
    x = (rnd(M, Cc, seed=60) * 2.0 + 0.5)
    gamma, beta = (rnd(Cc, seed=61) * 0.1 + 1.0), rnd(Cc, seed=62)
    xf = x.float()
    stats = torch.stack([xf.mean(1), (xf.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    dyl = rnd(M, Kr, seed=63); w = rnd(Kr, Cc, seed=64, scale=Kr ** -0.5)
    addend = rnd(M, Cc, seed=65) if acc else None
    dx = torch.empty(M, Cc, dtype=torch.bfloat16, device=dev)
    nrb = (M + 127) // 128
    pcol = torch.empty(nrb, 2, Cc, dtype=torch.float32, device=dev)
    for rep in range(3):
        lib.check(L.sdxl_op_linear_dgrad_ln_bwd(ptr(dyl), ptr(w), ptr(x), fp(stats), ptr(gamma), ptr(addend), ptr(dx), None, fp(pcol), M, Cc, Kr, st))
    torch.cuda.synchronize()
    dy = torch.empty(M, Cc, dtype=torch.bfloat16, device=dev)
    dx2 = addend.clone() if acc else torch.empty_like(x)
    dg = torch.zeros(Cc, dtype=torch.float32, device=dev); db = torch.zeros(Cc, dtype=torch.float32, device=dev)
    for rep in range(9):
        lib.check(L.sdxl_op_gemm(1, ptr(dyl), ptr(w), ptr(dy), M, Cc, Kr, None, None, 0, 1, st))
        if acc: dx2.copy_(addend)
        lib.check(L.sdxl_op_layernorm_bwd(ptr(x), ptr(dy), ptr(gamma), fp(stats.reshape(-1)), ptr(dx2), fp(dg), fp(db), M, Cc, acc, st))
    torch.cuda.synchronize()
    print(M, Cc, Kr, 'fused vs separate dx max diff', float((dx.float() - dx2.float()).abs().max()), 'ref max', float(dx2.float().abs().max()))
