#!/usr/bin/env python3
"""Stream-K kernel (gemm_sk.hip): correctness against fp32 matmul and timing next to the tile-per-workgroup kernels, on the
level-2 shapes of the B = 4, 1024^2 step, single problems and the fused dgrad + wgrad launches of a layer.

    python profiles/tools/sk_bench.py [--check-only] [--workers 256,240,160]
"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

lib.use_diag()      # the stream-K kernel lives in the diagnostics build (build.py --diag)

dev = torch.device("cuda:0")
L = lib.load()
FORMS = {"NT": 0, "NN": 1, "TN": 2}


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


class Prob:
    def __init__(self, form, M, N, K, bias=False, resid=False, accumulate=False):
        self.form, self.M, self.N, self.K = form, M, N, K
        if form == "NT":
            self.a, self.b = r(M, K), r(N, K, scale=K ** -0.5)
        elif form == "NN":
            self.a, self.b = r(M, K), r(K, N, scale=K ** -0.5)
        else:
            self.a, self.b = r(K, M), r(K, N)
        self.out = torch.zeros(M, N, device=dev, dtype=torch.float32 if form == "TN" else torch.bfloat16)
        self.bias = None
        if bias:
            self.bias = torch.zeros(M, device=dev, dtype=torch.float32) if form == "TN" else r(N)
        self.resid = r(M, N) if (resid and form != "TN") else None
        self.acc = accumulate

    def ref(self):
        a, b = self.a.float(), self.b.float()
        if self.form == "NT":
            y = a @ b.t()
        elif self.form == "NN":
            y = a @ b
        else:
            y = a.t() @ b
        if self.form != "TN":
            if self.bias is not None:
                y = y + self.bias.float()
            if self.resid is not None:
                y = y + self.resid.float()
        return y

    @property
    def flops(self):
        return 2.0 * self.M * self.N * self.K


def sk_args(ps):
    n = len(ps)
    I, V = C.c_int * n, C.c_void_p * n
    p = lambda t: t.data_ptr() if t is not None else None
    return (n, I(*[FORMS[q.form] for q in ps]), V(*[p(q.a) for q in ps]), V(*[p(q.b) for q in ps]), V(*[p(q.out) for q in ps]),
            I(*[q.M for q in ps]), I(*[q.N for q in ps]), I(*[q.K for q in ps]), V(*[p(q.bias) for q in ps]),
            V(*[p(q.resid) for q in ps]), I(*[int(q.acc) for q in ps]))


def run_sk(ps):
    args = sk_args(ps)
    return lambda: L.sdxl_op_gemm_sk(*args, st())


def run_old(q, mode):
    sk = 1
    if q.form == "TN":
        tiles = ((q.M + 127) // 128) * ((q.N + 127) // 128)
        sk = max(1, min(384 // tiles, q.K // 64 // 8, 32))
    args = (FORMS[q.form], q.a.data_ptr(), q.b.data_ptr(), q.out.data_ptr(), q.M, q.N, q.K, q.bias.data_ptr() if q.bias is not None else None,
            q.resid.data_ptr() if q.resid is not None else None, int(q.acc), sk, st())
    def f():
        L.sdxl_set_gemm_mode(mode)
        return L.sdxl_op_gemm(*args)
    return f


def sk_err():
    e = C.c_uint(0)
    lib.check(L.sdxl_sk_error(st(), C.byref(e)))
    return e.value


def check(ps, workers, tag):
    lib.check(L.sdxl_set_sk_mode(0, workers))
    for q in ps:
        q.out.zero_()
        if q.form == "TN" and q.bias is not None:
            q.bias.zero_()
    lib.check(run_sk(ps)())
    torch.cuda.synchronize()
    ok = True
    for q in ps:
        ref = q.ref()
        err = float((q.out.float() - ref).abs().max() / ref.abs().max())
        tol = 6e-3 if q.form != "TN" else 2e-5 * q.K ** 0.5 + 1e-5
        bad = not (err <= tol)
        msg = f"  {tag} workers={workers} {q.form} {q.M}x{q.N}x{q.K}: err {err:.2e} (tol {tol:.1e})"
        if q.form == "TN" and q.bias is not None:
            rb = q.a.float().sum(0)
            eb = float((q.bias - rb).abs().max() / rb.abs().max())
            msg += f" bias_grad err {eb:.2e}"
            bad = bad or eb > 1e-4
        print(msg + ("  <-- FAIL" if bad else ""), flush=True)
        ok = ok and not bad
    e = sk_err()
    if e:
        print(f"  sk error word = {e}  <-- FAIL")
        ok = False
    return ok


def main():
    torch.manual_seed(0)
    workers = [256]
    if "--workers" in sys.argv:
        workers = [int(x) for x in sys.argv[sys.argv.index("--workers") + 1].split(",")]
    ok = True
    # ---- correctness: small cases with awkward partitions, then the model's shapes ----
    small = [[Prob("NT", 256, 256, 64, bias=True, resid=True)], [Prob("NT", 512, 768, 640, bias=True)], [Prob("NN", 512, 512, 1280, resid=True)],
             [Prob("TN", 512, 256, 1024, bias=True)], [Prob("TN", 768, 512, 4096, accumulate=False)],
             [Prob("NN", 512, 256, 1280), Prob("TN", 1280, 256, 512, bias=True)],
             [Prob("NT", 1024, 1280, 320), Prob("NN", 256, 256, 2560), Prob("TN", 256, 512, 768), Prob("TN", 512, 256, 768)]]
    for ps in small:
        for wk in (1, 3, 7, 16, 37, 100, 255, 256, 0):
            ok = check(ps, wk, "small") and ok
    model = [[Prob("NT", 4096, 1280, 1280, bias=True, resid=True)], [Prob("NT", 4096, 3840, 1280)], [Prob("NT", 4096, 1280, 5120, bias=True, resid=True)],
             [Prob("NT", 4096, 10240, 1280, bias=True)],
             [Prob("NN", 4096, 1280, 10240), Prob("TN", 10240, 1280, 4096, bias=True)],
             [Prob("NN", 4096, 1280, 3840), Prob("TN", 3840, 1280, 4096)],
             [Prob("NN", 4096, 1280, 1280), Prob("TN", 1280, 1280, 4096, bias=True)],
             [Prob("NN", 4096, 5120, 1280), Prob("TN", 1280, 5120, 4096, bias=True)]]
    for ps in model:
        for wk in workers + [0]:
            ok = check(ps, wk, "model") and ok
    # reproducibility + stress: repeated launches, bit-identical outputs
    ps = model[4]
    lib.check(L.sdxl_set_sk_mode(0, 256))
    f = run_sk(ps)
    f(); torch.cuda.synchronize()
    base = [q.out.clone() for q in ps]
    same = True
    for it in range(50):
        for q in ps:
            q.out.fill_(7.0)
        f()
        torch.cuda.synchronize()
        same = same and all(torch.equal(q.out, b) for q, b in zip(ps, base))
    print(f"  50 repeated fused launches bit-identical: {same}; error word {sk_err()}" + ("" if same else "  <-- FAIL"), flush=True)
    ok = ok and same
    print("CHECK", "PASS" if ok else "FAIL", flush=True)
    if "--check-only" in sys.argv:
        return 0 if ok else 1
    # ---- timing ----
    print(f"{'problem(s)':58s} {'old us':>9s} {'TF/s':>7s} | " + " ".join(f"sk{w:>3d} us  TF/s" for w in workers) + " | policy us TF/s")
    for ps in model:
        fl = sum(q.flops for q in ps)
        told = 0.0
        for q in ps:
            told += min(bench(run_old(q, 1)), bench(run_old(q, 0)))
        line = (" + ".join(f"{q.form} {q.M}x{q.N}x{q.K}" for q in ps)).ljust(58) + f" {told:9.1f} {fl / told / 1e6:7.0f} |"
        for wk in workers + [0]:
            lib.check(L.sdxl_set_sk_mode(0, wk))
            t = bench(run_sk(ps))
            line += f" {t:8.1f} {fl / t / 1e6:5.0f} " + ("|" if wk == workers[-1] else "")
        print(line, flush=True)
    # single problems of the backward, for reference
    for q in [Prob("NN", 4096, 1280, 10240), Prob("TN", 10240, 1280, 4096), Prob("NN", 4096, 5120, 1280), Prob("TN", 1280, 5120, 4096), Prob("TN", 1280, 1280, 4096)]:
        told = min(bench(run_old(q, 1)), bench(run_old(q, 0)))
        line = f"{q.form} {q.M}x{q.N}x{q.K}".ljust(58) + f" {told:9.1f} {q.flops / told / 1e6:7.0f} |"
        for wk in workers + [0]:
            lib.check(L.sdxl_set_sk_mode(0, wk))
            t = bench(run_sk([q]))
            line += f" {t:8.1f} {q.flops / t / 1e6:5.0f} " + ("|" if wk == workers[-1] else "")
        print(line, flush=True)
    lib.check(L.sdxl_set_sk_mode(0, 0))
    lib.check(L.sdxl_set_gemm_mode(1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
