"""Worker of tests/test_gpu_multiproc.py::test_zero1_update_bit_equal_to_unsharded (one process per rank, all on cuda:0, gloo).

Two ranks with different gradients.  Path A (ZeRO-1, row f3): bucketed reduce-scatter -> squared norm of the owned slices + one
float all-reduced -> clip coefficient -> fused AdamW_BF16 on the owned slices -> all-gather of the parameters.  Path B: bucketed
all-reduce -> norm of the whole arena -> the same kernel over the whole arena.  The parameters after two updates must be
bit-identical, and so must every optimizer-state element a rank owns."""
import ctypes as C
import importlib
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import lib  # noqa: E402

D = importlib.import_module("sdxl-training-improvements_amd.distributed")
O = importlib.import_module("sdxl-training-improvements_amd.optimizer")


class Arena:
    def __init__(self, w):
        self.L = lib.load()
        self.weights = w.clone()
        self.grads = torch.zeros(w.numel(), dtype=torch.float32, device=w.device)

    def zero_grads(self):
        pass

    def param_ranges(self):
        return {"a": (0, 4096), "b": (4096, self.weights.numel() - 4096)}      # two "tensors" for the lazy-decay bookkeeping


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    D.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = lib.load()
    total = 3 * 65536 + 4096
    segs = [(2 * 65536 + 4096, 65536), (65536, 65536 + 4096), (0, 65536)]      # reverse execution order
    gw = torch.Generator().manual_seed(7)
    w0 = (torch.randn(total, generator=gw) * 0.05).to(torch.bfloat16).to(dev)
    nets = {"zero": Arena(w0), "full": Arena(w0)}
    opts = {k: O.AdamWBF16(n, lr=1e-2, weight_decay=0.6, seed=5) for k, n in nets.items()}   # lr * wd large: the lazy decay fires
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = True
    for step in range(2):
        g = torch.randn(total, generator=torch.Generator().manual_seed(100 * step + rank)).to(dev) * 3.0

        def cast(off, n, dst, g=g):
            dst.copy_((g[off:off + n] * (1.0 / world)).to(torch.bfloat16))

        # ---- path A: ZeRO-1
        if step == 0:
            zs = D.ShardedGradSync(total, cast, torch.bfloat16, dev)
            fs = D.GradSync(total, cast, torch.bfloat16, dev)
        zs.cast = fs.cast = cast
        for k, (off, n) in enumerate(segs):
            zs.on_segment(k, off, n)
        zs.finish()
        buf = torch.zeros(2, dtype=torch.float32, device=dev)
        sh = zs.reduced()
        lib.check(L.sdxl_sumsq(C.c_void_p(sh.data_ptr()), 1, sh.numel(), C.c_void_p(buf.data_ptr()), st()))
        zs.global_sumsq(buf[0:1])
        lib.check(L.sdxl_clip_coef(C.c_void_p(buf.data_ptr()), 1.0, C.c_void_p(buf.data_ptr() + 4), st()))
        opts["zero"].step(sh, grad_scale=buf[1:2], pieces=zs.pieces)
        zs.gather_params(nets["zero"].weights)
        # ---- path B: all-reduce + full update
        for k, (off, n) in enumerate(segs):
            fs.on_segment(k, off, n)
        fs.finish()
        buf2 = torch.zeros(2, dtype=torch.float32, device=dev)
        lib.check(L.sdxl_sumsq(C.c_void_p(fs.reduced().data_ptr()), 1, total, C.c_void_p(buf2.data_ptr()), st()))
        lib.check(L.sdxl_clip_coef(C.c_void_p(buf2.data_ptr()), 1.0, C.c_void_p(buf2.data_ptr() + 4), st()))
        # the norm of the whole arena agrees with the slice-wise one to fp32 summation order; clipping is active.  Every rank
        # applies the coefficient derived from the all-reduced slice norms, so path B is handed that same coefficient.
        ok = ok and abs(float(buf[0]) - float(buf2[0])) <= 1e-5 * float(buf2[0]) and float(buf2[1]) < 1.0
        opts["full"].step(fs.reduced(), grad_scale=buf[1:2])
        torch.cuda.synchronize()
        for off, n, goff in zs.pieces:      # the slices hold what the all-reduce holds
            ok = ok and torch.equal(sh[goff:goff + n], fs.reduced()[off:off + n])
        ok = ok and torch.equal(nets["zero"].weights, nets["full"].weights)          # identical bits, everywhere
        for off, n, _ in zs.pieces:         # and identical optimizer state on everything this rank owns
            for a in ("exp_avg", "exp_avg_sq", "shift"):
                ok = ok and torch.equal(getattr(opts["zero"], a)[off:off + n], getattr(opts["full"], a)[off:off + n])
        if not ok:
            print(f"rank {rank} step {step}: mismatch", flush=True)
        ok = ok and not torch.equal(nets["zero"].weights, w0)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("ZERO1_OK" if float(flag) == 1.0 else "ZERO1_MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
