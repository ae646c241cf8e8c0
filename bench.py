#!/usr/bin/env python3
"""Headline benchmark: images/sec of one SDXL-base training step (loss prep + UNet forward + loss + UNet backward,
gradients zeroed each step; optimizer excluded) on synthetic data, batch 4 per GPU at 1024^2 (latent 128x128), bf16.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; data parallel = each rank runs the same step on its own batch and the UNet gradients are
all-reduced (bf16, pre-scaled by 1/N) over RCCL/xGMI in ~190 MB buckets overlapped with the rest of backward.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family (the bf16 MFMA GEMM behind every Linear and
3x3 conv: forward, dgrad, wgrad), measured with HIP events around each of its launches in a separate profiled step;
`cpu_baseline` is the fp32 CPU oracle (the restatement of the reference's arithmetic) timed on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

FLOP_PER_IMAGE_1024 = 20.28e12      # fwd+bwd, BASELINE.md section 2 (2*M*N*K of every matmul, bwd = 2x fwd)
FLOP_PER_IMAGE_1344x768 = 19.93e12          # 1344x768 (latent 96x168)
FLOP_PER_IMAGE_512 = 4.77e12
PEAK_BF16_TFLOPS = 2516.6           # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU (MI355X dense bf16 MFMA)

WORKLOADS = {
    "ddpm_b4_1024": dict(method="ddpm", B=4, H=128, W=128, flop_per_image=FLOP_PER_IMAGE_1024,
                         desc="method=ddpm v_prediction + zero_terminal_snr + MinSNR(5), SDXL-base UNet fwd+bwd, "
                              "batch 4/GPU, 1024^2 (latent 128x128)"),
    "flow_b4_1024": dict(method="flow_matching", B=4, H=128, W=128, flop_per_image=FLOP_PER_IMAGE_1024,
                         desc="method=flow_matching (logit-normal t), SDXL-base UNet fwd+bwd, batch 4/GPU, 1024^2"),
    "flow_b4_1344x768": dict(method="flow_matching", B=4, H=96, W=168, flop_per_image=FLOP_PER_IMAGE_1344x768,
                             desc="method=flow_matching, SDXL-base UNet fwd+bwd, batch 4/GPU, 1344x768 bucket (latent 96x168)"),
    "ddpm_b1_512": dict(method="ddpm", B=1, H=64, W=64, flop_per_image=FLOP_PER_IMAGE_512,
                        desc="method=ddpm, SDXL-base UNet fwd+bwd, batch 1, 512^2 (latent 64x64)"),
}


def karras_table(n=1000, smin=0.002, smax=20000.0, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    return (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho


def make_batch(wl, rank, device, cross=2048, pooled=1280):
    g = torch.Generator().manual_seed(1234 + rank)
    B, H, W = wl["B"], wl["H"], wl["W"]
    r = lambda *s: torch.randn(*s, generator=g)
    b = dict(lat=r(B, 4, H, W), noise=r(B, 4, H, W), ehs=r(B, 77, cross), pooled=r(B, pooled),
             tid=torch.tensor([[8.0 * W, 8.0 * H, 0, 0, 8.0 * W, 8.0 * H]] * B))
    u = torch.rand(B, generator=g)
    if wl["method"] == "ddpm":
        ts = (u * 1000).long()
        b["timestep"] = ts.float()
        b["sigma_or_t"] = karras_table()[ts]
    else:
        t = torch.sigmoid(r(B))
        b["timestep"] = t
        b["sigma_or_t"] = t
    return {k: v.to(device) for k, v in b.items()}


def _physical_cores_per_socket():
    """physical cores of one socket of this host (lscpu), falling back to os.cpu_count()"""
    try:
        import subprocess
        info = dict(l.split(":", 1) for l in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines() if ":" in l)
        return int(info["Core(s) per socket"].strip())
    except Exception:
        return os.cpu_count() or 8


def cpu_baseline(threads_list=None, timed_steps: int = 3):
    """the fp32 CPU oracle (the restatement pinned to the reference's loss-side goldens), cfg 1 (ddpm, B=1, 512^2,
    single process): 1 warm-up + `timed_steps` timed forward+backward steps, at threads = one socket's physical cores and at
    threads = 8 (the authoring container's core count).  `value` is the better of the two, scaled to 1024^2-equivalent images by
    the FLOP ratio; both measurements are reported with their thread counts."""
    from oracle import loss_ref as R
    from oracle import unet_ref as U
    cfg = U.SDXL_BASE
    t0 = time.time()
    w = {}
    g = torch.Generator().manual_seed(0)
    pool = torch.rand(1 << 22, generator=g) - 0.5        # cheap init (tiled random pool): timing is value-independent
    for name, shape in U.param_shapes(cfg).items():
        std, mean = U.synth_std(name, shape)
        n = math.prod(shape)
        t = pool.repeat((n + pool.numel() - 1) // pool.numel())[:n].reshape(shape) * (3.4641 * std) + mean
        w[name] = t.requires_grad_(True)
    t_init = time.time() - t0
    B, H, W = 1, 64, 64
    lat, noise = torch.randn(B, 4, H, W, generator=g), torch.randn(B, 4, H, W, generator=g)
    batch = {"vae_latents": lat, "prompt_embeds": torch.randn(B, 77, 2048, generator=g),
             "pooled_prompt_embeds": torch.randn(B, 1280, generator=g),
             "time_ids": torch.tensor([[512.0, 512, 0, 0, 512, 512]])}
    ts = torch.tensor([500])
    unet_fn = lambda s, t, e, p, ti: U.unet_forward(w, s, t, e, p, ti, cfg)

    def one_step():
        for t in w.values():
            t.grad = None
        t0 = time.time()
        out = R.compute_loss_ddpm(unet_fn, batch, noise, ts)
        out["loss"].backward()
        return time.time() - t0

    if not threads_list:
        threads_list = sorted({_physical_cores_per_socket(), 8}, reverse=True)
    runs = []
    for n in threads_list:
        torch.set_num_threads(int(n))
        one_step()                                       # warm-up (allocator, thread pool, oneDNN primitive cache)
        dts = [one_step() for _ in range(timed_steps)]
        dt = sum(dts) / len(dts)
        runs.append({"threads": torch.get_num_threads(), "s_per_step_512": round(dt, 3), "img_per_s_512": round(B / dt, 4),
                     "tflops": round(FLOP_PER_IMAGE_512 / dt / 1e12, 3)})
    best = max(runs, key=lambda r: r["img_per_s_512"])
    return {"value": best["img_per_s_512"] * FLOP_PER_IMAGE_512 / FLOP_PER_IMAGE_1024, "unit": "images/sec", "cores": best["threads"],
            "kind": "port", "runs": runs,
            "sample": f"oracle (fp32 torch CPU restatement), cfg1: ddpm B=1 512^2 fwd+bwd, 1 warm-up + {timed_steps} timed steps per "
                      f"thread count {[r['threads'] for r in runs]}; value = best ({best['s_per_step_512']} s/step at {best['threads']} "
                      f"threads) scaled to 1024^2-equivalent images by the FLOP ratio 4.77/20.28; weight init {t_init:.1f} s untimed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ddpm_b4_1024", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the (untimed) fused-optimizer measurement")
    ap.add_argument("--cpu-threads", type=int, nargs="*", default=None, help="thread counts of the CPU baseline (default: one socket's cores, 8)")
    ap.add_argument("--exchange", default="zero1", choices=["zero1", "allreduce"],
                    help="N > 1: reduce-scatter of the gradient buckets (ZeRO-1, default) or all-reduce")
    ap.add_argument("--no-emit", action="store_true", help="N > 1 A/B: cast the fp32 gradient arena per bucket instead of bf16 wgrad epilogues")
    ap.add_argument("--profile-steps", type=int, default=1, help="extra steps with per-launch GEMM event timing")
    ap.add_argument("--gemm-mode", type=int, default=None, help="A/B runs: sdxl_set_gemm_mode (0 = 128-row kernel only)")
    ap.add_argument("--lib", default=None, help="A/B runs: another build of libsdxlstep.so (e.g. last round's) on the same box")
    args = ap.parse_args()

    import sdxl_amd  # noqa: F401
    from sdxl_amd import distributed as D
    from sdxl_amd import lib, synth
    from sdxl_amd import unet as NU

    rank, local_rank, world = D.env_rank_world()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
    # test hook (single-GPU boxes): SDXL_BENCH_BACKEND=gloo + SDXL_BENCH_ONE_DEVICE=1 run all ranks on cuda:0 to exercise the
    # N > 1 control flow (barriers, per-segment exchange, MAX over ranks) without RCCL; never used for reported numbers
    backend = os.environ.get("SDXL_BENCH_BACKEND", "nccl")
    if os.environ.get("SDXL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    D.init_process_group(backend if world > 1 else None)
    wl = WORKLOADS[args.workload]
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.lib:                     # an older build may lack newer entry points: bind what it has
        import ctypes
        lib.LIB_PATH = Path(args.lib).resolve()
        probe = ctypes.CDLL(str(lib.LIB_PATH))
        lib.SIGNATURES = {k: v for k, v in lib.SIGNATURES.items() if hasattr(probe, k)}
    if args.gemm_mode is not None:
        lib.check(lib.load().sdxl_set_gemm_mode(args.gemm_mode))
    net = NU.NativeUNet(NU.make_config(), device=local_rank)
    synth.load_synthetic(net, seed=0)                      # same weights on every rank
    net.plan(wl["B"], wl["H"], wl["W"], 77)
    b = make_batch(wl, rank, dev)
    L = net.L

    emit = world > 1 and not args.no_emit and hasattr(L, "sdxl_set_grad_emit")

    def cast(off, n, dst):
        if emit:           # the wgrad GEMMs write bf16 into the exchange arena themselves: only biases / norm parameters are cast
            net.cast_small(off, n, dst)
            return
        lib.check(L.sdxl_grads_to_bf16(net.h, off, n, C.c_void_p(dst.data_ptr()), 1.0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # data parallel: every rank runs the same step on its own batch; the gradient buckets are reduce-scattered (ZeRO-1: rank r
    # keeps slice r of every bucket, bf16, pre-scaled by 1/N) over RCCL under the rest of the backward.  The sharded optimizer
    # update + parameter all-gather that follow belong to the optimizer phase, which the metric excludes at every N (measured
    # separately below).
    Sync = D.ShardedGradSync if args.exchange == "zero1" else D.GradSync
    sync = Sync(net.param_elems, cast, torch.bfloat16, dev)
    scale = 1.0 / world
    if emit:
        net.set_grad_emit(sync.comm, 1.0)

    def step():
        net.zero_grads()
        net.forward_loss(wl["method"], b["lat"], b["noise"], b["sigma_or_t"], b["timestep"], b["ehs"], b["pooled"], b["tid"])
        net.backward(scale, True, on_segment=sync.on_segment if world > 1 else None, segment_stream=True)
        sync.finish()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss = net.read_loss()[0]
    images = world * wl["B"] * args.steps
    value = images / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # dominant-kernel roofline: per-launch HIP events around every GEMM-family launch (same stream), profiled steps
    roof = None
    if args.profile_steps > 0:
        lib.check(L.sdxl_profile_gemm_begin())
        for _ in range(args.profile_steps):
            step()
        fl, ms, n = C.c_double(), C.c_double(), C.c_int()
        lib.check(L.sdxl_profile_gemm_end(C.byref(fl), C.byref(ms), C.byref(n)))
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        traffic = None       # fabric-side bytes of the GEMM family per step (one launch set), from the committed PMC passes of this
        try:                 # workload: FETCH_SIZE x 2 (the gfx950 correction of the microarchitecture guide) + WRITE_SIZE, both in KiB
            with open(ROOT / "profiles" / "r02_pmc_step_summary.json") as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload:
                traffic = round(pm["gemm"]["hbm_bytes_per_step"])
        except Exception:
            pass
        roof = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_note": "bytes per step over all launches of the family (fabric side, MALL hits included); algorithmic "
                                "operand + result bytes ~94e9",
                "kernel": "gemm_kernel<NT|NN|TN, conv|linear> (128-row tiles) + gemm256_kernel (256x256 tiles): bf16 MFMA 16x16x32",
                "launches_per_step": n.value // args.profile_steps,
                "gemm_ms_per_step": round(ms.value / args.profile_steps, 2),
                "gemm_tflop_per_step": round(fl.value / args.profile_steps / 1e12, 2)}

    # row f1, measured OUTSIDE the timed region (the metric excludes the optimizer): the fused AdamW_BF16 update of all
    # parameters, HBM-bound: 20 algorithmic bytes per element (p, m, v, shift bf16 in + out, fp32 gradient in)
    opt_extra = None
    if not args.no_optimizer:
        from sdxl_amd.optimizer import AdamWBF16
        opt = AdamWBF16(net, lr=4e-7, weight_decay=0.01)
        sharded = world > 1 and args.exchange == "zero1"

        def update():
            if sharded:           # ZeRO-1: this rank's slices of every bucket, then the parameters are all-gathered
                opt.step(sync.reduced(), pieces=sync.pieces)
                sync.gather_params(net.weights)
            else:
                opt.step(sync.reduced() if world > 1 else None)

        for _ in range(2):
            update()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            update()
        e1.record()
        torch.cuda.synchronize()
        oms = e0.elapsed_time(e1) / 5
        nel = net.weights.numel()
        upd = nel // world if sharded else nel
        traffic = None                   # HBM bytes per launch from the committed PMC passes (profiles/r01j_pmc_adamw.json)
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01j_pmc_adamw.json")) as f:
                pm = json.load(f)
            traffic = round(pm["hbm_bytes_per_launch"] * upd / pm["elements"])
        except Exception:
            pass
        gbytes = (20.0 if world == 1 else 18.0) * upd        # p, m, v, shift in + out (16 B) + fp32 (4 B) or bf16 (2 B) gradient
        opt_extra = {"kernel": "adamw_bf16_kernel (fused AdamW_BF16 step" + (", this rank's ZeRO-1 slices + parameter all-gather)" if sharded
                               else ", all parameters in one launch)"),
                     "ms_per_update": round(oms, 2), "params_updated_per_rank": upd,
                     "roofline": {"bound": "hbm", "achieved": round(gbytes / oms / 1e6, 1), "peak": 8000.0,
                                  "unit": "GB/s", "frac": round(gbytes / oms / 1e6 / 8000.0, 4), "traffic": traffic,
                                  "note": "at N > 1 the time includes the parameter all-gather" if sharded else None},
                     "note": "not part of `value`; one update per gradient_accumulation_steps micro-steps"}
        del opt
    if rank == 0:
        step_tflops = value / world * wl["flop_per_image"] / 1e12
        out = {"metric": "images/sec/node SDXL-base 1024^2 bf16 fwd+bwd", "value": round(value, 3), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": wl["desc"], "global_batch": wl["B"] * world, "parallelism": f"dp{world}",
                          "exchange": None if world == 1 else ("reduce-scatter of bf16 gradient buckets overlapped with backward (ZeRO-1)"
                                                               if args.exchange == "zero1" else "all-reduce of bf16 gradient buckets overlapped with backward"),
                          "weights": "synthetic (counter-hash init of the 2,567,463,684-parameter SDXL-base UNet)"},
               "step_tflops_per_gpu": round(step_tflops, 1),
               "step_mfma_frac": round(step_tflops / PEAK_BF16_TFLOPS, 4),
               "loss": loss, "roofline": roof, "optimizer": opt_extra}
        if world == 1 and not args.no_cpu_baseline:
            del net, sync
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.cpu_threads or None)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
