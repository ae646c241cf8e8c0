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
ATTN_FLOP_PER_IMAGE_1024 = 2.351e12   # of those, attention QK^T / PV fwd + bwd (3.135 of 27.045 TFLOP per B = 4 forward)
PMC_SUMMARY = "r06_pmc_step_summary.json"     # committed PMC passes (profiles/tools/measure_step.sh), stamped with the commit they were taken at


def kernels_changed_since(commit):
    """True if the kernel sources differ between `commit` and the working tree (then the committed PMC traffic figure is stale);
    None when git is not available (the GPU box's snapshot has no .git: the stamp is reported, the comparison is the reader's)."""
    try:
        import subprocess
        # (outside a work tree -- the GPU box's snapshot -- `git diff --quiet` ALSO exits 1: ask for the commit first)
        have = subprocess.run(["git", "-C", str(ROOT), "cat-file", "-e", commit + "^{commit}"], capture_output=True, timeout=20)
        if have.returncode != 0:
            return None
        r = subprocess.run(["git", "-C", str(ROOT), "diff", "--quiet", commit, "--", "sdxl-training-improvements_amd/csrc"],
                           capture_output=True, timeout=20)
        return None if r.returncode not in (0, 1) else r.returncode == 1
    except Exception:
        return None

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
    # configs[4]: the two bucket shapes alternate 1:1 per micro-step (two plans sharing arenas and workspace), gradient
    # accumulation 4 -- gradients are zeroed at the start of a cycle and exchanged (N > 1) on every 4th micro-step only.
    # One "step" of the bench = one micro-step (4 images per GPU).
    "flow_mixed_accum4": dict(method="flow_matching", B=4, H=128, W=128, accum=4, buckets=[(128, 128), (96, 168)],
                              flop_per_image=0.5 * (FLOP_PER_IMAGE_1024 + FLOP_PER_IMAGE_1344x768),
                              desc="method=flow_matching, mixed aspect-ratio buckets 1024^2 / 1344x768 alternating per micro-step, "
                                   "grad-accum 4 (exchange on every 4th micro-step), batch 4/GPU; one step = one micro-step"),
}


def karras_table(n=1000, smin=0.002, smax=20000.0, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    return (smax ** (1 / rho) + ramp * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho


def make_batch(wl, rank, device, cross=2048, pooled=1280, hw=None):
    g = torch.Generator().manual_seed(1234 + rank)
    B, H, W = wl["B"], wl["H"], wl["W"]
    if hw is not None:
        H, W = hw
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


def _usable_cores():
    """cores this process may actually run on: the affinity mask, cut by the cgroup CPU quota (a container with 8 CPUs' worth of quota on a
    64-core host runs 64 threads SLOWER than 8: round 5's socket-wide run measured 0.40 TFLOP/s against 0.79 at 8 threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()) + 0.5)))
            break
        except Exception:
            continue
    return n


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

    usable = _usable_cores()
    if not threads_list:
        # one socket's physical cores, but never more threads than the process may run on (affinity mask, cgroup quota), and 8 (the
        # authoring container's count) for comparison; the process is pinned to that many CPUs of its mask for each run
        threads_list = sorted({min(_physical_cores_per_socket(), usable), min(8, usable)}, reverse=True)
    mask0 = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    runs = []
    for n in threads_list:
        torch.set_num_threads(int(n))
        if mask0 and len(mask0) >= int(n) and len(mask0) == usable:      # (under a quota without a mask the kernel places the threads)
            os.sched_setaffinity(0, set(mask0[:int(n)]))
        one_step()                                       # warm-up (allocator, thread pool, oneDNN primitive cache)
        dts = [one_step() for _ in range(timed_steps)]
        dt = sum(dts) / len(dts)
        runs.append({"threads": torch.get_num_threads(), "s_per_step_512": round(dt, 3), "img_per_s_512": round(B / dt, 4),
                     "tflops": round(FLOP_PER_IMAGE_512 / dt / 1e12, 3)})
    if mask0:
        os.sched_setaffinity(0, set(mask0))
    best = max(runs, key=lambda r: r["img_per_s_512"])
    return {"value": best["img_per_s_512"] * FLOP_PER_IMAGE_512 / FLOP_PER_IMAGE_1024, "unit": "images/sec", "cores": best["threads"],
            "kind": "port", "runs": runs, "usable_cores": usable,
            "sample": f"oracle (fp32 torch CPU restatement), cfg1: ddpm B=1 512^2 fwd+bwd, 1 warm-up + {timed_steps} timed steps per "
                      f"thread count {[r['threads'] for r in runs]}; value = best ({best['s_per_step_512']} s/step at {best['threads']} "
                      f"threads) scaled to 1024^2-equivalent images by the FLOP ratio 4.77/20.28; weight init {t_init:.1f} s untimed"}


def rccl_info(max_lines: int = 6):
    """what RCCL said about the algorithm / protocol / channels it picked (NCCL_DEBUG=INFO lines of this rank's log file), or None"""
    path = os.environ.get("SDXL_RCCL_LOG")
    if not path or not os.path.exists(path):
        return None
    import re
    out = []
    with open(path, errors="replace") as f:
        for line in f:
            if re.search(r"(?i)\b(algo|proto|ring|tree|channel|xgmi|p2p)", line):
                out.append(line.strip()[-200:])
                if len(out) >= max_lines:
                    break
    return out or None


def clock_probe(step, steps: int = 10):
    """Shader clock / package power while `steps` more (untimed) steps run: one rocm-smi call per ~0.3 s from a thread of this process.
    The step runs at the package power cap on this chip (profiles/r04j_clock_under_load.txt); the sustained clock scales every peak."""
    import re
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                txt = subprocess.run([exe, "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if m and w:
                samples.append((int(m.group(1)), float(w.group(1))))

    th = threading.Thread(target=sampler, daemon=True)
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        th.start()
        t_end = time.perf_counter() + 1.5
        n = 0
        while n < steps or time.perf_counter() < t_end:
            step()
            n += 1
            if n % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    finally:
        stop.set()
        th.join(timeout=15)
    busy = [x for x in samples if x[1] > 600.0]      # samples taken while the chip was under load
    if not busy:
        return None
    sclk = sorted(x[0] for x in busy)[len(busy) // 2]
    watts = sorted(x[1] for x in busy)[len(busy) // 2]
    return {"sclk_mhz_under_load": sclk, "package_w_under_load": watts, "samples": len(busy), "nominal_sclk_mhz": 2400,
            "how": "rocm-smi beside untimed steps after the timed region (median of the samples above 600 W)"}


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
                    help="N > 1: zero1 (default = the trainer's default, training.shard_optimizer: DESIGN.md section 6) = reduce-scatter of the bf16 "
                         "gradient buckets overlapped with the backward + the all-gather of the parameter slices, BOTH inside the timed step "
                         "(the same bytes on the wire as an all-reduce; the sharded update itself is reported apart), or allreduce = the "
                         "all-reduce of the buckets north_star names, kept as the reference point")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1: initialise a ONE-rank process group on the backend (nccl = RCCL) and run the complete exchange through it "
                         "every step (casts, collectives on the engine's side stream, all-gather): no byte leaves the GPU, but RCCL's "
                         "stream / dtype / alignment handling is the real one.  Not the headline configuration.")
    ap.add_argument("--exchange-shadow", default=None, metavar="CH[:LDS_KB[:GBPS]]",
                    help="N = 1 only: price the co-residency of the exchange's device kernels without a node -- after every backward segment "
                         "a stand-in kernel of CH workgroups x 256 threads (LDS_KB of LDS each, default 64) streams a bucket-sized buffer on "
                         "a third stream for bucket_bytes x 2 x 7/8 / GBPS (default 300 GB/s: reduce-scatter + all-gather over 7 xGMI links)")
    ap.add_argument("--max-nchannels", type=int, default=None, help="N > 1: NCCL_MAX_NCHANNELS for RCCL (its kernels take one workgroup per channel)")
    ap.add_argument("--no-emit", action="store_true", help="N > 1 A/B: cast the fp32 gradient arena per bucket instead of bf16 wgrad epilogues")
    ap.add_argument("--profile-steps", type=int, default=1, help="extra steps with per-launch GEMM event timing")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the rocm-smi clock / power samples after the timed region")
    ap.add_argument("--gemm-mode", type=int, default=None, help="A/B runs: sdxl_set_gemm_mode (0 = 128-row kernel only)")
    ap.add_argument("--knob", action="append", default=[], help="A/B runs: id=value for sdxl_set_knob (repeatable)")
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
    if args.max_nchannels:
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.max_nchannels)
    forced = bool(args.force_exchange) and world == 1
    if (world > 1 or forced) and backend == "nccl" and rank == 0 and "NCCL_DEBUG" not in os.environ:
        # rank 0 logs RCCL's topology / algorithm choices to a file; the lines end up in config.exchange.rccl
        os.environ["SDXL_RCCL_LOG"] = f"/tmp/sdxl_rccl_{os.getpid()}.log"
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=os.environ["SDXL_RCCL_LOG"])
    if forced:
        os.environ["SDXL_FORCE_EXCHANGE"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    D.init_process_group(backend if (world > 1 or forced) else None)
    wl = WORKLOADS[args.workload]
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.lib:                     # an older build may lack newer entry points: bind what it has
        import ctypes
        lib.LIB_PATH = Path(args.lib).resolve()
        probe = ctypes.CDLL(str(lib.LIB_PATH))
        lib.SIGNATURES = {k: v for k, v in lib.SIGNATURES.items() if hasattr(probe, k)}
        lib.TEST_HOOK_SIGNATURES = {k: v for k, v in lib.TEST_HOOK_SIGNATURES.items() if hasattr(probe, k)}
    if args.knob:
        lib.use_diag()          # knobs exist in the diagnostics build only (build.py --diag; include/sdxlstep_diag.h)
    if args.gemm_mode is not None:
        lib.check(lib.load().sdxl_set_gemm_mode(args.gemm_mode))
    for kv in args.knob:
        kid, kval = kv.split("=")
        lib.check(lib.load().sdxl_set_knob(int(kid), int(kval)))
    net = NU.NativeUNet(NU.make_config(), device=local_rank)
    synth.load_synthetic(net, seed=0)                      # same weights on every rank
    buckets = wl.get("buckets", [(wl["H"], wl["W"])])
    accum = int(wl.get("accum", 1))
    for (h_, w_) in buckets:
        net.plan(wl["B"], h_, w_, 77)
    batches = [make_batch(wl, rank, dev, hw=hw) for hw in buckets]
    L = net.L

    emit = (world > 1 or forced) and not args.no_emit and hasattr(L, "sdxl_set_grad_emit")

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
    sync = D.make_grad_sync(net.param_elems, cast, torch.bfloat16, dev, sharded=args.exchange == "zero1",
                            segment_sizes=[n for _o, n in net.segment_ranges()], force=forced)
    sharded_exchange = isinstance(sync, D.ShardedGradSync)
    scale = 1.0 / world / accum
    micro = [0]
    shadow = None
    if args.exchange_shadow and world == 1:
        f = args.exchange_shadow.split(":")
        shadow = {"channels": int(f[0]), "lds_kb": int(f[1]) if len(f) > 1 else 64, "gbps": float(f[2]) if len(f) > 2 else 300.0}
        shadow_buf = torch.zeros(max(n for _o, n in net.segment_ranges()), dtype=torch.bfloat16, device=dev)
        comm_stream = torch.cuda.Stream(device=dev)

        def shadow_on_segment(k, off, n):
            # called with the engine's side stream current (behind the segment's weight gradients), as the real cast + collective are;
            # RCCL's kernel then runs on the process group's own stream: a third stream here
            comm_stream.wait_stream(torch.cuda.current_stream())
            us = 2.0 * n * 2 * 7 / 8 / shadow["gbps"] / 1e3              # bytes / (GB/s) -> us
            lib.check(L.sdxl_op_exchange_shadow(C.c_void_p(shadow_buf.data_ptr()), n * 2, shadow["channels"], shadow["lds_kb"] * 1024,
                                                float(us), C.c_void_p(comm_stream.cuda_stream)))

    def step():
        """one micro-step: (cycle start: zero grads) + loss prep + UNet forward + loss + UNet backward (+ at N > 1, on the cycle's
        last micro-step, the COMPLETE gradient exchange: all-reduce, or reduce-scatter + parameter all-gather)"""
        i = micro[0]
        micro[0] += 1
        b = batches[i % len(batches)]
        first, last = i % accum == 0, i % accum == accum - 1
        if first:
            net.zero_grads()
        net.forward_loss(wl["method"], b["lat"], b["noise"], b["sigma_or_t"], b["timestep"], b["ehs"], b["pooled"], b["tid"])
        exch = (world > 1 or forced) and last
        if exch and emit:
            net.set_grad_emit(sync.comm, 1.0)
        net.backward(scale, first, on_segment=sync.on_segment if exch else (shadow_on_segment if shadow else None), segment_stream=True)
        if shadow:
            if shadow.get("_ev") is None:
                shadow["_ev"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            shadow["_ev"][0].record()                       # the backward's own end on the caller's stream ...
            shadow["_ev"][1].record(comm_stream)            # ... and the last stand-in kernel's: the un-overlapped tail of the exchange
            torch.cuda.current_stream().wait_stream(comm_stream)
        if exch:
            if emit:
                net.set_grad_emit(None)
            sync.finish()
            if sharded_exchange:          # the other half of the all-reduce's bytes: all-gather of this rank's parameter slices
                sync.gather_params(net.weights)

    for _ in range(args.warmup):
        step()
    micro[0] = 0
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # per-step HIP events on the stream the step is launched on (report the median), inside the wall-clock bracket
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    step_stats = {"median_ms": round(per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]), 3),
                  "min_ms": round(per_step[0], 3), "max_ms": round(per_step[-1], 3), "n": len(per_step),
                  "how": "HIP events on the launch stream around every step of the timed region"}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    if shadow and shadow.get("_ev"):
        ev = shadow.pop("_ev")
        shadow["tail_ms_last_step"] = round(max(0.0, ev[0].elapsed_time(ev[1])), 3)
    loss = net.read_loss()[0]
    net_param_elems = int(net.param_elems)       # bf16 gradient bytes sent (and, reduced, received) per rank and exchange = 2 x this
    images = world * wl["B"] * args.steps
    value = images / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # what the package lets the chip clock under this workload (untimed, after the timed region): rocm-smi sampled beside a few more steps
    clock = None
    if world == 1 and not args.no_clock_probe:
        clock = clock_probe(step)

    # dominant-kernel roofline: per-launch HIP events around every GEMM-family launch (same stream), profiled steps
    roof = None
    if args.profile_steps > 0:
        lib.check(L.sdxl_profile_gemm_begin())
        for _ in range(args.profile_steps):
            step()
        fl, ms, n = C.c_double(), C.c_double(), C.c_int()
        lib.check(L.sdxl_profile_gemm_end(C.byref(fl), C.byref(ms), C.byref(n)))
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        traffic, traffic_commit = None, None   # fabric-side bytes of the GEMM family per step (one launch set), from the committed PMC passes
        try:                 # of this workload: FETCH_SIZE x 2 (the gfx950 correction of the microarchitecture guide) + WRITE_SIZE, both in KiB
            with open(ROOT / "profiles" / PMC_SUMMARY) as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload:
                traffic = round(pm["gemm"]["hbm_bytes_per_step"])
                traffic_commit = pm.get("commit")
        except Exception:
            pass
        stale = kernels_changed_since(traffic_commit) if traffic_commit else None
        if stale:
            traffic = None
        # two bases, both stated: `achieved` / `frac` = the flops the family EXECUTES (the up-sampler convolutions run on the low-resolution
        # image: 16 instead of 36 tap-pixels per output quad) over the event-bracketed time; `frac_algorithmic` = the reference
        # formulation's flops of the family (BASELINE.md section 2 minus attention) over the same time -- the basis of `step_mfma_frac`.
        alg = None
        if wl["flop_per_image"] == FLOP_PER_IMAGE_1024 and len(buckets) == 1 and ms.value > 0:
            alg = (FLOP_PER_IMAGE_1024 - ATTN_FLOP_PER_IMAGE_1024) * wl["B"] * args.profile_steps / (ms.value * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "flops_basis": "executed (frac), algorithmic = reference formulation (frac_algorithmic)",
                "achieved_algorithmic": None if alg is None else round(alg, 1),
                "frac_algorithmic": None if alg is None else round(alg / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_note": ("the committed PMC summary was taken at commit %s and the kernel sources changed since: re-run "
                                 "profiles/tools/measure_step.sh" % traffic_commit) if stale else
                                ("bytes per step over all launches of the family (fabric side, MALL hits included), PMC passes at commit %s; "
                                 "algorithmic operand + result bytes ~94e9" % traffic_commit),
                "kernel": "gemm_kernel<NT|NN|TN, conv|linear> (128-row tiles) + gemm256_kernel (256x256 tiles) + conv_wgrad3_kernel (3x3 wgrad, three taps per workgroup) + wgrad256_kernel (long-reduction linear wgrad, 256x160 tiles) + cr256_kernel (co-resident 256-row tiles: level-2 linear wgrads) + pl_kernel (software-pipelined one-wave-per-SIMD 128x160 tiles: one-round linear forward / dgrad): bf16 MFMA 16x16x32",
                "launches_per_step": n.value // args.profile_steps,
                "gemm_ms_per_step": round(ms.value / args.profile_steps, 2),
                "gemm_tflop_per_step": round(fl.value / args.profile_steps / 1e12, 2)}

    # row f1, measured OUTSIDE the timed region (the metric excludes the optimizer): the fused AdamW_BF16 update of all
    # parameters, HBM-bound: 20 algorithmic bytes per element (p, m, v, shift bf16 in + out, fp32 gradient in)
    opt_extra = None
    if not args.no_optimizer:
        from sdxl_amd.optimizer import AdamWBF16
        opt = AdamWBF16(net, lr=4e-7, weight_decay=0.01)
        sharded = (world > 1 or forced) and sharded_exchange

        def update():
            if sharded:           # ZeRO-1: this rank's slices of every bucket, then the parameters are all-gathered
                opt.step(sync.reduced(), pieces=sync.pieces)
                sync.gather_params(net.weights)
            else:
                opt.step(sync.reduced() if (world > 1 or forced) else None)

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
                          "exchange": ({"shadow": shadow, "what": "single GPU: a stand-in kernel per backward segment on a third stream (no bytes "
                                        "leave the GPU); the step time beside it prices the exchange kernels' co-residency"} if shadow else None)
                          if (world == 1 and not forced) else {
                              "what": ("reduce-scatter of the bf16 gradient buckets overlapped with the backward + all-gather of the parameter slices, "
                                       "both inside the timed step (ZeRO-1 wire pattern; the sharded update is in `optimizer`)") if sharded_exchange
                                      else "all-reduce of the bf16 gradient buckets overlapped with the backward, complete inside the timed step",
                              "every_n_micro_steps": accum, "forced_single_rank": forced,
                              "exchange_bytes_timed": 2 * net_param_elems,     # bf16 gradient arena per rank and exchange (in: RS / AR; out: AG / AR)
                              "backend": backend, "rccl": rccl_info(),
                              "rccl_env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS",
                                                                         "RCCL_MSCCL_ENABLE") if os.environ.get(k) is not None}},
                          "weights": "synthetic (counter-hash init of the 2,567,463,684-parameter SDXL-base UNet)"},
               "step_time": step_stats,
               "step_tflops_per_gpu": round(step_tflops, 1),
               "step_mfma_frac": round(step_tflops / PEAK_BF16_TFLOPS, 4),
               "loss": loss, "roofline": roof, "optimizer": opt_extra}
        if clock:
            clock["peak_at_sclk_tflops"] = round(PEAK_BF16_TFLOPS * clock["sclk_mhz_under_load"] / clock["nominal_sclk_mhz"], 1)
            clock["step_mfma_frac_at_sclk"] = round(step_tflops / clock["peak_at_sclk_tflops"], 4)
            out["clock"] = clock
        if world == 1 and not args.no_cpu_baseline:
            del net, sync
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.cpu_threads or None)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
