"""Does running the batch as two concurrent half-batches hide the per-launch fixed costs?  (round-6 probe, not a product path)

Three timings on one box, ddpm, 1024^2 (latent 128 x 128), step = zero grads + forward + loss + backward, no exchange:
  a. ONE handle, B = 4                                  (the shipped step)
  b. ONE handle, B = 2                                  (how well a half batch runs alone)
  c. TWO handles, B = 2 each, on two streams, enqueued interleaved (A.fwd, B.fwd, A.bwd, B.bwd): each has its own weights, gradients,
     workspace and side stream -- a LOWER bound on what a batch-split engine could gain (its weight gradients would run once over the
     whole batch; here they run twice over half the reduction length, into two gradient arenas).
usage: python profiles/tools/split_batch_probe.py [steps]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
import sdxl_amd  # noqa: E402,F401
from sdxl_amd import synth, unet as NU  # noqa: E402


def make(B, dev, seed_rank):
    net = NU.NativeUNet(NU.make_config(), device=0)
    synth.load_synthetic(net, seed=0)
    net.plan(B, 128, 128, 77)
    wl = dict(bench.WORKLOADS["ddpm_b4_1024"]); wl["B"] = B
    return net, bench.make_batch(wl, seed_rank, dev)


def fwd(net, b):
    net.zero_grads()
    net.forward_loss("ddpm", b["lat"], b["noise"], b["sigma_or_t"], b["timestep"], b["ehs"], b["pooled"], b["tid"])


def timed(fn, steps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if len(sys.argv) > 2 and sys.argv[2].startswith("b"):      # `... 10 b8`: one handle at another batch size (the ceiling argument of DESIGN.md section 10)
        B = int(sys.argv[2][1:])
        n, bt = make(B, dev, 0)

        def stepB():
            fwd(n, bt); n.backward(1.0, True)
        t = timed(stepB, steps)
        print(f"one handle  B={B}: {t:8.2f} ms per step  ({B * 1e3 / t:.2f} img/s, {B * 20.28 / t / 2.5166 * 100:.1f} % of the bf16 MFMA roofline)", flush=True)
        return
    n4, b4 = make(4, dev, 0)

    def step4():
        fwd(n4, b4); n4.backward(1.0, True)
    t4 = timed(step4, steps)
    print(f"a. one handle  B=4: {t4:8.2f} ms per step  ({4e3 / t4:.2f} img/s)", flush=True)
    del n4
    torch.cuda.empty_cache()
    nA, bA = make(2, dev, 0)

    def step2():
        fwd(nA, bA); nA.backward(1.0, True)
    t2 = timed(step2, steps)
    print(f"b. one handle  B=2: {t2:8.2f} ms per step  ({2e3 / t2:.2f} img/s)", flush=True)
    nB, bB = make(2, dev, 1)
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def pair():
        with torch.cuda.stream(sA):
            fwd(nA, bA)
        with torch.cuda.stream(sB):
            fwd(nB, bB)
        with torch.cuda.stream(sA):
            nA.backward(1.0, True)
        with torch.cuda.stream(sB):
            nB.backward(1.0, True)
    tp = timed(pair, steps)
    print(f"c. two handles B=2+2 on two streams: {tp:8.2f} ms per pair ({4e3 / tp:.2f} img/s)", flush=True)

    def pair_stag():      # staggered: B's forward is enqueued behind A's backward start (A.bwd overlaps B.fwd)
        with torch.cuda.stream(sA):
            fwd(nA, bA)
            nA.backward(1.0, True)
        with torch.cuda.stream(sB):
            fwd(nB, bB)
            nB.backward(1.0, True)
    ts = timed(pair_stag, steps)
    print(f"d. two handles, each stream enqueued whole (A then B): {ts:8.2f} ms per pair ({4e3 / ts:.2f} img/s)", flush=True)


if __name__ == "__main__":
    main()
