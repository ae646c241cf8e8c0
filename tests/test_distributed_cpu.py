"""N>1 path on CPU: world_size-2 gloo processes run GradSync over a fake segmented gradient arena and must end up
with the average of the per-rank gradients, bucket by bucket, plus reduce_dict semantics of the reference."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sdxl_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import importlib
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    D.init_process_group("gloo")
    assert D.get_world_size() == world and D.is_main_process() == (rank == 0)
    total = 1000
    segs = [(700, 300), (256, 444), (0, 256)]           # reverse execution order, like the engine's segments
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(total, generator=g)

    def cast(off, n, dst):
        dst.copy_(grads[off:off + n] * sync.scale)

    sync = D.GradSync(total, cast, comm_dtype=torch.float32, device="cpu")
    for k, (off, n) in enumerate(segs):
        sync.on_segment(k, off, n)
    sync.finish()
    all_g = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    expect = sum(all_g) / world
    ok = torch.allclose(sync.reduced(), expect, atol=1e-6)
    # accumulation: exchange disabled on non-final micro-steps
    sync.enabled = False
    sync.on_segment(0, 0, 10)
    ok = ok and len(sync.pending) == 0
    red = D.reduce_dict({"loss": float(rank + 1), "b": 2.0 * rank})
    ok = ok and abs(red["loss"] - (sum(range(1, world + 1)) / world)) < 1e-6 and abs(red["b"] - (world - 1)) < 1e-6
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_sharded(rank, world, port, q):
    """ZeRO-1 exchange on CPU: reduce-scatter per bucket, local squared norms + 1-float all-reduce, a stand-in "optimizer"
    applied to the owned slices only, all-gather of the parameters -- must equal the all-reduce path on every rank."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import importlib
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    D.init_process_group("gloo")
    total = 1024
    segs = [(768, 256), (256, 512), (0, 256)]
    grads = torch.randn(total, generator=torch.Generator().manual_seed(100 + rank))
    sync = D.ShardedGradSync(total, lambda off, n, dst: dst.copy_(grads[off:off + n] * sync.scale), comm_dtype=torch.float32, device="cpu")
    for k, (off, n) in enumerate(segs):
        sync.on_segment(k, off, n)
    sync.finish()
    expect = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    ok = len(sync.pieces) == 3 and sum(p[1] for p in sync.pieces) == total // world
    for off, n, goff in sync.pieces:                            # this rank's slices hold the averaged gradients
        ok = ok and torch.allclose(sync.reduced()[goff:goff + n], expect[off:off + n], atol=1e-6)
        ok = ok and off == [s for s in segs if s[0] <= off < s[0] + s[1]][0][0] + rank * n
    sq = sync.global_sumsq(sync.reduced().pow(2).sum().reshape(1).clone())
    ok = ok and abs(float(sq) - float(expect.pow(2).sum())) < 1e-3
    weights = torch.arange(total, dtype=torch.float32)
    for off, n, goff in sync.pieces:                            # "optimizer": w -= g on the owned slices only
        weights[off:off + n] -= sync.reduced()[goff:goff + n]
    sync.gather_params(weights)
    ok = ok and torch.allclose(weights, torch.arange(total, dtype=torch.float32) - expect, atol=1e-6)
    # a second cycle reuses the buffers from the start
    sync.on_segment(0, 768, 256)
    sync.finish()
    ok = ok and len(sync.pieces) == 1 and sync.pieces[0][2] == 0
    try:
        sync.on_segment(1, 0, 100)                              # 100 elements do not split into 2 x whole 16-byte vectors
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradsync_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gradsync_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_passthrough():
    import importlib
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    assert D.get_world_size() == 1 and D.is_main_process()
    s = D.GradSync(16, lambda o, n, d: None, device="cpu")
    s.on_segment(0, 0, 16)
    s.finish()
    assert s.reduced() is None and D.reduce_dict({"a": 1.0}) == {"a": 1.0}
