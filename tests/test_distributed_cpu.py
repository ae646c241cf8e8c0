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
