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


def _worker_state_and_emit(rank, world, port, q):
    """(1) ZeRO-1 optimizer state: every rank owns (has updated) only its slices of an arena-layout tensor; gather_arena must give
    every rank the complete tensor (what save_checkpoint writes to optimizer.pt).  (2) make_grad_sync falls back to all-reduce
    where a segment does not split into world x 16-byte slices.  (3) trainer: emit mode (bf16 straight from the wgrad epilogues,
    fp32 arena untouched) only on a backward KNOWN to be the cycle's last; the direct .backward() loop with accumulation takes
    the fp32 accumulate + cast path on every micro-step."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import importlib
    import warnings
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    T = importlib.import_module("sdxl-training-improvements_amd.trainer")
    CFG = importlib.import_module("sdxl-training-improvements_amd.config")
    D.init_process_group("gloo")
    total = 1024
    segs = [(768, 256), (256, 512), (0, 256)]
    sync = D.make_grad_sync(total, lambda off, n, dst: dst.zero_(), torch.float32, "cpu", sharded=True, segment_sizes=[n for _o, n in segs])
    ok = isinstance(sync, D.ShardedGradSync)
    for k, (off, n) in enumerate(segs):
        sync.on_segment(k, off, n)
    sync.finish()
    state = torch.full((total,), -1.0)                       # stale everywhere ...
    for off, n, _g in sync.pieces:
        state[off:off + n] = torch.arange(off, off + n, dtype=torch.float32) + 1000.0 * rank     # ... but on the owned slices
    sync.gather_arena(state)
    expect = torch.empty(total)
    for off, cnt in segs:
        n = cnt // world
        for r in range(world):
            expect[off + r * n: off + (r + 1) * n] = torch.arange(off + r * n, off + (r + 1) * n, dtype=torch.float32) + 1000.0 * r
    ok = ok and torch.equal(state, expect)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        fb = D.make_grad_sync(total, lambda *a: None, torch.float32, "cpu", sharded=True, segment_sizes=[256, 24])
    ok = ok and type(fb) is D.GradSync and len(wl) == 1

    class Net:                                             # stand-in for NativeUNet: records what the trainer asks of it
        def __init__(self):
            self.param_elems, self.device = 64, "cpu"
            self.weights = torch.zeros(64, dtype=torch.bfloat16)
            self.grads = torch.zeros(64)
            self.log = []
        def segment_ranges(self): return [(32, 32), (0, 32)]
        def zero_grads(self): pass
        def forward_loss(self, *a, **k): pass
        def read_loss(self): return [0.5, 0, 8.0, 16.0, 4.0, 9.0, 25.0, 1.0]
        def set_grad_emit(self, arena, scale=1.0): self.log.append(("emit", arena is not None))
        def cast_small(self, off, n, dst, scale=1.0): dst.zero_()
        def backward(self, scale, first, on_segment=None, segment_stream=False):
            self.log.append(("bwd", first, on_segment is not None))
        # the trainer's full-cast path goes through libsdxlstep; give it a host stand-in
        L = None
        h = None

    cfg = CFG.Config()
    cfg.training.method = "ddpm"
    cfg.training.gradient_accumulation_steps = 2
    net = Net()
    from types import SimpleNamespace
    tr = T.NativeSDXLTrainer(net, optimizer=SimpleNamespace(param_groups=[{"lr": 1e-4}]), train_dataloader=None, device="cpu", config=cfg)
    tr._cast = lambda off, n, dst: dst.zero_()
    tr.sync.cast = tr._cast
    b = {"vae_latents": torch.randn(2, 4, 8, 8), "prompt_embeds": torch.randn(2, 77, 16), "pooled_prompt_embeds": torch.randn(2, 8),
         "time_ids": torch.zeros(2, 1, 6), "metadata": {}}
    # caller-owned loop, accumulation 2: no backward may run in emit mode
    for _ in range(2):
        tr.compute_loss(b)["loss"].backward()
    ok = ok and not any(e == ("emit", True) for e in net.log)
    tr._end_cycle()
    net.log.clear()
    # the trainer's own loop: only the last micro-step exchanges, and that one emits
    tr._execute_training_step(b, accumulate=True, is_last_accumulation_step=False)
    ok = ok and not any(e[0] == "emit" for e in net.log)
    tr._execute_training_step(b, accumulate=True, is_last_accumulation_step=True)
    ok = ok and ("emit", True) in net.log
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, timeout=180, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def _worker_checkpoint(rank, world, port, q, tmp):
    """save_checkpoint must not contain a collective (the reference calls it on rank 0 only), the loop's save
    decision must be the same on every rank although their losses differ, and the optimizer.pt rank 0 writes under ZeRO-1 must
    hold every rank's moments once prepare_checkpoint() ran on all ranks."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(tmp)
    import importlib
    import warnings
    from types import SimpleNamespace
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    T = importlib.import_module("sdxl-training-improvements_amd.trainer")
    O = importlib.import_module("sdxl-training-improvements_amd.optimizer")
    CFG = importlib.import_module("sdxl-training-improvements_amd.config")
    D.init_process_group("gloo")
    total = 128

    class Net:                                            # stand-in for NativeUNet
        L = None
        h = None
        def __init__(self):
            self.param_elems, self.device = total, "cpu"
            self.weights = torch.zeros(total, dtype=torch.bfloat16)
            self.grads = torch.zeros(total)
            self.step = 0
        def segment_ranges(self): return [(64, 64), (0, 64)]
        def zero_grads(self): pass
        def forward_loss(self, *a, **k): self.step += 1
        # epoch 1 (steps 1, 2): rank 0 sees 1.0, rank 1 sees 0.2 -> mean 0.6; epoch 2: 0.5 / 2.0 -> mean 1.25.  Local decisions would
        # differ in epoch 2 (rank 0 improves on its own 1.0, rank 1 does not improve on its 0.2): the ranks must agree NOT to save.
        def read_loss(self):
            ep = (self.step - 1) // 2
            return [[1.0, 0.2][rank] if ep == 0 else [0.5, 2.0][rank], 0, 8.0, 16.0, 4.0, 9.0, 25.0, 1.0]
        def set_grad_emit(self, arena, scale=1.0): pass
        def cast_small(self, off, n, dst, scale=1.0): dst.fill_(1.0)
        def backward(self, scale, first, on_segment=None, segment_stream=False):
            if on_segment is not None:
                for k, (off, n) in enumerate(self.segment_ranges()):
                    on_segment(k, off, n)
        def state_dict(self, dtype=torch.bfloat16): return {"w": self.weights.clone()}

    class FakeFused(O.AdamWBF16):                         # the fused optimizer's host surface without libsdxlstep
        def __init__(self, net):
            self.net, self.L = net, None
            self.param_groups = [dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)]
            self.exp_avg = torch.full((total,), -1.0, dtype=torch.bfloat16)
            self.exp_avg_sq = torch.full((total,), -1.0, dtype=torch.bfloat16)
            self.shift = torch.full((total,), -1.0, dtype=torch.bfloat16)
            self.step_count, self.accumulated_decay, self._post_step_hooks = 0, {}, []
        def step(self, grads=None, grad_scale=None, pieces=None, **kw):
            self.step_count += 1
            for off, n, _g in pieces:                     # this rank's slices only: what the sharded update touches
                for t in (self.exp_avg, self.exp_avg_sq, self.shift):
                    t[off:off + n] = float(10 * self.step_count + rank)
            for fn in self._post_step_hooks:
                fn(self)

    cfg = CFG.Config()
    cfg.training.method = "ddpm"
    cfg.training.gradient_accumulation_steps = 1
    cfg.training.clip_grad_norm = 0
    net = Net()
    b = {"vae_latents": torch.randn(2, 4, 8, 8), "prompt_embeds": torch.randn(2, 77, 16), "pooled_prompt_embeds": torch.randn(2, 8),
         "time_ids": torch.zeros(2, 1, 6), "metadata": {}}
    tr = T.NativeSDXLTrainer(net, optimizer=FakeFused(net), train_dataloader=[b, b], device="cpu", config=cfg)
    tr._cast = lambda off, n, dst: dst.fill_(1.0)
    tr.sync.cast = tr._cast
    ok = tr.sharded and isinstance(tr.sync, D.ShardedGradSync)
    saved = []
    orig = tr.save_checkpoint
    tr.save_checkpoint = lambda e=0, is_final=False: (saved.append((e, is_final)), orig(e, is_final))[1]
    tr.wandb_logger = SimpleNamespace(log_metrics=lambda *a, **k: None)
    tr.train(2, save_checkpoints=True)                    # would hang (mismatched collectives) with per-rank decisions
    ok = ok and saved == [(1, False), (2, True)]          # epoch 1 improves on inf, epoch 2 does not improve on the MEAN, then the final one
    if rank == 0:
        from pathlib import Path
        sd = torch.load(str(Path("outputs") / "final_checkpoint" / "optimizer.pt"), weights_only=True)
        st = sd["state"]
        exp = torch.empty(total)
        for off, cnt in net.segment_ranges():
            n = cnt // world
            for r in range(world):
                exp[off + r * n: off + (r + 1) * n] = 10.0 * 4 + r           # 4 optimizer steps; every rank's slices present
        ok = ok and "zero1_partial" not in sd and all(torch.equal(st[k].float(), exp) for k in ("exp_avg", "exp_avg_sq", "shift"))
    dist.barrier()
    # rank 0 alone, as the reference calls it (no prepare on the others) after a further step: fails loudly AT SAVE TIME -- the
    # weights and config.json are on disk, no optimizer.pt that could not be resumed; the other ranks do nothing
    tr.optimizer.step(pieces=tr.sync.pieces)
    if rank == 0:
        from pathlib import Path
        d = Path("outputs") / "checkpoint-0007"
        try:
            orig(7, False)
            ok = False
        except RuntimeError as e:
            ok = ok and "prepare_checkpoint" in str(e) and (d / "config.json").exists() and not (d / "optimizer.pt").exists()
        # a file written by an older build with the partial marker is still refused on resume
        torch.save({"state": {}, "zero1_partial": {"rank": 0, "world": world, "pieces": []}}, str(d / "optimizer.pt"))
        try:
            tr.load_optimizer_state(d)
            ok = False
        except ValueError as e:
            ok = ok and "zero1_partial" not in str(e) and "partial" in str(e)
    else:
        ok = ok and orig(7, False) is None
    dist.barrier()
    # an epoch without a single step: every rank still enters the save decision's reduction (the guard is on the GLOBAL count, so no
    # rank can skip the collective on its own), nothing improves on `best`, the final checkpoint is still written
    tr.train_dataloader = []
    saved.clear()
    tr.save_checkpoint = lambda e=0, is_final=False: (saved.append((e, is_final)), None)[1]
    tr.train(1, save_checkpoints=True)
    ok = ok and saved == [(1, True)]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_checkpoint_no_collective_and_rank_consistent_decision_world2_gloo(tmp_path):
    assert _spawn(_worker_checkpoint, 2, extra=(str(tmp_path),)) == [(0, True), (1, True)]


def _sdxl_segments():
    import json
    from pathlib import Path
    d = json.loads((Path(__file__).parent / "golden" / "sdxl_segments.json").read_text())
    return int(d["param_elems"]), [(int(o), int(n)) for o, n in d["segments"]]


def test_world8_slices_of_the_real_sdxl_segments():
    """The ZeRO-1 slicing arithmetic over the REAL SDXL-base segment sizes (tests/golden/sdxl_segments.json, written on a GPU by
    profiles/tools/dump_segments.py and pinned against the live engine by tests/test_gpu_model.py) at world = 2, 4, 8: every
    segment splits into `world` slices of whole 16-byte vectors (no all-reduce fallback), the slices tile the arena exactly, each
    starts on a 16-byte boundary of the bf16 exchange arena, and a rank's shard fits the `gshard` buffer the class allocates."""
    import importlib
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    total, segs = _sdxl_segments()
    assert total == 2567486784 or total > 2_567_463_684          # parameters + 64-element padding
    assert sorted(o for o, _n in segs)[0] == 0 and sum(n for _o, n in segs) == total
    for world in (2, 4, 8):
        assert all(n % (8 * world) == 0 for _o, n in segs), "make_grad_sync would fall back to all-reduce"
        cover = []
        for rank in range(world):
            shard = 0
            for off, cnt in segs:
                poff, n = D.ShardedGradSync.slice_of(off, cnt, world, rank)
                assert (poff * 2) % 16 == 0 and (n * 2) % 16 == 0 and n == cnt // world
                cover.append((poff, n))
                shard += n
            assert shard <= (total + world - 1) // world + 8      # the gshard allocation of ShardedGradSync.__init__
        cover.sort()
        pos = 0
        for poff, n in cover:
            assert poff == pos
            pos += n
        assert pos == total


def _worker_world8(rank, world, port, q):
    """the collectives themselves at world = 8 (gloo): the real segment list scaled down 4096 x (multiples of 64 elements kept),
    through make_grad_sync / on_segment / finish / global_sumsq / gather_arena, against locally recomputed expectations."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib
    import warnings
    D = importlib.import_module("sdxl-training-improvements_amd.distributed")
    D.init_process_group("gloo")
    _total, real = _sdxl_segments()
    sizes = [max(64, n // 4096 // 64 * 64) for _o, n in real]          # exchange order preserved
    total = sum(sizes)
    offs, pos = [], total
    for n in sizes:                                                      # reverse execution order: the last parameters first
        pos -= n
        offs.append(pos)
    segs = list(zip(offs, sizes))
    grads = [torch.randn(total, generator=torch.Generator().manual_seed(500 + r)) for r in range(world)]
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        sync = D.make_grad_sync(total, lambda off, n, dst: dst.copy_(grads[rank][off:off + n] / world), torch.float32, "cpu",
                                sharded=True, segment_sizes=sizes)
    ok = isinstance(sync, D.ShardedGradSync) and len(wl) == 0
    for k, (off, n) in enumerate(segs):
        sync.on_segment(k, off, n)
    sync.finish()
    mean = sum(grads) / world
    red = sync.reduced()
    for poff, n, goff in sync.pieces:
        ok = ok and torch.allclose(red[goff:goff + n], mean[poff:poff + n], atol=1e-5)
    sq = sync.global_sumsq(red.double().pow(2).sum().float().reshape(1))
    ok = ok and abs(float(sq) - float(mean.double().pow(2).sum())) < 1e-3 * float(mean.double().pow(2).sum())
    w = torch.full((total,), -1.0)
    for poff, n, _g in sync.pieces:
        w[poff:poff + n] = float(rank)
    sync.gather_params(w)
    for off, cnt in segs:
        n = cnt // world
        for r in range(world):
            ok = ok and bool((w[off + r * n: off + (r + 1) * n] == float(r)).all())
    # the all-reduce path over the same buckets
    ar = D.GradSync(total, lambda off, n, dst: dst.copy_(grads[rank][off:off + n] / world), comm_dtype=torch.float32, device="cpu")
    for k, (off, n) in enumerate(segs):
        ar.on_segment(k, off, n)
    ar.finish()
    ok = ok and torch.allclose(ar.reduced(), mean, atol=1e-5)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world8_gloo_over_the_sdxl_segment_list():
    assert _spawn(_worker_world8, 8, timeout=300) == [(r, True) for r in range(8)]


def test_optimizer_state_gather_fallback_and_emit_gating_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_state_and_emit, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


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
