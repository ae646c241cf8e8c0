"""Data-parallel gradient exchange: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl").

Replaces the reference's DistributedDataParallel wrap (core/distributed.py:142-163; gradient hooks fire inside
loss.backward()).  Here the backward is split into segments of ~96M parameters in reverse execution order; as soon
as a segment's kernels are enqueued its fp32 gradients are cast to bf16 (pre-scaled by 1/world, the reference's
gradients are bf16 too) and all-reduced asynchronously, so the exchange of the 5.1 GB of UNet gradients runs under
the remaining backward.  Buckets of ~190 MB keep RCCL in its bandwidth regime on the 7-link xGMI mesh.
With gradient accumulation the exchange happens on the last micro-step only.

ShardedGradSync (row f3, ZeRO-1) is the same bucketing with REDUCE-SCATTER instead of all-reduce: rank r keeps slice r of
every bucket, computes the squared norm of its slices (one float all-reduced -> the clip coefficient is identical on all
ranks by construction), runs the fused AdamW only on its slices (1/world of the 51 GB optimizer traffic) and the updated
parameters are all-gathered bucket by bucket.  On the point-to-point xGMI mesh reduce-scatter + all-gather move the same
bytes as the all-reduce they replace, without the redundant full-arena update and norm on every rank.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None) -> None:
    """env:// rendezvous as launched by torchrun (reference: core/distributed.py:88-131, backend "nccl")."""
    if dist.is_initialized():
        return
    rank, local_rank, world = env_rank_world()
    if world <= 1 and os.environ.get("SDXL_FORCE_EXCHANGE", "0") != "1":      # (a forced single-rank group: GradSync(force=True))
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)


def is_main_process() -> bool:                      # core/distributed.py:165
    return not dist.is_initialized() or dist.get_rank() == 0


def get_world_size() -> int:                        # core/distributed.py:169
    return dist.get_world_size() if dist.is_initialized() else 1


def reduce_dict(d: dict, average: bool = True) -> dict:
    """core/distributed.py:173-203: all-reduce a dict of scalars (sorted keys, one stacked tensor)."""
    if get_world_size() < 2:
        return d
    names = sorted(d)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    v = torch.tensor([float(d[k]) for k in names], dtype=torch.float32, device=dev)
    dist.all_reduce(v)
    if average:
        v /= get_world_size()
    return {k: float(x) for k, x in zip(names, v.cpu())}


class GradSync:
    """Bucketed, overlapped all-reduce of a flat gradient arena.

    cast(offset, count, dst) writes scale*grads[offset:offset+count] into dst (the HIP cast kernel on the GPU, a
    torch copy in the CPU tests); buckets are reduced in the order their segments finish."""

    def __init__(self, total_elems: int, cast: Callable[[int, int, torch.Tensor], None], comm_dtype=torch.bfloat16,
                 device="cuda", group=None, force: bool = False):
        self.world = get_world_size()
        self.group = group
        self.cast = cast
        # active: the exchange really runs.  At world size 1 there is nothing to exchange and every method is a no-op -- unless `force`
        # is set on an initialised process group: then a single rank drives the SAME calls (casts into the exchange arena, collectives on
        # the engine's side stream, sharded update, all-gather) through the backend.  That is how one GPU exercises RCCL's stream, dtype
        # and alignment checks (tests/test_gpu_rccl.py); the result equals the unexchanged step on the bf16-rounded gradients.
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.comm = torch.zeros(total_elems, dtype=comm_dtype, device=device) if self.active else None
        self.pending: List = []
        self.enabled = True

    @property
    def scale(self) -> float:
        return 1.0 / self.world

    def on_segment(self, k: int, offset: int, count: int) -> None:
        if not self.active or not self.enabled:
            return
        buf = self.comm[offset:offset + count]
        self.cast(offset, count, buf)
        self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        for w in self.pending:
            w.wait()
        self.pending.clear()

    def reduced(self) -> Optional[torch.Tensor]:
        """The averaged gradients (bf16 arena layout) after finish(); None at world size 1."""
        return self.comm


def _via_host(fn, out: torch.Tensor, inp: torch.Tensor, **kw):
    """gloo has no device-side reduce-scatter / all-gather: the CPU tests and the one-GPU control-flow test stage through host
    memory (synchronously); RCCL (backend "nccl") takes the tensors as they are."""
    if dist.get_backend(kw.get("group")) == "gloo" and out.is_cuda:
        o, i = out.cpu(), inp.cpu()
        fn(o, i, **{k: v for k, v in kw.items() if k != "async_op"})
        out.copy_(o)
        return None
    return fn(out, inp, **kw)


class ShardedGradSync(GradSync):
    """Bucketed, overlapped REDUCE-SCATTER of the gradient arena + all-gather of the updated parameters (ZeRO-1).

    Every bucket (= backward segment, sizes multiples of 8 * world elements) is reduce-scattered as soon as its segment
    is enqueued; this rank's slices land back to back in `gshard` (bf16, total / world elements)."""

    def __init__(self, total_elems: int, cast, comm_dtype=torch.bfloat16, device="cuda", group=None, force: bool = False):
        super().__init__(total_elems, cast, comm_dtype, device, group, force)
        self.rank = dist.get_rank(group) if self.active else 0
        self.gshard = torch.zeros((total_elems + self.world - 1) // self.world + 8, dtype=comm_dtype, device=device) \
            if self.active else None
        self.pieces: List = []          # (arena offset of this rank's slice, count, offset into gshard), in exchange order
        self.buckets: List = []         # (arena offset, count) of every bucket
        self._cursor = 0

    def begin(self) -> None:
        self.pieces.clear()
        self.buckets.clear()
        self._cursor = 0

    @staticmethod
    def slice_of(offset: int, count: int, world: int, rank: int):
        """(arena offset, elements) of rank `rank`'s slice of the bucket [offset, offset + count): equal slices of whole 16-byte
        vectors (count % (8 * world) == 0, checked up front by make_grad_sync), in rank order -- what reduce-scatter delivers."""
        if count % (8 * world):
            raise ValueError(f"bucket of {count} elements does not split into {world} slices of whole 16-byte vectors")
        n = count // world
        return offset + rank * n, n

    def on_segment(self, k: int, offset: int, count: int) -> None:
        if not self.active or not self.enabled:
            return
        if k == 0:
            self.begin()
        poff, n = self.slice_of(offset, count, self.world, self.rank)     # (make_grad_sync checks the sizes up front and falls back)
        buf = self.comm[offset:offset + count]
        self.cast(offset, count, buf)
        out = self.gshard[self._cursor:self._cursor + n]
        w = _via_host(dist.reduce_scatter_tensor, out, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if w is not None:
            self.pending.append(w)
        self.pieces.append((poff, n, self._cursor))
        self.buckets.append((offset, count))
        self._cursor += n

    def reduced(self) -> Optional[torch.Tensor]:
        """This rank's averaged gradient slices (bf16), `pieces` says where each lives in the arena."""
        return None if not self.active else self.gshard[:self._cursor]

    def global_sumsq(self, local_sumsq: torch.Tensor) -> torch.Tensor:
        """sum over ranks of the slices' squared norms: one float on the wire, the same bits on every rank."""
        if self.active:
            dist.all_reduce(local_sumsq, op=dist.ReduceOp.SUM, group=self.group)
        return local_sumsq

    def gather_params(self, weights: torch.Tensor) -> None:
        """all-gather the updated slices back into every rank's full weight arena, bucket by bucket."""
        self.gather_arena(weights)

    def gather_arena(self, arena: torch.Tensor) -> None:
        """all-gather this rank's slices of any arena-layout tensor (parameters after the sharded update; the optimizer's
        exp_avg / exp_avg_sq / shift before a checkpoint, so that optimizer.pt holds every rank's state, not rank 0's stale copy)
        into every rank's full copy, bucket by bucket (same buckets / slices as the exchange).  Collective: every rank calls it."""
        if not self.active:
            return
        works = []
        for (off, cnt), (poff, n, _g) in zip(self.buckets, self.pieces):
            mine = arena[poff:poff + n].clone()         # (in-place all-gather is an RCCL feature, not a gloo one)
            w = _via_host(dist.all_gather_into_tensor, arena[off:off + cnt], mine, group=self.group, async_op=True)
            if w is not None:
                works.append(w)
        for w in works:
            w.wait()


def make_grad_sync(total_elems: int, cast, comm_dtype=torch.bfloat16, device="cuda", sharded: bool = True,
                   segment_sizes: Optional[List[int]] = None, group=None, force: bool = False) -> GradSync:
    """ShardedGradSync (ZeRO-1) where every backward segment splits into `world` slices of whole 16-byte vectors, else the
    all-reduce GradSync.  Segments are multiples of 64 elements (the arena's alignment), so reduce-scatter works for world sizes
    that divide 8 (2, 4, 8); any other world size (3, 5, 6, 7, 16 = 2 nodes x 8, ...) falls back to all-reduce + the full update
    with a warning instead of failing at the first exchange."""
    world = get_world_size()
    if sharded and world > 1 and segment_sizes is not None and any(int(c) % (8 * world) for c in segment_sizes):
        import warnings
        warnings.warn(f"ZeRO-1 gradient exchange needs segment sizes divisible by 8 * world = {8 * world}; "
                      f"falling back to all-reduce + unsharded optimizer update (world size {world})")
        sharded = False
    cls = ShardedGradSync if sharded else GradSync
    return cls(total_elems, cast, comm_dtype, device, group, force)
