"""Batch producer for the reference's cached-latent format (row f2): the step in front of compute_loss().

Reads what `CacheManager.save_latents` wrote (src/data/preprocessing/cache_manager.py:289-403):

    <cache_dir>/cache_index.json                  zlib-compressed JSON (or plain JSON, the old format, :656-676):
                                                  {"entries": {md5(image path): {"vae_latent_path", "clip_latent_path",
                                                   "metadata_path" (relative to latents/), "bucket_info", "tag_info", ...}}}
    <cache_dir>/latents/vae/<md5>.pt              torch.save({"vae_latents": [4,h,w], "time_ids": [1,6]})
    <cache_dir>/latents/clip/<md5>.pt             torch.save({"prompt_embeds": [77,2048], "pooled_prompt_embeds": [1280]})
    <cache_dir>/latents/metadata/<md5>.json       {"text", "bucket_info", ...}

and yields exactly the dicts the reference feeds its trainers: `load_tensors` (:404-510), bucket grouping for cached
entries (bucket_utils.py:205-216), `BucketBatchSampler` (samplers.py:8-61: one bucket per batch, incomplete batches
dropped, whole batches shuffled with the `random` module) and `collate_fn` (dataset.py:197-249).  On top of that, what the
reference leaves to a DataLoader: a pinned, double-buffered host->device prefetcher so the copy of batch k+1 overlaps the
step of batch k.  Pinned fixture: tests/golden/latent_cache (written and read back by the reference's own code).

Reference defect handled (SURVEY D-list): `__getitem__` hands the *image path* to `load_tensors`, whose index is keyed by
md5(path) (dataset.py:172-179 vs cache_manager.py:410) -- `load_tensors` here accepts either.
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import zlib
from pathlib import Path
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import torch

REQUIRED_KEYS = ("vae_latents", "prompt_embeds", "pooled_prompt_embeds", "time_ids", "metadata")
_EMPTY_TAGS = {"tags": {"subject": [], "style": [], "quality": [], "technical": [], "meta": []}}


def cache_key(path) -> str:
    """cache_manager.py:726-729: md5 of the path string."""
    return hashlib.md5(str(Path(path)).encode()).hexdigest()


class LatentCache:
    def __init__(self, cache_dir, device: str = "cpu"):
        self.cache_dir = Path(cache_dir).absolute()
        self.latents_dir = self.cache_dir / "latents"
        self.device = device
        self.index = self._load_index(self.cache_dir / "cache_index.json")
        self.entries: Dict[str, Dict[str, Any]] = self.index.get("entries", {})

    @staticmethod
    def _load_index(path: Path) -> Dict[str, Any]:
        if not path.exists():
            raise FileNotFoundError(f"no cache index at {path}")
        raw = path.read_bytes()
        try:
            return json.loads(zlib.decompress(raw))
        except zlib.error:                         # old uncompressed format
            return json.loads(raw.decode("utf-8"))

    def __len__(self) -> int:
        return len(self.entries)

    def keys(self) -> List[str]:
        return list(self.entries)

    def _entry(self, key_or_path: str) -> Tuple[str, Dict[str, Any]]:
        e = self.entries.get(key_or_path)
        if e is not None:
            return key_or_path, e
        k = cache_key(key_or_path)
        e = self.entries.get(k)
        if e is None:
            raise RuntimeError(f"Cache entry not found for key: {key_or_path}")      # cache_manager.py:412
        return k, e

    def load_tensors(self, key_or_path: str) -> Dict[str, Any]:
        """The reference's `load_tensors`: same keys, same validation, same errors (RuntimeError)."""
        _, entry = self._entry(key_or_path)
        paths = {}
        for name, ek in (("vae", "vae_latent_path"), ("clip", "clip_latent_path"), ("metadata", "metadata_path")):
            p = self.latents_dir / entry[ek]
            if not p.exists():
                raise RuntimeError(f"File does not exist: {p}")
            if p.stat().st_size == 0:
                raise RuntimeError(f"File is empty: {p}")
            paths[name] = p
        vae = torch.load(paths["vae"], map_location=self.device)
        clip = torch.load(paths["clip"], map_location=self.device)
        with open(paths["metadata"], "r", encoding="utf-8") as f:
            meta = json.loads(f.read())
        for d, req, what in ((vae, ("vae_latents", "time_ids"), "VAE"), (clip, ("prompt_embeds", "pooled_prompt_embeds"), "CLIP"),
                             (meta, ("text", "bucket_info"), "metadata")):
            missing = [k for k in req if k not in d]
            if missing:
                raise RuntimeError(f"Invalid {what} data structure. Missing keys: {missing}")
        return {"vae_latents": vae["vae_latents"], "prompt_embeds": clip["prompt_embeds"],
                "pooled_prompt_embeds": clip["pooled_prompt_embeds"], "time_ids": vae["time_ids"],
                "metadata": {"text": meta.get("text"), "bucket_info": entry["bucket_info"],
                             "tag_info": entry.get("tag_info", _EMPTY_TAGS)}}

    def bucket_indices(self, keys_or_paths: Optional[Sequence[str]] = None) -> Dict[Tuple[int, int, int], List[int]]:
        """{(C, H, W) latent dims: [dataset indices]} for cached entries (bucket_utils.py:205-216); bucket_info's
        latent_dims are stored (W, H)."""
        items = list(keys_or_paths) if keys_or_paths is not None else self.keys()
        out: Dict[Tuple[int, int, int], List[int]] = {}
        for idx, k in enumerate(items):
            bi = self._entry(k)[1]["bucket_info"]
            out.setdefault((4, bi["latent_dims"][1], bi["latent_dims"][0]), []).append(idx)
        return out


class BucketBatchSampler:
    """samplers.py:8-61, same batch construction and the same use of the `random` module for the shuffle."""

    def __init__(self, bucket_indices: Dict[Tuple[int, ...], List[int]], batch_size: int, drop_last: bool = True,
                 shuffle: bool = True, seed: Optional[int] = None):
        """seed None: `random.shuffle` on the process-global state, exactly as the reference (single process).  With a seed the
        permutation of epoch e comes from a private random.Random(seed + e): every rank of a data-parallel job then draws the
        SAME batch order (set_epoch advances it), which rank-strided sharding needs."""
        self.bucket_indices, self.batch_size, self.drop_last, self.shuffle = bucket_indices, batch_size, drop_last, shuffle
        self.seed, self.epoch = seed, 0
        self.batches: List[List[int]] = []
        for _shape, indices in bucket_indices.items():
            if len(indices) < batch_size and drop_last:
                continue
            bb = [indices[i:i + batch_size] for i in range(0, len(indices), batch_size)]
            if drop_last and len(bb[-1]) < batch_size:
                bb = bb[:-1]
            self.batches.extend(bb)
        if not self.batches:
            raise ValueError("No valid batches created - check bucket sizes and batch size")

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __iter__(self) -> Iterator[List[int]]:
        if self.shuffle and self.seed is None:
            random.shuffle(self.batches)                       # reference behaviour (samplers.py:52-55)
            return iter(self.batches)
        if self.shuffle:
            order = list(self.batches)                          # same starting order on every rank and epoch
            random.Random(self.seed + self.epoch).shuffle(order)
            return iter(order)
        return iter(self.batches)

    def __len__(self) -> int:
        return len(self.batches)


def collate(batch: List[Optional[Dict[str, Any]]]) -> Optional[Dict[str, Any]]:
    """dataset.py:197-249: drop None items, require the five keys, stack; None when nothing valid."""
    valid = [b for b in batch if b is not None]
    if not valid:
        return None
    for item in valid:
        if not all(k in item for k in REQUIRED_KEYS):
            return None
    try:
        return {"vae_latents": torch.stack([b["vae_latents"] for b in valid]),
                "prompt_embeds": torch.stack([b["prompt_embeds"] for b in valid]),
                "pooled_prompt_embeds": torch.stack([b["pooled_prompt_embeds"] for b in valid]),
                "time_ids": torch.stack([b["time_ids"] for b in valid]),
                "metadata": [b["metadata"] for b in valid]}
    except Exception:
        return None


class CachedLatentLoader:
    """Iterable of collated batches: what `DataLoader(dataset, batch_sampler=..., collate_fn=...)` yields in the
    reference (main.py:64-71), read straight from the cache, optionally restricted to this rank's share of the batches
    (batch k goes to rank k mod world, so every rank sees same-bucket batches and the same number of steps)."""

    def __init__(self, cache: LatentCache, batch_size: int, items: Optional[Sequence[str]] = None, drop_last: bool = True,
                 shuffle: bool = True, rank: int = 0, world: int = 1, seed: Optional[int] = None):
        """world > 1 needs the same permutation on every rank: a seed is then mandatory in effect -- seed None falls back to 0
        (never to the per-process global `random` state, which differs between torchrun processes); world == 1 with seed None
        keeps the reference's global-`random` shuffle."""
        self.cache = cache
        self.items = list(items) if items is not None else cache.keys()
        if world > 1 and seed is None:
            seed = 0
        self.sampler = BucketBatchSampler(cache.bucket_indices(self.items), batch_size, drop_last, shuffle, seed=seed)
        self.rank, self.world = rank, world

    def set_epoch(self, epoch: int) -> None:
        """new permutation for the next pass (call once per epoch on every rank, like DistributedSampler.set_epoch)"""
        self.sampler.set_epoch(epoch)

    def __len__(self) -> int:
        return len(self.sampler) // self.world

    def __iter__(self):
        batches = list(iter(self.sampler))
        n = (len(batches) // self.world) * self.world
        for b in batches[self.rank:n:self.world]:
            out = collate([self._get(i) for i in b])
            if out is not None:
                yield out

    def _get(self, i: int):
        try:
            return self.cache.load_tensors(self.items[i])
        except Exception:
            return None                                 # dataset.py:181-193: a failed item is dropped, not fatal


class DevicePrefetcher:
    """Pinned, double-buffered H2D: batch k+1 is staged in page-locked memory and copied on a side stream while the
    step of batch k runs; the consumer's stream waits on the copy's event only."""

    TENSOR_KEYS = ("vae_latents", "prompt_embeds", "pooled_prompt_embeds", "time_ids")

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None

    def _stage(self, batch):
        if batch is None:
            return None
        out = dict(batch)
        if self.cuda:
            with torch.cuda.stream(self.stream):
                for k in self.TENSOR_KEYS:
                    out[k] = batch[k].pin_memory().to(self.device, non_blocking=True)
            out["_ready"] = self.stream.record_event()
        return out

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(next(it, None))
        while nxt is not None:
            cur, nxt = nxt, self._stage(next(it, None))
            ev = cur.pop("_ready", None)
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
            yield cur


def write_entry(cache_dir, image_path: str, tensors: Dict[str, torch.Tensor], text: str, bucket_info: Dict[str, Any],
                tag_info: Optional[Dict[str, Any]] = None, index: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Write one sample in the reference's on-disk layout (for synthetic caches / tests); returns the updated index
    dict, `save_index` persists it."""
    cache_dir = Path(cache_dir)
    lat = cache_dir / "latents"
    for sub in ("vae", "clip", "metadata"):
        (lat / sub).mkdir(parents=True, exist_ok=True)
    k = cache_key(image_path)
    torch.save({"vae_latents": tensors["vae_latents"].cpu(), "time_ids": tensors["time_ids"].cpu()}, lat / "vae" / f"{k}.pt")
    torch.save({"prompt_embeds": tensors["prompt_embeds"].cpu(), "pooled_prompt_embeds": tensors["pooled_prompt_embeds"].cpu()},
               lat / "clip" / f"{k}.pt")
    with open(lat / "metadata" / f"{k}.json", "w", encoding="utf-8") as f:
        json.dump({"text": text, "bucket_info": bucket_info}, f)
    index = index if index is not None else {"version": "1.0", "entries": {}}
    index["entries"][k] = {"vae_latent_path": f"vae/{k}.pt", "clip_latent_path": f"clip/{k}.pt", "metadata_path": f"metadata/{k}.json",
                           "is_valid": True, "bucket_info": bucket_info, "tag_info": tag_info}
    return index


def save_index(cache_dir, index: Dict[str, Any]) -> None:
    """cache_manager.py:613-654: compact JSON, zlib level 1, atomic replace."""
    p = Path(cache_dir) / "cache_index.json"
    tmp = p.with_suffix(".tmp")
    tmp.write_bytes(zlib.compress(json.dumps(index, separators=(",", ":"), ensure_ascii=False).encode("utf-8"), level=1))
    os.replace(tmp, p)
