"""Generate tests/golden/latent_cache/ (row f2 fixture) with the REAL reference code.

Authoring container only (needs /root/reference).  Uses, unchanged:
  src.data.preprocessing.cache_manager.CacheManager.save_latents / load_tensors  (on-disk format + reader)
  src.data.preprocessing.samplers.BucketBatchSampler                             (batch construction + shuffle)
  src.data.dataset.AspectBucketDataset.collate_fn                                (batch dict)
The cache directory it writes IS the fixture (a data format sample: .pt / .json / zlib index, a few KB); next to it
`expected.pt` holds what the reference's reader, sampler and collate return for it.  Output is data only.

Usage:  python oracle/make_cache_goldens.py
"""
from __future__ import annotations

import json
import os
import random
import shutil
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_goldens import _stub, _Blank, _Dummy, REF  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "latent_cache"


def main():
    _stub("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None, finish=lambda *a, **k: None, Image=_Dummy, run=None)
    _stub("colorama", Fore=_Blank(), Style=_Blank(), Back=_Blank(), init=lambda *a, **k: None)
    _stub("spacy", load=lambda *a, **k: None)
    _stub("diffusers", DDPMScheduler=_Dummy, StableDiffusionXLPipeline=_Dummy, AutoencoderKL=_Dummy, UNet2DConditionModel=_Dummy)
    _stub("xformers"); _stub("xformers.ops")
    scratch = tempfile.mkdtemp(prefix="refimport_")
    os.chdir(scratch)
    sys.path.insert(0, str(REF))
    from src.data.preprocessing.cache_manager import CacheManager
    from src.data.preprocessing.samplers import BucketBatchSampler
    from src.data.preprocessing.bucket_types import BucketDimensions, BucketInfo
    from src.data import dataset as ds

    if OUT.exists():
        shutil.rmtree(OUT)
    cache_dir = OUT / "cache"
    cm = CacheManager(cache_dir, config=None, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(0)
    # 7 samples in two buckets: 5 of 64x64 px (latent 8x8) and 2 of 80x48 px (latent 6 high x 10 wide)
    image_paths, items = [], []
    for i in range(7):
        w, h = (64, 64) if i not in (2, 5) else (80, 48)
        dims = BucketDimensions.from_pixels(w, h)
        info = BucketInfo(dimensions=dims, pixel_dims=(dims.width, dims.height), latent_dims=(dims.width_latent, dims.height_latent),
                          bucket_index=0 if w == 64 else 1, size_class="small", aspect_class="square" if w == h else "landscape")
        path = f"/data/train/img_{i:03d}.png"                 # only ever hashed, never opened
        tensors = {"vae_latents": torch.randn(4, h // 8, w // 8, generator=g),
                   "time_ids": torch.tensor([[h, w, 0, 0, h, w]], dtype=torch.float32),
                   "prompt_embeds": torch.randn(77, 32, generator=g).to(torch.bfloat16),
                   "pooled_prompt_embeds": torch.randn(16, generator=g).to(torch.bfloat16)}
        tags = {"tags": {"subject": ["cat"], "style": [], "quality": ["best"], "technical": [], "meta": []}} if i % 3 == 0 else None
        assert cm.save_latents(tensors, path, {"text": f"caption number {i}"}, bucket_info=info, tag_info=tags)
        image_paths.append(path)
    # what the reference reads back (fresh manager = from disk), keyed by md5 as load_tensors expects
    cm2 = CacheManager(cache_dir, config=None, device=torch.device("cpu"))
    keys = [cm2.get_cache_key(p) for p in image_paths]
    loaded = [cm2.load_tensors(k) for k in keys]
    # bucket grouping as group_images_by_bucket does for cached entries (bucket_utils.py:205-216)
    bucket_indices = {}
    for idx, k in enumerate(keys):
        bi = cm2.cache_index["entries"][k]["bucket_info"]
        bucket_indices.setdefault((4, bi["latent_dims"][1], bi["latent_dims"][0]), []).append(idx)
    random.seed(1234)
    sampler = BucketBatchSampler(bucket_indices, batch_size=2, drop_last=True, shuffle=True)
    epochs = [list(iter(sampler)), list(iter(sampler))]
    nodrop = BucketBatchSampler(bucket_indices, batch_size=2, drop_last=False, shuffle=False)
    collated = [ds.AspectBucketDataset.collate_fn(None, [loaded[i] for i in b] + [None]) for b in epochs[0]]
    torch.save({"image_paths": image_paths, "keys": keys, "loaded": loaded,
                "bucket_indices": {str(k): v for k, v in bucket_indices.items()},
                "sampler_seed": 1234, "epochs": epochs, "nodrop_batches": list(iter(nodrop)), "collated": collated},
               OUT / "expected.pt")
    # the metadata json files carry absolute paths of this container and timestamps; they are part of the format sample
    size = sum(f.stat().st_size for f in OUT.rglob("*") if f.is_file())
    print("wrote", OUT, size, "bytes", len(list(OUT.rglob("*"))), "files")


if __name__ == "__main__":
    main()
