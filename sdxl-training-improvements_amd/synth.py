"""Deterministic synthetic UNet weights (no checkpoint exists offline).

Counter-hash RNG in pure integer arithmetic (int64 torch ops), so CPU and GPU -- and every torch version --
produce identical bytes: u = hash32(index, stream) ; w = (u-0.5)*2*sqrt(3)*std + mean, rounded to bf16.
The same recipe is restated independently in oracle/unet_ref.py (tests check they agree bit for bit).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch


def hash_uniform(n: int, stream: int, device="cpu") -> torch.Tensor:
    idx = torch.arange(n, dtype=torch.int64, device=device)
    h = (idx * 2654435761 + (stream * 40503 + 12345)) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h.to(torch.float64) + 0.5) * (1.0 / 4294967296.0)


def init_std(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(std, mean): ~1/sqrt(fan_in) for matmul weights (halved on residual-branch outputs), norms near identity."""
    if name.endswith(".bias"):
        return (0.02, 0.0)
    if ".norm" in name or name.startswith("conv_norm_out"):
        return (0.05, 1.0)
    fan_in = math.prod(shape[1:])
    gain = 0.5 if any(s in name for s in ("to_out.0", "ff.net.2", "conv2", "proj_out")) else 1.0
    return (gain / math.sqrt(fan_in), 0.0)


def synth_tensor(index: int, name: str, shape: Tuple[int, ...], seed: int = 0, device="cpu") -> torch.Tensor:
    std, mean = init_std(name, shape)
    u = hash_uniform(math.prod(shape), stream=seed * 100003 + index, device=device)
    x = (u - 0.5) * (2.0 * math.sqrt(3.0) * std) + mean
    return x.to(torch.float32).reshape(shape).to(torch.bfloat16)


def iter_synth(shapes: Dict[str, Tuple[int, ...]], order: Iterable[str], seed: int = 0, device="cpu"):
    """Yields (name, bf16 tensor).  `order` fixes the per-tensor stream index (use the diffusers key order)."""
    for i, name in enumerate(order):
        yield name, synth_tensor(i, name, shapes[name], seed, device)


def diffusers_key_order(shapes: Dict[str, Tuple[int, ...]]):
    """Canonical diffusers state-dict order of the UNet keys (what the oracle enumerates), independent of the order
    in which the native library registers them."""
    def block_keys(prefix):
        return [k for k in shapes if k.startswith(prefix)]

    def resnet(p):
        names = ["norm1", "conv1", "time_emb_proj", "norm2", "conv2", "conv_shortcut"]
        return [f"{p}.{n}.{s}" for n in names for s in ("weight", "bias") if f"{p}.{n}.{s}" in shapes]

    def transformer(p):
        out = [f"{p}.norm.weight", f"{p}.norm.bias", f"{p}.proj_in.weight", f"{p}.proj_in.bias"]
        k = 0
        while f"{p}.transformer_blocks.{k}.norm1.weight" in shapes:
            b = f"{p}.transformer_blocks.{k}"
            for n in ("norm1", "norm2", "norm3"):
                out += [f"{b}.{n}.weight", f"{b}.{n}.bias"]
            for a in ("attn1", "attn2"):
                out += [f"{b}.{a}.to_q.weight", f"{b}.{a}.to_k.weight", f"{b}.{a}.to_v.weight",
                        f"{b}.{a}.to_out.0.weight", f"{b}.{a}.to_out.0.bias"]
            out += [f"{b}.ff.net.0.proj.weight", f"{b}.ff.net.0.proj.bias", f"{b}.ff.net.2.weight", f"{b}.ff.net.2.bias"]
            k += 1
        return out + [f"{p}.proj_out.weight", f"{p}.proj_out.bias"]

    order = ["conv_in.weight", "conv_in.bias"]
    for e in ("time_embedding", "add_embedding"):
        for l in ("linear_1", "linear_2"):
            order += [f"{e}.{l}.weight", f"{e}.{l}.bias"]
    for stage in ("down_blocks", "mid_block", "up_blocks"):
        idxs = [None] if stage == "mid_block" else range(3)
        for i in idxs:
            base = stage if i is None else f"{stage}.{i}"
            if stage == "mid_block":
                order += resnet(f"{base}.resnets.0") + transformer(f"{base}.attentions.0") + resnet(f"{base}.resnets.1")
                continue
            j = 0
            while f"{base}.resnets.{j}.norm1.weight" in shapes:
                order += resnet(f"{base}.resnets.{j}")
                if f"{base}.attentions.{j}.norm.weight" in shapes:
                    order += transformer(f"{base}.attentions.{j}")
                j += 1
            for s in ("downsamplers", "upsamplers"):
                if f"{base}.{s}.0.conv.weight" in shapes:
                    order += [f"{base}.{s}.0.conv.weight", f"{base}.{s}.0.conv.bias"]
    order += ["conv_norm_out.weight", "conv_norm_out.bias", "conv_out.weight", "conv_out.bias"]
    assert sorted(order) == sorted(shapes), "key enumeration does not cover the parameter table"
    return order


def load_synthetic(net, seed: int = 0) -> None:
    """Fill a NativeUNet with synthetic weights generated on its device."""
    shapes = net.param_shapes()
    for name, t in iter_synth(shapes, diffusers_key_order(shapes), seed, device=net.device):
        net.load_weight(name, t)
