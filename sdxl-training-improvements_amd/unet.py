"""Host-side owner of one libsdxlstep handle: packed parameters, plans per bucket shape, the training step.

Replaces what the reference reaches through `model.unet` (models/sdxl.py:40-62): parameters()/state_dict(),
the forward call at ddpm_trainer.py:320-325 / flow_matching_trainer.py:400-405, and loss.backward().
PyTorch is used for device memory and streams only; every FLOP of the step runs in libsdxlstep.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import lib

SDXL_BASE_CFG = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                     transformer_layers=(0, 2, 10), head_dim=64, cross_attention_dim=2048, norm_num_groups=32,
                     addition_time_embed_dim=256, pooled_dim=1280, resnet_eps=1e-5, tf_gn_eps=1e-6, ln_eps=1e-5)

METHODS = {"ddpm": 0, "flow_matching": 1}
PRED_TYPES = {"epsilon": 0, "v_prediction": 1}


def make_config(**over) -> lib.UNetConfig:
    d = dict(SDXL_BASE_CFG)
    d.update(over)
    c = lib.UNetConfig()
    for k, v in d.items():
        if isinstance(v, (tuple, list)):
            arr = getattr(c, k)
            for i, x in enumerate(v):
                arr[i] = int(x)
        else:
            setattr(c, k, v)
    return c


def config_from_unet(unet, sd=None) -> lib.UNetConfig:
    """sdxl_unet_config of a PyTorch / diffusers UNet2DConditionModel: from its `.config` where present (diffusers), else
    from the shapes of its diffusers-keyed state dict (what models/sdxl.py:25-40 loads)."""
    sd = sd if sd is not None else unet.state_dict()
    c = getattr(unet, "config", None)
    get = (lambda k, d=None: (c.get(k, d) if isinstance(c, dict) else getattr(c, k, d))) if c is not None else (lambda k, d=None: d)
    ch = get("block_out_channels")
    if ch is None:
        ch = [sd["conv_in.weight"].shape[0]]
        for i in (1, 2):
            k = f"down_blocks.{i}.resnets.0.conv1.weight"
            if k in sd:
                ch.append(sd[k].shape[0])
    tl = get("transformer_layers_per_block")
    if tl is None or isinstance(tl, int):
        tl = []
        for i in range(len(ch)):
            n = 0
            while f"down_blocks.{i}.attentions.0.transformer_blocks.{n}.norm1.weight" in sd:
                n += 1
            tl.append(n)
    cross = get("cross_attention_dim")
    if cross is None or isinstance(cross, (list, tuple)):
        k = next(k for k in sd if k.endswith("attn2.to_k.weight"))
        cross = sd[k].shape[1]
    ad = get("addition_time_embed_dim", 256)
    add_in = get("projection_class_embeddings_input_dim", None) or sd["add_embedding.linear_1.weight"].shape[1]
    head = get("attention_head_dim", 64)
    head = 64 if isinstance(head, (list, tuple)) else head           # SDXL: heads = C / 64 at every level
    return make_config(block_out_channels=tuple(int(x) for x in ch), transformer_layers=tuple(int(x) for x in tl),
                       layers_per_block=int(get("layers_per_block", 2)), cross_attention_dim=int(cross),
                       addition_time_embed_dim=int(ad), pooled_dim=int(add_in) - 6 * int(ad),
                       norm_num_groups=int(get("norm_num_groups", 32)), head_dim=64 if head in (5, 10, 20) else int(head))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeUNet:
    """SDXL UNet + loss, forward and backward, on one MI355X.

    Memory (all torch-allocated so torch.distributed can reduce the gradient arena in place):
      weights  bf16 packed arena (5.1 GB for SDXL-base), grads fp32 arena (10.3 GB),
      workspace = activations + activation gradients + statistics of the largest planned bucket.
    """

    def __init__(self, cfg: Optional[lib.UNetConfig] = None, device: int = 0):
        if not torch.cuda.is_available():
            raise lib.SdxlError("NativeUNet needs a GPU (no CPU fallback)")
        self.L = lib.load()
        self.cfg = cfg if cfg is not None else make_config()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        lib.check(self.L.sdxl_create(C.byref(self.cfg), device, C.byref(h)), "sdxl_create")
        self.h = h
        wb, gb = C.c_size_t(), C.c_size_t()
        lib.check(self.L.sdxl_param_bytes(self.h, C.byref(wb), C.byref(gb)))
        self.param_elems = wb.value // 2
        self.weights = torch.zeros(self.param_elems, dtype=torch.bfloat16, device=self.device)
        self.grads = torch.zeros(self.param_elems, dtype=torch.float32, device=self.device)
        lib.check(self.L.sdxl_bind_params(self.h, _ptr(self.weights), _ptr(self.grads)), "sdxl_bind_params")
        self.workspace: Optional[torch.Tensor] = None
        self._plans: Dict[Tuple[int, int, int, int], int] = {}
        self._cur: Optional[Tuple[int, int, int, int]] = None
        self._keep = []          # tensors whose device pointers the library still references
        self.param_table = self._read_param_table()

    # ------------------------------------------------------------------ parameters
    def _read_param_table(self):
        n = self.L.sdxl_num_params(self.h)
        out = {}
        buf = C.create_string_buffer(256)
        nd = C.c_int()
        shp = (C.c_long * 4)()
        for i in range(n):
            lib.check(self.L.sdxl_param_info(self.h, i, buf, 256, C.byref(nd), shp))
            out[buf.value.decode()] = tuple(int(shp[k]) for k in range(nd.value))
        return out

    def param_ranges(self) -> Dict[str, Tuple[int, int]]:
        """{diffusers key: (element offset, element count)} of every tensor inside the packed weight / gradient arenas."""
        out = {}
        off, cnt = C.c_size_t(), C.c_size_t()
        for i, name in enumerate(self.param_table):
            lib.check(self.L.sdxl_param_range(self.h, i, C.byref(off), C.byref(cnt)))
            out[name] = (int(off.value), int(cnt.value))
        return out

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """{diffusers state-dict key: shape} -- same keys/shapes as unet.state_dict() in the reference."""
        return dict(self.param_table)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        missing = [k for k in self.param_table if k not in sd]
        extra = [k for k in sd if k not in self.param_table]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]}... extra {extra[:5]}...")
        for k, t in sd.items():
            if k in self.param_table:
                self.load_weight(k, t)
        torch.cuda.current_stream().synchronize()

    def load_weight(self, name: str, t: torch.Tensor) -> None:
        if tuple(t.shape) != self.param_table[name]:
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {self.param_table[name]}")
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.to(self.device).contiguous()
        lib.check(self.L.sdxl_load_weight(self.h, name.encode(), _ptr(t), 0 if t.dtype == torch.float32 else 1,
                                          _stream()), f"load {name}")
        torch.cuda.current_stream().synchronize()   # t may be a temporary

    def export(self, name: str, grad: bool = False, dtype=torch.float32) -> torch.Tensor:
        out = torch.empty(self.param_table[name], dtype=dtype, device=self.device)
        fn = self.L.sdxl_export_grad if grad else self.L.sdxl_export_weight
        lib.check(fn(self.h, name.encode(), _ptr(out), 0 if dtype == torch.float32 else 1, _stream()))
        return out

    def state_dict(self, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
        return {k: self.export(k, False, dtype) for k in self.param_table}

    def grad_dict(self, dtype=torch.float32) -> Dict[str, torch.Tensor]:
        return {k: self.export(k, True, dtype) for k in self.param_table}

    # ------------------------------------------------------------------ plans
    def plan(self, B: int, H: int, W: int, ctx: int = 77) -> None:
        key = (B, H, W, ctx)
        if self._cur == key:
            return
        need = C.c_size_t()
        lib.check(self.L.sdxl_plan(self.h, B, H, W, ctx, C.byref(need)), "sdxl_plan")
        self._plans[key] = need.value
        if self.workspace is None or self.workspace.numel() < need.value:
            torch.cuda.current_stream().synchronize()
            self.workspace = None
            self.workspace = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        lib.check(self.L.sdxl_bind_workspace(self.h, _ptr(self.workspace), self.workspace.numel()))
        self._cur = key

    @property
    def num_segments(self) -> int:
        return self.L.sdxl_num_segments(self.h)

    def segment_range(self, k: int) -> Tuple[int, int]:
        off, n = C.c_size_t(), C.c_size_t()
        lib.check(self.L.sdxl_segment_range(self.h, k, C.byref(off), C.byref(n)))
        return off.value, n.value

    def segment_ranges(self):
        """[(offset, count)] of every backward segment, in exchange order."""
        return [self.segment_range(k) for k in range(self.num_segments)]

    # ------------------------------------------------------------------ the step
    def zero_grads(self) -> None:
        lib.check(self.L.sdxl_zero_grads(self.h, _stream()))

    def _batch(self, latents, noise, sigma_or_t, timestep, prompt_embeds, pooled, time_ids, tag_weights):
        d = self.device
        B, _, H, W = latents.shape
        f32 = lambda t: None if t is None else t.to(d, torch.float32).contiguous()
        b16 = lambda t: None if t is None else t.to(d, torch.bfloat16).contiguous()
        ts = [f32(latents), f32(noise), f32(sigma_or_t).reshape(-1), f32(timestep).reshape(-1), b16(prompt_embeds),
              b16(pooled).reshape(B, -1), f32(time_ids).reshape(B, 6), f32(tag_weights)]
        ctx = ts[4].shape[1]
        self.plan(B, H, W, ctx)
        self._keep = ts
        return lib.Batch(B, H, W, ctx, *[None if t is None else t.data_ptr() for t in ts])

    def forward_loss(self, method: str, latents, noise, sigma_or_t, timestep, prompt_embeds, pooled, time_ids,
                     tag_weights=None, prediction_type="v_prediction", min_snr_gamma: Optional[float] = 5.0,
                     use_ztsnr=True) -> None:
        """loss preparation + UNet forward + loss; results stay on the device until read_loss()."""
        lc = lib.LossConfig(METHODS[method], PRED_TYPES.get(prediction_type, 0), int(min_snr_gamma is not None),
                            float(min_snr_gamma or 0.0), int(bool(use_ztsnr)))
        b = self._batch(latents, noise, sigma_or_t, timestep, prompt_embeds, pooled, time_ids, tag_weights)
        lib.check(self.L.sdxl_forward_loss(self.h, C.byref(lc), C.byref(b), _stream()), "sdxl_forward_loss")

    def backward(self, grad_scale: float = 1.0, first_micro: bool = True, on_segment=None, segment_stream: bool = False) -> None:
        """All backward segments in reverse execution order; `on_segment(k, offset, count)` is called after segment
        k's kernels are enqueued (used to start that bucket's gradient exchange under the rest of backward).
        segment_stream: the callbacks run with the engine's SIDE stream as torch's current stream (join mode 2): the
        bucket's cast and collective are ordered behind the segment's weight gradients there, and the caller's stream --
        the backward's critical path -- neither waits for the side stream nor runs the casts (3.5 ms of them per step)."""
        side = self._side_stream() if (segment_stream and on_segment is not None) else None
        lib.check(self.L.sdxl_set_join_mode(self.h, 0 if on_segment is not None else 1) if side is None
                  else self.L.sdxl_set_join_mode(self.h, 2))
        if on_segment is None and hasattr(self.L, "sdxl_backward_all"):     # no per-segment exchange: one call for the whole backward
            lib.check(self.L.sdxl_backward_all(self.h, float(grad_scale), int(first_micro), _stream()), "backward")
            return
        for k in range(self.num_segments):
            lib.check(self.L.sdxl_backward_segment(self.h, k, float(grad_scale), int(first_micro), _stream()),
                      f"backward segment {k}")
            if on_segment is not None:
                if side is None:
                    on_segment(k, *self.segment_range(k))
                else:
                    with torch.cuda.stream(side):
                        on_segment(k, *self.segment_range(k))
        if side is not None:      # whatever the callbacks left on the side stream (the last cast; host-staged copies of the
            torch.cuda.current_stream().wait_stream(side)      # gloo test transport) is ordered before the caller's next work

    def set_grad_emit(self, arena: Optional[torch.Tensor], scale: float = 1.0) -> None:
        """Exchange micro-step without the cast pass: the weight-gradient GEMMs of the next backward write their final
        value x scale as bf16 into `arena` (param_elems bf16, the exchange arena) and leave the fp32 arena alone; None = off.
        `cast_small` then covers the biases / norm parameters of a segment."""
        if arena is not None:
            assert arena.dtype == torch.bfloat16 and arena.numel() >= self.param_elems and arena.is_contiguous()
        lib.check(self.L.sdxl_set_grad_emit(self.h, None if arena is None else _ptr(arena), float(scale)))

    def cast_small(self, off: int, n: int, dst: torch.Tensor, scale: float = 1.0) -> None:
        lib.check(self.L.sdxl_small_grads_to_bf16(self.h, off, n, _ptr(dst), float(scale), _stream()))

    def _side_stream(self):
        if getattr(self, "_side_ext", None) is None:
            p = C.c_void_p()
            lib.check(self.L.sdxl_side_stream(self.h, C.byref(p)))
            self._side_ext = torch.cuda.ExternalStream(p.value, device=self.device) if p.value else False
        return self._side_ext or None

    def set_graph_mode(self, on: bool) -> None:
        """hipGraph replay of forward / backward (default off: eager two-stream launches measured faster on ROCm 7.2)."""
        lib.check(self.L.sdxl_set_graph_mode(self.h, int(bool(on))))

    def read_loss(self):
        out = (C.c_float * 8)()
        lib.check(self.L.sdxl_read_loss(self.h, out, _stream()))
        return [float(x) for x in out]

    # UNet only (sample NCHW fp32/bf16 in, NCHW fp32 out) -- for parity tests and validation sampling
    def unet_forward(self, sample, timestep, prompt_embeds, pooled, time_ids) -> torch.Tensor:
        B, Cc, H, W = sample.shape
        d = self.device
        dummy = torch.zeros(B, 4, H, W, device=d)
        b = self._batch(dummy, dummy, torch.zeros(B), timestep, prompt_embeds, pooled, time_ids, None)
        x8 = torch.zeros(B * H * W, 8, dtype=torch.bfloat16, device=d)
        x8[:, :4] = sample.to(d).permute(0, 2, 3, 1).reshape(B * H * W, 4).to(torch.bfloat16)
        out8 = torch.empty_like(x8)
        lib.check(self.L.sdxl_unet_forward(self.h, _ptr(x8), C.byref(b), _ptr(out8), _stream()), "sdxl_unet_forward")
        return out8[:, :4].float().reshape(B, H, W, 4).permute(0, 3, 1, 2).contiguous()

    def unet_backward(self, dpred_nchw: torch.Tensor, first_micro: bool = True) -> None:
        B, Cc, H, W = dpred_nchw.shape
        d8 = torch.zeros(B * H * W, 8, dtype=torch.bfloat16, device=self.device)
        d8[:, :4] = dpred_nchw.to(self.device).permute(0, 2, 3, 1).reshape(B * H * W, 4).to(torch.bfloat16)
        lib.check(self.L.sdxl_unet_backward(self.h, _ptr(d8), int(first_micro), _stream()), "sdxl_unet_backward")
        torch.cuda.current_stream().synchronize()

    def grad_norm(self) -> float:
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        lib.check(self.L.sdxl_grad_sumsq(self.h, _ptr(out), _stream()))
        return float(out.sqrt())

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize()
            self.L.sdxl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
