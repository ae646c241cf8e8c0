"""AdamWBF16 on the packed arenas: host mirror of the reference optimizer class, arithmetic in csrc/optimizer.hip.

Reference: src/training/optimizers/adamw_bfloat16/__init__.py (class AdamWBF16, `_make_step`) and stochastic/__init__.py.
Same constructor arguments, `step()`, `zero_grad()`, `state_dict()` / `load_state_dict()`, `param_groups` (a real list
here -- the reference's property returns an iterator).  State is three bf16 arenas (exp_avg, exp_avg_sq,
shift) laid out like the weight arena, so one fused launch updates all 2.567 B parameters; the reference's per-tensor
lazy weight decay (`accumulated_decay`, paid when it exceeds 5e-3, random per-tensor phase) is kept per tensor on the
host and paid with a small per-range kernel on the rare steps it is due.
There is no PyTorch fallback: the step fails loudly without libsdxlstep.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import torch

from . import lib


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class AdamWBF16:
    decay_threshold = 5e-3                                   # adamw_bfloat16/__init__.py:27

    def __init__(self, net, *, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, reference_ema: bool = True,
                 grad_round_bf16: bool = False, seed: int = 0):
        """reference_ema (config key optimizer.reference_ema, default True): True reproduces the reference's ACTUAL first
        moment, m <- SR(g + (1-beta1) * beta1 * m) -- `add_stochastic_(_input, other, alpha)` computes other + alpha*_input
        (stochastic/__init__.py:96), i.e. almost no momentum; False selects the documented EMA
        m <- SR(beta1 * m + (1-beta1) * g)."""
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        self.net = net
        self.L = getattr(net, "L", None)                      # the loaded libsdxlstep; step() refuses to run without it
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        w = net.weights
        assert w.dtype == torch.bfloat16, "only bfloat16 is supported."          # :98
        self.exp_avg = torch.zeros_like(w)
        self.exp_avg_sq = torch.zeros_like(w)
        self.shift = torch.zeros_like(w)                      # true value is p + shift (:108-112)
        self.step_count = 0
        self.reference_ema = bool(reference_ema)
        self.grad_round_bf16 = bool(grad_round_bf16)
        self.seed = int(seed)
        self.ranges = net.param_ranges() if hasattr(net, "param_ranges") else {}
        # each tensor starts its decay account at a random phase so that they do not all pay at once (:116-119)
        g = torch.Generator().manual_seed(self.seed)
        self.accumulated_decay = {k: float(torch.rand([], generator=g) * self.decay_threshold) for k in self.ranges}
        self._post_step_hooks = []

    def register_step_post_hook(self, fn) -> None:
        """fn(optimizer) after every step() (same idea as torch.optim.Optimizer.register_step_post_hook)."""
        self._post_step_hooks.append(fn)

    # ------------------------------------------------------------------ reference surface
    def zero_grad(self, set_to_none: bool = False) -> None:
        self.net.zero_grads()

    def state_dict(self) -> Dict[str, Any]:
        return {"state": {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                          "shift": self.shift, "accumulated_decay": dict(self.accumulated_decay)},
                "param_groups": self.param_groups}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        st = sd["state"]
        self.step_count = int(st["step"])
        for k in ("exp_avg", "exp_avg_sq", "shift"):
            getattr(self, k).copy_(st[k])
        self.accumulated_decay = dict(st["accumulated_decay"])
        self.param_groups = sd["param_groups"]

    @torch.no_grad()
    def step(self, grads: Optional[torch.Tensor] = None, grad_scale: Optional[torch.Tensor] = None,
             zero_grad: bool = False, _rand: Optional[torch.Tensor] = None, pieces=None) -> None:
        """One update of every parameter.  grads: None = the net's fp32 gradient arena, or a bf16 / fp32 tensor in
        arena layout (the all-reduced bf16 gradients under data parallelism).  grad_scale: optional 1-element device
        tensor multiplied into the gradient inside the kernel (clip coefficient, 1/accumulation).
        pieces (ZeRO-1, distributed.ShardedGradSync.pieces): [(arena offset, count, offset into `grads`)] -- only those
        ranges of the arenas are updated, `grads` then holds just this rank's reduce-scattered shard; the stochastic-rounding
        counters are keyed by arena index, so the union over ranks is bit-identical to the unsharded update."""
        if self.L is None:
            raise lib.SdxlError("AdamWBF16.step needs libsdxlstep.so (there is no PyTorch fallback for the optimizer step)")
        grp = self.param_groups[0]
        g = self.net.grads if grads is None else grads
        if g.dtype not in (torch.float32, torch.bfloat16) or (pieces is None and g.numel() != self.net.weights.numel()):
            raise ValueError("grads must be an fp32 or bf16 tensor in arena layout")
        self.step_count += 1
        lr, wd = float(grp["lr"]), float(grp["weight_decay"])
        due = []
        for k in self.accumulated_decay:                      # :121-126, per tensor
            acc = self.accumulated_decay[k] + wd * lr
            d = acc if acc > self.decay_threshold else 0.0
            self.accumulated_decay[k] = acc - d
            if d > 0:
                due.append((k, d))
        cfg = lib.AdamWConfig()
        lib.check(self.L.sdxl_adamw_default_config(C.byref(cfg)))
        cfg.lr, (cfg.beta1, cfg.beta2), cfg.eps = lr, grp["betas"], float(grp["eps"])
        cfg.step = float(self.step_count)
        cfg.decay_this_iteration = 0.0
        cfg.reference_ema = int(self.reference_ema)
        cfg.grad_round_bf16 = int(self.grad_round_bf16)
        cfg.seed = self.seed
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if self.net.weights.is_cuda else None
        gsz = g.element_size()
        todo = [(0, self.net.weights.numel(), 0)] if pieces is None else list(pieces)
        for off, cnt, goff in todo:
            cfg.elem_offset = off
            at = lambda t, o=off: C.c_void_p(t.data_ptr() + 2 * o)
            lib.check(self.L.sdxl_adamw_bf16_step(at(self.net.weights), C.c_void_p(g.data_ptr() + gsz * goff),
                                                  0 if g.dtype == torch.float32 else 1, at(self.exp_avg), at(self.exp_avg_sq),
                                                  at(self.shift), cnt, C.byref(cfg), _ptr(grad_scale), _ptr(_rand), st),
                      "sdxl_adamw_bf16_step")
        for k, d in due:                                       # :191-193 `shift.add_(p, alpha=-decay)`, on the owned part
            toff, tcnt = self.ranges[k]
            for off, cnt, _g in todo:
                lo, hi = max(off, toff), min(off + cnt, toff + tcnt)
                if lo < hi:
                    lib.check(self.L.sdxl_adamw_decay(C.c_void_p(self.shift.data_ptr() + 2 * lo),
                                                      C.c_void_p(self.net.weights.data_ptr() + 2 * lo), hi - lo, d, st), "sdxl_adamw_decay")
        if zero_grad:
            self.net.zero_grads()
        for fn in self._post_step_hooks:
            fn(self)
