"""CPU oracle for row f1: the reference's AdamW_BF16 element arithmetic, restated in numpy.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline may import this; the product never).

Follows /root/reference/src/training/optimizers/adamw_bfloat16/__init__.py:146-197 (`_make_step`) and
stochastic/__init__.py:46-124.  Pinned: tests/test_oracle_adamw.py checks it BIT-EXACTLY against
tests/golden/adamw_bf16.npz, which oracle/make_adamw_goldens.py produced by running the reference's own `_make_step`
(the random integers the reference drew for its stochastic rounding are part of the fixture).

All state is bf16 (uint16 bit patterns here); every op is float32 arithmetic on the widened values followed by the
rounding the reference's torch op applies:
  RN  = round-to-nearest-even to bf16 (torch in-place ops on bf16 tensors)
  SR  = the reference's stochastic rounding: add r in [0, 2^16) to the fp32 bit pattern, clear the low 16 bits
        (stochastic/__init__.py:55-68)
Reference quirk reproduced (flagged D17 in DESIGN.md): `add_stochastic_(_input, other, alpha)` computes
other + alpha * _input (stochastic/__init__.py:96 `result.add_(_input, alpha=alpha)`), not _input + alpha * other as its
docstring says, so the first-moment update is  m <- SR(g + (1 - beta1) * (beta1 * m)).
Rounding details were established against torch 2.10 CPU ops (the generator of the fixture) and are marked inline:
`a + alpha * b` is one fused multiply-add in the vectorised kernels (fma32 below; the <16-element scalar tail of a CPU
tensor is not fused, so the fixture uses lengths that are multiples of 64), python scalars that meet a bf16 tensor in
add_ are cast to bf16 first, mul_ keeps them in fp32.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_rn(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = u + (np.uint32(0x7FFF) + ((u >> 16) & np.uint32(1)))
    out = (r >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, np.uint16(0x7FC0), out)
    return out


def f32_to_bf16_sr(x: np.ndarray, r16: np.ndarray) -> np.ndarray:
    """stochastic/__init__.py:46-71: int32(x bits) + r, & 0xFFFF0000 (wrap-around int32 add)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + r16.astype(np.uint32)) >> 16).astype(np.uint16)


def fma32(a, b, c):
    """float32 fused multiply-add (exact product in float64, one rounding of the sum, then to float32)."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32) if np.isscalar(b) else \
        (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _bf16_scalar(x: float) -> np.float32:
    return bf16_to_f32(f32_to_bf16_rn(np.array([x], dtype=np.float32)))[0]


def make_step(grad, p, shift, m, v, rand, *, beta1, beta2, step, lr, eps, decay, reference_ema=True):
    """One `_make_step` on uint16 bf16 arrays; rand = uint16 [4][n] (draw order: m, shift, p, shift).
    grad may be bf16 bits (uint16) or float32 (the native gradient arena).  Returns (p, shift, m, v) bit arrays."""
    g = bf16_to_f32(grad) if grad.dtype == np.uint16 else grad.astype(np.float32)
    pf, sf, mf, vf = (bf16_to_f32(a) for a in (p, shift, m, v))
    b1, b2 = F32(beta1), F32(beta2)
    # exp_avg.mul_(beta1)                                                    (__init__.py:162)
    m1 = bf16_to_f32(f32_to_bf16_rn(mf * b1))
    # add_stochastic_(exp_avg, grad, alpha=1-beta1)                          (:163, stochastic:89-98)
    a1 = F32(1.0 - beta1)
    if reference_ema:
        r = fma32(m1, a1, g)                  # grad + alpha * exp_avg   (the reference's actual argument order)
    else:
        r = fma32(g, a1, m1)                  # exp_avg + alpha * grad   (what the docstring describes)
    m2b = f32_to_bf16_sr(r, rand[0])
    m2 = bf16_to_f32(m2b)
    # exp_avg_sq.mul_(beta2).addcmul_(grad, grad.conj(), value=1-beta2)      (:164)
    v1 = bf16_to_f32(f32_to_bf16_rn(vf * b2))
    v2b = f32_to_bf16_rn(fma32(F32(1.0 - beta2) * g, g, v1))       # torch CPU: self + (value * t1) * t2, last step fused
    v2 = bf16_to_f32(v2b)
    # denominator: exp_avg_sq.sqrt().add_(eps, alpha=1)                      (:176-181)
    den = bf16_to_f32(f32_to_bf16_rn(np.sqrt(v2)))
    den = bf16_to_f32(f32_to_bf16_rn(den + _bf16_scalar(eps)))    # a python scalar added to a bf16 tensor is cast to bf16 first
    # addcdiv_stochastic_(shift, exp_avg, denom, value=-lr*denom_correction) (stochastic:106-124)
    value = F32(-lr * (1.0 - beta2 ** step) ** 0.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = sf + (value * m2) / den
    s1b = f32_to_bf16_sr(r, rand[1])
    s1 = bf16_to_f32(s1b)
    # buffer = p.clone(); add_stochastic_(p, shift)                          (:183-186)
    p1b = f32_to_bf16_sr(s1 + pf, rand[2])
    p1 = bf16_to_f32(p1b)
    # add_stochastic_(shift, buffer.sub_(p))                                 (:188-190)
    diff = bf16_to_f32(f32_to_bf16_rn(pf - p1))
    s2b = f32_to_bf16_sr(diff + s1, rand[3])
    # lazy decay: shift.add_(p, alpha=-decay)                                (:192-193)
    if decay > 0:
        s2 = bf16_to_f32(s2b)
        s2b = f32_to_bf16_rn(fma32(p1, _bf16_scalar(-decay), s2))   # alpha of a bf16 add_ is cast to bf16 as well
    return p1b, s2b, m2b, v2b


class LazyDecay:
    """Host-side bookkeeping of AdamWBF16.step (:118-126): decay is owed per step and paid once it exceeds 5e-3."""
    threshold = 5e-3

    def __init__(self, start: float = 0.0):
        self.acc = float(start)

    def next(self, weight_decay: float, lr: float) -> float:
        self.acc += weight_decay * lr
        d = self.acc if self.acc > self.threshold else 0.0
        self.acc -= d
        return d
