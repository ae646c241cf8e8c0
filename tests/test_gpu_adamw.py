"""Row f1: fused AdamW_BF16 kernel (csrc/optimizer.hip) through the C ABI.
Bit-exact against (a) the fixtures the reference's own `_make_step` produced (tests/golden/adamw_bf16.npz) and
(b) the numpy oracle on large seeded inputs, with the stochastic-rounding integers injected; plus properties of the
built-in Philox mode (unbiased rounding, determinism, decorrelation across steps)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sdxl_amd  # noqa: E402
from sdxl_amd import lib  # noqa: E402

G = np.load(Path(__file__).parent / "golden" / "adamw_bf16.npz")


@pytest.fixture(scope="module")
def L():
    return lib.load()


def dev():
    return torch.device("cuda:0")


def to_dev_bits(a):          # uint16 bit patterns -> bf16 device tensor
    return torch.from_numpy(a.astype(np.int16)).to(dev()).view(torch.bfloat16)


def bits(t):
    return t.detach().cpu().view(torch.int16).numpy().astype(np.uint16)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_step(L, p, grad, m, v, s, *, lr, b1, b2, eps, step, decay=0.0, rand=None, ref_ema=1, grad_dtype=1, round_bf16=0,
             scale=None, seed=0):
    cfg = lib.AdamWConfig()
    lib.check(L.sdxl_adamw_default_config(C.byref(cfg)))
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.step, cfg.decay_this_iteration = lr, b1, b2, eps, float(step), decay
    cfg.reference_ema, cfg.grad_round_bf16, cfg.seed = ref_ema, round_bf16, seed
    lib.check(L.sdxl_adamw_bf16_step(ptr(p), ptr(grad), grad_dtype, ptr(m), ptr(v), ptr(s), p.numel(), C.byref(cfg),
                                     ptr(scale), ptr(rand), stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", [str(c) for c in G["cases"]])
def test_reference_fixture_bit_exact(L, name):
    lr, b1, b2, eps = (float(x) for x in G[f"{name}_hyper"])
    p, m, v, s = (to_dev_bits(G[f"{name}_{k}0"]) for k in "pmvs")
    for st in range(1, int(G[f"{name}_steps"]) + 1):
        grad = to_dev_bits(G[f"{name}_grad{st}"])
        rand = torch.from_numpy(G[f"{name}_rand{st}"].astype(np.int16)).to(dev())
        run_step(L, p, grad, m, v, s, lr=lr, b1=b1, b2=b2, eps=eps, step=st, decay=float(G[f"{name}_decay{st}"]), rand=rand)
        for k, t in (("p", p), ("m", m), ("v", v), ("s", s)):
            bad = int((bits(t) != G[f"{name}_{k}{st}"]).sum())
            assert bad == 0, f"{name} step {st} {k}: {bad} elements differ from the reference"


@pytest.mark.parametrize("ref_ema", [1, 0])
def test_vs_oracle_large_fp32_grads(L, ref_ema):
    """1 M elements, fp32 gradients from the native arena format, 3 steps, gradient scale + bf16 rounding fused."""
    from oracle import adamw_ref as R
    rng = np.random.default_rng(5 + ref_ema)
    n = 1 << 20
    pb = R.f32_to_bf16_rn((rng.standard_normal(n) * 0.05).astype(np.float32))
    mb = R.f32_to_bf16_rn((rng.standard_normal(n) * 3e-4).astype(np.float32))
    vb = R.f32_to_bf16_rn((rng.random(n) * 1e-6).astype(np.float32))
    sb = R.f32_to_bf16_rn((rng.standard_normal(n) * 1e-5).astype(np.float32))
    p, m, v, s = (to_dev_bits(a) for a in (pb, mb, vb, sb))
    scale = torch.tensor([0.25], dtype=torch.float32, device=dev())
    for st in (1, 2, 3):
        g32 = (rng.standard_normal(n) * 4e-3).astype(np.float32)
        r = rng.integers(0, 1 << 16, size=(4, n), dtype=np.uint16)
        decay = 0.006 if st == 2 else 0.0
        run_step(L, p, torch.from_numpy(g32).to(dev()), m, v, s, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, step=st, decay=decay,
                 rand=torch.from_numpy(r.astype(np.int16)).to(dev()), ref_ema=ref_ema, grad_dtype=0, round_bf16=1, scale=scale)
        gq = R.f32_to_bf16_rn(g32 * np.float32(0.25))
        pb, sb, mb, vb = R.make_step(gq, pb, sb, mb, vb, r, beta1=0.9, beta2=0.999, step=float(st), lr=1e-4, eps=1e-8,
                                     decay=decay, reference_ema=bool(ref_ema))
        for k, t, w in (("p", p, pb), ("m", m, mb), ("v", v, vb), ("s", s, sb)):
            bad = int((bits(t) != w).sum())
            assert bad == 0, f"step {st} {k}: {bad}/{n} elements differ from the oracle"


def test_philox_mode_properties(L):
    """Built-in generator: deterministic for a (seed, step), different across steps / seeds, and the stochastic
    rounding is unbiased: E[p + shift] moves by the fp32 update although |update| << ulp(p)."""
    n = 1 << 20
    def fresh():
        p = torch.full((n,), 1.0, dtype=torch.bfloat16, device=dev())
        z = lambda: torch.zeros(n, dtype=torch.bfloat16, device=dev())
        return p, z(), z(), z()
    g = torch.full((n,), 1e-3, dtype=torch.float32, device=dev())
    outs = []
    for seed, step in ((1, 1), (1, 1), (1, 2), (2, 1)):
        p, m, v, s = fresh()
        run_step(L, p, g, m, v, s, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, step=step, grad_dtype=0, seed=seed)
        outs.append((bits(p).copy(), bits(s).copy(), bits(m).copy()))
    assert all((a == b).all() for a, b in zip(outs[0], outs[1])), "same (seed, step) must reproduce"
    assert (outs[0][1] != outs[2][1]).mean() > 0.2 and (outs[0][1] != outs[3][1]).mean() > 0.2
    # unbiasedness on the first moment: m <- SR(g + 0.1 * 0) = SR(1e-3): mean of the bf16 results ~ 1e-3
    p, m, v, s = fresh()
    run_step(L, p, g, m, v, s, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, step=1, grad_dtype=0, seed=7)
    mm = m.float().mean().item()
    assert abs(mm - 1e-3) < 1e-3 * 2e-3, mm                       # bf16 spacing at 1e-3 is 7.6e-6 (0.8 %): SR removes the bias
    assert m.float().unique().numel() == 2                        # the two neighbouring bf16 values
    # true value p + shift: step 1 update = -lr*sqrt(1-b2) * m/(sqrt(v)+eps) with v = 1e-3*g^2 -> ~ -1e-3 * sqrt(1e-3)*... finite, unbiased
    tv = (p.double() + s.double()).mean().item()
    assert abs(tv - 1.0) < 5e-3 and tv < 1.0


def test_decay_range(L):
    from oracle import adamw_ref as R
    rng = np.random.default_rng(3)
    n = 4096 + 24
    pb = R.f32_to_bf16_rn((rng.standard_normal(n) * 0.05).astype(np.float32))
    sb = R.f32_to_bf16_rn((rng.standard_normal(n) * 1e-4).astype(np.float32))
    p, s = to_dev_bits(pb), to_dev_bits(sb)
    off, cnt = 104, 1000                                        # an unaligned interior range, as a single tensor would be
    lib.check(L.sdxl_adamw_decay(C.c_void_p(s.data_ptr() + 2 * off), C.c_void_p(p.data_ptr() + 2 * off), cnt, 0.0075, stream()))
    torch.cuda.synchronize()
    want = sb.copy()
    al = R._bf16_scalar(-0.0075)
    want[off:off + cnt] = R.f32_to_bf16_rn(R.fma32(R.bf16_to_f32(pb[off:off + cnt]), al, R.bf16_to_f32(sb[off:off + cnt])))
    assert (bits(s) == want).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sumsq_and_clip_coef(L, dtype):
    """Row f3 pieces: device-side squared norm (fp32 / bf16 arrays, ragged tail) and clip_grad_norm_'s coefficient."""
    g = torch.Generator().manual_seed(11)
    n = (1 << 20) + 3
    x = (torch.randn(n + 8, generator=g) * 0.02).to(dtype).to(dev())[:n]       # 16-byte aligned start, ragged length
    buf = torch.zeros(2, dtype=torch.float32, device=dev())
    lib.check(L.sdxl_sumsq(ptr(x), 0 if dtype == torch.float32 else 1, n, ptr(buf), stream()))
    want = float((x.double() ** 2).sum())
    torch.cuda.synchronize()
    assert abs(float(buf[0]) - want) <= 2e-5 * want
    for max_norm in (1.0, 1e9):
        lib.check(L.sdxl_clip_coef(ptr(buf), max_norm, C.c_void_p(buf.data_ptr() + 4), stream()))
        torch.cuda.synchronize()
        norm = want ** 0.5
        ref = min(1.0, max_norm / (norm + 1e-6))
        assert abs(float(buf[1]) - ref) <= 1e-5 * ref
