"""Generate tests/golden/adamw_bf16.npz by running the REAL reference optimizer arithmetic.

Authoring container only (needs /root/reference).  Imports the reference's own
`src/training/optimizers/adamw_bfloat16/__init__.py::_make_step` (unchanged) and calls it on seeded bf16 tensors.
The only intervention: `torch.randint_like` is wrapped so that the int32 tensors the reference draws for its
stochastic rounding (stochastic/__init__.py:55-60) are RECORDED and stored with the fixture -- the draws themselves
are the reference's own (torch CPU generator under manual_seed).  Output is data only: inputs, the recorded random
integers and the resulting (p, exp_avg, exp_avg_sq, shift) after each step.

Usage:  python oracle/make_adamw_goldens.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_goldens import _stub, _Blank, _Dummy, REF  # noqa: E402  (same stand-in modules for missing third parties)

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "adamw_bf16.npz"


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> uint16 bit patterns (npz has no bf16)."""
    return t.detach().contiguous().view(torch.int16).numpy().astype(np.uint16)


def main():
    import os, tempfile
    _stub("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None, finish=lambda *a, **k: None, Image=_Dummy, run=None)
    _stub("colorama", Fore=_Blank(), Style=_Blank(), Back=_Blank(), init=lambda *a, **k: None)
    _stub("spacy", load=lambda *a, **k: None)
    _stub("diffusers", DDPMScheduler=_Dummy, StableDiffusionXLPipeline=_Dummy, AutoencoderKL=_Dummy, UNet2DConditionModel=_Dummy)
    _stub("xformers"); _stub("xformers.ops")
    os.chdir(tempfile.mkdtemp(prefix="refimport_"))
    sys.path.insert(0, str(REF))
    from src.training.optimizers import adamw_bfloat16 as ref      # the reference module, unchanged

    recorded = []
    real = torch.randint_like

    def recording(*a, **k):
        r = real(*a, **k)
        recorded.append(r.clone())
        return r

    g = {}
    cases = [
        # name, n, lr, beta1, beta2, eps, steps, decay at step (1-based, value), grad scale, zero-initialised state
        ("default", 4096, 4e-7, 0.9, 0.999, 1e-8, 3, (None, 0.0), 1e-3, True),
        ("biglr", 2048, 1e-3, 0.9, 0.999, 1e-8, 3, (2, 0.006), 1e-2, False),
        ("betas", 1024, 1e-4, 0.8, 0.95, 1e-6, 2, (1, 0.0075), 1.0, False),
    ]
    for ci, (name, n, lr, b1, b2, eps, steps, (dstep, dval), gs, zero_state) in enumerate(cases):
        gen = torch.Generator().manual_seed(1000 + ci)
        p = (torch.randn(n, generator=gen) * 0.05).to(torch.bfloat16)
        if zero_state:
            m = torch.zeros(n, dtype=torch.bfloat16); v = torch.zeros(n, dtype=torch.bfloat16); sh = torch.zeros(n, dtype=torch.bfloat16)
        else:
            m = (torch.randn(n, generator=gen) * gs * 0.3).to(torch.bfloat16)
            v = (torch.rand(n, generator=gen) * gs * gs).to(torch.bfloat16)
            sh = (torch.randn(n, generator=gen) * 1e-4).to(torch.bfloat16)
        g[f"{name}_hyper"] = np.array([lr, b1, b2, eps], dtype=np.float64)
        g[f"{name}_p0"] = bits(p); g[f"{name}_m0"] = bits(m); g[f"{name}_v0"] = bits(v); g[f"{name}_s0"] = bits(sh)
        torch.manual_seed(77 + ci)
        for st in range(1, steps + 1):
            grad = (torch.randn(n, generator=gen) * gs).to(torch.bfloat16)
            decay = dval if dstep == st else 0.0
            recorded.clear()
            torch.randint_like = recording
            try:
                ref._make_step(grad, p, sh, m, v, beta1=b1, beta2=b2, step=float(st), lr=lr, eps=eps,
                               decay_this_iteration=decay, zero_grad=False)
            finally:
                torch.randint_like = real
            assert len(recorded) == 4, len(recorded)
            g[f"{name}_grad{st}"] = bits(grad)
            g[f"{name}_decay{st}"] = np.array(decay, dtype=np.float64)
            g[f"{name}_rand{st}"] = torch.stack(recorded).numpy().astype(np.uint16)      # [4][n], values < 2^16
            g[f"{name}_p{st}"] = bits(p); g[f"{name}_m{st}"] = bits(m); g[f"{name}_v{st}"] = bits(v); g[f"{name}_s{st}"] = bits(sh)
        g[f"{name}_steps"] = np.array(steps)
    g["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
