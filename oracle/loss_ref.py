"""ORACLE (test infrastructure only) -- fp32 CPU restatement of the reference's loss-side
arithmetic for SDXLTrainer.compute_loss().

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned against the real reference: oracle/make_goldens.py imports the reference's own
functions (with stub modules for the missing wandb/diffusers/...) in the authoring
container and writes tests/golden/loss_side.npz; tests/test_oracle_loss.py checks every
function below against those vectors.

Each function cites the reference lines it restates (paths under /root/reference/src).
Deviations ("repairs") are the D-ledger of SURVEY.md section 2.3 and are flagged inline.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

SIGMA_DATA = 1.0          # training/schedulers/novelai_v3.py:17
RHO_DEFAULT = 7.0         # D1: config.model.rho does not exist -> default of get_karras_sigmas (:164)
ZTSNR_SIGMA_MAX = 20000.0  # novelai_v3.py:106
LOSS_CAP = 1000.0         # ddpm_trainer.py:380-384, flow_matching_trainer.py:331-335


def karras_sigmas(n_sigmas: int = 1000, sigma_min: float = 0.002, sigma_max: float = 20000.0,
                  rho: float = RHO_DEFAULT) -> torch.Tensor:
    """training/schedulers/novelai_v3.py:160-184 (get_karras_sigmas), fp32 like the reference."""
    ramp = torch.linspace(0, 1, n_sigmas)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho


def timestep_to_sigma(timesteps: torch.Tensor, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """novelai_v3.py:134-137 (the reference rebuilds the table on every call)."""
    if table is None:
        table = karras_sigmas()
    return table[timesteps]


def sample_timesteps_from_u(u: torch.Tensor, num_timesteps: int = 1000) -> torch.Tensor:
    """novelai_v3.py:139-146 ZTSNR branch with the uniform draw injected: floor(u*T).long()."""
    return (u * num_timesteps).long()


def add_noise(sample: torch.Tensor, noise: torch.Tensor, sigmas: torch.Tensor, use_ztsnr: bool = True):
    """novelai_v3.py:111-120: x + sigma*noise, clamp +-20000 under ZTSNR."""
    noisy = sample + sigmas.view(-1, 1, 1, 1) * noise
    if use_ztsnr:
        noisy = torch.clamp(noisy, -20000.0, 20000.0)
    return noisy


def get_velocity(sample: torch.Tensor, noise: torch.Tensor, sigmas: torch.Tensor) -> torch.Tensor:
    """novelai_v3.py:122-127: (noise - x) / sqrt(sigma^2)  (D8: non-standard, reproduced)."""
    return (noise - sample) / (sigmas.view(-1, 1, 1, 1) ** 2).sqrt()


def get_snr(sigmas: torch.Tensor) -> torch.Tensor:
    """novelai_v3.py:129-132: (sigma_data / sigma)^2."""
    return (SIGMA_DATA / sigmas) ** 2


def loss_guard(loss: torch.Tensor) -> torch.Tensor:
    """ddpm_trainer.py:379-384 / flow_matching_trainer.py:330-335."""
    if not torch.isfinite(loss):
        return torch.tensor(LOSS_CAP, dtype=loss.dtype)
    return torch.clamp(loss, max=LOSS_CAP)


def ddpm_loss(model_pred: torch.Tensor, latents: torch.Tensor, noise: torch.Tensor,
              timesteps: torch.Tensor, prediction_type: str = "v_prediction",
              min_snr_gamma: Optional[float] = 5.0, tag_weights: Optional[torch.Tensor] = None,
              table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ddpm_trainer.py:327-384.

    D3 repair: the reference multiplies mse[B,C,H,W] by min(snr,gamma)[B], which broadcasts
    on the LAST dim (identical to this form at B=1, a RuntimeError at B>1 unless W==B);
    here the weight is per sample, w.view(-1,1,1,1).
    """
    sig = timestep_to_sigma(timesteps, table)
    if prediction_type == "v_prediction":
        target = get_velocity(latents, noise, sig)
    else:                                           # "epsilon" and the fall-through (:328-333)
        target = noise
    if min_snr_gamma is not None:
        snr = get_snr(sig)
        w = torch.minimum(snr, torch.ones_like(snr) * min_snr_gamma).float()
        loss = ((model_pred - target) ** 2 * w.view(-1, 1, 1, 1)).mean()
    else:
        loss = torch.nn.functional.mse_loss(model_pred, target)
    if tag_weights is not None:                     # D14: optional [B] tensor
        loss = loss * tag_weights.mean()
    return loss_guard(loss)


def sample_logit_normal_from_z(z: torch.Tensor, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """flow_matching_trainer.py:373-385 with the normal draw injected."""
    return torch.sigmoid(mean + std * z)


def optimal_transport_path(x0: torch.Tensor, x1: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """flow_matching_trainer.py:387-390."""
    t = t.view(-1, 1, 1, 1)
    return (1 - t) * x0 + t * x1


def flow_matching_loss(v_pred: torch.Tensor, x0: torch.Tensor, x1: torch.Tensor,
                       tag_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """flow_matching_trainer.py:407-419 + :323-335: per-sample mean of (v_pred-(x1-x0))^2, batch mean,
    optional tag weight, guard."""
    v_true = x1 - x0
    loss = ((v_pred - v_true) ** 2).mean([1, 2, 3]).mean()
    if tag_weights is not None:
        loss = loss * tag_weights.mean()
    return loss_guard(loss)


# ----------------------------------------------------------------------------------------------
# whole compute_loss() restatements driving the oracle UNet (used for model-level parity + cpu_baseline)
# ----------------------------------------------------------------------------------------------
def compute_loss_ddpm(unet_fn, batch: Dict[str, torch.Tensor], noise: torch.Tensor, timesteps: torch.Tensor,
                      prediction_type="v_prediction", min_snr_gamma: Optional[float] = 5.0,
                      use_ztsnr: bool = True) -> Dict:
    """ddpm_trainer.py:280-405 (training_step) with noise/timesteps injected (D2)."""
    lat = batch["vae_latents"].float()
    B = lat.shape[0]
    table = karras_sigmas(1000, 0.002, ZTSNR_SIGMA_MAX if use_ztsnr else 20000.0)
    sig = table[timesteps]
    noisy = add_noise(lat, noise, sig, use_ztsnr)
    pred = unet_fn(noisy, timesteps, batch["prompt_embeds"].float(),
                   batch["pooled_prompt_embeds"].float(), batch["time_ids"].float())
    loss = ddpm_loss(pred, lat, noise, timesteps, prediction_type, min_snr_gamma,
                     batch.get("tag_weights"), table)
    metrics = {"loss": float(loss.detach()), "timestep_mean": float(timesteps.float().mean()),
               "noise_scale": float(noise.abs().mean()), "pred_scale": float(pred.detach().abs().mean()),
               "batch_size": B}
    if B > 1:
        metrics["timestep_std"] = float(timesteps.float().std())
    return {"loss": loss, "metrics": metrics, "pred": pred, "noisy": noisy}


def compute_loss_flow(unet_fn, batch: Dict[str, torch.Tensor], x0: torch.Tensor, t: torch.Tensor) -> Dict:
    """flow_matching_trainer.py:267-356 with x0 / t injected; one UNet forward (D5); t unscaled (D6)."""
    x1 = batch["vae_latents"].float()
    B = x1.shape[0]
    xt = optimal_transport_path(x0, x1, t)
    v = unet_fn(xt, t, batch["prompt_embeds"].float(), batch["pooled_prompt_embeds"].float(),
                batch["time_ids"].float())
    loss = flow_matching_loss(v, x0, x1, batch.get("tag_weights"))
    metrics = {"loss": float(loss.detach()), "x0_norm": float(x0.norm()), "x1_norm": float(x1.norm()),
               "time_mean": float(t.mean()), "time_std": float(t.std()) if B > 1 else float("nan"),
               "velocity_norm": float(v.detach().norm()), "batch_size": B}
    return {"loss": loss, "metrics": metrics, "pred": v, "xt": xt}
