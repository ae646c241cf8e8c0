"""Host-side noise schedule: same API as the reference's NoiseScheduler (training/schedulers/novelai_v3.py),
training functions only.  The Karras table is built ONCE (the reference rebuilds it three times per step)."""
from __future__ import annotations

import torch


def get_karras_sigmas(n_sigmas: int, sigma_min: float, sigma_max: float, rho: float = 7.0, device=None) -> torch.Tensor:
    """novelai_v3.py:160-184 -- identical torch ops, so the fp32 table is bit-identical to the reference's."""
    ramp = torch.linspace(0, 1, n_sigmas, device=device)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    return (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho


class NoiseScheduler:
    def __init__(self, config, device="cpu"):
        self.config = config
        self.device = device
        self.sigma_data = 1.0                                        # novelai_v3.py:17
        m = config.model
        smax = 20000.0 if m.use_ztsnr else m.sigma_max               # novelai_v3.py:106
        self.sigmas = get_karras_sigmas(m.num_timesteps, m.sigma_min, smax, getattr(m, "rho", 7.0)).to(device)

    def get_sigmas(self, n_sigmas: int) -> torch.Tensor:             # :101-109
        if n_sigmas == self.sigmas.numel():
            return self.sigmas
        m = self.config.model
        return get_karras_sigmas(n_sigmas, m.sigma_min, 20000.0 if m.use_ztsnr else m.sigma_max,
                                 getattr(m, "rho", 7.0), self.device)

    def timestep_to_sigma(self, timesteps: torch.Tensor) -> torch.Tensor:   # :134-137
        return self.sigmas[timesteps.to(self.sigmas.device)]

    def sample_timesteps(self, batch_size: int, generator=None) -> torch.Tensor:   # :139-151 (D2: no device kwarg)
        n = self.config.model.num_timesteps
        if self.config.model.use_ztsnr:
            u = torch.rand(batch_size, generator=generator)
            return (u * n).long()
        return torch.randint(0, n, (batch_size,), generator=generator)

    # the three below run on the device inside the loss kernels; kept for API parity / host-side checks
    def add_noise(self, sample, noise, timesteps):                    # :111-120
        noisy = sample + self.timestep_to_sigma(timesteps).view(-1, 1, 1, 1).to(sample.device) * noise
        return torch.clamp(noisy, -20000.0, 20000.0) if self.config.model.use_ztsnr else noisy

    def get_velocity(self, sample, noise, timesteps):                 # :122-127
        s = self.timestep_to_sigma(timesteps).view(-1, 1, 1, 1).to(sample.device)
        return (noise - sample) / (s ** 2).sqrt()

    def get_snr(self, timesteps):                                     # :129-132
        return (self.sigma_data / self.timestep_to_sigma(timesteps)) ** 2
