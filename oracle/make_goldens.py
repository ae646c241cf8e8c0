"""Generate tests/golden/loss_side.npz by running the REAL reference functions.

Runs only in the authoring container (needs /root/reference); never on the GPU box.
The reference cannot be imported as-is (wandb / colorama / spacy / diffusers / xformers are not
installed), so this script pre-seeds sys.modules with empty stand-in *modules* for those
third-party imports -- none of them is on the loss-side arithmetic being captured -- then
imports the reference's own files unchanged and calls:

    src.training.schedulers.novelai_v3: get_karras_sigmas, NoiseScheduler.{add_noise,get_velocity,
        get_snr,timestep_to_sigma,sample_timesteps}
    src.training.trainers.methods.flow_matching_trainer.FlowMatchingTrainer.{sample_logit_normal,
        optimal_transport_path,compute_flow_matching_loss,_compute_loss_impl}
    the MinSNR / guard lines of ddpm_trainer.py:336-384 (B=1, unmodified) via DDPMTrainer.training_step
        on a stand-in "unet" (a fixed linear map, so the captured loss is a pure function of inputs)

Output is data only (inputs + expected outputs).  Usage:  python oracle/make_goldens.py
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "loss_side.npz"


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Blank:
    def __getattr__(self, k):
        return ""


class _Dummy:
    def __init__(self, *a, **kw):
        self.num_train_timesteps = kw.get("num_train_timesteps", 1000)

    def set_timesteps(self, *a, **kw):
        pass


def import_reference():
    _stub("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None, finish=lambda *a, **k: None,
          Image=_Dummy, run=None)
    _stub("colorama", Fore=_Blank(), Style=_Blank(), Back=_Blank(), init=lambda *a, **k: None)
    _stub("spacy", load=lambda *a, **k: None)
    _stub("diffusers", DDPMScheduler=_Dummy, StableDiffusionXLPipeline=_Dummy, AutoencoderKL=_Dummy,
          UNet2DConditionModel=_Dummy)
    _stub("xformers"); _stub("xformers.ops")
    scratch = tempfile.mkdtemp(prefix="refimport_")
    os.chdir(scratch)                       # reference loggers create outputs/logs relative to cwd
    sys.path.insert(0, str(REF))
    from src.training.schedulers import novelai_v3                     # noqa
    from src.training.trainers.methods import flow_matching_trainer    # noqa
    from src.training.trainers.methods import ddpm_trainer             # noqa
    from src.data.config import Config                                 # noqa
    return novelai_v3, flow_matching_trainer, ddpm_trainer, Config


class _UNetOut:
    def __init__(self, s):
        self.sample = s


class StandInUNet(torch.nn.Module):
    """Fixed, parameter-light map so captured losses are pure functions of the inputs:
    pred = a*x + b*mean(ehs) + c*t  (per sample)."""
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.tensor(0.75))

    def forward(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None, **kw):
        B = x.shape[0]
        tt = t.reshape(-1).to(x.dtype) if torch.is_tensor(t) else torch.full((B,), float(t))
        e = encoder_hidden_states.reshape(B, -1).mean(1)
        return _UNetOut(self.a * x + 0.1 * e.view(B, 1, 1, 1) + 0.001 * tt.view(B, 1, 1, 1))


def main():
    nv3, fmt, ddt, Config = import_reference()
    g = {}
    cfg = Config()
    cfg.model.rho = 7.0                      # D1: attribute missing in ModelConfig
    sched = nv3.NoiseScheduler(cfg, "cpu")

    # --- Karras table ---------------------------------------------------------------------
    g["karras_table"] = nv3.get_karras_sigmas(1000, 0.002, 20000.0, 7.0).numpy()
    g["karras_table_sched"] = sched.get_sigmas(1000).numpy()

    # --- scheduler functions on seeded inputs ----------------------------------------------
    case = 0
    for seed in (0, 1, 2):
        for B in (1, 4):
            for hw in ((8, 64) if (seed == 2 and B == 1) else (8, 12)):
                gen = torch.Generator().manual_seed(seed * 100 + B * 10 + hw)
                x = torch.randn(B, 4, hw, hw, generator=gen)
                n = torch.randn(B, 4, hw, hw, generator=gen)
                u = torch.rand(B, generator=gen)
                ts = (u * 1000).long()
                if case == 0:
                    ts = torch.tensor([0])             # extreme: sigma = 20000 (clamp active)
                if case == 1:
                    ts = torch.tensor([999, 0, 500, 998])
                k = f"sch{case}"
                g[f"{k}_x"] = x.numpy(); g[f"{k}_noise"] = n.numpy(); g[f"{k}_u"] = u.numpy()
                g[f"{k}_t"] = ts.numpy()
                g[f"{k}_sigma"] = sched.timestep_to_sigma(ts).numpy()
                g[f"{k}_noisy"] = sched.add_noise(x, n, ts).numpy()
                g[f"{k}_vel"] = sched.get_velocity(x, n, ts).numpy()
                g[f"{k}_snr"] = sched.get_snr(ts).numpy()
                case += 1
    g["n_sched_cases"] = np.array(case)

    # sample_timesteps (ZTSNR branch): torch.rand under manual_seed
    torch.manual_seed(123)
    g["sample_timesteps_seed123_B8"] = sched.sample_timesteps(8).numpy()
    torch.manual_seed(123)
    g["sample_timesteps_seed123_B8_u"] = torch.rand(8).numpy()

    # --- flow matching pieces ---------------------------------------------------------------
    FM = fmt.FlowMatchingTrainer
    fm = object.__new__(FM)
    gen = torch.Generator().manual_seed(0)
    g["logit_normal_seed0"] = fm.sample_logit_normal((4,), "cpu", torch.float32, generator=gen).numpy()
    gen = torch.Generator().manual_seed(0)
    g["logit_normal_seed0_z"] = torch.randn((4,), generator=gen).numpy()
    unet = StandInUNet()
    case = 0
    for seed in (0, 1, 2):
        for B in (1, 4):
            gen = torch.Generator().manual_seed(1000 + seed * 10 + B)
            x0 = torch.randn(B, 4, 16, 16, generator=gen)
            x1 = torch.randn(B, 4, 16, 16, generator=gen)
            t = torch.sigmoid(torch.randn(B, generator=gen))
            ehs = torch.randn(B, 77, 32, generator=gen)
            k = f"fm{case}"
            g[f"{k}_x0"] = x0.numpy(); g[f"{k}_x1"] = x1.numpy(); g[f"{k}_t"] = t.numpy(); g[f"{k}_ehs"] = ehs.numpy()
            g[f"{k}_xt"] = fm.optimal_transport_path(x0, x1, t).numpy()
            cond = {"prompt_embeds": ehs, "added_cond_kwargs": {"text_embeds": None, "time_ids": None}}
            with torch.no_grad():
                lb = fm.compute_flow_matching_loss(unet, x0, x1, t, cond)
                g[f"{k}_vpred"] = unet(fm.optimal_transport_path(x0, x1, t), t, encoder_hidden_states=ehs).sample.numpy()
            g[f"{k}_loss_per_sample"] = lb.numpy()
            case += 1
    g["n_fm_cases"] = np.array(case)

    # full _compute_loss_impl (D4 repaired by handing it an object whose __call__ is the unet)
    class _Model:
        def __init__(self, u):
            self.unet = u
        def __call__(self, *a, **k):
            return self.unet(*a, **k)
        def parameters(self):
            return self.unet.parameters()
    class _Opt:
        param_groups = [{"lr": 4e-7}]
    fm2 = object.__new__(FM)
    fm2.model = _Model(unet); fm2.device = "cpu"; fm2.optimizer = _Opt()
    gen = torch.Generator().manual_seed(77)
    batch = {"vae_latents": torch.randn(4, 4, 16, 16, generator=gen),
             "prompt_embeds": torch.randn(4, 77, 32, generator=gen),
             "pooled_prompt_embeds": torch.randn(4, 16, generator=gen),
             "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * 4), "metadata": {},
             "tag_weights": torch.tensor([0.5, 1.0, 2.0, 1.5])}
    # make the in-function RNG reproducible: generator for t, global seed for randn_like(x0)
    torch.manual_seed(4242)
    gen_t = torch.Generator().manual_seed(5)
    out = fm2._compute_loss_impl(fm2.model, batch, generator=gen_t)
    torch.manual_seed(4242)
    x0_used = torch.randn_like(batch["vae_latents"])
    gen_t = torch.Generator().manual_seed(5)
    t_used = torch.sigmoid(torch.randn((4,), generator=gen_t))
    for k2 in ("vae_latents", "prompt_embeds", "pooled_prompt_embeds", "time_ids", "tag_weights"):
        g[f"fmfull_{k2}"] = batch[k2].numpy()
    g["fmfull_x0"] = x0_used.numpy(); g["fmfull_t"] = t_used.numpy()
    g["fmfull_loss"] = np.array(out["loss"].item())
    g["fmfull_metric_keys"] = np.array(sorted(out["metrics"].keys()))
    for mk in ("x0_norm", "x1_norm", "time_mean", "time_std", "velocity_norm"):
        g[f"fmfull_m_{mk}"] = np.array(out["metrics"][mk])

    # --- DDPM training_step lines 327-384, unmodified reference at B=1 -----------------------
    DD = ddt.DDPMTrainer
    dd = object.__new__(DD)
    class _S2(nv3.NoiseScheduler):
        def sample_timesteps(self, batch_size, device=None):          # D2: signature repair
            return self._inject
    s2 = _S2(cfg, "cpu")
    dd.noise_scheduler = s2; dd.device = "cpu"; dd.config = cfg; dd.optimizer = _Opt()
    dd.model = _Model(unet); dd.wandb_logger = None
    cfg.training.prediction_type = "v_prediction"
    case = 0
    for seed, tval in ((0, 0), (1, 500), (2, 999), (3, 850), (4, 17)):
        gen = torch.Generator().manual_seed(9000 + seed)
        b1 = {"vae_latents": torch.randn(1, 4, 16, 16, generator=gen),
              "prompt_embeds": torch.randn(1, 77, 32, generator=gen),
              "pooled_prompt_embeds": torch.randn(1, 16, generator=gen),
              "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]]), "metadata": {}}
        s2._inject = torch.tensor([tval])
        torch.manual_seed(31337 + seed)
        out = dd.training_step(b1)
        torch.manual_seed(31337 + seed)
        noise_used = torch.randn_like(b1["vae_latents"])
        k = f"dd{case}"
        for k2 in ("vae_latents", "prompt_embeds"):
            g[f"{k}_{k2}"] = b1[k2].numpy()
        g[f"{k}_noise"] = noise_used.numpy(); g[f"{k}_t"] = np.array([tval])
        g[f"{k}_loss"] = np.array(out["loss"].item())
        g[f"{k}_m_noise_scale"] = np.array(out["metrics"]["noise_scale"])
        g[f"{k}_m_pred_scale"] = np.array(out["metrics"]["pred_scale"])
        g[f"{k}_metric_keys"] = np.array(sorted(out["metrics"].keys()))
        case += 1
    g["n_dd_cases"] = np.array(case)
    # MinSNR off branch (plain mse_loss) and epsilon target
    cfg.model.min_snr_gamma = None
    s2._inject = torch.tensor([500]); torch.manual_seed(1); out = dd.training_step(b1)
    g["dd_nogamma_loss"] = np.array(out["loss"].item())
    torch.manual_seed(1); g["dd_nogamma_noise"] = torch.randn_like(b1["vae_latents"]).numpy()
    cfg.model.min_snr_gamma = 5.0; cfg.training.prediction_type = "epsilon"
    s2._inject = torch.tensor([500]); torch.manual_seed(2); out = dd.training_step(b1)
    g["dd_eps_loss"] = np.array(out["loss"].item())
    torch.manual_seed(2); g["dd_eps_noise"] = torch.randn_like(b1["vae_latents"]).numpy()
    g["dd_last_vae_latents"] = b1["vae_latents"].numpy(); g["dd_last_prompt_embeds"] = b1["prompt_embeds"].numpy()
    # guard: non-finite -> 1000 ; > 1000 -> 1000
    cfg.training.prediction_type = "v_prediction"
    binf = dict(b1); binf["vae_latents"] = b1["vae_latents"].clone(); binf["vae_latents"][0, 0, 0, 0] = float("inf")
    s2._inject = torch.tensor([500]); torch.manual_seed(3); out = dd.training_step(binf)
    g["dd_guard_inf_loss"] = np.array(out["loss"].item())
    bbig = dict(b1); bbig["vae_latents"] = b1["vae_latents"] * 1e4
    s2._inject = torch.tensor([999]); torch.manual_seed(3); out = dd.training_step(bbig)
    g["dd_guard_big_loss"] = np.array(out["loss"].item())

    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **g)
    print(f"wrote {OUT} ({OUT.stat().st_size/1024:.1f} KiB, {len(g)} arrays)")


if __name__ == "__main__":
    main()
