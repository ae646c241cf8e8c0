"""Drop-in boundary on the GPU: the reference's model object in, trained weights back out, checkpoints reload."""
import importlib

import pytest
import torch

import sdxl_amd  # noqa: F401
from oracle import unet_ref as U
from sdxl_amd import unet as NU

pytestmark = pytest.mark.gpu
T = importlib.import_module("sdxl-training-improvements_amd.trainer")
CFG = importlib.import_module("sdxl-training-improvements_amd.config")


def module_from_state_dict(sd):
    """a torch.nn.Module tree whose state_dict() has exactly these (diffusers) keys -- what `model.unet` is in the reference"""
    root = torch.nn.Module()
    for key, t in sd.items():
        *path, leaf = key.split(".")
        m = root
        for p in path:
            if p not in m._modules:
                m.add_module(p, torch.nn.Module())
            m = m._modules[p]
        m.register_parameter(leaf, torch.nn.Parameter(t.clone(), requires_grad=False))
    return root


def _batch(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    bfr = lambda t: t.to(torch.bfloat16).float()
    return {"vae_latents": r(B, 4, 16, 16), "prompt_embeds": bfr(r(B, 77, cfg.cross_attention_dim)),
            "pooled_prompt_embeds": bfr(r(B, cfg.pooled_dim)), "time_ids": torch.tensor([[[128.0, 128, 0, 0, 128, 128]]] * B),
            "metadata": {}}


def test_reference_model_object_trains_and_round_trips(tmp_path):
    cfg = U.tiny_config()
    w = U.synth_weights(cfg, seed=3)                                       # fp32 tensors holding bf16-exact values
    torch_unet = module_from_state_dict(w)
    torch_unet.config = {"block_out_channels": list(cfg.block_out_channels), "transformer_layers_per_block": list(cfg.transformer_layers_per_block),
                         "cross_attention_dim": cfg.cross_attention_dim, "addition_time_embed_dim": cfg.addition_time_embed_dim,
                         "projection_class_embeddings_input_dim": cfg.add_in_dim, "attention_head_dim": [1, 2, 4]}
    assert set(torch_unet.state_dict()) == set(w)

    class Model:                                                           # models/sdxl.py: .unet (+ save_pretrained in the real one)
        unet = torch_unet

    c = CFG.Config()
    c.training.method = "ddpm"
    c.optimizer.learning_rate = 1e-3
    tr = T.NativeSDXLTrainer(Model(), config=c, device=torch.device("cuda", 0))
    assert tr._torch_unet is torch_unet and tuple(tr.net.cfg.block_out_channels) == cfg.block_out_channels
    # import is bit-exact for EVERY key (conv repack, fused q|k|v rows, interleaved GEGLU projection included)
    back = tr.net.state_dict(torch.float32)
    assert set(back) == set(w)
    for k in w:
        assert torch.equal(back[k].cpu(), w[k]), k
    b = _batch(cfg, 2, 11)
    ts, noise = torch.tensor([300, 650]), torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    l0 = float(tr.compute_loss(b, timesteps=ts, noise=noise)["loss"])
    for _ in range(3):
        tr._execute_training_step(b, timesteps=ts, noise=noise)
        tr.optimizer_step()
    l1 = float(tr.compute_loss(b, timesteps=ts, noise=noise)["loss"])
    assert l1 < l0
    tr.sync_to_model()                                                     # trained weights are now in the caller's module
    sd = torch_unet.state_dict()
    native = tr.net.state_dict(torch.float32)
    changed = 0
    for k in w:
        assert sd[k].dtype == torch.float32 and torch.equal(sd[k], native[k].cpu()), k
        changed += int(not torch.equal(sd[k], w[k]))
    assert changed > len(w) // 2
    # checkpoint: diffusers-keyed safetensors + optimizer.pt + config.json; a fresh engine loaded from it computes the same loss
    d = tr.save_checkpoint(tmp_path / "ck", is_final=True)
    from safetensors.torch import load_file
    ck = load_file(str(d / "unet" / "diffusion_pytorch_model.safetensors"))
    assert set(ck) == set(w) and all(t.dtype == torch.bfloat16 for t in ck.values())
    net2 = NU.NativeUNet(NU.config_from_unet(torch_unet))
    net2.load_state_dict(ck)
    tr2 = T.NativeSDXLTrainer(type("M", (), {"unet": net2})(), config=c)
    l2 = float(tr2.compute_loss(b, timesteps=ts, noise=noise)["loss"])
    assert l2 == l1
    tr2.load_optimizer_state(d)
    assert tr2.optimizer.step_count == 3 and torch.equal(tr2.optimizer.exp_avg_sq.cpu(), tr.optimizer.exp_avg_sq.cpu())
    net2.close()
    tr.net.close()
