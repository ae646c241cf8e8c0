"""Opportunistic pin of the UNet oracle ("parity unpinned" -- the UNet arithmetic is diffusers', which is neither
vendored by the reference, `requirements.txt:2`, nor installed in this image).  Wherever `diffusers` IS importable this test
builds `UNet2DConditionModel` from the SDXL-base `unet/config.json` literals (the module the reference loads at
`src/models/sdxl.py:25-40` and calls at `ddpm_trainer.py:320-325` / `flow_matching_trainer.py:400-405`), loads the oracle's
synthetic weights by diffusers key name and requires `oracle.unet_ref.unet_forward` to agree in fp32.  It skips here."""
import pytest
import torch

diffusers = pytest.importorskip("diffusers")

from oracle import unet_ref as U  # noqa: E402


def _diffusers_unet(cfg: U.UNetConfig):
    heads = tuple(c // cfg.head_dim for c in cfg.block_out_channels)
    return diffusers.UNet2DConditionModel(
        sample_size=128, in_channels=cfg.in_channels, out_channels=cfg.out_channels, center_input_sample=False,
        flip_sin_to_cos=True, freq_shift=0,
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        mid_block_type="UNetMidBlock2DCrossAttn",
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block, downsample_padding=1,
        mid_block_scale_factor=1, act_fn="silu", norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.resnet_eps,
        cross_attention_dim=cfg.cross_attention_dim,
        transformer_layers_per_block=tuple(max(1, t) for t in cfg.transformer_layers_per_block),   # level 0 has no attention blocks
        attention_head_dim=heads, use_linear_projection=True, upcast_attention=False, resnet_time_scale_shift="default",
        addition_embed_type="text_time", addition_time_embed_dim=cfg.addition_time_embed_dim,
        projection_class_embeddings_input_dim=cfg.add_in_dim)


@pytest.mark.parametrize("which", ["tiny", "sdxl_base"])
def test_oracle_forward_equals_diffusers(which):
    cfg = U.tiny_config() if which == "tiny" else U.SDXL_BASE
    torch.manual_seed(0)
    net = _diffusers_unet(cfg).float().eval()
    w = U.synth_weights(cfg, seed=0)
    sd = net.state_dict()
    assert set(sd) == set(w), (sorted(set(sd) - set(w))[:5], sorted(set(w) - set(sd))[:5])
    net.load_state_dict({k: v.reshape(sd[k].shape) for k, v in w.items()}, strict=True)
    g = torch.Generator().manual_seed(1)
    B, H = (2, 16) if which == "tiny" else (1, 32)
    x = torch.randn(B, cfg.in_channels, H, H, generator=g)
    ehs = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    for t in (torch.tensor([10, 900][:B]), torch.tensor([0.3, 0.8][:B])):      # DDPM indices and flow-matching t in (0, 1)
        with torch.no_grad():
            ref = net(x, t, ehs, added_cond_kwargs={"text_embeds": pooled, "time_ids": tid}).sample
            got = U.unet_forward(w, x, t, ehs, pooled, tid, cfg)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, (which, err)
