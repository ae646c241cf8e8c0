"""Drop-in boundary, host logic (no GPU): the trainer accepts the REFERENCE's model object (`model.unet` = a PyTorch module
with a diffusers-keyed state_dict, models/sdxl.py:11-62), imports its weights, writes the trained ones back, saves
checkpoints the way sdxl_trainer.py:162-210 does; the method module `native_mi355x.py` is constructed with the reference's
own constructor call; accumulation state is correct on the caller-owned `compute_loss(...)["loss"].backward()` loop."""
import importlib
import json
from types import SimpleNamespace

import pytest
import torch

import sdxl_amd  # noqa: F401

T = importlib.import_module("sdxl-training-improvements_amd.trainer")
NM = importlib.import_module("sdxl-training-improvements_amd.native_mi355x")
CFG = importlib.import_module("sdxl-training-improvements_amd.config")
UN = importlib.import_module("sdxl-training-improvements_amd.unet")


class StubTorchUNet(torch.nn.Module):
    """a module whose state_dict has diffusers key names (a handful are enough for the host logic)"""

    def __init__(self):
        super().__init__()
        self.conv_in = torch.nn.Conv2d(4, 8, 3, padding=1)
        self.conv_out = torch.nn.Conv2d(8, 4, 3, padding=1)


class FakeNative:
    """stand-in for NativeUNet on a machine without a GPU: records the boundary calls, keeps `weights` like the arena"""

    def __init__(self, cfg):
        self.cfg, self.calls, self.loaded = cfg, [], None
        self.param_elems = 16
        self.weights = torch.zeros(16, dtype=torch.bfloat16)
        self.grads = torch.zeros(16)

    def load_state_dict(self, sd, strict=True):
        if strict and set(sd) != {"conv_in.weight", "conv_in.bias", "conv_out.weight", "conv_out.bias"}:
            raise KeyError("state_dict mismatch")
        self.loaded = {k: v.detach().clone() for k, v in sd.items()}

    def state_dict(self, dtype=torch.bfloat16):
        return {k: (v + 1.0).to(dtype) for k, v in self.loaded.items()}       # "trained": every weight moved by +1

    def zero_grads(self):
        self.calls.append(("zero",))

    def forward_loss(self, method, *a, **k):
        self.calls.append(("fwd", method))

    def backward(self, scale, first, on_segment=None):
        self.calls.append(("bwd", round(scale, 6), first, on_segment is not None))

    def read_loss(self):
        return [0.5, 0, 8.0, 16.0, 4.0, 9.0, 25.0, 1.0]


class RefModel:
    """what the reference hands a trainer (models/sdxl.py): .unet + save_pretrained"""

    def __init__(self):
        self.unet = StubTorchUNet()
        self.saved = None

    def save_pretrained(self, d, safe_serialization=True):
        self.saved = (d, safe_serialization, {k: v.detach().clone() for k, v in self.unet.state_dict().items()})


def _batch(B=2):
    return {"vae_latents": torch.randn(B, 4, 8, 8), "prompt_embeds": torch.randn(B, 77, 16),
            "pooled_prompt_embeds": torch.randn(B, 8), "time_ids": torch.zeros(B, 1, 6), "metadata": {}}


def _mk(method="ddpm", accum=1, model=None):
    cfg = CFG.Config()
    cfg.training.method = method
    cfg.training.gradient_accumulation_steps = accum
    model = model or RefModel()
    tr = T.NativeSDXLTrainer(model, optimizer=None, train_dataloader=None, device="cpu", config=cfg,
                             native_factory=FakeNative, native_config=UN.SDXL_BASE_CFG)
    return tr, model


def test_reference_model_object_is_imported_and_written_back(tmp_path, monkeypatch):
    tr, model = _mk()
    before = {k: v.detach().clone() for k, v in model.unet.state_dict().items()}
    assert set(tr.net.loaded) == set(before) and all(torch.equal(tr.net.loaded[k], before[k]) for k in before)
    tr.sync_to_model()                                                     # trained weights (= +1, rounded to bf16) land in the module
    after = model.unet.state_dict()
    for k in before:
        assert after[k].dtype == before[k].dtype
        assert torch.equal(after[k], (before[k] + 1.0).to(torch.bfloat16).to(before[k].dtype))
    # save_checkpoint(epoch) -> outputs/checkpoint-0003 via the model's own save_pretrained, optimizer.pt, config.json
    monkeypatch.chdir(tmp_path)
    tr.optimizer = SimpleNamespace(state_dict=lambda: {"state": {"step": 7}, "param_groups": [{"lr": 1e-6}]},
                                   param_groups=[{"lr": 1e-6}])
    d = tr.save_checkpoint(3)
    assert d == (tmp_path / "outputs" / "checkpoint-0003").relative_to(tmp_path) or d.resolve() == (tmp_path / "outputs" / "checkpoint-0003")
    assert model.saved[0].endswith("checkpoint-0003") and model.saved[1] is True
    assert torch.load(d / "optimizer.pt", weights_only=False)["state"]["step"] == 7
    assert json.loads((d / "config.json").read_text())["training"]["method"] == "ddpm"
    d2 = tr.save_checkpoint(10, is_final=True)
    assert d2.name == "final_checkpoint"
    d3 = tr.save_checkpoint(tmp_path / "custom_dir", is_final=True)       # main.py:111 hands over a directory
    assert d3 == tmp_path / "custom_dir" and (d3 / "config.json").exists()


def test_model_without_save_pretrained_gets_safetensors(tmp_path):
    cfg = CFG.Config()
    tr = T.NativeSDXLTrainer(SimpleNamespace(unet=StubTorchUNet()), optimizer=None, device="cpu", config=cfg,
                             native_factory=FakeNative, native_config=UN.SDXL_BASE_CFG)
    tr.optimizer = None
    d = tr.save_checkpoint(tmp_path / "ck")
    from safetensors.torch import load_file
    sd = load_file(str(d / "unet" / "diffusion_pytorch_model.safetensors"))
    assert set(sd) == {"conv_in.weight", "conv_in.bias", "conv_out.weight", "conv_out.bias"} and sd["conv_in.weight"].dtype == torch.bfloat16


def test_unusable_model_is_rejected():
    with pytest.raises(TypeError):
        T.NativeSDXLTrainer(object(), config=CFG.Config())
    class Bad(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.other = torch.nn.Linear(2, 2)
    with pytest.raises(KeyError):
        T.NativeSDXLTrainer(SimpleNamespace(unet=Bad()), config=CFG.Config(), device="cpu", native_factory=FakeNative,
                            native_config=UN.SDXL_BASE_CFG)


def test_direct_backward_loop_keeps_accumulation_state():
    """`compute_loss(batch)["loss"].backward()` in a caller-owned loop must pass first_micro only on the first
    micro-step of a cycle, zero the small-parameter gradients once per cycle, and start a new cycle after optimizer.step()."""
    tr, _ = _mk()
    tr.optimizer = None
    for _ in range(3):
        (tr.compute_loss(_batch())["loss"] / 3).backward()
    bw = [c for c in tr.net.calls if c[0] == "bwd"]
    assert [b[2] for b in bw] == [True, False, False] and all(abs(b[1] - 1 / 3) < 1e-6 for b in bw)
    assert [c[0] for c in tr.net.calls].count("zero") == 1 and tr.net.calls.index(("zero",)) < tr.net.calls.index(bw[0])
    tr.optimizer_step()                                   # ends the cycle
    tr.compute_loss(_batch())["loss"].backward()
    bw = [c for c in tr.net.calls if c[0] == "bwd"]
    assert bw[-1][2] is True and [c[0] for c in tr.net.calls].count("zero") == 2
    # explicit zero_grad also starts a cycle and is not repeated by the next backward
    tr.zero_grad()
    tr.compute_loss(_batch())["loss"].backward()
    assert [c[0] for c in tr.net.calls].count("zero") == 3 and [c for c in tr.net.calls if c[0] == "bwd"][-1][2] is True


def test_optimizer_post_step_hook_ends_cycle():
    tr, _ = _mk()
    class Opt:
        param_groups = [{"lr": 1e-6}]
        def __init__(self):
            self.hooks = []
        def register_step_post_hook(self, fn):
            self.hooks.append(fn)
        def step(self):
            for h in self.hooks:
                h(self)
    cfg = CFG.Config()
    o = Opt()
    tr = T.NativeSDXLTrainer(RefModel(), optimizer=o, device="cpu", config=cfg, native_factory=FakeNative,
                             native_config=UN.SDXL_BASE_CFG)
    tr.compute_loss(_batch())["loss"].backward()
    tr.compute_loss(_batch())["loss"].backward()
    o.step()                                              # the caller steps the optimizer itself
    tr.compute_loss(_batch())["loss"].backward()
    assert [c[2] for c in tr.net.calls if c[0] == "bwd"] == [True, False, True]


def test_native_mi355x_method_module(tmp_path, monkeypatch):
    """constructed exactly like sdxl_trainer.py:130-150 builds its method trainers, from a reference-style config object"""
    ref_cfg = SimpleNamespace(model=SimpleNamespace(model_type="sdxl", min_snr_gamma=5.0, use_ztsnr=True),
                              optimizer=SimpleNamespace(learning_rate=4e-7, weight_decay=0.01),
                              training=SimpleNamespace(method="native_mi355x", native_objective="flow_matching",
                                                       gradient_accumulation_steps=4, clip_grad_norm=1.0, mixed_precision="bf16"))
    ref_opt = SimpleNamespace(param_groups=[{"lr": 2e-6, "betas": (0.8, 0.95), "eps": 1e-7, "weight_decay": 0.02}])
    parent = SimpleNamespace(saved=[], save_checkpoint=lambda e, f=False: parent.saved.append((e, f)))
    tr = NM.NativeMI355XTrainer(model=RefModel(), optimizer=ref_opt, train_dataloader=None, device="cpu", wandb_logger=None,
                                config=ref_cfg, parent_trainer=parent, native_factory=FakeNative,
                                native_config=UN.SDXL_BASE_CFG, )
    assert tr.name == "native_mi355x" and tr.method == "flow_matching" and tr.gradient_accumulation_steps == 4
    g = tr.optimizer.param_groups[0]
    assert (g["lr"], g["betas"], g["eps"], g["weight_decay"]) == (2e-6, (0.8, 0.95), 1e-7, 0.02)
    out = tr.compute_loss(tr.model, _batch(), torch.Generator().manual_seed(0))
    assert set(out["metrics"]) == {"loss", "x0_norm", "x1_norm", "time_mean", "time_std", "velocity_norm", "batch_size", "lr"}
    monkeypatch.chdir(tmp_path)                           # (the reference's checkpoint directories are relative to the working directory)
    tr.save_checkpoint(2, False)                          # ddpm_trainer.py:236-253: through the parent trainer, weights synced first
    assert parent.saved == [(2, False)]
    assert (tmp_path / "outputs" / "checkpoint-0002" / "optimizer.pt").exists()      # the fused optimizer's state, in the reference's layout


def test_config_from_unet_shapes():
    """sdxl_unet_config from a state dict's shapes when the module has no diffusers `.config`"""
    from oracle import unet_ref as U
    cfg = U.tiny_config()
    shapes = U.param_shapes(cfg)
    sd = {k: torch.empty(s) for k, s in shapes.items()}
    # (the width of one time-id embedding is not recoverable from shapes: it comes from `.config`, default 256 as in SDXL-base)
    c = UN.config_from_unet(SimpleNamespace(config=SimpleNamespace(addition_time_embed_dim=cfg.addition_time_embed_dim)), sd)
    assert tuple(c.block_out_channels) == cfg.block_out_channels and tuple(c.transformer_layers) == cfg.transformer_layers_per_block
    assert c.cross_attention_dim == cfg.cross_attention_dim and c.pooled_dim == cfg.pooled_dim
    assert c.addition_time_embed_dim == cfg.addition_time_embed_dim
    # diffusers-style config dict wins over shapes
    c2 = UN.config_from_unet(SimpleNamespace(config={"block_out_channels": [320, 640, 1280], "transformer_layers_per_block": [0, 2, 10],
                                                     "cross_attention_dim": 2048, "addition_time_embed_dim": 256,
                                                     "projection_class_embeddings_input_dim": 2816, "attention_head_dim": [5, 10, 20]}), sd)
    assert tuple(c2.block_out_channels) == (320, 640, 1280) and tuple(c2.transformer_layers) == (0, 2, 10)
    assert c2.pooled_dim == 1280 and c2.head_dim == 64


def test_checkpoint_dir_is_the_references_layout(tmp_path):
    """sdxl_trainer.py:171-178: outputs/checkpoint-<epoch:04d> / outputs/final_checkpoint relative to the working directory; a path is taken as is"""
    from pathlib import Path
    assert T.checkpoint_dir(3) == Path("outputs") / "checkpoint-0003"
    assert T.checkpoint_dir(12, is_final=True) == Path("outputs") / "final_checkpoint"
    assert T.checkpoint_dir(tmp_path / "x", True) == tmp_path / "x" and T.checkpoint_dir(str(tmp_path / "y")) == tmp_path / "y"
