"""config.yaml -> dataclasses, restricted to the keys the hot path reads (reference: src/data/config.py).

Same loading rule as the reference's Config.from_yaml (data/config.py:327-420): start from defaults, overlay only
keys that exist, ignore unknown keys silently, missing file -> defaults.  Shipped defaults = src/config.yaml."""
from __future__ import annotations

from dataclasses import asdict, dataclass, field, fields, is_dataclass
from pathlib import Path
from typing import List, Optional, Union

import yaml


@dataclass
class ModelConfig:                       # data/config.py:11-22
    pretrained_model_name: str = "stabilityai/stable-diffusion-xl-base-1.0"
    model_type: str = "sdxl"
    prediction_type: str = "v_prediction"
    num_timesteps: int = 1000
    sigma_min: float = 0.002
    sigma_max: float = 20000.0
    use_ztsnr: bool = True
    min_snr_gamma: Optional[float] = 5.0
    rho: float = 7.0                     # D1: read by novelai_v3.py:107 but absent from the reference dataclass


@dataclass
class OptimizerConfig:                   # data/config.py:42-49
    learning_rate: float = 1e-6
    weight_decay: float = 0.01
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-8
    optimizer_type: str = "adamw_bf16"
    reference_ema: bool = True           # build-only key: True = the reference's actual first-moment update (its
                                         # add_stochastic_ operand order, SURVEY D17: almost no momentum), False = the
                                         # documented EMA; see optimizer.py::AdamWBF16


@dataclass
class TrainingConfig:                    # data/config.py:152-168
    method: str = "ddpm"                 # "ddpm" | "flow_matching" (sdxl_trainer.py:128-152)
    num_epochs: int = 10
    batch_size: int = 4
    gradient_accumulation_steps: int = 1
    mixed_precision: str = "bf16"
    enable_xformers: bool = True         # accepted, meaningless here (attention is the HIP flash kernel)
    prediction_type: str = "v_prediction"
    clip_grad_norm: float = 1.0
    num_workers: int = 4
    save_final_model: bool = True        # data/config.py:170, config.yaml:39: main.py:108 saves on rank 0 after train() when set
    shard_optimizer: bool = True         # build-only key: data parallel = ZeRO-1 (reduce-scatter -> sharded fused AdamW ->
                                         # all-gather) instead of all-reduce + a full update on every rank
    force_exchange: bool = False         # build-only key: drive the gradient exchange through the backend even at world size 1
                                         # (one-GPU RCCL test, tests/test_gpu_rccl.py); SDXL_FORCE_EXCHANGE=1 does the same


@dataclass
class ImageConfig:                       # data/config.py:182-199
    supported_dims: List[List[int]] = field(default_factory=lambda: [
        [640, 1536], [768, 1344], [832, 1216], [896, 1152], [1024, 1024], [1152, 896], [1216, 832], [1344, 768],
        [1536, 640]])


@dataclass
class GlobalConfig:
    image: ImageConfig = field(default_factory=ImageConfig)


@dataclass
class Config:
    model: ModelConfig = field(default_factory=ModelConfig)
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    global_config: GlobalConfig = field(default_factory=GlobalConfig)

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_yaml(cls, path: Union[str, Path]) -> "Config":
        path = Path(path)
        cfg = cls()
        if not path.exists():
            return cfg
        raw = yaml.safe_load(path.read_text()) or {}
        _overlay(cfg, raw)
        return cfg


def _overlay(obj, data):
    if not isinstance(data, dict):
        return
    names = {f.name for f in fields(obj)}
    for k, v in data.items():
        if k not in names:
            continue                      # unknown keys ignored (data/config.py:351-360)
        cur = getattr(obj, k)
        if is_dataclass(cur):
            _overlay(cur, v)
        else:
            setattr(obj, k, v)
