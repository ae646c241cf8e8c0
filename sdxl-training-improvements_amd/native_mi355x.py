"""Method module for the reference's plugin procedure (README.md:22-29: copy `methods/example_method.py`, set `name`,
implement `compute_loss`, select it in config.yaml).

Drop this file into `src/training/trainers/methods/` of DataCTE/SDXL-Training-Improvements (with this package importable) and
add the dispatcher lines of INTEGRATION.md section 1(b) to `sdxl_trainer.py`; `training.method: "native_mi355x"` then builds
this class with the reference's own constructor call (sdxl_trainer.py:130-150):

    NativeMI355XTrainer(model=model, optimizer=optimizer, train_dataloader=train_dataloader, device=device,
                        wandb_logger=wandb_logger, config=config, parent_trainer=self)

The objective comes from `training.native_objective` ("ddpm" | "flow_matching", default "ddpm"), everything else from the
reference's config.  `model.unet` (the diffusers UNet loaded by models/sdxl.py:25-40) is imported into the native engine at
construction and written back by `save_checkpoint` / `sync_to_model`.  The reference's AdamWBF16 instance (main.py:73-86)
holds per-tensor PyTorch state and is replaced by the fused equivalent on the packed arena (same hyper-parameters, read from
its `param_groups`)."""
from __future__ import annotations

import copy

from .config import Config
from .trainer import NativeSDXLTrainer


def _get(obj, path, default=None):
    for k in path.split("."):
        obj = obj.get(k) if isinstance(obj, dict) else getattr(obj, k, None)
        if obj is None:
            return default
    return obj


class NativeMI355XTrainer(NativeSDXLTrainer):
    name = "native_mi355x"

    def __init__(self, model, optimizer=None, train_dataloader=None, device=None, wandb_logger=None, config=None,
                 parent_trainer=None, **kwargs):
        cfg = Config()
        if config is not None:                      # the reference's Config object (or a dict): copy the keys the hot path reads
            for sec in ("model", "optimizer", "training"):
                dst = getattr(cfg, sec)
                for f in list(vars(dst)):
                    v = _get(config, f"{sec}.{f}")
                    if v is not None:
                        setattr(dst, f, copy.deepcopy(v))
        method = str(_get(config, "training.method", "native_mi355x")).lower()
        cfg.training.method = str(_get(config, "training.native_objective", "ddpm")).lower() if method == self.name else method
        native_opt = None
        groups = getattr(optimizer, "param_groups", None)
        if groups is not None:                      # take the hyper-parameters of the optimizer main.py built
            g = next(iter(groups))
            cfg.optimizer.learning_rate = float(g.get("lr", cfg.optimizer.learning_rate))
            b = g.get("betas", (cfg.optimizer.beta1, cfg.optimizer.beta2))
            cfg.optimizer.beta1, cfg.optimizer.beta2 = float(b[0]), float(b[1])
            cfg.optimizer.epsilon = float(g.get("eps", cfg.optimizer.epsilon))
            cfg.optimizer.weight_decay = float(g.get("weight_decay", cfg.optimizer.weight_decay))
        self.parent_trainer = parent_trainer
        super().__init__(model, native_opt, train_dataloader, device, wandb_logger, cfg, **kwargs)

    def save_checkpoint(self, epoch_or_path=0, is_final: bool = False):
        """No collective (the reference calls this on rank 0 only): under ZeRO-1 every rank calls prepare_checkpoint() first."""
        self.sync_to_model()
        is_path = isinstance(epoch_or_path, (str, bytes)) or hasattr(epoch_or_path, "__fspath__")
        if self.parent_trainer is not None and callable(getattr(self.parent_trainer, "save_checkpoint", None)) and not is_path:
            out = self.parent_trainer.save_checkpoint(epoch_or_path, is_final)       # ddpm_trainer.py:236-253
            # the parent wrote ITS optimizer's state (the reference's torch AdamWBF16, which never stepped); the state that
            # trained the weights is the fused optimizer's: replace optimizer.pt in the directory the parent reports, else in
            # the reference's own layout (trainer.checkpoint_dir = sdxl_trainer.py:171-178)
            from pathlib import Path
            from .trainer import checkpoint_dir
            save_dir = Path(out) if isinstance(out, (str, Path)) else checkpoint_dir(epoch_or_path, is_final)
            save_dir.mkdir(parents=True, exist_ok=True)
            self.save_optimizer_state(save_dir)
            return out
        return super().save_checkpoint(epoch_or_path, is_final)
