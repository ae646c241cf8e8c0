"""Trainer-plugin surface of the native path -- the drop-in for the reference's method trainers.

Reference contract mirrored here:
  * constructed like every trainer: (model, optimizer, train_dataloader, device, wandb_logger=None, config=None, **kw)
    (training/trainers/base_router.py:14-31, sdxl_trainer.py:18-41);
  * `compute_loss(batch) -> {"loss": 0-d tensor with .backward(), "metrics": dict}` (methods/example_method.py:108-122;
    flow matching signature `(model, batch, generator=None)` is accepted too, flow_matching_trainer.py:261);
  * `training_step(batch)` = the DDPM name of the same thing (ddpm_trainer.py:280);
  * `_execute_training_step(batch, accumulate, is_last_accumulation_step) -> (loss, metrics)`
    (ddpm_trainer.py:256-278 / flow_matching_trainer.py:237-259), with the D10 repair: gradients are zeroed at the
    START of an accumulation cycle, not before its last micro-step;
  * `train(num_epochs)`: the template loop (methods/example_method.py:150-230): every N micro-steps clip -> step -> zero;
  * batch dict keys {"vae_latents","prompt_embeds","pooled_prompt_embeds","time_ids","metadata"} else ValueError
    (ddpm_trainer.py:284-290); optional "tag_weights" [B];
  * metric keys: DDPM loss, lr, timestep_mean, timestep_std, noise_scale, pred_scale, batch_size (ddpm_trainer.py:386-396);
    flow matching loss, x0_norm, x1_norm, time_mean, time_std, velocity_norm, batch_size, lr
    (flow_matching_trainer.py:338-347).
Selected by `training.method` in config.yaml exactly like the reference (sdxl_trainer.py:128-152): "ddpm" or
"flow_matching"; anything else raises ValueError.  (The module `native_mi355x.py` is the file a maintainer drops into the
reference's `methods/` directory; it registers this class under the method name "native_mi355x".)

`model` is what the reference hands every trainer (models/sdxl.py:11-62): an object whose `.unet` is the PyTorch /
diffusers UNet.  Its diffusers-keyed `state_dict()` is imported into the packed native arena at construction
(`sdxl_load_weight`), and `sync_to_model()` / `save_checkpoint()` write the trained weights back into that module, so the
reference's `model.save_pretrained` (models/sdxl.py:246-288) keeps producing a loadable diffusers checkpoint.  A
`NativeUNet` may be passed directly as well.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import time
from collections import defaultdict
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from . import distributed as D
from . import lib
from .config import Config
from .optimizer import AdamWBF16
from .scheduler import NoiseScheduler
from .unet import NativeUNet, config_from_unet

REQUIRED_KEYS = {"vae_latents", "prompt_embeds", "pooled_prompt_embeds", "time_ids", "metadata"}


class _NativeLoss(torch.autograd.Function):
    """0-d loss whose backward runs the HIP backward with the incoming scale (so `(loss / N).backward()` works)."""

    @staticmethod
    def forward(ctx, anchor, trainer, value):
        ctx.trainer = trainer
        return anchor.new_tensor(value)

    @staticmethod
    def backward(ctx, grad_out):
        ctx.trainer._native_backward(float(grad_out))
        return None, None, None


class NativeSDXLTrainer:
    """SDXL trainer whose compute_loss / backward run in libsdxlstep (HIP, gfx950)."""

    name = "native_mi355x"

    def __init__(self, model, optimizer=None, train_dataloader=None, device=None, wandb_logger=None,
                 config: Optional[Config] = None, **kwargs):
        self.model = model
        self.train_dataloader = train_dataloader
        self.device = device if device is not None else torch.device("cuda", 0)
        self.wandb_logger = wandb_logger
        self.config = config if config is not None else Config()
        method = str(self.config.training.method).lower()
        if method not in ("ddpm", "flow_matching"):
            raise ValueError(f"Unsupported training method: {self.config.training.method}")   # sdxl_trainer.py:151
        self.method = method
        self.gradient_accumulation_steps = int(self.config.training.gradient_accumulation_steps)
        unet = model.unet if hasattr(model, "unet") else model
        native_attrs = ("forward_loss", "backward", "read_loss", "zero_grads", "param_elems")
        self._torch_unet = None
        if all(hasattr(unet, a) for a in native_attrs):
            self.net = unet                                            # already a NativeUNet
        elif callable(getattr(unet, "state_dict", None)):
            # the reference's model object: import the PyTorch UNet's diffusers-keyed weights (models/sdxl.py:40, :92)
            sd = unet.state_dict()
            factory = kwargs.pop("native_factory", None)               # (tests: a stand-in for machines without a GPU)
            ncfg = kwargs.pop("native_config", None) or config_from_unet(unet, sd)
            dev = self.device.index if isinstance(self.device, torch.device) and self.device.index is not None else 0
            self.net = factory(ncfg) if factory is not None else NativeUNet(ncfg, device=dev)
            self.net.load_state_dict(sd, strict=True)                   # KeyError on any missing / unexpected key
            self._torch_unet = unet
        else:
            raise TypeError("model.unet must be a NativeUNet or a module with a diffusers-keyed state_dict(); got "
                            f"{type(unet).__name__}")
        self.noise_scheduler = NoiseScheduler(self.config, "cpu")
        self.optimizer = optimizer if optimizer is not None else AdamWBF16(       # main.py:73-86 (optimizer_type adamw_bf16)
            self.net, lr=self.config.optimizer.learning_rate, betas=(self.config.optimizer.beta1, self.config.optimizer.beta2),
            eps=self.config.optimizer.epsilon, weight_decay=self.config.optimizer.weight_decay,
            reference_ema=bool(getattr(self.config.optimizer, "reference_ema", True)))
        self._clip_coef = None
        # data parallel: ZeRO-1 (reduce-scatter, sharded fused AdamW, all-gather) with the fused optimizer, else all-reduce
        want_sharded = bool(getattr(self.config.training, "shard_optimizer", True)) and isinstance(self.optimizer, AdamWBF16)
        seg_sizes = [n for _off, n in self.net.segment_ranges()] if hasattr(self.net, "segment_ranges") else None
        # force_exchange (build-only key / SDXL_FORCE_EXCHANGE=1): run the exchange through the backend even at world size 1
        force = bool(getattr(self.config.training, "force_exchange", False)) or os.environ.get("SDXL_FORCE_EXCHANGE", "0") == "1"
        self.sync = D.make_grad_sync(self.net.param_elems, self._cast, torch.bfloat16, getattr(self.net, "device", "cpu"),
                                     sharded=want_sharded, segment_sizes=seg_sizes, force=force)
        self.sharded = isinstance(self.sync, D.ShardedGradSync)      # (falls back to all-reduce where the segments do not split)
        self._emit = False                   # this backward's weight-gradient GEMMs write the bf16 exchange arena themselves
        self._micro = 0                      # micro-step index inside the accumulation cycle
        self._zeroed = False                 # gradients already zeroed for the cycle in progress
        self._anchor = torch.zeros((), requires_grad=True)
        self._exchange = True
        self._final_known = False            # this backward is known to be the cycle's last micro-step (emit mode is safe)
        hook = getattr(self.optimizer, "register_step_post_hook", None)
        if callable(hook):                   # an optimizer step ends the accumulation cycle, whoever calls it
            hook(lambda *_a, **_k: self._end_cycle())

    # -------------------------------------------------------------------------------- loss
    def _cast(self, off, n, dst):
        # the exchange micro-step's weight-gradient GEMMs wrote bf16 into the comm arena themselves (set_grad_emit): what is
        # left to cast per bucket are the biases / norm parameters (fp32 atomic accumulators)
        if self._emit:
            self.net.cast_small(off, n, dst)
            return
        lib.check(self.net.L.sdxl_grads_to_bf16(self.net.h, off, n, C.c_void_p(dst.data_ptr()), 1.0,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def compute_loss(self, *args, generator: Optional[torch.Generator] = None, timesteps=None, noise=None) -> Dict[str, Any]:
        """compute_loss(batch) or compute_loss(model, batch[, generator]).  `timesteps` (ddpm: int64 indices, flow
        matching: t in (0,1)) and `noise` (ddpm noise / flow-matching x0) may be injected for reproducible fixtures;
        otherwise they are drawn as the reference draws them."""
        batch = args[-1] if not isinstance(args[-1], torch.Generator) else args[-2]
        if isinstance(args[-1], torch.Generator):
            generator = args[-1]
        if not all(k in batch for k in REQUIRED_KEYS):
            raise ValueError(f"Batch missing required keys: {REQUIRED_KEYS - set(batch.keys())}")
        lat = batch["vae_latents"].float()
        B = lat.shape[0]
        cm = self.config.model
        if noise is None:
            noise = torch.randn(lat.shape, generator=generator)
        tag = batch.get("tag_weights")
        if self.method == "ddpm":
            ts = timesteps if timesteps is not None else self.noise_scheduler.sample_timesteps(B, generator)
            sig = self.noise_scheduler.timestep_to_sigma(ts)
            self.net.forward_loss("ddpm", lat, noise, sig, ts.float(), batch["prompt_embeds"],
                                  batch["pooled_prompt_embeds"], batch["time_ids"], tag,
                                  prediction_type=self.config.training.prediction_type,
                                  min_snr_gamma=cm.min_snr_gamma, use_ztsnr=cm.use_ztsnr)
        else:
            if timesteps is None:                                      # sample_logit_normal, :373-385
                timesteps = torch.sigmoid(torch.randn(B, generator=generator))
            t = timesteps.float()
            if str(self.config.training.mixed_precision) == "bf16":     # D6: t is handed to the UNet in model dtype
                t_unet = t.to(torch.bfloat16).float()
            else:
                t_unet = t
            self.net.forward_loss("flow_matching", lat, noise, t, t_unet, batch["prompt_embeds"],
                                  batch["pooled_prompt_embeds"], batch["time_ids"], tag)
            ts = t
        o = self.net.read_loss()                                        # the single host sync of the step
        numel = lat.numel()
        lr = self.optimizer.param_groups[0]["lr"] if self.optimizer is not None else 0.0
        if self.method == "ddpm":
            metrics = {"loss": o[0], "lr": lr, "timestep_mean": float(ts.float().mean()),
                       "noise_scale": o[4] / numel, "pred_scale": o[2] / numel, "batch_size": B}
            if B > 1:
                metrics["timestep_std"] = float(ts.float().std())
        else:
            metrics = {"loss": o[0], "x0_norm": math.sqrt(o[5]), "x1_norm": math.sqrt(o[6]),
                       "time_mean": float(ts.mean()), "time_std": float(ts.std()) if B > 1 else float("nan"),
                       "velocity_norm": math.sqrt(o[3]), "batch_size": B, "lr": lr}
        loss = _NativeLoss.apply(self._anchor, self, o[0])
        return {"loss": loss, "metrics": metrics}

    training_step = compute_loss                                        # DDPM trainer's name for it

    def _end_cycle(self) -> None:
        self._micro = 0
        self._zeroed = False

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Start a new accumulation cycle (the direct `compute_loss(...)["loss"].backward()` loop calls this or
        `optimizer.step()` between cycles, like the reference's template loop, example_method.py:191-206)."""
        self.net.zero_grads()
        self._micro = 0
        self._zeroed = True

    def _native_backward(self, grad_scale: float) -> None:
        """backward of the 0-d loss: runs from `_execute_training_step` and from a caller-owned
        `compute_loss(batch)["loss"].backward()` loop alike, so the accumulation state lives here."""
        first = self._micro == 0
        if first and not self._zeroed:       # the small-parameter gradients accumulate with atomics: zero them per cycle
            self.net.zero_grads()
        world = self.sync.world
        exchange = self._exchange and self.sync.active
        self.sync.enabled = exchange
        # per-segment joins (side stream -> caller's stream) only on the micro-step that exchanges gradients
        if exchange:    # bucket casts + collectives ride the engine's side stream, behind the segment's weight gradients
            # emit mode (wgrad GEMMs write bf16 straight into the exchange arena, the fp32 arena is NOT written) is only correct
            # on the LAST micro-step of a cycle: a later micro-step would add to an fp32 arena that never received this one.
            # `_execute_training_step` knows that; a caller-owned `compute_loss(...)["loss"].backward()` loop with accumulation does
            # not say which backward is the last, so every one of its micro-steps takes the fp32-accumulate + cast path.
            final = self._final_known or self.gradient_accumulation_steps == 1
            self._emit = final and hasattr(self.net, "set_grad_emit") and self.sync.comm is not None
            if self._emit:
                self.net.set_grad_emit(self.sync.comm, 1.0)
            try:
                self.net.backward(grad_scale / world, first, on_segment=self.sync.on_segment, segment_stream=True)
            finally:
                if self._emit:
                    self.net.set_grad_emit(None)
        else:
            self.net.backward(grad_scale / world, first)
        self._micro += 1
        self._zeroed = False

    # -------------------------------------------------------------------------------- loop pieces
    def _execute_training_step(self, batch, accumulate: bool = False, is_last_accumulation_step: bool = True, **kw):
        N = self.gradient_accumulation_steps if accumulate else 1
        if self._micro == 0:
            self.net.zero_grads()                                      # start of the cycle (D10 repair)
            self._zeroed = True
        self._exchange = (not accumulate) or is_last_accumulation_step
        self._final_known = self._exchange
        out = self.compute_loss(batch, **kw)
        loss = out["loss"] / N if accumulate else out["loss"]
        try:
            loss.backward()
        finally:
            self._final_known = False
        if self._exchange:
            self._end_cycle()
            self.sync.finish()
        self._exchange = True
        return loss.detach() * N, out["metrics"]

    def clip_grad_norm_(self, max_norm: float) -> float:
        """torch.nn.utils.clip_grad_norm_ over the flat arena (flow_matching_trainer.py:181-186).  Under data parallelism
        the norm is taken over the exchanged gradients: the whole all-reduced arena, or (ZeRO-1) this rank's
        reduce-scattered slices + one float all-reduced -- the coefficient is then the same bits on every rank."""
        self.sync.finish()                                   # the asynchronous exchange must have landed
        fused = isinstance(self.optimizer, AdamWBF16)        # the coefficient rides into the fused optimizer kernel
        g = self.sync.reduced() if self.sync.active else self.net.grads
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if g.is_cuda else None
        if g.is_cuda:                                        # squared norm + coefficient on the device (HIP kernels)
            buf = torch.empty(2, dtype=torch.float32, device=g.device)
            n = (g.numel() // 8) * 8
            lib.check(self.net.L.sdxl_sumsq(C.c_void_p(g.data_ptr()), 0 if g.dtype == torch.float32 else 1, n,
                                            C.c_void_p(buf.data_ptr()), st), "sdxl_sumsq")
            if self.sharded and self.sync.active:
                self.sync.global_sumsq(buf[0:1])
            lib.check(self.net.L.sdxl_clip_coef(C.c_void_p(buf.data_ptr()), float(max_norm),
                                                C.c_void_p(buf.data_ptr() + 4), st), "sdxl_clip_coef")
            if fused:
                self._clip_coef = buf[1:2]
            else:
                g.mul_(buf[1])
            return float(buf[0].sqrt())                      # the reference logs the norm (one read-back)
        sq = g.float().pow(2).sum().reshape(1)               # host-logic tests with a stand-in net (no GPU)
        if self.sharded and self.sync.active:
            self.sync.global_sumsq(sq)
        norm = float(sq.sqrt())
        coef = max_norm / (norm + 1e-6) if norm > max_norm else 1.0
        if fused:
            self._clip_coef = torch.tensor([coef], dtype=torch.float32) if coef != 1.0 else None
        elif coef != 1.0:
            g.mul_(coef)
        return norm

    def optimizer_step(self) -> Optional[float]:
        self.sync.finish()
        gn = None
        if self.config.training.clip_grad_norm and self.config.training.clip_grad_norm > 0:
            gn = self.clip_grad_norm_(float(self.config.training.clip_grad_norm))
        if self.optimizer is not None:
            if isinstance(self.optimizer, AdamWBF16):
                if self.sync.active and self.sharded:        # ZeRO-1: update this rank's slices, then all-gather the parameters
                    self.optimizer.step(self.sync.reduced(), grad_scale=self._clip_coef, pieces=self.sync.pieces)
                    self.sync.gather_params(self.net.weights)
                else:
                    self.optimizer.step(self.sync.reduced() if self.sync.active else None, grad_scale=self._clip_coef)
                self._clip_coef = None
            else:
                self.optimizer.step()
        self._end_cycle()
        return gn

    def train(self, num_epochs: int, save_checkpoints: bool = False) -> None:
        """The template loop.  save_checkpoints: the reference's cadence (flow_matching_trainer.py:211-234): a checkpoint
        whenever the epoch's mean loss improves, and `final_checkpoint` at the end (off by default: writing the 5 GB UNet
        is the caller's decision, `SDXLTrainer.save_checkpoint` in the reference)."""
        N = self.gradient_accumulation_steps
        global_step = 0
        best = float("inf")
        for epoch in range(num_epochs):
            acc_loss, acc_metrics = 0.0, defaultdict(float)
            ep_loss, ep_n = 0.0, 0
            for step, batch in enumerate(self.train_dataloader):
                t0 = time.time()
                last = (step + 1) % N == 0
                loss, metrics = self._execute_training_step(batch, accumulate=True, is_last_accumulation_step=last)
                acc_loss += float(loss)
                ep_loss, ep_n = ep_loss + float(loss), ep_n + 1
                for k, v in metrics.items():
                    acc_metrics[k] += v
                if last:
                    eff = {k: v / N for k, v in acc_metrics.items()}
                    gn = self.optimizer_step()
                    if gn is not None:
                        eff["grad_norm"] = gn
                    eff.update(epoch=epoch + 1, step=global_step, loss=acc_loss / N, step_time=time.time() - t0)
                    if D.is_main_process():
                        if self.wandb_logger is not None:
                            self.wandb_logger.log_metrics(eff, step=global_step)
                        else:
                            print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in eff.items()}, flush=True)
                    acc_loss, acc_metrics = 0.0, defaultdict(float)
                global_step += 1
            if save_checkpoints:
                # the decision must be the same on every rank (each sees its own data): the epoch's loss sum and step count, both summed
                # over ranks -- EVERY rank enters the reduction, also one whose shard produced no step this epoch (an uneven loader
                # shard; a guard on the local count would leave the others waiting in the collective).
                # prepare_checkpoint() is the collective part (every rank), save_checkpoint() itself has none (rank 0 writes).
                tot = D.reduce_dict({"loss_sum": ep_loss, "n": float(ep_n)}, average=False)
                mean = tot["loss_sum"] / tot["n"] if tot["n"] > 0 else float("inf")
                if mean < best:
                    best = mean
                    self.prepare_checkpoint()
                    self.save_checkpoint(epoch + 1, is_final=False)
        # The reference's main.py:108-111 calls save_checkpoint(path) on rank 0 only right after train() returns, when
        # `training.save_final_model` is set (config.yaml:39, the default).  Under ZeRO-1 that save needs every rank's slices of the
        # moments, and save_checkpoint itself may hold no collective -- so the gather (three arenas, every rank) happens here, but only
        # when a save can follow: this loop's own final checkpoint, or the caller's under save_final_model.  Without either, train() ends
        # without any collective beyond the steps', and a rank-0-only save afterwards fails loudly (save_checkpoint) instead of writing
        # a state that could not be resumed.
        if save_checkpoints or bool(getattr(self.config.training, "save_final_model", True)):
            self.prepare_checkpoint()
        if save_checkpoints:
            self.save_checkpoint(num_epochs, is_final=True)

    # -------------------------------------------------------------------------------- weights out (row f4)
    def sync_to_model(self) -> None:
        """Write the trained native weights back into the caller's PyTorch UNet (diffusers keys, the module's own dtypes), so
        everything the reference does with `model.unet` afterwards -- `save_pretrained` (models/sdxl.py:246-288), validation
        sampling -- sees them."""
        if self._torch_unet is None:
            return
        sd = self.net.state_dict()
        ref = self._torch_unet.state_dict()
        self._torch_unet.load_state_dict({k: v.to(device=ref[k].device, dtype=ref[k].dtype) for k, v in sd.items()}, strict=True)

    def save_checkpoint(self, epoch_or_path=0, is_final: bool = False) -> Optional[Path]:
        """sdxl_trainer.py:162-210: `outputs/checkpoint-<epoch>` or `outputs/final_checkpoint` (main.py:111 passes a
        directory instead of an epoch: accepted too); the model through `model.save_pretrained(dir, safe_serialization=True)`
        when the caller's model has it (weights synced back first), else the UNet as diffusers-keyed safetensors;
        `optimizer.pt` = optimizer.state_dict(); `config.json` = the training config."""
        # No collective in here: the reference calls save_checkpoint on rank 0 only (main.py:110-111, flow_matching_trainer.py:218-219,
        # ddpm_trainer.py:235-237), so a gather at this point would leave rank 0 alone in it.  Under ZeRO-1 the complete optimizer
        # state needs prepare_checkpoint() on EVERY rank first (train() does that); without it optimizer.pt holds rank 0's slices of
        # exp_avg / exp_avg_sq / shift only: that raises (after the weights and config.json were written).
        if not D.is_main_process():
            return None
        save_dir = checkpoint_dir(epoch_or_path, is_final)
        save_dir.mkdir(parents=True, exist_ok=True)
        self.sync_to_model()
        if self._torch_unet is not None and callable(getattr(self.model, "save_pretrained", None)):
            self.model.save_pretrained(str(save_dir), safe_serialization=True)
        else:
            from safetensors.torch import save_file
            (save_dir / "unet").mkdir(exist_ok=True)
            save_file({k: v.cpu().contiguous() for k, v in self.net.state_dict().items()},
                      str(save_dir / "unet" / "diffusion_pytorch_model.safetensors"))
        with open(save_dir / "config.json", "w") as f:
            json.dump(self.config.to_dict(), f, indent=2)
        if self.optimizer is not None and callable(getattr(self.optimizer, "state_dict", None)):
            torch.save(self._optimizer_state_for_save(), str(save_dir / "optimizer.pt"))      # (raises under ZeRO-1 on a stale state)
        return save_dir

    def _zero1_active(self) -> bool:
        return bool(self.sharded and self.sync.active and isinstance(self.optimizer, AdamWBF16) and getattr(self.sync, "buckets", None))

    def prepare_checkpoint(self) -> None:
        """COLLECTIVE -- every rank calls it, at the same point of its loop, before rank 0 calls save_checkpoint().  Under ZeRO-1
        each rank has updated exp_avg / exp_avg_sq / shift on its own slices only: all-gather them so that the optimizer.pt rank 0
        writes holds the complete state.  No-op otherwise.  The gathered state stays valid until the next optimizer step."""
        if self._zero1_active():
            for arena in (self.optimizer.exp_avg, self.optimizer.exp_avg_sq, self.optimizer.shift):
                self.sync.gather_arena(arena)
        self._opt_state_step = self._step_counter()

    def _step_counter(self):
        return getattr(self.optimizer, "step_count", None) if self.optimizer is not None else None

    def _optimizer_state_for_save(self) -> dict:
        osd = self.optimizer.state_dict()
        if isinstance(osd.get("state"), dict):
            osd["state"] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in osd["state"].items()}
        if self._zero1_active() and getattr(self, "_opt_state_step", object()) != self._step_counter():
            # Surfaces at SAVE time, not at resume time: a file with this rank's slices only could never be loaded again
            # (load_optimizer_state refuses it), and the run that could still have gathered the state would be long gone.
            raise RuntimeError(
                "save_checkpoint under ZeRO-1 without prepare_checkpoint() on every rank since the last optimizer step: this rank "
                f"(rank {int(self.sync.rank)} of {int(self.sync.world)}) holds only its own slices of exp_avg / exp_avg_sq / shift. "
                "Call prepare_checkpoint() on EVERY rank first (train() does, before a final save), or train with "
                "training.shard_optimizer = false when the caller's loop saves on rank 0 only (INTEGRATION.md section 3). "
                "The model weights and config.json of this checkpoint were written; optimizer.pt was not.")
        return osd

    def save_optimizer_state(self, save_dir) -> None:
        """optimizer.pt of the NATIVE fused optimizer.  No collective (see save_checkpoint): complete under ZeRO-1 after
        prepare_checkpoint() on every rank."""
        if not D.is_main_process() or self.optimizer is None or not callable(getattr(self.optimizer, "state_dict", None)):
            return
        torch.save(self._optimizer_state_for_save(), str(Path(save_dir) / "optimizer.pt"))

    def load_optimizer_state(self, checkpoint_dir) -> None:
        """resume: optimizer.pt written by save_checkpoint (the UNet weights come back through the model object).  The state
        holds tensors, numbers and dicts only, so the safe loader is enough."""
        sd = torch.load(str(Path(checkpoint_dir) / "optimizer.pt"), map_location="cpu", weights_only=True)
        part = sd.pop("zero1_partial", None) if isinstance(sd, dict) else None
        if part is not None:
            # written under ZeRO-1 without prepare_checkpoint() on every rank (the drop-in path: the reference's loop calls
            # save_checkpoint on rank 0 only): the moments of the other ranks' slices in it are stale.  Resuming from it would be
            # silently wrong on (world - 1) / world of the arena -- refuse.
            raise ValueError(
                f"optimizer.pt is a ZeRO-1 partial state (rank {part.get('rank')} of {part.get('world')}: only that rank's slices of "
                "exp_avg / exp_avg_sq / shift are current). Call prepare_checkpoint() on every rank before save_checkpoint(), or train "
                "with training.shard_optimizer = false when the caller's loop saves on rank 0 only (INTEGRATION.md section 3)")
        self.optimizer.load_state_dict(sd)


def checkpoint_dir(epoch_or_path, is_final: bool = False) -> Path:
    """The directory sdxl_trainer.py:171-178 writes a checkpoint to: `outputs/final_checkpoint` or `outputs/checkpoint-<epoch:04d>`
    (relative to the working directory, as in the reference); a path is taken as it is (main.py:111)."""
    if isinstance(epoch_or_path, (str, bytes, Path)) or hasattr(epoch_or_path, "__fspath__"):
        return Path(epoch_or_path)
    return Path("outputs") / ("final_checkpoint" if is_final else f"checkpoint-{int(epoch_or_path):04d}")


def create_trainer(model, optimizer=None, train_dataloader=None, device=None, wandb_logger=None, config=None, **kw):
    """BaseRouter.create equivalent for model_type == "sdxl" (base_router.py:48-84)."""
    if config is not None and str(config.model.model_type).lower() != "sdxl":
        raise ValueError(f"Unsupported model type: {config.model.model_type}")
    return NativeSDXLTrainer(model, optimizer, train_dataloader, device, wandb_logger, config, **kw)
