"""Worker of tests/test_gpu_rccl.py: ONE rank, backend "nccl" (= RCCL), the exchange forced on (GradSync(force=True)).

No byte leaves the GPU at world size 1, but every call is the real one: the process group is RCCL's, the per-segment
`all_reduce` / `reduce_scatter_tensor` are issued asynchronously with the engine's side stream as the current (external) stream
while the backward's remaining kernels are enqueued, the weight-gradient GEMMs write bf16 straight into the exchange arena (emit
mode), then `global_sumsq` (one float all-reduced), the ranged fused AdamW and `all_gather_into_tensor` of the parameter slices
(and of the optimizer arenas: `prepare_checkpoint`).  RCCL's stream semantics, dtype and alignment checks see exactly what an
8-rank run hands them.  (Reference call sites replaced: core/distributed.py:88-131 `setup_distributed`, :153-157 DDP wrap.)

Checked:
  A. GradSync / ShardedGradSync on a flat arena: all_reduce, reduce_scatter_tensor, gather_arena, global_sumsq give the input back
     (world 1: sum over one rank) -- through RCCL, on a side torch stream.
  B. one trainer step (tiny UNet, ddpm, clip on) through ShardedGradSync and one through GradSync, each against a step WITHOUT any
     exchange that is handed the same bf16-rounded gradients: parameters and optimizer state bit for bit.
"""
import ctypes as C
import importlib
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sdxl_amd  # noqa: E402,F401
from oracle import unet_ref as U  # noqa: E402   (the tiny configuration and its synthetic weights: test infrastructure)
from sdxl_amd import lib  # noqa: E402
from sdxl_amd import unet as NU  # noqa: E402

D = importlib.import_module("sdxl-training-improvements_amd.distributed")
T = importlib.import_module("sdxl-training-improvements_amd.trainer")
CFG = importlib.import_module("sdxl-training-improvements_amd.config")
O = importlib.import_module("sdxl-training-improvements_amd.optimizer")


def part_a(dev):
    total = 3 * 65536 + 4096
    segs = [(2 * 65536 + 4096, 65536), (65536, 65536 + 4096), (0, 65536)]
    g = torch.randn(total, generator=torch.Generator().manual_seed(3)).to(dev)
    ref = g.to(torch.bfloat16)

    def cast(off, n, dst):
        dst.copy_(g[off:off + n].to(torch.bfloat16))

    side = torch.cuda.Stream(device=dev)
    ok = True
    for cls in (D.GradSync, D.ShardedGradSync):
        s = cls(total, cast, torch.bfloat16, dev, force=True)
        ok = ok and s.active and s.world == 1
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # collectives are enqueued with a non-default stream current, as in the step
            for k, (off, n) in enumerate(segs):
                s.on_segment(k, off, n)
        s.finish()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if cls is D.GradSync:
            ok = ok and torch.equal(s.reduced(), ref)
        else:
            for off, n, goff in s.pieces:
                ok = ok and torch.equal(s.reduced()[goff:goff + n], ref[off:off + n])
            ok = ok and sum(n for _o, n, _g in s.pieces) == total
            sq = torch.tensor([3.5], device=dev)
            ok = ok and float(s.global_sumsq(sq)) == 3.5
            arena = ref.clone()
            s.gather_arena(arena)
            torch.cuda.synchronize()
            ok = ok and torch.equal(arena, ref)
    return ok


def make_net(cfg, w):
    net = NU.NativeUNet(NU.make_config(block_out_channels=cfg.block_out_channels, transformer_layers=cfg.transformer_layers_per_block,
                                       cross_attention_dim=cfg.cross_attention_dim, addition_time_embed_dim=cfg.addition_time_embed_dim,
                                       pooled_dim=cfg.pooled_dim))
    net.load_state_dict(w)
    return net


def part_b(dev, sharded):
    cfg = U.tiny_config()
    w = U.synth_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g)
    bfr = lambda t: t.to(torch.bfloat16).float()
    B = 2
    batch = {"vae_latents": r(B, 4, 16, 16), "prompt_embeds": bfr(r(B, 77, cfg.cross_attention_dim)),
             "pooled_prompt_embeds": bfr(r(B, cfg.pooled_dim)), "time_ids": torch.tensor([[[128.0, 128, 0, 0, 128, 128]]] * B), "metadata": {}}
    noise = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(12))
    ts = torch.tensor([250, 777])

    def trainer(force):
        c = CFG.Config()
        c.training.method = "ddpm"
        c.training.mixed_precision = "no"
        c.training.clip_grad_norm = 1e-3                   # small: the clip coefficient is active (< 1)
        c.training.shard_optimizer = sharded
        c.training.force_exchange = force
        net = make_net(cfg, w)
        opt = O.AdamWBF16(net, lr=1e-2, weight_decay=0.1, seed=5)
        class M: unet = net
        return T.NativeSDXLTrainer(M(), optimizer=opt, config=c), net, opt

    os.environ.pop("SDXL_FORCE_EXCHANGE", None)            # the trainers below say it themselves
    # ---- the step through RCCL
    tr, net, opt = trainer(True)
    ok = tr.sync.active and isinstance(tr.sync, D.ShardedGradSync) == sharded and tr.sync.comm is not None
    loss, _m = tr._execute_training_step(batch, timesteps=ts, noise=noise)
    ok = ok and tr._emit                                    # emit mode was on: the wgrad GEMMs wrote the exchange arena themselves
    gn = tr.optimizer_step()
    torch.cuda.synchronize()
    if sharded:
        ok = ok and tr._zero1_active()
        tr.prepare_checkpoint()                             # all-gather of the three optimizer arenas: through RCCL as well
    # ---- the same step with NO exchange, handed the bf16-rounded gradients (what the exchange arena holds at world size 1)
    tr2, net2, opt2 = trainer(False)
    ok = ok and not tr2.sync.active
    loss2, _m2 = tr2._execute_training_step(batch, timesteps=ts, noise=noise)
    gb = torch.empty(net2.param_elems, dtype=torch.bfloat16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.check(net2.L.sdxl_grads_to_bf16(net2.h, 0, net2.param_elems, C.c_void_p(gb.data_ptr()), 1.0, st))
    buf = torch.zeros(2, dtype=torch.float32, device=dev)
    n8 = (gb.numel() // 8) * 8
    lib.check(net2.L.sdxl_sumsq(C.c_void_p(gb.data_ptr()), 1, n8, C.c_void_p(buf.data_ptr()), st))
    lib.check(net2.L.sdxl_clip_coef(C.c_void_p(buf.data_ptr()), 1e-3, C.c_void_p(buf.data_ptr() + 4), st))
    opt2.step(gb, grad_scale=buf[1:2])
    torch.cuda.synchronize()
    ok = ok and float(loss) == float(loss2) and float(buf[1]) < 1.0
    # squared norm: slice-wise + all-reduce against one pass over the arena -- the same summands, fp32 order may differ in the last bits
    ok = ok and abs(gn - float(buf[0].sqrt())) <= 1e-5 * gn
    same_w = torch.equal(net.weights, net2.weights)
    same_s = all(torch.equal(getattr(opt, a), getattr(opt2, a)) for a in ("exp_avg", "exp_avg_sq", "shift"))
    moved = not torch.equal(net.weights, make_net(cfg, w).weights)
    print(f"[rccl] sharded={sharded} loss {float(loss):.6f} grad-norm {gn:.6e} weights-equal {same_w} state-equal {same_s} moved {moved}", flush=True)
    return ok and same_w and same_s and moved


def main():
    os.environ["SDXL_FORCE_EXCHANGE"] = "1"
    D.init_process_group("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ok = part_a(dev)
    print(f"[rccl] backend {dist.get_backend()} world {dist.get_world_size()} part A {'ok' if ok else 'MISMATCH'}", flush=True)
    okb = part_b(dev, True)
    okc = part_b(dev, False)
    print("RCCL_OK" if (ok and okb and okc) else "RCCL_MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
