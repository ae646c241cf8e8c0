#!/usr/bin/env python3
"""batch-vs-per-sample gradient decomposition at B = 5, 512^2 with the fused Delta epilogue (knob 21 = 0) and without it (knob 21 = 1): diagnostics build"""
import os, sys
os.environ["SDXL_DIAG"] = "1"
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import sdxl_amd  # noqa
from sdxl_amd import lib, synth, unet as NU
from oracle import loss_ref as R
L = lib.load()
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.check(L.sdxl_set_knob(21, knob))
net = NU.NativeUNet(NU.make_config()); synth.load_synthetic(net, seed=0)
B, H, W = 5, 64, 64
g = torch.Generator().manual_seed(1200 + B)
r = lambda *s: torch.randn(*s, generator=g)
bfr = lambda t: t.to(torch.bfloat16).float()
x = dict(lat=r(B, 4, H, W), noise=r(B, 4, H, W), ehs=bfr(r(B, 77, 2048)), pooled=bfr(r(B, 1280)), tid=torch.tensor([[8.0 * W, 8.0 * H, 0, 0, 8.0 * W, 8.0 * H]] * B))
ts = torch.tensor([450, 613, 700, 820, 377]); sig = R.karras_sigmas()[ts]
probes = ["down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight", "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_out.0.weight", "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.2.weight"]
def run(idx, scale, first):
    s = slice(idx, idx + 1) if idx is not None else slice(None)
    net.forward_loss("ddpm", x["lat"][s], x["noise"][s], sig[s], ts[s].float(), x["ehs"][s], x["pooled"][s], x["tid"][s])
    net.backward(scale, first)
net.zero_grads(); run(None, 1.0, True)
gb = {k: net.export(k, grad=True).clone() for k in probes}
net.zero_grads()
for i in range(B): run(i, 1.0 / B, i == 0)
for k in probes:
    gs = net.export(k, grad=True)
    print(f"knob21={knob} {k}: rel {float((gs - gb[k]).norm() / gb[k].norm()):.3e}")
