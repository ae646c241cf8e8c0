"""Product-side synthetic weights agree bit for bit with the oracle's independent restatement of the recipe."""
import torch

import sdxl_amd  # noqa: F401
from oracle import unet_ref as U
from sdxl_amd import synth


def test_key_order_and_values_match_oracle():
    for cfg in (U.tiny_config(), U.SDXL_BASE):
        shapes = {k: tuple(v) for k, v in U.param_shapes(cfg).items()}
        order = synth.diffusers_key_order(shapes)
        assert order == list(U.param_shapes(cfg).keys())
    cfg = U.tiny_config()
    shapes = {k: tuple(v) for k, v in U.param_shapes(cfg).items()}
    ref = U.synth_weights(cfg, seed=3)
    n = 0
    for name, t in synth.iter_synth(shapes, synth.diffusers_key_order(shapes), seed=3):
        assert torch.equal(t.float(), ref[name]), name
        n += 1
    assert n == len(ref)
