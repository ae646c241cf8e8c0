"""ctypes binding of libsdxlstep.so (include/sdxlstep.h).  No fallback: if the HIP library is missing or a
call fails this raises -- the product path never silently degrades to PyTorch ops or to the CPU oracle.

SDXL_DIAG=1 in the environment (or use_diag() before the first load()) selects libsdxlstep_diag.so, the -DSDXL_DIAG build with the
experiment ABI of include/sdxlstep_diag.h part 2 (knobs, stream-K, ...): A/B tooling and the tests marked `diag` only."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

HERE = Path(__file__).resolve().parent
DIAG = os.environ.get("SDXL_DIAG", "") == "1"
LIB_PATH = HERE / ("libsdxlstep_diag.so" if DIAG else "libsdxlstep.so")


def use_diag() -> None:
    """select the diagnostics build (before the first load())"""
    global DIAG, LIB_PATH
    if _lib is not None and not DIAG:
        raise SdxlError("use_diag() after the product library has been loaded")
    DIAG, LIB_PATH = True, HERE / "libsdxlstep_diag.so"



class SdxlError(RuntimeError):
    pass


class UNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 3),
                ("layers_per_block", C.c_int), ("transformer_layers", C.c_int * 3), ("head_dim", C.c_int),
                ("cross_attention_dim", C.c_int), ("norm_num_groups", C.c_int),
                ("addition_time_embed_dim", C.c_int), ("pooled_dim", C.c_int),
                ("resnet_eps", C.c_float), ("tf_gn_eps", C.c_float), ("ln_eps", C.c_float)]


class LossConfig(C.Structure):
    _fields_ = [("method", C.c_int), ("prediction_type", C.c_int), ("use_min_snr", C.c_int),
                ("min_snr_gamma", C.c_float), ("use_ztsnr", C.c_int)]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("ctx_len", C.c_int),
                ("latents", C.c_void_p), ("noise", C.c_void_p), ("sigma_or_t", C.c_void_p),
                ("timestep", C.c_void_p), ("prompt_embeds", C.c_void_p), ("pooled", C.c_void_p),
                ("time_ids", C.c_void_p), ("tag_weights", C.c_void_p)]


class AdamWConfig(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("step", C.c_double), ("decay_this_iteration", C.c_double), ("reference_ema", C.c_int),
                ("grad_round_bf16", C.c_int), ("seed", C.c_ulonglong), ("elem_offset", C.c_ulonglong)]


_vp, _i, _f, _l, _sz = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t
_P = C.POINTER

# name -> argtypes   (every function returns int except sdxl_last_error)
SIGNATURES = {
    "sdxl_version": [],
    "sdxl_default_config": [_P(UNetConfig)],
    "sdxl_create": [_P(UNetConfig), _i, _P(_vp)],
    "sdxl_destroy": [_vp],
    "sdxl_param_bytes": [_vp, _P(_sz), _P(_sz)],
    "sdxl_bind_params": [_vp, _vp, _vp],
    "sdxl_num_params": [_vp],
    "sdxl_param_info": [_vp, _i, C.c_char_p, _i, _P(_i), _P(_l)],
    "sdxl_load_weight": [_vp, C.c_char_p, _vp, _i, _vp],
    "sdxl_export_weight": [_vp, C.c_char_p, _vp, _i, _vp],
    "sdxl_export_grad": [_vp, C.c_char_p, _vp, _i, _vp],
    "sdxl_plan": [_vp, _i, _i, _i, _i, _P(_sz)],
    "sdxl_bind_workspace": [_vp, _vp, _sz],
    "sdxl_zero_grads": [_vp, _vp],
    "sdxl_forward_loss": [_vp, _P(LossConfig), _P(Batch), _vp],
    "sdxl_num_segments": [_vp],
    "sdxl_segment_range": [_vp, _i, _P(_sz), _P(_sz)],
    "sdxl_backward_segment": [_vp, _i, _f, _i, _vp],
    "sdxl_loss_fwd_bwd": [_vp, _P(LossConfig), _P(Batch), _f, _i, _vp],
    "sdxl_backward_all": [_vp, _f, _i, _vp],
    "sdxl_set_graph_mode": [_vp, _i],
    "sdxl_read_loss": [_vp, _P(_f), _vp],
    "sdxl_unet_forward": [_vp, _vp, _P(Batch), _vp, _vp],
    "sdxl_unet_backward": [_vp, _vp, _i, _vp],
    "sdxl_grads_to_bf16": [_vp, _sz, _sz, _vp, _f, _vp],
    "sdxl_set_grad_emit": [_vp, _vp, _f],
    "sdxl_small_grads_to_bf16": [_vp, _sz, _sz, _vp, _f, _vp],
    "sdxl_grad_sumsq": [_vp, _vp, _vp],
    "sdxl_op_gemm": [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "sdxl_op_wgrad_group": [_i, _P(_vp), _P(_vp), _P(_vp), _P(_vp), _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_upconv3x3_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_upconv3x3_dgrad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_s2_dgrad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_upconv3x3_wgrad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_dgrad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_wgrad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_wgrad2": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_attention_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _vp],
    "sdxl_op_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _vp],
    "sdxl_op_groupnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "sdxl_op_groupnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "sdxl_op_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sdxl_set_join_mode": [_vp, _i],
    "sdxl_side_stream": [_vp, _P(_vp)],
    "sdxl_sumsq": [_vp, _i, _sz, _vp, _vp],
    "sdxl_clip_coef": [_vp, _f, _vp, _vp],
    "sdxl_param_range": [_vp, _i, _P(_sz), _P(_sz)],
    "sdxl_adamw_default_config": [C.POINTER(AdamWConfig)],
    "sdxl_adamw_bf16_step": [_vp, _vp, _i, _vp, _vp, _vp, _sz, C.POINTER(AdamWConfig), _vp, _vp, _vp],
    "sdxl_adamw_decay": [_vp, _vp, _sz, _f, _vp],
    "sdxl_op_ff_geglu_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sdxl_op_ff_geglu_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sdxl_op_loss": [_P(LossConfig), _P(Batch), _vp, _vp, _vp, _f, _vp, _i, _vp],
    "sdxl_profile_gemm_begin": [],
    "sdxl_profile_gemm_end": [_P(C.c_double), _P(C.c_double), _P(_i)],
}
# include/sdxlstep_diag.h part 1: test hooks, exported by the product library too
TEST_HOOK_SIGNATURES = {
    "sdxl_probe_layout": [_vp, _vp],
    "sdxl_set_gemm_mode": [_i],
    "sdxl_debug_act_checksums": [_vp, _P(C.c_ulonglong), _i, _P(_i), _i],
    "sdxl_op_exchange_shadow": [_vp, _sz, _i, _i, _f, _vp],
    "sdxl_op_gemm_ld": [_i, _vp, _vp, _vp, _i, _i, _i, C.c_long, C.c_long, C.c_long, _vp, _vp, C.c_long, _i, _vp],
    "sdxl_op_linear_dgrad_delta": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
}
# include/sdxlstep_diag.h part 2: experiment ABI, exported by libsdxlstep_diag.so only
DIAG_SIGNATURES = {
    "sdxl_op_linear_dgrad_ln_bwd": [_vp, _vp, _vp, _P(_f), _vp, _vp, _vp, _vp, _P(_f), _i, _i, _i, _vp],
    "sdxl_set_knob": [_i, _i],
    "sdxl_set_sk_mode": [_i, _i],
    "sdxl_sk_error": [_vp, _P(C.c_uint)],
    "sdxl_ln_error": [_P(C.c_uint)],
    "sdxl_op_pl_prefetch_b": [_i, _vp, _i, _i, _i, C.c_long, _i, _vp],
    "sdxl_op_gemm_sk": [_i, _P(_i), _P(_vp), _P(_vp), _P(_vp), _P(_i), _P(_i), _P(_i), _P(_vp), _P(_vp), _P(_i), _vp],
    "sdxl_op_conv3x3_s2_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sdxl_op_conv3x3_s2_wgrad": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
}

_lib = None


def load() -> C.CDLL:
    """Load libsdxlstep.so and declare every prototype.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SdxlError(f"{LIB_PATH} is missing -- build it with `python {HERE / 'build.py'}{' --diag' if DIAG else ''}` "
                        "(there is no PyTorch/CPU fallback for the training step)")
    lib = C.CDLL(str(LIB_PATH))
    lib.sdxl_last_error.restype = C.c_char_p
    lib.sdxl_last_error.argtypes = []
    sigs = dict(SIGNATURES)
    sigs.update(TEST_HOOK_SIGNATURES)
    if DIAG:
        sigs.update(DIAG_SIGNATURES)
    for name, args in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    if DIAG:      # diagnostics build: SDXL_KNOBS=id=value,id=value preset the experiment knobs for the whole process (A/B runs of tools AND tests)
        for kv in os.environ.get("SDXL_KNOBS", "").split(","):
            if kv:
                rc = lib.sdxl_set_knob(int(kv.split("=")[0]), int(kv.split("=")[1]))
                if rc != 0:
                    raise SdxlError(f"SDXL_KNOBS: sdxl_set_knob({kv}) failed")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().sdxl_last_error()
        raise SdxlError(f"{what or 'libsdxlstep'} failed (rc={rc}): {msg.decode() if msg else '?'}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
