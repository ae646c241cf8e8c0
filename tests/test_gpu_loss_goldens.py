"""HIP loss-side kernels (csrc/loss.hip, through the C ABI `sdxl_op_loss`) on the vectors the REFERENCE itself produced
(tests/golden/loss_side.npz, written by oracle/make_goldens.py from the reference's own NoiseScheduler /
FlowMatchingTrainer / DDPMTrainer functions): every `sch*`, `fm*`, `dd*` case, the sigma = 20000 clamp, timesteps
{0, 998, 999}, the guards and the epsilon / no-MinSNR variants.

What is exact and what is not:
  * `loss_prepare` (add_noise + clamp, optimal_transport_path) is computed with individually rounded fp32 products and
    sums like the reference's torch ops, and the library hands the UNet a bf16 tensor: it must equal the
    round-to-nearest-even bf16 image of the reference's fp32 output BIT FOR BIT (no max-relative tolerance:
    at sigma = 2e4 that would be +-100 absolute).
  * target (get_velocity / x1 - x0) and weight (min(snr, gamma)) live only inside the loss sum; with pred = 0 the raw
    sum is mean(w * target^2), compared with the same expression built from the reference's `vel` / `snr` arrays to
    fp32 reduction-order accuracy (2e-5).
  * full losses: the UNet output crosses the boundary as bf16, so the stand-in prediction is rounded to bf16 first; the
    kernel must match the oracle on the same rounded prediction to 1e-5 and the reference's own scalar (fp32
    prediction) within north_star's 1e-3.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import sdxl_amd  # noqa: F401
from oracle import loss_ref as R
from sdxl_amd import lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available()
    return lib.load()


def T(a):
    return torch.from_numpy(np.asarray(a))


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_rows8(x_nchw):
    """NCHW fp32 -> token-major [B*HW][8] bf16 (channels 4..7 zero), the UNet's output layout."""
    B, Cc, H, W = x_nchw.shape
    o = torch.zeros(B * H * W, 8, dtype=torch.bfloat16, device=DEV)
    o[:, :4] = x_nchw.to(torch.bfloat16).permute(0, 2, 3, 1).reshape(B * H * W, 4).to(DEV)
    return o


def from_rows8(r, B, H, W):
    return r.view(B, H * W, 8)[..., :4].permute(0, 2, 1).reshape(B, 4, H, W).cpu()


class Case:
    """One call set of sdxl_op_loss on fixed (latents, noise / x0, sigma / t)."""

    def __init__(self, L, method, lat, noise, sig, pred_type=1, use_min_snr=1, gamma=5.0, ztsnr=1, tag=None):
        self.L, self.B, self.H, self.W = L, lat.shape[0], lat.shape[2], lat.shape[3]
        self.keep = [lat.float().contiguous().to(DEV), noise.float().contiguous().to(DEV), sig.float().contiguous().to(DEV),
                     None if tag is None else tag.float().contiguous().to(DEV)]
        self.lc = lib.LossConfig(method, pred_type, use_min_snr, gamma, ztsnr)
        self.b = lib.Batch(self.B, self.H, self.W, 77, self.keep[0].data_ptr(), self.keep[1].data_ptr(), self.keep[2].data_ptr(),
                           None, None, None, None, None if tag is None else self.keep[3].data_ptr())

    def prepare(self):
        xin = torch.empty(self.B * self.H * self.W, 8, dtype=torch.bfloat16, device=DEV)
        lib.check(self.L.sdxl_op_loss(C.byref(self.lc), C.byref(self.b), C.c_void_p(xin.data_ptr()), None, None, 1.0, None, 0, _st()))
        assert float(xin[:, 4:].float().abs().max()) == 0.0
        return from_rows8(xin, self.B, self.H, self.W)

    def loss(self, pred_nchw):
        p8 = to_rows8(pred_nchw)
        out = torch.zeros(8, dtype=torch.float32, device=DEV)
        lib.check(self.L.sdxl_op_loss(C.byref(self.lc), C.byref(self.b), None, C.c_void_p(p8.data_ptr()), None, 1.0,
                                      C.c_void_p(out.data_ptr()), 1, _st()))
        return [float(v) for v in out.cpu()]

    def dpred(self, pred_nchw, scale=1.0):
        p8 = to_rows8(pred_nchw)
        out = torch.zeros(8, dtype=torch.float32, device=DEV)
        dp = torch.empty_like(p8)
        lib.check(self.L.sdxl_op_loss(C.byref(self.lc), C.byref(self.b), None, C.c_void_p(p8.data_ptr()), None, 1.0,
                                      C.c_void_p(out.data_ptr()), 1, _st()))
        lib.check(self.L.sdxl_op_loss(C.byref(self.lc), C.byref(self.b), None, C.c_void_p(p8.data_ptr()), C.c_void_p(dp.data_ptr()),
                                      scale, C.c_void_p(out.data_ptr()), 2, _st()))
        return from_rows8(dp, self.B, self.H, self.W).float()


def assert_bf16_image(got_bf16, ref_f32, what):
    """got must be the RNE bf16 rounding of the reference's fp32 tensor, bit for bit."""
    want = ref_f32.to(torch.bfloat16)
    same = torch.equal(got_bf16.view(torch.int16), want.view(torch.int16))
    if not same:
        bad = (got_bf16.float() != want.float())
        i = int(bad.flatten().nonzero()[0])
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements differ from bf16(reference); first: got "
                             f"{float(got_bf16.flatten()[i])} want {float(want.flatten()[i])} (fp32 ref {float(ref_f32.flatten()[i])})")


def _standin_unet(x, t, ehs):
    B = x.shape[0]
    return 0.75 * x + 0.1 * ehs.reshape(B, -1).mean(1).view(B, 1, 1, 1) + 0.001 * t.reshape(-1).float().view(B, 1, 1, 1)


def test_scheduler_cases_prepare_target_weight(L, golden):
    n = int(golden["n_sched_cases"])
    assert n >= 12
    seen_t = set()
    for c in range(n):
        k = f"sch{c}"
        x, noise, ts = T(golden[f"{k}_x"]), T(golden[f"{k}_noise"]), T(golden[f"{k}_t"])
        sig, noisy, vel, snr = T(golden[f"{k}_sigma"]), T(golden[f"{k}_noisy"]), T(golden[f"{k}_vel"]), T(golden[f"{k}_snr"])
        B = sig.shape[0]
        if x.shape[0] != B:                     # case 1: the reference broadcast one sample against 4 timesteps
            x, noise = x.expand(B, -1, -1, -1).contiguous(), noise.expand(B, -1, -1, -1).contiguous()
        seen_t.update(int(v) for v in ts)
        cs = Case(L, 0, x, noise, sig)
        assert_bf16_image(cs.prepare().to(torch.bfloat16), noisy, f"{k} add_noise")
        zero = torch.zeros_like(x)
        # v-prediction + MinSNR(5): raw = mean(min(snr, 5)_b * vel^2)
        w = torch.minimum(snr, torch.full_like(snr, 5.0)).view(-1, 1, 1, 1)
        want = float((w.double() * vel.double() ** 2).mean())
        got = cs.loss(zero)[1] / x.numel()
        assert abs(got - want) <= 2e-5 * abs(want), (k, got, want)
        # epsilon target, no MinSNR: raw = mean(noise^2)
        ce = Case(L, 0, x, noise, sig, pred_type=0, use_min_snr=0)
        want = float((noise.double() ** 2).mean())
        got = ce.loss(zero)[1] / x.numel()
        assert abs(got - want) <= 2e-5 * abs(want), (k, got, want)
    assert {0, 998, 999} <= seen_t, seen_t          # sigma = 20000 (clamp) and the two smallest sigmas are in the set
    assert float(np.abs(golden["sch0_noisy"]).max()) == 20000.0


def test_flow_matching_cases(L, golden):
    for c in range(int(golden["n_fm_cases"])):
        k = f"fm{c}"
        x0, x1, t = T(golden[f"{k}_x0"]), T(golden[f"{k}_x1"]), T(golden[f"{k}_t"])
        vpred, per = T(golden[f"{k}_vpred"]), golden[f"{k}_loss_per_sample"]
        cs = Case(L, 1, x1, x0, t)
        assert_bf16_image(cs.prepare().to(torch.bfloat16), T(golden[f"{k}_xt"]), f"{k} optimal_transport_path")
        o = cs.loss(vpred)
        vb = vpred.to(torch.bfloat16).float()
        oracle = float(R.flow_matching_loss(vb, x0, x1))
        assert abs(o[0] - oracle) <= 1e-5 * abs(oracle), (k, o[0], oracle)
        assert abs(o[0] - float(per.mean())) <= 1e-3 * float(per.mean()), (k, o[0], float(per.mean()))
        # d loss / d pred = 2 (pred - (x1 - x0)) / numel
        gd = cs.dpred(vpred)
        ref = 2.0 * (vb - (x1 - x0)) / vb.numel()
        assert float((gd - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
        assert float(((gd - ref).abs() / ref.abs().clamp_min(1e-12)).median()) <= 2.0 ** -8


def test_flow_full_compute_loss_with_tag_weights(L, golden):
    lat, x0, t = T(golden["fmfull_vae_latents"]), T(golden["fmfull_x0"]), T(golden["fmfull_t"])
    tag, ehs = T(golden["fmfull_tag_weights"]), T(golden["fmfull_prompt_embeds"])
    cs = Case(L, 1, lat, x0, t, tag=tag)
    xt = R.optimal_transport_path(x0, lat, t)
    assert_bf16_image(cs.prepare().to(torch.bfloat16), xt, "fmfull xt")
    v = _standin_unet(xt, t, ehs)
    o = cs.loss(v)
    want = float(golden["fmfull_loss"])
    assert abs(o[0] - want) <= 1e-3 * want, (o[0], want)
    oracle = float(R.flow_matching_loss(v.to(torch.bfloat16).float(), x0, lat, tag))
    assert abs(o[0] - oracle) <= 1e-5 * oracle
    # metric sums: x0_norm, x1_norm from the fp32 inputs (exact data), velocity_norm from the bf16 prediction
    assert abs(o[5] ** 0.5 - float(golden["fmfull_m_x0_norm"])) <= 1e-5 * float(golden["fmfull_m_x0_norm"])
    assert abs(o[6] ** 0.5 - float(golden["fmfull_m_x1_norm"])) <= 1e-5 * float(golden["fmfull_m_x1_norm"])
    assert abs(o[3] ** 0.5 - float(golden["fmfull_m_velocity_norm"])) <= 2e-3 * float(golden["fmfull_m_velocity_norm"])


def test_ddpm_training_step_cases(L, golden):
    tab = R.karras_sigmas()
    for c in range(int(golden["n_dd_cases"])):
        k = f"dd{c}"
        lat, noise, ts, ehs = T(golden[f"{k}_vae_latents"]), T(golden[f"{k}_noise"]), T(golden[f"{k}_t"]), T(golden[f"{k}_prompt_embeds"])
        sig = tab[ts]
        cs = Case(L, 0, lat, noise, sig)
        noisy = R.add_noise(lat, noise, sig)
        assert_bf16_image(cs.prepare().to(torch.bfloat16), noisy, f"{k} add_noise")
        pred = _standin_unet(noisy, ts, ehs)
        o = cs.loss(pred)
        want = float(golden[f"{k}_loss"])
        oracle = float(R.ddpm_loss(pred.to(torch.bfloat16).float(), lat, noise, ts))
        print(f"[parity] {k} t={int(ts[0])}: hip {o[0]:.6e} oracle(bf16 pred) {oracle:.6e} reference(fp32 pred) {want:.6e}")
        assert abs(o[0] - oracle) <= 1e-5 * abs(oracle) + 1e-9, (k, o[0], oracle)
        # the golden was produced with an fp32 stand-in prediction; the device kernel (like the reference's bf16 UNet) consumes
        # a bf16 prediction.  At t = 0 (sigma = 2e4) the loss is min(snr, 5) * pred^2 with |pred| ~ 1e4: the rounding of the
        # prediction alone moves it by up to one bf16 epsilon (2^-8); the arithmetic itself is pinned by the line above.
        assert abs(o[0] - want) <= 4e-3 * abs(want), (k, o[0], want)
        assert abs(o[4] / lat.numel() - float(golden[f"{k}_m_noise_scale"])) <= 1e-5 * float(golden[f"{k}_m_noise_scale"])
        assert abs(o[2] / lat.numel() - float(golden[f"{k}_m_pred_scale"])) <= 2e-3 * float(golden[f"{k}_m_pred_scale"])


def test_ddpm_variants_and_guards(L, golden):
    tab = R.karras_sigmas()
    lat, ehs = T(golden["dd_last_vae_latents"]), T(golden["dd_last_prompt_embeds"])
    ts = torch.tensor([500])
    sig = tab[ts]
    # MinSNR off (plain mse_loss)
    noise = T(golden["dd_nogamma_noise"])
    pred = _standin_unet(R.add_noise(lat, noise, sig), ts, ehs)
    o = Case(L, 0, lat, noise, sig, use_min_snr=0).loss(pred)
    assert abs(o[0] - float(golden["dd_nogamma_loss"])) <= 1e-3 * float(golden["dd_nogamma_loss"])
    # epsilon target
    noise = T(golden["dd_eps_noise"])
    pred = _standin_unet(R.add_noise(lat, noise, sig), ts, ehs)
    o = Case(L, 0, lat, noise, sig, pred_type=0).loss(pred)
    assert abs(o[0] - float(golden["dd_eps_loss"])) <= 1e-3 * float(golden["dd_eps_loss"])
    # guards, as the reference ran them: an inf latent at t = 500 ; latents x 1e4 at t = 999 (noise = randn_like under seed 3)
    torch.manual_seed(3)
    noise = torch.randn_like(lat)
    binf = lat.clone()
    binf[0, 0, 0, 0] = float("inf")
    cs = Case(L, 0, binf, noise, sig)
    pred = _standin_unet(R.add_noise(binf, noise, sig), ts, ehs)
    pred[~torch.isfinite(pred)] = 3.0e38                       # (bf16 keeps the inf anyway; make the sum's inf come from the target)
    o = cs.loss(pred)
    assert o[0] == float(golden["dd_guard_inf_loss"]) == 1000.0 and o[7] == 0.0
    assert float(cs.dpred(pred).abs().max()) == 0.0             # no gradient through the guard
    ts9 = torch.tensor([999])
    big = lat * 1e4
    cs = Case(L, 0, big, noise, tab[ts9])
    pred = _standin_unet(R.add_noise(big, noise, tab[ts9]), ts9, ehs)
    o = cs.loss(pred)
    assert o[0] == float(golden["dd_guard_big_loss"]) == 1000.0 and o[7] == 0.0
    assert float(cs.dpred(pred).abs().max()) == 0.0
