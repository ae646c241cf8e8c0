// Loss side of compute_loss() on device, fp32 arithmetic.
//
// DDPM (reference ddpm_trainer.py:303-384 + novelai_v3.py:111-137):
//   noisy  = clamp(x + sigma_b*noise, +-20000)                  (add_noise, ZTSNR clamp)
//   target = (noise - x)/sqrt(sigma_b^2)   | noise              (v_prediction | epsilon)
//   w_b    = min((1/sigma_b)^2, gamma)     | 1                  (MinSNR, per-sample: SURVEY D3)
//   loss   = mean(w_b * (pred-target)^2) [* mean(tag_w)]  -> guard
// Flow matching (flow_matching_trainer.py:298-335, :387-419):
//   xt = (1-t_b)*x0 + t_b*x1 ; target = x1 - x0 ; loss = mean_b(mean_chw((pred-target)^2)) [* mean(tag_w)] -> guard
// Guard: non-finite -> 1000 (no gradient) ; clamp(max=1000) (zero gradient above the cap).
//
// Inputs arrive as the reference hands them over: NCHW fp32 latents / noise.  The UNet consumes and
// produces token-major [B*HW][8] bf16 (4 real channels + 4 zero pad so every row is one 16-byte vector).
#include "kernels.h"

// a * b + c and friends with every operation rounded on its own, as separate torch ops do (hipcc contracts a * b + c into
// an FMA by default, and __fmul_rn / __fadd_rn are plain operators to it)
__device__ __forceinline__ float mul_add_rn(float a, float b, float c) {
#pragma clang fp contract(off)
  const float t = a * b;
  return t + c;
}
__device__ __forceinline__ float mul_mul_add_rn(float a, float b, float c, float d) {   // a * b + c * d
#pragma clang fp contract(off)
  const float t = a * b, u = c * d;
  return t + u;
}

__device__ __forceinline__ void loss_target(const LossP& p, int b, float x, float n, float sg, float* target,
                                            float* w) {
  if (p.method == 0) {
    *target = p.prediction_type == 1 ? (n - x) / sqrtf(sg * sg) : n;
    const float inv = 1.f / sg;
    float snr = inv * inv;
    *w = p.use_min_snr ? fminf(snr, p.min_snr_gamma) : 1.f;
  } else {
    *target = x - n;  // x1 - x0
    *w = 1.f;
  }
}

__global__ void loss_prepare_kernel(const LossP p) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
  if (i >= (long)p.B * p.HW) return;
  int b = (int)(i / p.HW), hw = (int)(i - (long)b * p.HW);
  float sg = p.sigma[b];
  bf16x8 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    long idx = ((long)b * 4 + c) * p.HW + hw;
    float x = p.latents[idx], n = p.noise[idx];
    float v;
    // every product and sum rounded on its own (no FMA contraction), exactly as the reference's separate torch ops do:
    // the bf16 UNet input is then the round-to-nearest-even image of the reference's fp32 tensor, bit for bit
    if (p.method == 0) {
      v = mul_add_rn(sg, n, x);
      if (p.use_ztsnr) v = fminf(fmaxf(v, -20000.f), 20000.f);
    } else {
      v = mul_mul_add_rn(1.f - sg, n, sg, x);
    }
    o[c] = (bf16)v;
    o[c + 4] = (bf16)0.f;
  }
  *(bf16x8*)(p.unet_in + i * 8) = o;
}

__global__ void loss_fwd_kernel(const LossP p) {
  __shared__ float sred[4][8];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < (long)p.B * p.HW) {
    int b = (int)(i / p.HW), hw = (int)(i - (long)b * p.HW);
    float sg = p.sigma[b];
    bf16x8 pv = *(const bf16x8*)(p.pred + i * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      long idx = ((long)b * 4 + c) * p.HW + hw;
      float x = p.latents[idx], n = p.noise[idx], pr = (float)pv[c];
      float tg, w;
      loss_target(p, b, x, n, sg, &tg, &w);
      float d = pr - tg;
      acc[0] += w * d * d;
      acc[1] += fabsf(pr);
      acc[2] += pr * pr;
      acc[3] += fabsf(n);
      acc[4] += n * n;
      acc[5] += x * x;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float s = wave_sum(acc[k]);
    if (lane == 0) sred[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6)      // the block's six sums as a partial row: loss_finalize_kernel adds the rows in a fixed order (bitwise reproducible loss)
    p.part[(long)blockIdx.x * 6 + threadIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
}

// one wave: lane l adds the partial rows l, l + 64, ... in order, lane 0 then adds the 64 lane sums in order
__global__ void loss_finalize_kernel(const LossP p) {
  __shared__ float sl[6][64];
  const long nb = ((long)p.B * p.HW + 255) / 256;
  for (int k = 0; k < 6; ++k) {
    float s = 0.f;
    for (long r = threadIdx.x; r < nb; r += 64) s += p.part[r * 6 + k];
    sl[k][threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 6; ++k) {
    float s = 0.f;
    for (int l = 0; l < 64; ++l) s += sl[k][l];
    p.out[1 + k] = s;
  }
  float numel = (float)p.B * 4.f * (float)p.HW;
  float l = p.out[1] / numel;
  float tm = 1.f;
  if (p.tag_w) {
    float s = 0.f;
    for (int b = 0; b < p.B; ++b) s += p.tag_w[b];
    tm = s / (float)p.B;
    l *= tm;
  }
  if (!isfinite(l)) { p.out[0] = 1000.f; p.out[7] = 0.f; }
  else if (l > 1000.f) { p.out[0] = 1000.f; p.out[7] = 0.f; }
  else { p.out[0] = l; p.out[7] = tm; }
}

__global__ void loss_bwd_kernel(const LossP p) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)p.B * p.HW) return;
  int b = (int)(i / p.HW), hw = (int)(i - (long)b * p.HW);
  float sg = p.sigma[b];
  float k = p.out[7] * p.grad_scale * 2.f / ((float)p.B * 4.f * (float)p.HW);
  bf16x8 pv = *(const bf16x8*)(p.pred + i * 8), o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    long idx = ((long)b * 4 + c) * p.HW + hw;
    float tg, w;
    loss_target(p, b, p.latents[idx], p.noise[idx], sg, &tg, &w);
    // guard taken (non-finite or clamped loss, out[7] == 0): an exact zero gradient, also where pred - target is inf / nan
    o[c] = k == 0.f ? (bf16)0.f : (bf16)(k * w * ((float)pv[c] - tg));
    o[c + 4] = (bf16)0.f;
  }
  *(bf16x8*)(p.dpred + i * 8) = o;
}

static int check_loss(const LossP& p) {
  ARG_CHECK(p.C == 4, "loss: latent channels must be 4 (got %d)", p.C);
  ARG_CHECK(p.B > 0 && p.HW > 0, "loss: empty batch");
  ARG_CHECK(p.latents && p.noise && p.sigma, "loss: missing inputs");
  return 0;
}
int launch_loss_prepare(const LossP& p, hipStream_t st) {
  if (int e = check_loss(p)) return e;
  long n = (long)p.B * p.HW;
  hipLaunchKernelGGL(loss_prepare_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_loss_fwd(const LossP& p, hipStream_t st) {
  if (int e = check_loss(p)) return e;
  long n = (long)p.B * p.HW;
  ARG_CHECK(p.part != nullptr, "loss: missing partial-row scratch (loss_part_floats)");
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_loss_bwd(const LossP& p, hipStream_t st) {
  if (int e = check_loss(p)) return e;
  long n = (long)p.B * p.HW;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
