// Fused AdamW_BF16 step for gfx950 (row f1): one pass over the parameter arena.
//
// Reference: src/training/optimizers/adamw_bfloat16/__init__.py:146-197 (`_make_step`) + stochastic/__init__.py:46-124:
// bf16 parameters, bf16 first / second moments, a bf16 error-feedback `shift` (true value = p + shift), three
// stochastically rounded accumulations + one more for the feedback, lazy weight decay.  The reference runs it as ~15
// separate torch kernels per parameter tensor (1 680 tensors); here it is one launch over the packed arena:
// per element 12 B read (p, m, v, shift bf16 + fp32 gradient) and 8 B written = 20 B -> 51.3 GB per step for the
// 2.567 B-parameter UNet -> HBM-bound, >= 6.4 ms at 8 TB/s.
//
// The arithmetic is the reference's float32 sequence with the rounding of each torch op reproduced exactly (see
// oracle/adamw_ref.py, pinned bit-for-bit to fixtures produced by the reference itself): fused multiply-add where
// torch's kernels fuse, bf16-rounded scalars where torch casts them, IEEE divide / sqrt.  `#pragma clang fp
// contract(off)` keeps the compiler from fusing anything else.  Stochastic rounding: r in [0, 2^16) is added to the
// fp32 bit pattern and the low half dropped; r comes from a counter-based generator (Philox-4x32-7 keyed by the
// seed, counter = (element-pair index, step)) or, for the parity tests, from a caller-supplied table.
#include "kernels.h"

#pragma clang fp contract(off)

__device__ __forceinline__ unsigned mulhi32(unsigned a, unsigned b) { return __umulhi(a, b); }
// Philox-4x32-7 (Salmon et al. 2011; 7 rounds is the variant that passes BigCrush with margin): 128 random bits per
// (counter, key) = the 8 sixteen-bit integers of two elements
__device__ __forceinline__ void philox4x32_7(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                             unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const unsigned h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const unsigned h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0;
    c1 = l1;
    c2 = h0 ^ c3 ^ k1;
    c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float bf(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 rn(float x) { return (bf16)x; }                       // round-to-nearest-even
__device__ __forceinline__ bf16 sr(float x, unsigned r16) {                           // stochastic/__init__.py:55-68
  const unsigned u = __float_as_uint(x) + r16;
  return __builtin_bit_cast(bf16, (unsigned short)(u >> 16));
}

template <bool INJECT>
__global__ __launch_bounds__(256) void adamw_bf16_kernel(const AdamWP q) {
  const size_t nvec = q.n / 8;
  const float gscale = q.grad_scale ? *q.grad_scale : 1.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = i * 8;
    bf16x8 pv = __builtin_nontemporal_load((const bf16x8*)(q.p + e0)), mv = __builtin_nontemporal_load((const bf16x8*)(q.m + e0)),
           vv = __builtin_nontemporal_load((const bf16x8*)(q.v + e0)), sv = __builtin_nontemporal_load((const bf16x8*)(q.shift + e0));
    float g[8];
    if (q.grad_f32) {
      const f32x4 a = __builtin_nontemporal_load((const f32x4*)(q.grad_f32 + e0)), b = __builtin_nontemporal_load((const f32x4*)(q.grad_f32 + e0 + 4));
      g[0] = a[0]; g[1] = a[1]; g[2] = a[2]; g[3] = a[3]; g[4] = b[0]; g[5] = b[1]; g[6] = b[2]; g[7] = b[3];
    } else {
      const bf16x8 gv = *(const bf16x8*)(q.grad_bf16 + e0);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = bf(gv[e]);
    }
    unsigned rnd[4][4];     // [element pair][word]: counter = (pair index, step), key = seed
    if (!INJECT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t pair = q.elem_offset / 2 + i * 4 + k;
        philox4x32_7((unsigned)pair, (unsigned)(pair >> 32), q.step_counter, 0x5D71A3B1u, q.seed_lo, q.seed_hi, rnd[k]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned r0, r1, r2, r3;
      if (INJECT) {
        r0 = q.rand[e0 + e]; r1 = q.rand[q.n + e0 + e]; r2 = q.rand[2 * q.n + e0 + e]; r3 = q.rand[3 * q.n + e0 + e];
      } else {
        const unsigned a = rnd[e >> 1][(e & 1) * 2], b = rnd[e >> 1][(e & 1) * 2 + 1];
        r0 = a & 0xFFFFu; r1 = a >> 16; r2 = b & 0xFFFFu; r3 = b >> 16;
      }
      float gr = g[e] * gscale;                                  // fused unscale / clip coefficient
      if (q.grad_round_bf16) gr = bf(rn(gr));                    // the reference's gradients are bf16 tensors
      const float pf = bf(pv[e]), sf = bf(sv[e]);
      // exp_avg.mul_(beta1); add_stochastic_(exp_avg, grad, alpha=1-beta1)       (__init__.py:162-163)
      const float m1 = bf(rn(bf(mv[e]) * q.beta1));
      const float rm = q.reference_ema ? __builtin_fmaf(m1, q.one_minus_beta1, gr)      // grad + alpha * exp_avg (D17)
                                       : __builtin_fmaf(gr, q.one_minus_beta1, m1);     // exp_avg + alpha * grad
      const bf16 m2b = sr(rm, r0);
      const float m2 = bf(m2b);
      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)                 (:164)
      const float v1 = bf(rn(bf(vv[e]) * q.beta2));
      const bf16 v2b = rn(__builtin_fmaf(q.one_minus_beta2 * gr, gr, v1));
      // denom = exp_avg_sq.sqrt().add_(eps)                                         (:176-181)
      float den = bf(rn(__builtin_sqrtf(bf(v2b))));
      den = bf(rn(den + q.eps_bf16));
      // addcdiv_stochastic_(shift, exp_avg, denom, value=-lr*sqrt(1-beta2^t))       (stochastic:106-124)
      const bf16 s1b = sr(sf + (q.value * m2) / den, r1);
      const float s1 = bf(s1b);
      // buffer = p.clone(); add_stochastic_(p, shift); add_stochastic_(shift, buffer - p)   (:183-190)
      const bf16 p1b = sr(s1 + pf, r2);
      const float diff = bf(rn(pf - bf(p1b)));
      bf16 s2b = sr(diff + s1, r3);
      // lazy decay, when this tensor's accumulated decay is due: shift.add_(p, alpha=-decay)  (:192-193)
      if (q.decay_alpha_bf16 != 0.f) s2b = rn(__builtin_fmaf(bf(p1b), q.decay_alpha_bf16, bf(s2b)));
      pv[e] = p1b; mv[e] = m2b; vv[e] = v2b; sv[e] = s2b;
    }
    __builtin_nontemporal_store(pv, (bf16x8*)(q.p + e0));      // streamed once per update: keep them out of the caches
    __builtin_nontemporal_store(mv, (bf16x8*)(q.m + e0));
    __builtin_nontemporal_store(vv, (bf16x8*)(q.v + e0));
    __builtin_nontemporal_store(sv, (bf16x8*)(q.shift + e0));
  }
}

// shift.add_(p, alpha) over one parameter tensor's range (alpha already rounded to bf16, as torch does)
__global__ void adamw_decay_kernel(bf16* shift, const bf16* p, size_t n, float alpha) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    shift[i] = rn(__builtin_fmaf(bf(p[i]), alpha, bf(shift[i])));
}

int launch_adamw_bf16(const AdamWP& q, hipStream_t st) {
  ARG_CHECK(q.p && q.m && q.v && q.shift && (q.grad_f32 || q.grad_bf16), "adamw: missing buffers");
  ARG_CHECK(q.n % 8 == 0 && q.elem_offset % 8 == 0, "adamw: n=%zu, elem_offset=%zu: each must be a multiple of 8", q.n, q.elem_offset);
  ARG_CHECK((((uintptr_t)q.p | (uintptr_t)q.m | (uintptr_t)q.v | (uintptr_t)q.shift | (uintptr_t)q.grad_f32 | (uintptr_t)q.grad_bf16) & 15) == 0,
            "adamw: buffers must be 16-byte aligned");
  if (q.n == 0) return 0;
  size_t nvec = q.n / 8;
  size_t blocks = (nvec + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;      // 16 workgroups per CU, grid-stride the rest
  if (q.rand) hipLaunchKernelGGL(adamw_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, q);
  else hipLaunchKernelGGL(adamw_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_adamw_decay(bf16* shift, const bf16* p, size_t n, float alpha_bf16, hipStream_t st) {
  if (n == 0) return 0;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_decay_kernel, dim3((unsigned)blocks), dim3(256), 0, st, shift, p, n, alpha_bf16);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
