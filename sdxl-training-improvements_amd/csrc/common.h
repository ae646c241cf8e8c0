// Shared device/host helpers for the gfx950 (MI355X, CDNA4) SDXL training-step library.
// Everything here is written for wave64 / gfx950 only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// last error text, readable through sdxl_last_error()
void sdxl_set_error(const char* fmt, ...);

#define HIP_CHECK_RET(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      sdxl_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

#define ARG_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      sdxl_set_error(__VA_ARGS__);      \
      return 1;                         \
    }                                   \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// exact (erf) GELU pieces in ~14 instructions: Phi(x) and phi(x) share one exponential.
// erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below bf16 resolution): with z = |x| / sqrt(2),
//   erf(z) = 1 - (a1 k + a2 k^2 + a3 k^3 + a4 k^4 + a5 k^5) exp(-z^2),  k = 1 / (1 + 0.3275911 z)
__device__ __forceinline__ void gelu_cdf_pdf(float x, float* cdf, float* pdf) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);      // exp(-x^2 / 2)
  const float k = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.23164189f, 1.f));    // 0.3275911 / sqrt(2)
  float q = fmaf(k, 1.061405429f, -1.453152027f);
  q = fmaf(k, q, 1.421413741f);
  q = fmaf(k, q, -0.284496736f);
  q = fmaf(k, q, 0.254829592f);
  const float h = 0.5f - 0.5f * (q * k * e);                                    // erf(|x|/sqrt2) / 2
  *cdf = 0.5f + copysignf(h, x);
  *pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float c, d;
  gelu_cdf_pdf(x, &c, &d);
  return x * c;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float c, d;
  gelu_cdf_pdf(x, &c, &d);
  return c + x * d;
}
#endif
