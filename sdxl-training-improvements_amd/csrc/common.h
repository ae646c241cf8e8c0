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
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
#endif
