// Shared device/host helpers for the gfx950 (MI355X, CDNA4) SDXL training-step library.
// Everything here is written for wave64 / gfx950 only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
// ---- LDS-DMA (global / buffer -> LDS, 16 bytes per lane, lane-linear destination) issued from inline asm ----
// hipcc treats the LDS-DMA builtins as stores to LDS that any later LDS read may alias: before the first
// ds_read_b64_tr_b16 (builtin) after a DMA it inserts s_waitcnt vmcnt(0), which drains every DMA in flight -- in a pipeline
// that keeps later tiles in flight on purpose that wait is the whole memory latency, every phase.  Issued from asm the DMA is
// invisible to the compiler; completion is then the author's job everywhere (counted s_waitcnt vmcnt + barrier before the
// data is read, vmcnt(0) before the workgroup ends).  M0 holds the wave-uniform LDS destination and is compiler-reserved:
// it is saved, written and restored inside the one statement that uses it.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_dma16_global(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void lds_dma4_global(const void* gsrc, unsigned lds_dst_uniform) {   // 4 bytes per lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
typedef __attribute__((ext_vector_type(4))) int i32x4;
// raw buffer descriptor: 48-bit base, stride 0, num_records bytes, DST_SEL / format word of a raw dword buffer
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned num_records) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)(unsigned)((a >> 32) & 0xFFFFu);
  r[2] = (int)num_records;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16_buffer(i32x4 srd_uniform, unsigned voffset, unsigned soffset_uniform, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voffset), "s"(srd_uniform), "s"(soffset_uniform), "s"(lds_dst_uniform) : "memory");
}

// whole-kernel wave priority (the immediate must be a literal; `prio` is a kernel argument: wave-uniform)
__device__ __forceinline__ void set_wave_prio(int prio) {
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);
}
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
