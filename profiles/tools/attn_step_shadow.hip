// What the instruction mix of ONE phase of the pipelined attention forward (attention_pl.hip: per step MFMA, 2 x v_exp, MFMA, 2 x v_add,
// v_cvt_pk -- 8 steps per unit) costs on one SIMD with NO memory, LDS or barrier in the loop: one wave per SIMD (256 threads, 1 workgroup
// per CU, 512 registers) or two (launch bounds 256 x 2), all CUs busy.  Variants knock out one instruction class each.
//   hipcc --offload-arch=gfx950 -O3 -o attn_step_shadow attn_step_shadow.hip && ./attn_step_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V, bool OA>
__device__ __forceinline__ void step(f32x4& s, const bf16x8& ka, const bf16x8& qb, float x0, float x1, f32x4& o, const bf16x8& va, const bf16x8& pb,
                                     float& e0, float& e1, float pe0, float pe1, float& la, float& lb, unsigned& pd, bool ks0) {
  // V bits: 1 = S MFMA, 2 = exps, 4 = O MFMA, 8 = adds, 16 = cvt, 32 = consumers read THIS step's exps (no delay)
#define M1 "v_mfma_f32_16x16x32_bf16 %0, %7, %8, %0\n\t"
#define EX "v_exp_f32 %1, %9\n\tv_exp_f32 %2, %10\n\t"
#define M2 "v_mfma_f32_16x16x32_bf16 %3, %11, %12, %3\n\t"
#define AD "v_add_f32 %4, %4, %13\n\tv_add_f32 %5, %5, %14\n\t"
#define CV "v_cvt_pk_bf16_f32 %6, %13, %14\n\t"
#define AD0 "v_add_f32 %4, %4, %1\n\tv_add_f32 %5, %5, %2\n\t"
#define CV0 "v_cvt_pk_bf16_f32 %6, %1, %2\n\t"
#define OUTS(OC) : "+v"(s), "+v"(e0), "+v"(e1), OC(o), "+v"(la), "+v"(lb), "+v"(pd) : "v"(ka), "v"(qb), "v"(x0), "v"(x1), "v"(va), "v"(pb), "v"(pe0), "v"(pe1)
#define OCA(x) "+a"(x)
#define OCV(x) "+v"(x)
  if (OA) asm volatile(
      "" /* */
      : : );
  if (V == 63 - 32) { if (OA) asm volatile(M1 EX M2 AD CV OUTS(OCA)); else asm volatile(M1 EX M2 AD CV OUTS(OCV)); }
  if (V == 63)      { if (OA) asm volatile(M1 EX M2 AD0 CV0 OUTS(OCA)); else asm volatile(M1 EX M2 AD0 CV0 OUTS(OCV)); }
  if (V == 5)       { if (OA) asm volatile(M1 M2 OUTS(OCA)); else asm volatile(M1 M2 OUTS(OCV)); }
  if (V == 26)      { if (OA) asm volatile(EX AD CV OUTS(OCA)); else asm volatile(EX AD CV OUTS(OCV)); }
  if (V == 7)       { if (OA) asm volatile(M1 EX M2 OUTS(OCA)); else asm volatile(M1 EX M2 OUTS(OCV)); }
  if (V == 29)      { if (OA) asm volatile(M1 M2 AD CV OUTS(OCA)); else asm volatile(M1 M2 AD CV OUTS(OCV)); }
  if (V == 2)       { if (OA) asm volatile(EX OUTS(OCA)); else asm volatile(EX OUTS(OCV)); }
  if (V == 24)      { if (OA) asm volatile(AD CV OUTS(OCA)); else asm volatile(AD CV OUTS(OCV)); }
}

template <int V, bool OA, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x;
  bf16x8 ka[4], qb[2], va[4], pb[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ka[i][e] = (__bf16)(0.01f * (lane + i + e)); va[i][e] = (__bf16)(0.02f * (lane - i + e)); }
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { qb[i][e] = (__bf16)(0.03f * (lane + i)); pb[i][e] = (__bf16)(0.5f); }
  f32x4 S[4], O[4];
  for (int i = 0; i < 4; ++i) { S[i] = (f32x4){0.1f * lane, 0.2f, 0.3f, 0.4f}; O[i] = (f32x4){0, 0, 0, 0}; }
  float x[16]; for (int i = 0; i < 16; ++i) x[i] = -0.001f * (lane + i);
  float la = 0, lb = 0, lc = 0, ld = 0, e0 = 0, e1 = 0, pe0 = 0.5f, pe1 = 0.25f;
  unsigned pd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      step<V, OA>(S[kk & 3], ka[kk & 3], qb[kk >> 2], x[2 * kk], x[2 * kk + 1], O[kk & 3], va[kk & 3], pb[kk >> 2], e0, e1, pe0, pe1,
                  (kk & 1) ? lc : la, (kk & 1) ? ld : lb, pd[kk], false);
      pe0 = e0; pe1 = e1;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
  float acc = la + lb + lc + ld + e0 + e1;
  for (int i = 0; i < 4; ++i) acc += S[i][0] + O[i][1];
  for (int i = 0; i < 8; ++i) acc += (float)pd[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V, bool OA, int WPS>
void run(const char* name, int grid) {
  float* out; unsigned long long* cyc; hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<V, OA, WPS><<<grid, 256>>>(out, 2000, cyc);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<V, OA, WPS><<<grid, 256>>>(out, iters, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // per unit = one pass of 8 steps of ONE wave; with 2 waves per SIMD a SIMD does two units in that time
  printf("%-44s %s waves/SIMD %d: %7.1f ns per unit-pass  (%6.1f memtime ticks @100MHz = %5.0f ns)  -> %5.0f cycles at 2.1 GHz\n", name, OA ? "O in AGPR" : "O in VGPR", WPS,
         ms * 1e6 / iters, (double)c / iters, (double)c / iters * 10.0, ms * 1e6 / iters * 2.1);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int wps = 1; wps <= 2; ++wps) {
    const int grid = 256 * wps;
#define RUN(V, NAME) if (wps == 1) { run<V, true, 1>(NAME, grid); run<V, false, 1>(NAME, grid); } else { run<V, true, 2>(NAME, grid); run<V, false, 2>(NAME, grid); }
    RUN(31, "full step, consumers one step behind");
    RUN(63, "full step, consumers in the same step");
    RUN(5, "MFMAs only (16)");
    RUN(26, "VALU only (16 exp, 16 add, 8 cvt)");
    RUN(7, "MFMAs + exps");
    RUN(29, "MFMAs + adds + cvts");
    RUN(2, "exps only (16)");
    RUN(24, "adds + cvts only");
  }
  return 0;
}
