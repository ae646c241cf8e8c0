// Instruction mix of one UNIT of the attention backward (d = 64), no memory / LDS / barrier, every CU busy:
//   dQ body  (16 queries x 64 keys): 24 MFMAs (S, dP, dQ) + 16 v_exp + 16 v_mul + 8 v_cvt_pk
//   dKV body (64 queries x 16 keys): 32 MFMAs (S, dP, dV, dK) + 16 v_exp + 16 v_mul + 16 v_cvt_pk
// 8 steps per unit; per step: [MFMA, exp, exp, MFMA, mul, mul, (MFMA,) MFMA / cvt ...].  One or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o attn_bwd_shadow attn_bwd_shadow.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MF(D, A, B) "v_mfma_f32_16x16x32_bf16 " D ", " A ", " B ", " D "\n\t"
// operands: %0 s (S acc), %1 dp (dP acc), %2 o (dQ / dV acc, AGPR), %3 o2 (dK acc, AGPR), %4 e0, %5 e1, %6 pd, %7 pd2 | %8 ka, %9 qb, %10 x0, %11 x1, %12 d0, %13 d1
template <int V>
__device__ __forceinline__ void step(f32x4& s, f32x4& dp, f32x4& o, f32x4& o2, float& e0, float& e1, unsigned& pd, unsigned& pd2,
                                     const bf16x8& ka, const bf16x8& qb, float x0, float x1, float d0, float d1) {
#define OUTS : "+v"(s), "+v"(dp), "+a"(o), "+a"(o2), "+v"(e0), "+v"(e1), "+v"(pd), "+v"(pd2) : "v"(ka), "v"(qb), "v"(x0), "v"(x1), "v"(d0), "v"(d1)
  if (V == 0)   // dQ unit, full: 3 MFMA + 2 exp + 2 mul + 1 cvt per step
    asm volatile("v_exp_f32 %4, %10\n\tv_exp_f32 %5, %11\n\t" MF("%0", "%8", "%9") "v_mul_f32 %4, %4, %12\n\tv_mul_f32 %5, %5, %13\n\t" MF("%1", "%8", "%9")
                 "v_cvt_pk_bf16_f32 %6, %4, %5\n\t" MF("%2", "%8", "%9") OUTS);
  if (V == 1)   // dQ unit, MFMAs only
    asm volatile(MF("%0", "%8", "%9") MF("%1", "%8", "%9") MF("%2", "%8", "%9") OUTS);
  if (V == 2)   // dQ unit, VALU only
    asm volatile("v_exp_f32 %4, %10\n\tv_exp_f32 %5, %11\n\tv_mul_f32 %4, %4, %12\n\tv_mul_f32 %5, %5, %13\n\tv_cvt_pk_bf16_f32 %6, %4, %5\n\t" OUTS);
  if (V == 3)   // dKV unit, full: 4 MFMA + 2 exp + 2 mul + 2 cvt per step
    asm volatile("v_exp_f32 %4, %10\n\tv_exp_f32 %5, %11\n\t" MF("%0", "%8", "%9") "v_cvt_pk_bf16_f32 %6, %4, %5\n\t" MF("%1", "%8", "%9")
                 "v_mul_f32 %4, %4, %12\n\tv_mul_f32 %5, %5, %13\n\t" MF("%2", "%8", "%9") "v_cvt_pk_bf16_f32 %7, %4, %5\n\t" MF("%3", "%8", "%9") OUTS);
  if (V == 4)   // dKV unit, MFMAs only
    asm volatile(MF("%0", "%8", "%9") MF("%1", "%8", "%9") MF("%2", "%8", "%9") MF("%3", "%8", "%9") OUTS);
  if (V == 5)   // dQ unit, consumers one step behind (exp of this step; mul / cvt of the previous step's pair in d0/d1 -> here x0/x1 stand in)
    asm volatile("v_exp_f32 %4, %10\n\tv_exp_f32 %5, %11\n\t" MF("%0", "%8", "%9") "v_mul_f32 %6, %12, %10\n\tv_mul_f32 %7, %13, %11\n\t" MF("%1", "%8", "%9")
                 "v_cvt_pk_bf16_f32 %6, %6, %7\n\t" MF("%2", "%8", "%9") OUTS);
}
template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 ka[4], qb[2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) ka[i][e] = (__bf16)(0.01f * (lane + i + e));
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) qb[i][e] = (__bf16)(0.03f * (lane + i));
  f32x4 S[4], DP[4], O[4], O2[4];
  for (int i = 0; i < 4; ++i) { S[i] = (f32x4){0.1f * lane, 0.2f, 0.3f, 0.4f}; DP[i] = S[i]; O[i] = (f32x4){0, 0, 0, 0}; O2[i] = O[i]; }
  float x[16]; for (int i = 0; i < 16; ++i) x[i] = -0.001f * (lane + i);
  float e0 = 0, e1 = 0; unsigned pd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pd2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) step<V>(S[kk & 3], DP[kk & 3], O[kk & 3], O2[kk & 3], e0, e1, pd[kk], pd2[kk], ka[kk & 3], qb[kk >> 2], x[2 * kk], x[2 * kk + 1], x[kk], x[15 - kk]);
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
  float acc = e0 + e1;
  for (int i = 0; i < 4; ++i) acc += S[i][0] + DP[i][1] + O[i][1] + O2[i][2];
  for (int i = 0; i < 8; ++i) acc += (float)pd[i] + (float)pd2[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int V, int WPS>
void run(const char* name, int nmfma) {
  const int grid = 256 * WPS;
  float* out; (void)hipMalloc(&out, grid * 256 * 4);
  const int iters = 20000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<V, WPS><<<grid, 256>>>(out, 2000);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<V, WPS><<<grid, 256>>>(out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double ns = ms * 1e6 / iters, per_unit = ns / WPS;
  printf("%-46s waves/SIMD %d: %7.1f ns per pass = %6.1f ns per unit per SIMD; %2d MFMAs/unit -> MFMA pipe busy %4.1f %% at 2.1 GHz\n", name, WPS, ns, per_unit, nmfma,
         100.0 * nmfma * 16 / 2.1 / per_unit);
  (void)hipFree(out);
}
int main() {
  run<0, 1>("dQ unit (24 MFMA + 16 exp + 16 mul + 8 cvt)", 24);  run<0, 2>("dQ unit (24 MFMA + 16 exp + 16 mul + 8 cvt)", 24);
  run<5, 1>("dQ unit, consumers one step behind", 24);           run<5, 2>("dQ unit, consumers one step behind", 24);
  run<1, 1>("dQ unit, MFMAs only", 24);                           run<1, 2>("dQ unit, MFMAs only", 24);
  run<2, 1>("dQ unit, VALU only", 0);                             run<2, 2>("dQ unit, VALU only", 0);
  run<3, 1>("dKV unit (32 MFMA + 16 exp + 16 mul + 16 cvt)", 32); run<3, 2>("dKV unit (32 MFMA + 16 exp + 16 mul + 16 cvt)", 32);
  run<4, 1>("dKV unit, MFMAs only", 32);                          run<4, 2>("dKV unit, MFMAs only", 32);
  return 0;
}
