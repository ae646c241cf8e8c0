// Does the matrix pipe overlap with VALU work (a) across the waves of one SIMD when each wave runs its MFMAs and its VALU
// in separate blocks, (b) inside one wave when they are interleaved?  Per iteration: 16 independent v_mfma_f32_16x16x32_bf16
// + NV independent VALU (v_fma_f32 / v_exp_f32 mix).  hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF32(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[i]) : "v"(fa), "v"(fb));
#define VA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
#define VE(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
template <int MODE>   // 0: MFMA block then VALU block; 1: interleaved 1 MFMA : 3 VALU; 2: MFMA only; 3: VALU only;
                      // 4 / 5 / 6: the same FLOPs as 8 v_mfma_f32_32x32x16_bf16: block + block / interleaved 1 : 6 / MFMA only
__global__ void k(float* out, int iters) {
  float a[48];
  for (int i = 0; i < 48; ++i) a[i] = threadIdx.x * 1e-3f + i;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(threadIdx.x + i); fb[i] = (__bf16)(threadIdx.x - i); }
  f32x16 acc32[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4 || MODE == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MF32(i & 3)
    }
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 48; ++i) { if (i % 3 == 0) VE(i) else VA(i) }
    }
    if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { MF32(i & 3) VE(6 * i) VA(6 * i + 1) VA(6 * i + 2) VE(6 * i + 3) VA(6 * i + 4) VA(6 * i + 5) }
    }
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) MF(i)
    }
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < 48; ++i) { if (i % 3 == 0) VE(i) else VA(i) }
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { MF(i) VE(3 * i) VA(3 * i + 1) VA(3 * i + 2) }
    }
  }
  float s = 0;
  for (int i = 0; i < 48; ++i) s += a[i];
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  for (int i = 0; i < 4; ++i) s += acc32[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  const int iters = 20000;
  const char* names[] = {"MFMA block, then VALU block", "interleaved 1 MFMA : 1 exp + 2 fma", "MFMA only", "VALU only",
                         "32x32x16: MFMA block, then VALU block", "32x32x16: interleaved 1 : 2 exp + 4 fma", "32x32x16: MFMA only"};
  for (int mode = 0; mode < 7; ++mode)
    for (int wps = 1; wps <= 3; ++wps) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        dim3 grid(256), blk(256 * wps);
        switch (mode) { case 0: hipLaunchKernelGGL(k<0>, grid, blk, 0, 0, out, iters); break; case 1: hipLaunchKernelGGL(k<1>, grid, blk, 0, 0, out, iters); break;
                        case 2: hipLaunchKernelGGL(k<2>, grid, blk, 0, 0, out, iters); break; case 3: hipLaunchKernelGGL(k<3>, grid, blk, 0, 0, out, iters); break;
                        case 4: hipLaunchKernelGGL(k<4>, grid, blk, 0, 0, out, iters); break; case 5: hipLaunchKernelGGL(k<5>, grid, blk, 0, 0, out, iters); break;
                        case 6: hipLaunchKernelGGL(k<6>, grid, blk, 0, 0, out, iters); break; }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("%-40s waves/SIMD %d: %7.1f ns per iteration per wave, %7.1f ns per iteration per SIMD\n", names[mode], wps, 1e6 * ms / iters, 1e6 * ms / iters / wps);
    }
  return 0;
}
