// Issue-rate microbenchmark of the VALU instructions the attention kernels live on (gfx950): cycles per wave64 instruction
// at 1, 2, 3 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP 64
template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
  float a[8], b = threadIdx.x * 1e-3f;
  for (int i = 0; i < 8; ++i) a[i] = b + i;
  f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(b + i); fb[i] = (__bf16)(b - i); }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      }
      if (OP == 4) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          f32x2 v = {a[i], a[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
          a[i] = v[0]; a[i + 1] = v[1];
        }
      }
      if (OP == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8 * 4096);
  const char* names[] = {"v_exp_f32", "v_fma_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "mfma_16x16x32_bf16", "v_rcp_f32", "v_exp_f16"};
  const int iters = 2000;
  for (int op = 0; op < 8; ++op)
    for (int wps = 1; wps <= 3; ++wps) {   // waves per SIMD: block of 256 * wps threads, one block per CU
      dim3 grid(256), blk(256 * wps);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define L(OP) hipLaunchKernelGGL(k<OP>, grid, blk, 0, 0, out, cyc, iters)
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (op) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break; case 7: L(7); break; }
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double c = 0; for (auto v : h) c += v; c /= 256;
      // s_memtime ticks at 100 MHz? report both: ticks per instruction and ns per instruction per wave
      const double ninst = (double)iters * REP;
      printf("%-20s waves/SIMD %d: %.3f memtime-ticks/inst  %.3f ns/inst/wave -> %.2f ns per inst per SIMD\n", names[op], wps, c / ninst, 1e6 * ms / ninst, 1e6 * ms / ninst / wps);
    }
  return 0;
}
