// scratch diagnostic: which component bounds the GEMM K-step?  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdxl-training-improvements_amd/csrc -DSDXL_GEMM_DIAG=<bits> -o diag profiles/tools/diag.hip
#ifndef SDXL_GEMM_DIAG
#define SDXL_GEMM_DIAG 0
#endif
#include "../../sdxl-training-improvements_amd/csrc/gemm.hip"
#include <stdio.h>
void sdxl_set_error(const char* fmt, ...) {}
int g_knobs[SDXL_NKNOBS];
__global__ void null_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main(int argc, char** argv) {
  int form = argc > 1 ? atoi(argv[1]) : 0;
  if (argc > 2) gemm_set_mode((atoi(argv[2]) << 2) | 1);      // force a configuration of the 128-row kernel
  struct { int M, N, K; } shapes[] = {{1280, 1280, 4096}, {3840, 1280, 4096}, {10240, 1280, 4096}, {4096, 1280, 64}, {4096, 1280, 1280}, {4096, 1280, 5120}, {4096, 10240, 1280}, {4096, 1280, 10240}, {4096, 5120, 1280}, {8192, 8192, 4096}};
  bf16 *A, *B; void* C;
  size_t big = (size_t)8192 * 10240 * 2 * 2;
  hipMalloc(&A, big); hipMalloc(&B, big); hipMalloc(&C, big * 2);
  hipMemset(A, 0x11, big); hipMemset(B, 0x11, big);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(null_kernel, dim3(320), dim3(256), 0, st, (int*)nullptr);
    hipEventRecord(e0, st);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(null_kernel, dim3(320), dim3(256), 0, st, (int*)nullptr);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("null kernel back-to-back: %.2f us per launch\n", ms * 10);
  }
  for (auto sh : shapes) {
    for (int dbg = SDXL_GEMM_DIAG; dbg <= SDXL_GEMM_DIAG; ++dbg) {
      GemmP g; gemm_defaults(&g);
      g.form = form; g.A = A; g.B = B; g.C = C; g.M = sh.M; g.N = sh.N; g.K = sh.K;
      if (form == 0) { g.lda = sh.K; g.ldb = sh.K; g.ldc = sh.N; }
      if (form == 1) { g.lda = sh.K; g.ldb = sh.N; g.ldc = sh.N; }
      if (form == 2) { g.lda = sh.M; g.ldb = sh.N; g.ldc = sh.N; g.out_f32 = 1; }
      for (int i = 0; i < 3; ++i) launch_gemm(g, st);
      hipEventRecord(e0, st);
      for (int i = 0; i < 20; ++i) launch_gemm(g, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
      printf("form %d %5dx%5dx%5d dbg=%d%s%s%s  %8.1f us  %7.1f TF/s-equiv\n", form, sh.M, sh.N, sh.K, dbg, dbg & 1 ? " noMFMA" : "       ",
             dbg & 2 ? " noDMA" : "      ", dbg & 4 ? " noLDSread" : "          ", ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms / 1e9);
    }
  }
  return 0;
}
