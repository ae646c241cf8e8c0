// LDS read rate of the GEMM kernels' fragment pattern: W waves per workgroup (one workgroup per CU), each issuing R ds_read_b128 per
// s_waitcnt in a loop, (a) linear addresses (lane * 16), (b) the K-contiguous tile pattern of gemm_tiles.h (row = lane & 15, 16-byte chunk
// (lane >> 4) ^ (row & 7) of a 128-byte row), with and without a workgroup barrier per "K-step".  Reports bytes per clock per CU at the
// clock rocm-smi shows under load is not known here: prints GB/s per CU and, for 2.2 GHz, B/clk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PATTERN, bool BARRIER, int R>
__global__ __launch_bounds__(512) void lds_kernel(unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 36 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const int row = lane & 15, g = lane >> 4;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int off;
      if (PATTERN == 0) off = ((r + wave) & 31) * 1024 + lane * 16;                                   // linear 1 KiB blocks
      else off = (((r + wave) & 15) * 16 + row) * 128 + ((((r >> 4) * 4 + g) ^ (row & 7)) << 4);     // [rows][128 B], swizzled chunk
      v[r] = *(const u32x4*)(smem + off);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc ^= v[r][0] ^ v[r][3];
    if (BARRIER) __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x1234567u) sink[0] = acc;
}
template <int PATTERN, bool BARRIER, int R>
static void run(unsigned* sink, int waves, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((lds_kernel<PATTERN, BARRIER, R>), dim3(256), dim3(waves * 64), 36 * 1024, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (double)waves * R * 1024 * iters;      // per CU
  printf("%-26s %d waves, %2d reads / wait%s: %7.1f GB/s per CU = %5.1f B/clk at 2.2 GHz  (%.3f us per iteration)\n", name, waves, R,
         BARRIER ? " + barrier" : "          ", bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 2.2e9, best * 1e3 / iters);
}
int main() {
  unsigned* sink; hipMalloc(&sink, 64);
  for (int w : {4, 8}) {
    run<0, false, 16>(sink, w, "linear");
    run<1, false, 16>(sink, w, "K-contiguous tile pattern");
    run<1, true, 16>(sink, w, "K-contiguous tile pattern");
    run<1, true, 14>(sink, w, "K-contiguous tile pattern");
    run<1, false, 32>(sink, w, "K-contiguous tile pattern");
  }
  return 0;
}
