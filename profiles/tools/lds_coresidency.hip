// Which pairs of dynamic-LDS sizes let two workgroups (of two kernels on two streams) share a CU on MI355X?
// Each kernel: 256 workgroups x 256 threads sleeping ~60 us (a fixed number of s_sleep 127); both launched back to back on two streams: ~60 us = co-resident, ~120 = not.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* out, long long ticks) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = threadIdx.x;
  for (long long i = 0; i < ticks; ++i) asm volatile("s_sleep 127");      // ~8k cycles each
  out[blockIdx.x * blockDim.x + threadIdx.x] = sm[(threadIdx.x + 1) & 255];
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int pairs[][2] = {{65, 73}, {64, 64}, {80, 80}, {96, 64}, {96, 55}, {96, 48}, {96, 32}, {112, 48}, {128, 32}, {128, 16}, {145, 8}, {81, 79}, {100, 60}, {90, 70}, {88, 64}, {73, 73}, {73, 80}, {73, 87}};
  for (auto& pr : pairs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), pr[0] * 1024, s1, out, 15LL);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), pr[1] * 1024, s2, out + (1 << 20), 15LL);
      hipDeviceSynchronize();
      hipEventRecord(e1, s1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("LDS %3d KiB + %3d KiB: %6.1f us  %s\n", pr[0], pr[1], best * 1e3, best * 1e3 < 95 ? "co-resident" : "serialized");
  }
  return 0;
}
