// Does an XCD's L2 keep what a kernel wrote (or read) for the NEXT kernel?  Producer: workgroup i writes chunk i of a buffer.  Consumer: workgroup i
// reads chunk (i + shift) % n: shift 0 = the chunk the same XCD wrote (workgroup i runs on XCD i % 8 in both launches), shift 1 = a chunk another XCD
// wrote.  If kernel boundaries flush / invalidate the L2, the two take the same time.  Also: consumer after a READ-ONLY producer (shift 0 / 1).
//   hipcc --offload-arch=gfx950 -O3 -o xcd_affinity xcd_affinity.hip && ./xcd_affinity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void produce(uint4* buf, long vec_per_chunk, unsigned seed) {
  uint4* c = buf + (long)blockIdx.x * vec_per_chunk;
  for (long v = threadIdx.x; v < vec_per_chunk; v += 256) c[v] = make_uint4(seed + v, seed, blockIdx.x, 1u);
}
__global__ __launch_bounds__(256) void consume(const uint4* buf, long vec_per_chunk, int shift, int n, unsigned* sink) {
  const uint4* c = buf + (long)((blockIdx.x + shift) % n) * vec_per_chunk;
  unsigned acc = 0;
  for (long v = threadIdx.x; v < vec_per_chunk; v += 256) { const uint4 x = c[v]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
  if (acc == 0x12345u) *sink = acc;
}
int main() {
  const int n = 2048;                       // workgroups = chunks
  unsigned* sink; CHECK(hipMalloc(&sink, 4));
  for (long chunk_bytes : {4096L, 8192L, 16384L}) {     // total 8 / 16 / 32 MB: 1 / 2 / 4 MB per XCD (L2 = 4 MB)
    const long vpc = chunk_bytes / 16;
    const int NB = 24;                      // rotate over buffers so that nothing survives in the memory-side cache either (NB x total > 256 MB for the larger sizes)
    std::vector<uint4*> bufs(NB);
    for (auto& b : bufs) CHECK(hipMalloc(&b, n * chunk_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int producer = 0; producer < 2; ++producer)      // 0: the producer WRITES the buffer, 1: it READS it (consume with shift 0)
      for (int shift : {0, 1, 8}) {
        float tot = 0;
        for (int it = 0; it < 3 * NB; ++it) {
          uint4* b = bufs[it % NB];
          if (producer == 0) hipLaunchKernelGGL(produce, dim3(n), dim3(256), 0, 0, b, vpc, (unsigned)it);
          else hipLaunchKernelGGL(consume, dim3(n), dim3(256), 0, 0, b, vpc, 0, n, sink);
          CHECK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(consume, dim3(n), dim3(256), 0, 0, b, vpc, shift, n, sink);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (it >= NB) tot += ms;
        }
        printf("%5.1f MB (%.1f MB per XCD)  after a %s of it, consumer shift %d: %6.2f us\n", n * chunk_bytes / 1e6, n * chunk_bytes / 8e6,
               producer == 0 ? "WRITE" : "READ ", shift, tot / (2 * NB) * 1e3);
      }
    for (auto& b : bufs) CHECK(hipFree(b));
  }
  return 0;
}
