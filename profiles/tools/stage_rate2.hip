// What can ONE CU bring from the L2 into its LDS per microsecond, and does the answer depend on the operand LAYOUT?
// (round-4 review, item 2: profiles/r04l_stage_rate.txt measured 84.7 KB/us per CU = 21.7 TB/s chip-wide with row-strided K-contiguous
//  operands only, against the guide's ~34.5 TB/s of aggregate L2 bandwidth.)
//
// The operand stream of the forward's one-round GEMM and nothing else: NT 4096 x 1280 x K on 128 x 160 tiles = 256 workgroups, one per CU,
// every K-step (64 deep) a workgroup brings 128 + 160 rows x 128 B = 36 KiB through LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave
// instruction) into a 4-deep ring, counted vmcnt + one barrier per step -- no fragment reads, no MFMA.  Arms:
//   layout   strided   rows of 128 B at the operand's row stride (what gemm.hip stages: 8 rows per wave instruction); the row stride is a
//                      parameter: K elements (the model's tensors: 2560 / 10240 B), K + 64 (+128 B), K + 128 (+256 B)
//            packed    tile-major: the [rows x 64] tile of a K-step is one linear 16 / 20 KiB block (1 KiB linear bursts per wave instruction)
//   policy   default / sc1 / nt / sc0 sc1 on the loads
//   map      XCD-aware tile order (each XCD's 32 CUs share 8 A row-panels x 4 B panels, as xcd_tile_map does) or the plain order
//   hits     every workgroup stages tile (0, 0): all L2 hits after the first touch
// Reports us per launch (best of 5 x 10 back-to-back launches), the slope per K-step between two reduction lengths (fixed cost removed)
// and KB/us per CU from the slope.
//     hipcc --offload-arch=gfx950 -O3 -I sdxl-training-improvements_amd/csrc profiles/tools/stage_rate2.hip -o profiles/tools/stage_rate2
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int RA = 128, RB = 160, BK = 64, SEG = BK * 2;     // rows of the A / B tile, K-step, bytes per row and step
constexpr int NPC = (RA + RB) * SEG / 1024;                   // 36 pieces of 1 KiB per K-step
constexpr int STAGE = NPC * 1024;
constexpr int S = 4;

template <int POL>
__device__ __forceinline__ void dma(i32x4 srd, unsigned voff, unsigned soff, unsigned dst) {
  unsigned keep;
  if (POL == 0)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(dst) : "memory");
  else if (POL == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(dst) : "memory");
  else if (POL == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(dst) : "memory");
}

// NW waves (4 or 8); PACKED layout or strided; POL cache policy
template <int NW, bool PACKED, int POL>
__global__ __launch_bounds__(NW * 64 + 64, 1) void stage_kernel(const char* __restrict__ A, const char* __restrict__ B, unsigned lda_b, unsigned ldb_b,
                                                           int T, int xcd_map, int same_tile, unsigned* sink, int pf_lead = 0) {
  constexpr int PPW = NPC / NW;                     // pieces per wave and K-step (9 or 4.5 -> 36 / 8 is not whole: 8 waves issue 4 or 5)
  constexpr int PPWC = (NPC + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, by;
  const int id = blockIdx.x;
  if (xcd_map) { const int xcd = id & 7, li = id >> 3; bx = (xcd & 1) * 4 + (li & 3); by = (xcd >> 1) * 8 + (li >> 2); }
  else { bx = id & 7; by = id >> 3; }
  if (same_tile == 1) { bx = 0; by = 0; }
  if (same_tile == 2) { bx = id; by = id; }      // every workgroup its own operand panels: nothing shared, all of it crosses the fabric (packed arms only)
  if (wave == NW) {      // the extra L2-prefetch wave (launched only with pf_lead > 0): this workgroup's share of the XCD's lines, pf_lead steps ahead
    const int ln = bx & 3, lm = by & 7;
    unsigned snk = 0;
    auto touch = [&](int tp) {
      if (tp >= T) return;
      const char* src = nullptr;
      if (lane < 32) src = A + (size_t)(by * RA + ln * 32 + lane) * lda_b + (size_t)tp * SEG;
      else if (lane < 52) src = B + (size_t)(bx * RB + lm * 20 + (lane - 32)) * ldb_b + (size_t)tp * SEG;
      if (src) {
        if (pf_lead < 100) asm volatile("global_load_dword %0, %1, off sc1" : "+v"(snk) : "v"(src) : "memory");
        else asm volatile("global_load_dword %0, %1, off" : "+v"(snk) : "v"(src) : "memory");      // (lead + 100: plain loads, L1 + L2 allocate)
      }
    };
    const int lead = pf_lead % 100;
    for (int tp = S - 1; tp < lead; ++tp) touch(tp);
    for (int t = 0; t < T; ++t) { __builtin_amdgcn_s_barrier(); touch(t + lead); }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(snk) :: "memory");
    __syncthreads();
    if (snk == 0x12345678u) sink[1] = snk;
    return;
  }
  const i32x4 ra = make_srd(A, 0x7FFFFFFFu), rb = make_srd(B, 0x7FFFFFFFu);
  const unsigned lds_base = lds_addr_of(smem);
  // per-lane offset inside a piece: strided = 8 rows x 128 B (vector index xor-swizzled by the row, as gemm.hip stages); packed = linear
  const unsigned vo_a = PACKED ? lane * 16u : (unsigned)(lane >> 3) * lda_b + (((lane & 7) ^ (lane >> 3)) << 4);
  const unsigned vo_b = PACKED ? lane * 16u : (unsigned)(lane >> 3) * ldb_b + (((lane & 7) ^ (lane >> 3)) << 4);
  (void)PPW;
  // pf_lead == 1000: every XCD walks K from its own starting point (xcd * T / 8, wrapping): the 2 (A panels) / 4 (B panels) XCDs that share a
  // panel never ask the memory side for the same lines at the same time
  const int rot = pf_lead == 1000 ? ((id & 7) * T) / 8 : 0;
  auto issue = [&](int t0, int slot) {
    const int t = t0 + rot >= T ? t0 + rot - T : t0 + rot;
#pragma unroll
    for (int j = 0; j < PPWC; ++j) {
      const int pc = wave + NW * j;                 // 0 .. 35: pieces 0-15 = A tile, 16-35 = B tile
      if (pc < NPC) {
        const bool isa = pc < RA * SEG / 1024;
        const int pl = isa ? pc : pc - RA * SEG / 1024;
        unsigned so;
        if (PACKED) so = isa ? ((unsigned)(by * T + t) * (RA * SEG) + pl * 1024u) : ((unsigned)(bx * T + t) * (RB * SEG) + pl * 1024u);
        else so = isa ? (unsigned)(by * RA + pl * 8) * lda_b + (unsigned)t * SEG : (unsigned)(bx * RB + pl * 8) * ldb_b + (unsigned)t * SEG;
        dma<POL>(isa ? ra : rb, isa ? vo_a : vo_b, so, lds_base + slot * STAGE + pc * 1024);
      }
    }
  };
  const bool full = wave + NW * (PPWC - 1) < NPC;   // this wave issues PPWC pieces per step (else PPWC - 1)
  for (int d = 0; d < S - 1; ++d) issue(d < T ? d : 0, d);
  int wr = S - 1;
  for (int t = 0; t < T; ++t) {
    // step t landed: at most the S - 2 later steps' pieces of this wave outstanding
    if (NW == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else { if (full) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    issue(t + S - 1 < T ? t + S - 1 : 0, wr);
    wr = wr + 1 == S ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned acc = ((unsigned*)smem)[tid];
  if (acc == 0x12345678u) sink[0] = acc;
}

struct Arm { const char* name; int nw; bool packed; int pol; int pad_el; int xcd; int same; };

static int g_pf = 0;                 // > 0: launch the extra prefetch wave with this lead
static int g_nset = 1;               // > 1: "cold" -- launch i uses operand set i % g_nset (the sets are g_stride bytes apart: > L2 + memory-side cache in total)
static size_t g_strideA = 0, g_strideB = 0;
template <int NW, bool PACKED, int POL>
static float time_one(const char* A, const char* B, unsigned lda_b, unsigned ldb_b, int T, int xcd, int same, unsigned* sink) {
  const int smem = S * STAGE;
  if (g_nset > 1) {
    hipFuncSetAttribute((const void*)stage_kernel<NW, PACKED, POL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < g_nset; ++i)
        hipLaunchKernelGGL((stage_kernel<NW, PACKED, POL>), dim3(256), dim3(NW * 64 + ((g_pf && g_pf != 1000) ? 64 : 0)), smem, 0, A + (size_t)i * g_strideA, B + (size_t)i * g_strideB, lda_b, ldb_b, T, xcd, same, sink, g_pf);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms / g_nset < best) best = ms / g_nset;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best * 1e3f;
  }
  hipFuncSetAttribute((const void*)stage_kernel<NW, PACKED, POL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stage_kernel<NW, PACKED, POL>), dim3(256), dim3(NW * 64 + ((g_pf && g_pf != 1000) ? 64 : 0)), smem, 0, A, B, lda_b, ldb_b, T, xcd, same, sink, g_pf);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 < best) best = ms / 10;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best * 1e3f;   // us
}
static float dispatch(const Arm& a, const char* A, const char* B, unsigned lda_b, unsigned ldb_b, int T, unsigned* sink) {
#define CASE(NW, P, POL) if (a.nw == NW && a.packed == P && a.pol == POL) return time_one<NW, P, POL>(A, B, lda_b, ldb_b, T, a.xcd, a.same, sink);
  CASE(8, false, 0) CASE(8, true, 0) CASE(4, false, 0) CASE(4, true, 0)
  CASE(8, false, 1) CASE(8, true, 1) CASE(8, false, 2) CASE(8, true, 2) CASE(8, false, 3) CASE(8, true, 3)
  CASE(4, true, 2) CASE(4, false, 2)
#undef CASE
  return -1.f;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;            // run ONE arm (for rocprofv3 --pmc passes): its index
  const bool cold = argc > 2 && !strcmp(argv[2], "cold");    // operands from HBM: 24 operand sets (1.3 GB) in rotation
  const int M = 4096, N = 1280;
  const int Ks[2] = {2560, 5120};
  size_t abytes = (size_t)M * (5120 + 128) * 2, bbytes = (size_t)N * (5120 + 128) * 2;
  if (cold) { g_nset = 24; g_strideA = abytes; g_strideB = bbytes; abytes *= g_nset; bbytes *= g_nset; printf("# COLD: %d operand sets in rotation\n", g_nset); }
  char *A, *B; unsigned* sink;
  {   // the nothing-shared arm: 256 workgroups x 80 K-steps x (16 + 20) KiB
    const size_t ua = (size_t)256 * 80 * RA * SEG, ub = (size_t)256 * 80 * RB * SEG;
    if (abytes < ua) abytes = ua;
    if (bbytes < ub) bbytes = ub;
  }
  hipMalloc(&A, abytes); hipMalloc(&B, bbytes); hipMalloc(&sink, 64);
  {   // random bytes: the clock under load depends on the data (guide, rule 25)
    const size_t fill = (size_t)M * (5120 + 128) * 2;
    std::vector<unsigned> h(fill / 4);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    for (size_t o = 0; o + fill <= abytes; o += fill) hipMemcpy(A + o, h.data(), fill, hipMemcpyHostToDevice);
    for (size_t o = 0; o + fill <= bbytes; o += fill) hipMemcpy(B + o, h.data(), fill, hipMemcpyHostToDevice);
  }
  const Arm arms[] = {
      {"strided ld=K        8w xcd", 8, false, 0, 0, 1, 0},   {"strided ld=K+64     8w xcd", 8, false, 0, 64, 1, 0},
      {"strided ld=K+128    8w xcd", 8, false, 0, 128, 1, 0}, {"strided ld=K+32     8w xcd", 8, false, 0, 32, 1, 0},
      {"packed              8w xcd", 8, true, 0, 0, 1, 0},    {"strided ld=K        4w xcd", 4, false, 0, 0, 1, 0},
      {"packed              4w xcd", 4, true, 0, 0, 1, 0},    {"strided ld=K        8w plain order", 8, false, 0, 0, 0, 0},
      {"packed              8w plain order", 8, true, 0, 0, 0, 0},
      {"strided ld=K        8w all tile (0,0)", 8, false, 0, 0, 1, 1}, {"packed              8w all tile (0,0)", 8, true, 0, 0, 1, 1},
      {"strided ld=K   sc1  8w xcd", 8, false, 1, 0, 1, 0},   {"packed         sc1  8w xcd", 8, true, 1, 0, 1, 0},
      {"strided ld=K   nt   8w xcd", 8, false, 2, 0, 1, 0},   {"packed         nt   8w xcd", 8, true, 2, 0, 1, 0},
      {"strided ld=K sc0sc1 8w xcd", 8, false, 3, 0, 1, 0},   {"packed       sc0sc1 8w xcd", 8, true, 3, 0, 1, 0},
      {"strided ld=K+64     4w xcd", 4, false, 0, 64, 1, 0},  {"packed         nt   4w xcd", 4, true, 2, 0, 1, 0},
      {"packed  NOTHING SHARED 8w (755 MB)", 8, true, 0, 0, 1, 2},
  };
  if (argc > 4 && argv[4][0] == 'A') { g_strideB = 0; printf("# only the A operand (activations) rotates\n"); }
  if (argc > 4 && argv[4][0] == 'B') { g_strideA = 0; printf("# only the B operand (weights) rotates\n"); }
  g_pf = argc > 3 ? atoi(argv[3]) : 0;                     // argv[3]: lead of the extra L2-prefetch wave (strided arms; 0 = none)
  if (g_pf) printf("# with an L2-prefetch wave, lead %d K-steps (strided arms only are meaningful)\n", g_pf);
  const int narms = sizeof(arms) / sizeof(arms[0]);
  printf("%-40s %10s %10s %12s %12s %10s\n", "arm (NT 4096 x 1280 x K, 128x160 tiles)", "K=2560 us", "K=5120 us", "us / K-step", "KB/us per CU", "TB/s chip");
  for (int i = 0; i < narms; ++i) {
    if (only >= 0 && i != only) continue;
    const Arm& a = arms[i];
    float us[2];
    for (int k = 0; k < 2; ++k) {
      const int K = Ks[k], T = K / BK;
      const unsigned ld = a.packed ? 0u : (unsigned)(K + a.pad_el) * 2u;
      us[k] = dispatch(a, A, B, ld, ld, T, sink);
    }
    const double slope = (us[1] - us[0]) / ((Ks[1] - Ks[0]) / BK);
    const double kbus = STAGE / 1e3 / slope;
    printf("[%2d] %-35s %10.1f %10.1f %12.3f %12.1f %10.2f\n", i, a.name, us[0], us[1], slope, kbus, kbus * 256 / 1e3);
    fflush(stdout);
  }
  return 0;
}
