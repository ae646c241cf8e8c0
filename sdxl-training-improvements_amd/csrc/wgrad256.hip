// Linear weight gradient with a LONG reduction and a small output (the 64 x 64-token level: 16 384 rows, 640-wide outputs):
//   dW[m][n] (+)= sum_k dY[k][m] * X[k][n]              (TN form, fp32 output)
// The 128 x 160-tile kernel of gemm.hip launches 80-240 workgroups for these shapes: one latency-bound workgroup per CU walking up
// to 256 K-steps at ~1.2 us each (300-510 TFLOP/s; 7 ms of serialized kernel time and, by the knock-out timing, 3.9 ms of the step).
// Same remedy as conv_wgrad3.hip -- more products per staged byte and per barrier: a workgroup owns a 256 x 160 output tile, i.e. TWO
// stacked 128-column dY tiles against ONE X tile; 8 waves (4 x 2), wave tile 64 x 80 (80 accumulator VGPRs), 40 MFMAs per wave and
// K-step behind 7 DMA pieces, 3-deep LDS ring (157 KiB), asm-issued LDS-DMA + counted vmcnt + one raw barrier per step.
// The workgroup owns its CU; split-K aims at ~144 workgroups per launch.  Contract (slab + fixed-order reduce, Cb emit, accumulate, bias
// gradient) as the TN form of gemm.hip.
#include "gemm_tiles.h"

namespace {

constexpr int WL_BM = 256, WL_BN = 160, WL_BK = 64, WL_S = 3;
constexpr int WL_A_BYTES = WL_BK * WL_BM * 2;            // two [64 k][128 m] sub-tiles     32 KiB
constexpr int WL_B_BYTES = WL_BK * WL_BN * 2;            // [64 k][160 n]                   20 KiB
constexpr int WL_STAGE = WL_A_BYTES + WL_B_BYTES;
constexpr int WL_SMEM = WL_S * WL_STAGE + 1024;          // + 1 KiB that absorbs the padding pieces
constexpr int WL_NL = 7;                                 // DMA pieces per wave and K-step: 4 (A) + 3 (B)

__global__ __launch_bounds__(512, 2) void wgrad256_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 waves, wave tile 64 x 80
  const int l16 = lane & 15, g = lane >> 4;
  int bx, by, split;
  xcd_seq_map(p.xcd_bh, bx, by, split);      // (gemm_tiles.h: XCD-aware order of tiles and reduction splits)
  const int n0 = bx * WL_BN, m0 = by * WL_BM;
  const int ktiles = p.K / WL_BK;
  const int chunk = (ktiles + p.splitk - 1) / p.splitk;
  const int kt_begin = split * chunk;
  const int kt_end = min(ktiles, kt_begin + chunk);
  const int T = kt_end - kt_begin;
  const bf16* zsrc = (const bf16*)g_zero16;
  const unsigned lds_base = lds_addr_of(smem);
  const unsigned pad_dst = lds_base + WL_S * WL_STAGE;

  // A: two [64][128] sub-tiles (N-contiguous image of gemm.hip each); chunk c (0..31): sub-tile c >> 4, k-rows 4 (c & 15) ..; 4 per wave
  const bf16* pa[4];
  long sa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = wave + 8 * j;
    const int sub = c >> 4, cc = c & 15;
    const int krow = cc * 4 + (lane >> 4);
    const int m = m0 + sub * 128 + (nc_logical<128>(krow, lane & 15) << 3);
    const bool ok = m < p.M;
    pa[j] = ok ? p.A + ((long)kt_begin * WL_BK + krow) * p.lda + m : zsrc;
    sa[j] = ok ? (long)WL_BK * p.lda : 0;
  }
  // B: [64][160]: chunk c (0..19) = vectors 64c .. 64c+63 of the [64][20] vector grid; 3 slots per wave (ids >= 20: padding)
  const bf16* pb[3];
  long sb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = wave + 8 * j;
    const int q = c * 64 + lane;
    const int krow = q / 20, pv = q - krow * 20;
    const int n = n0 + (nc_logical<160>(krow, pv) << 3);
    const bool ok = c < 20 && n < p.N;
    pb[j] = ok ? p.B + ((long)kt_begin * WL_BK + krow) * p.ldb + n : zsrc;
    sb[j] = ok ? (long)WL_BK * p.ldb : 0;
  }
  auto issue_piece = [&](int pc, int slot, bool live) {
    const unsigned At = lds_base + slot * WL_STAGE, Bt = At + WL_A_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (pc == j) {
        lds_dma16_global(live ? (const void*)pa[j] : (const void*)zsrc, At + (wave + 8 * j) * 1024);
        pa[j] += sa[j];
      }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (pc == 4 + j) {
        const int c = wave + 8 * j;
        lds_dma16_global(live ? (const void*)pb[j] : (const void*)zsrc, c < 20 ? Bt + c * 1024 : pad_dst);
        pb[j] += sb[j];
      }
  };
  auto stage = [&](int slot, bool live) {
#pragma unroll
    for (int pc = 0; pc < WL_NL; ++pc) issue_piece(pc, slot, live);
  };

  f32x4 acc[4][5];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.bias_grad != nullptr && bx == 0 && wn == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  stage(0, 0 < T);
  stage(1, 1 < T);
  int rd = 0, wr = 2;
  for (int t = 0; t < T; ++t) {
    wait_vmcnt<WL_NL>();                    // this wave's pieces of step t have landed (step t + 1 may be outstanding)
    __builtin_amdgcn_s_barrier();           // ... everyone's; and every wave is done reading slot `wr` (step t - 1)
    const bool live = t + 2 < T;
    const char* At = smem + rd * WL_STAGE + (wm >> 1) * (WL_A_BYTES / 2);     // this wave's 128-column sub-tile
    const char* Bt = smem + rd * WL_STAGE + WL_A_BYTES;
    int pc = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[5];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag_nc<128>(At, ks * 32 + g * 8, (wm & 1) * 64 + i * 16, l16);
#pragma unroll
      for (int j = 0; j < 5; ++j) fb[j] = frag_nc<160>(Bt, ks * 32 + g * 8, wn * 80 + j * 16, l16);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        // one DMA piece of step t + 2 behind every group of 5 products (7 pieces over the 8 groups of a K-step)
        if (pc < WL_NL) { issue_piece(pc, wr, live); }
        ++pc;
      }
    }
    rd = rd + 1 == WL_S ? 0 : rd + 1;
    wr = wr + 1 == WL_S ? 0 : wr + 1;
  }
  wait_vmcnt<0>();     // tail pieces must not outlive the workgroup's LDS allocation

  if (do_bias && l16 == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + g * 4 + r;
        if (m < p.M) gemm_bias_out(p.bias_grad, p.slab, p.slab_ld, p.splitk, split, p.M, m, accb[i][r]);
      }
  }
  // ---- epilogue: lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int n = n0 + wn * 80 + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4 x = acc[i][j];
      float* c = (float*)p.C + (long)m * p.ldc + n;
      if (p.splitk > 1) {
        *(f32x4*)(p.slab + ((long)split * p.M + m) * p.slab_ld + n) = x;
      } else if (p.Cb) {
        if (p.accumulate) {
          const f32x4 a = *(const f32x4*)c;
          x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3];
        }
        bf16x4 o;
        o[0] = (bf16)(x[0] * p.cb_scale); o[1] = (bf16)(x[1] * p.cb_scale);
        o[2] = (bf16)(x[2] * p.cb_scale); o[3] = (bf16)(x[3] * p.cb_scale);
        *(bf16x4*)(p.Cb + (long)m * p.ldc + n) = o;
      } else if (p.accumulate) {
        f32x4 a = *(f32x4*)c;
        a[0] += x[0]; a[1] += x[1]; a[2] += x[2]; a[3] += x[3];
        *(f32x4*)c = a;
      } else {
        *(f32x4*)c = x;
      }
    }
  }
}

bool g_wl_enabled = true;

}  // namespace

void wgrad256_set_enabled(bool on) { g_wl_enabled = on; }

// the plan's choice: linear weight gradients over >= 16 384 rows (the 64 x 64-token level and above)
bool wgrad256_policy(int M, int N, long red) {
  return g_wl_enabled && red >= 16384 && red % WL_BK == 0 && M % 8 == 0 && N % 8 == 0;
}
bool wgrad256_applicable(const GemmP& p) {
  if (p.form != GEMM_TN || p.taps != 1 || p.group > 1 || !p.out_f32) return false;
  if (p.K % WL_BK || p.M % 8 || p.N % 8 || p.lda % 8 || p.ldb % 8 || p.ldc % 4) return false;
  return true;
}
int wgrad256_pick_splitk(int M, int N, long red) {
  const long tiles = (long)cdiv(M, WL_BM) * cdiv(N, WL_BN);
  const long ktiles = red / WL_BK;
  const long target = KNOB(13) > 0 ? KNOB(13) : 144;
  long s = (target + tiles / 2) / tiles;
  if (s < 1) s = 1;
  while (s > 1 && ktiles / s < 16) --s;
  if (s > 32) s = 32;
  return (int)s;
}

int launch_wgrad256(const GemmP& p, hipStream_t st) {
  ARG_CHECK(wgrad256_applicable(p), "wgrad256: problem does not fit (TN, one tap, K %% 64 == 0)");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WL_SMEM));
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, WL_BN), cdiv(p.M, WL_BM), p.splitk);
  GemmP q = p;
  q.xcd_bh = KNOB(34) == 1 ? 0 : xcd_band_rows(grid.x, grid.y, grid.z, WL_BM, WL_BN);
  GEMM_LAUNCH(wgrad256_kernel, grid, dim3(512), WL_SMEM, st, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
