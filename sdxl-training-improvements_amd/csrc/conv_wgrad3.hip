// Weight gradient of the same-size stride-1 3x3 convolution, THREE TAPS PER WORKGROUP (one row of the 3x3 stencil).
//
//   dW[co][tap][ci] (+)= sum over pixels p of dY[p][co] * X[p + shift(tap)][ci]          (zero outside the image)
//
// The general kernel (gemm.hip, TN form with the gather) gives every tap its own workgroups: nine times the dY tile through
// L2 -> LDS, and at the long reductions of the 128 x 128 / 64 x 64 levels (65 536 / 16 384 pixels) a handful of 128 x 160 output
// tiles per tap, i.e. one latency-bound workgroup per CU walking 256 K-steps at ~1.2 us each: 300-390 TFLOP/s, 12 ms of the step's
// serialized kernel time -- and 7 ms of the step itself (timing experiment without them, profiles/README.md r03).
// Here a workgroup owns one output tile (128 co x 160 ci) for the taps dx = -1, 0, +1 of ONE stencil row dy:
//   * the dY tile of a K-step (64 pixels) is staged once and multiplied three times;
//   * the X tile is staged with a halo of one pixel on each side (66 rows): the three taps read it at row offsets 0 / 1 / 2.
//     A K-step is a piece of ONE image row (W = 64: the row, W = 128: half of it -- the launcher requires W % 64 == 0), so the
//     halo rows are either the neighbouring pixels of the same row or out of the row: zero-filled by the DMA source select;
//   * 8 waves (4 x 2), wave tile 32 x 80, accumulators for 3 taps = 120 VGPRs; 60 MFMAs per wave and K-step behind 5 DMA pieces:
//     the K-step is matrix-pipe bound (~1 us) instead of latency bound, three taps' worth of products per step;
//   * 3-deep LDS ring (2 steps of DMA in flight), asm-issued LDS-DMA + counted vmcnt + one raw barrier per step.
// The workgroup owns its CU (8 waves x ~220 VGPRs, 112 KiB): these launches do not co-reside with the other stream's kernels,
// they take fewer CUs for a much shorter time instead.
// Split-K, slab + fixed-order reduce, bf16 emit (Cb), accumulate, bias gradient: same contract as the TN form of gemm.hip.
#include "gemm_tiles.h"

namespace {

constexpr int W3_BM = 128, W3_BN = 160, W3_BK = 64, W3_S = 3;
constexpr int W3_A_BYTES = W3_BK * W3_BM * 2;            // [64 k][128 m]            16 KiB
// SUB = image rows per K-step: 1 (W = 64, 128, ...: a K-step is a piece of one row) or 2 (W = 32: two whole rows, each with its own halo)
constexpr int w3_b_rows(int SUB) { return W3_BK + 2 * SUB; }
constexpr int w3_b_chunks(int SUB) { return (w3_b_rows(SUB) * (W3_BN / 8) + 63) / 64; }      // 21 / 22 chunks of 1 KiB (the last one partly padding)
constexpr int w3_stage(int SUB) { return W3_A_BYTES + w3_b_chunks(SUB) * 1024; }
constexpr int w3_smem(int SUB) { return W3_S * w3_stage(SUB) + 1024; }                       // + 1 KiB that absorbs the padding pieces
constexpr int W3_NL = 5;                                 // DMA pieces per wave and K-step: 2 (A) + 3 (B)

template <int SUB>
__global__ __launch_bounds__(512, 2) void conv_wgrad3_kernel(const GemmP p) {
  constexpr int W3_B_VEC = w3_b_rows(SUB) * (W3_BN / 8);
  constexpr int W3_NCB = w3_b_chunks(SUB);
  constexpr int W3_STAGE = w3_stage(SUB);
  constexpr int RPS = W3_BK / SUB + 2;                   // physical rows per image row of the K-step (pixels + halo)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 waves, wave tile 32 x 80
  const int l16 = lane & 15, g = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, dyrow = blockIdx.z / p.splitk, split = blockIdx.z - dyrow * p.splitk;     // stencil row 0..2, reduction split
  if (p.xcd_bh > 0) xcd_seq_map(p.xcd_bh, 3, bx, by, dyrow, split);      // (gemm_tiles.h: a tile's three stencil rows side by side, splits apart)
  const int n0 = bx * W3_BN, m0 = by * W3_BM;
  const int tdy = dyrow - 1;
  const int Wm = p.Wm, Hm = p.Hm;
  const int ktiles = p.K / W3_BK;
  const int chunk = (ktiles + p.splitk - 1) / p.splitk;
  const int kt_begin = split * chunk;
  const int kt_end = min(ktiles, kt_begin + chunk);
  const int T = kt_end - kt_begin;
  const bf16* zsrc = (const bf16*)g_zero16;
  const unsigned lds_base = lds_addr_of(smem);
  const unsigned pad_dst = lds_base + W3_S * w3_stage(SUB);

  // ---- per-lane DMA descriptors ----
  // A (dY^T tile, [64 k][128 m], N-contiguous image as in gemm.hip): chunk c = k-rows 4c .. 4c+3; 2 chunks per wave
  const bf16* pa[2];
  long sa[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = wave + 8 * j;
    const int krow = c * 4 + (lane >> 4);
    const int m = m0 + (nc_logical<128>(krow, lane & 15) << 3);
    const bool ok = m < p.M;
    pa[j] = ok ? p.A + ((long)kt_begin * W3_BK + krow) * p.lda + m : zsrc;
    sa[j] = ok ? (long)W3_BK * p.lda : 0;
  }
  // B (X tile with halo, [66 (68) rows][160 ci]): chunk c = vectors 64c .. 64c+63 of the [rows][20] vector grid; physical row
  // r = sub * RPS + rr holds pixel (first pixel of the K-step) + sub * W + rr - 1 of image row sy + sub; 3 chunk slots per wave
  // (ids wave + 8j; ids >= W3_NCB and vectors beyond the grid are padding)
  const bf16* pb[3];
  int brr[3], bsub[3];         // rr (-1: padding / column out of range -> always the zero vector), sub
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = wave + 8 * j;
    const int q = c * 64 + lane;
    const int r = q / 20, pv = q - r * 20;
    const int sub = r / RPS, rr = r - sub * RPS;
    const int n = n0 + (nc_logical<160>(r, pv) << 3);
    const bool ok = c < W3_NCB && q < W3_B_VEC && n < p.N;
    brr[j] = ok ? rr : -1;
    bsub[j] = sub;
    pb[j] = p.B + ((long)(sub * Wm + rr - 1) + (long)tdy * Wm) * p.ldb + n;     // + the K-step's first pixel (ub); dereferenced only when valid
  }
  // wave-uniform position of the K-step being staged: first pixel kk (linear), its (y, x0) in the image
  long ub = (long)kt_begin * W3_BK * p.ldb;
  int sy, sx;
  {
    const long kk0 = (long)kt_begin * W3_BK;
    const long hw = (long)Hm * Wm;
    const int rem = (int)(kk0 % hw);
    sy = rem / Wm;
    sx = rem - sy * Wm;
  }
  auto issue_piece = [&](int pc, int slot, bool live) {
    const unsigned At = lds_base + slot * W3_STAGE, Bt = At + W3_A_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (pc == j) {
        lds_dma16_global(live ? (const void*)pa[j] : (const void*)zsrc, At + (wave + 8 * j) * 1024);
        pa[j] += sa[j];
      }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (pc == 2 + j) {
        const int c = wave + 8 * j;
        const int ys = sy + bsub[j] + tdy, lx = sx + brr[j] - 1;
        const bool v = live && brr[j] >= 0 && ys >= 0 && ys < Hm && lx >= 0 && lx < Wm;
        lds_dma16_global(v ? (const void*)(pb[j] + ub) : (const void*)zsrc, c < W3_NCB ? Bt + c * 1024 : pad_dst);
      }
  };
  auto advance = [&]() {       // the next K-step: 64 pixels further (SUB = 1: along the row, W % 64 == 0; SUB = 2: two rows down); images wrap
    ub += (long)W3_BK * p.ldb;
    if (SUB == 1) {
      sx += W3_BK;
      if (sx >= Wm) { sx = 0; if (++sy >= Hm) sy = 0; }
    } else {
      sy += SUB;
      if (sy >= Hm) sy -= Hm;
    }
  };
  auto stage = [&](int slot, bool live) {
#pragma unroll
    for (int pc = 0; pc < W3_NL; ++pc) issue_piece(pc, slot, live);
    advance();
  };

  f32x4 acc[3][2][5];
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[t3][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.bias_grad != nullptr && bx == 0 && dyrow == 0 && wn == 0;
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // B fragment of tap shift sh (= 1 + dx: physical row offset) for the 32-deep half ks, 16 columns at col0
  // (SUB = 2: the second 32-deep half is the K-step's second image row, behind the first row's two halo rows)
  auto frag_b = [&](const char* Bt, int ks, int sh, int col0) -> bf16x8 {
    const int r0 = ks * 32 + g * 8 + (l16 >> 2) + sh + (SUB == 2 ? 2 * ks : 0), r1 = r0 + 4;
    const int v = (col0 >> 3) + ((l16 >> 1) & 1);
    const char* p0 = Bt + r0 * (W3_BN * 2) + (nc_phys<160>(r0, v) << 4) + (l16 & 1) * 8;
    const char* p1 = Bt + r1 * (W3_BN * 2) + (nc_phys<160>(r1, v) << 4) + (l16 & 1) * 8;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return u.v;
  };

  // ---- prologue: two K-steps in flight ----
  stage(0, 0 < T);
  stage(1, 1 < T);
  int rd = 0, wr = 2;
  for (int t = 0; t < T; ++t) {
    wait_vmcnt<W3_NL>();                    // this wave's pieces of step t have landed (step t + 1 may be outstanding)
    __builtin_amdgcn_s_barrier();           // ... everyone's; and every wave is done reading slot `wr` (step t - 1)
    const bool live = t + 2 < T;
    const char* At = smem + rd * W3_STAGE;
    const char* Bt = At + W3_A_BYTES;
    int pc = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = frag_nc<128>(At, ks * 32 + g * 8, wm * 32 + i * 16, l16);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
      }
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        bf16x8 fb[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = frag_b(Bt, ks, t3, wn * 80 + j * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j)
            acc[t3][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[t3][i][j], 0, 0, 0);
        // one DMA piece of step t + 2 behind every group of 10 products (5 pieces over the 6 groups of a K-step)
        if (pc < W3_NL) { issue_piece(pc, wr, live); }
        ++pc;
      }
    }
    advance();
    rd = rd + 1 == W3_S ? 0 : rd + 1;
    wr = wr + 1 == W3_S ? 0 : wr + 1;
  }
  wait_vmcnt<0>();     // tail pieces must not outlive the workgroup's LDS allocation

  if (do_bias && l16 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + g * 4 + r;
        if (m < p.M) gemm_bias_out(p.bias_grad, p.slab, p.slab_ld, p.splitk, split, p.M, m, accb[i][r]);
      }
  }
  // ---- epilogue: lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile, for each of the three taps ----
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    const long tap_off = (long)(dyrow * 3 + t3) * p.c_tap_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + i * 16 + l16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int n = n0 + wn * 80 + j * 16 + g * 4;
        if (n >= p.N) continue;
        f32x4 x = acc[t3][i][j];
        float* c = (float*)p.C + (long)m * p.ldc + tap_off + n;
        if (p.splitk > 1) {
          *(f32x4*)(p.slab + ((long)split * p.M + m) * p.slab_ld + tap_off + n) = x;
        } else if (p.Cb) {
          if (p.accumulate) {
            const f32x4 a = *(const f32x4*)c;
            x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3];
          }
          bf16x4 o;
          o[0] = (bf16)(x[0] * p.cb_scale); o[1] = (bf16)(x[1] * p.cb_scale);
          o[2] = (bf16)(x[2] * p.cb_scale); o[3] = (bf16)(x[3] * p.cb_scale);
          *(bf16x4*)(p.Cb + (long)m * p.ldc + tap_off + n) = o;
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
}

bool g_w3_enabled = true;

}  // namespace

void conv_wgrad3_set_enabled(bool on) { g_w3_enabled = on; }

// p as normalised by launch_gemm (TN, taps == 9, fp32 output, slab_ld set)
bool conv_wgrad3_applicable(const GemmP& p) {
  if (!g_w3_enabled) return false;
  if (p.form != GEMM_TN || p.taps != 9 || p.group > 1 || p.up2) return false;
  if (p.sm != 1 || p.sd != 1 || p.Hm != p.Hs || p.Wm != p.Ws) return false;      // same-size stride-1
  if ((p.Wm % W3_BK != 0 && !(SDXL_UP2_3 && p.Wm == 32 && p.Hm % 2 == 0)) || p.K % W3_BK != 0 || p.K % ((long)p.Hm * p.Wm) != 0) return false;
  if (p.M % 8 || p.N % 8 || p.lda % 8 || p.ldb % 8 || p.ldc % 4) return false;
  return true;
}

bool conv_wgrad3_policy(int M, int N, long red, int Wm, int stride) {
  // the 128^2 / 64^2 levels (>= 16 384 pixels).  The W = 32 form (1280-channel level) is 1.3-1.4x faster standalone (1 112 vs 795 TFLOP/s)
  // but costs the step 0.3 ms: in the transformer-heavy stretch of the backward a workgroup that owns its CU displaces the
  // co-resident critical-path kernels, which the 4-wave one-tap kernel does not (knob 14 = 2 enables it: A/B runs)
  const bool w32 = Wm == 32 && KNOB(14) == 2;
  return g_w3_enabled && stride == 1 && (Wm % W3_BK == 0 || w32) && red >= (w32 ? 4096 : 16384) && M % 8 == 0 && N % 8 == 0;
}
// split-K factor: the kernel owns its CUs, so aim at ~half the chip per launch (the other stream keeps the rest), at least 16
// K-steps per split
int conv_wgrad3_pick_splitk(int M, int N, long red) {
  const long tiles = (long)cdiv(M, W3_BM) * cdiv(N, W3_BN) * 3;
  const long ktiles = red / W3_BK;
  const long target = KNOB(13) > 0 ? KNOB(13) : 256;
  long s = (target + tiles / 2) / tiles;
  if (s < 1) s = 1;
  while (s > 1 && ktiles / s < 16) --s;
  if (s > 32) s = 32;
  return (int)s;
}

int launch_conv_wgrad3(const GemmP& p, hipStream_t st) {
  ARG_CHECK(conv_wgrad3_applicable(p), "conv_wgrad3: problem does not fit (same-size 3x3, W %% 64 == 0 or W == 32)");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, w3_smem(1)));
#ifdef SDXL_DIAG      // the W = 32 form (two image rows per K-step) exists in the diagnostics build only
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, w3_smem(2)));
#endif
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, W3_BN), cdiv(p.M, W3_BM), 3 * p.splitk);
  GemmP q = p;
  q.xcd_bh = KNOB(34) == 1 ? 0 : xcd_band_rows(grid.x, grid.y, grid.z, W3_BM, W3_BN, 3);
#ifdef SDXL_DIAG
  if (p.Wm == 32) hipLaunchKernelGGL(conv_wgrad3_kernel<2>, grid, dim3(512), w3_smem(2), st, q);
  else
#endif
  GEMM_LAUNCH(conv_wgrad3_kernel<1>, grid, dim3(512), w3_smem(1), st, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
