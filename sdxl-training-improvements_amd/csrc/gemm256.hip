// bf16 MFMA GEMM for gfx950, 256 x 256 x 64 workgroup tile, 8 waves (2 x 4), wave tile 128 x 64, 8-phase schedule.
//
// Why a second kernel: with 128-row tiles (gemm.hip) every MFMA flop pulls (1/128 + 1/BN) operand elements through the
// L2 -> LDS path, which saturates at ~39 B/clk/CU on this chip -- that stream, not the matrix pipe, set the pace
// (profiles/r01f, r01m: 2.8x the algorithmic bytes fetched).  A 256 x 256 tile halves those bytes per flop and a
// 128 x 64 wave tile needs 0.375 LDS fragment reads per MFMA instead of 0.5, so the LDS read phase of one wave group
// fits inside the MFMA phase of the other.
//
// Structure (per workgroup, one per CU, 128 KiB LDS):
//   * the A tile [256][64] and B tile [256][64] of a K-step live in LDS as four 16 KiB REGIONS: A-a0 / A-a1 (the first /
//     second 64 rows of both wave rows), B-b0 / B-b1 (the first / second 32 columns of all four wave columns); two
//     buffers (even / odd K-tile).
//   * a K-tile is multiplied in two PHASES, one 64 x 64 half of every wave's accumulator tile each (32 MFMAs):
//       A: rows 0..63   reads A-a0 (8 b128) + B-b0, B-b1 (8)          B: rows 64..127   reads A-a1 (8), B kept in registers
//     phase = { ds_reads of the phase ; counted s_waitcnt vmcnt ; s_barrier ; s_setprio 1 ; 32 MFMA with the phase's LDS-DMA
//               pieces (regions of later K-tiles) issued between them ; s_setprio 0 ; s_barrier }.
//     Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while the other issues
//     its LDS reads and waits at the barrier -- the matrix pipe and the LDS path alternate by construction.  (A four-phase
//     version -- 16 MFMAs per phase, the guide's template -- was measured first: the 12 ds_read_b128 of its first phase take
//     longer than the partner's 16 MFMAs and every s_barrier costs ~55 cycles of latency, profiles/r02*_g256_stamps.)
//   * region schedule (regions are restaged as soon as they are dead):  phase A(t) <- A-a1(t+1), phase B(t) <- A-a0, B-b0,
//     B-b1 of K-tile t+2.  Hazards, with the DMA of a phase issued in its MFMA section (after the phase's first barrier):
//     WAR -- a region last read in phase p has been read by every wave before the second barrier of phase p of waves 4-7,
//     which precedes every MFMA section of phase p+1;  RAW -- a piece issued in phase q is waited for (counted vmcnt: only
//     the pieces of phase q+1 may be outstanding) before the first barrier of phase q+2 of its issuing wave and first read
//     in phase q+3.  One K-tile of MFMA time of lead.
//   * LDS-DMA = buffer_load_dwordx4 ... lds through a raw buffer descriptor: per-lane 32-bit offset (constant for the
//     whole kernel) + scalar offset per piece; out-of-range (tail) pieces return zeros without touching memory.
//   * bank swizzles, fragment reads, transposed reads for the N-contiguous operands: gemm_tiles.h (same as gemm.hip).
//   * epilogue straight from registers (operand-swapped products: a lane holds 4 consecutive columns of one row; pairs of
//     fragments trade halves with v_permlane16_swap for 16-byte bf16 stores).  With the GEGLU column map a wave's 64
//     columns are 32 value + 32 gate columns of the SAME 32 channels, so value * gelu(gate) needs no exchange at all.
//
// Applicability (launcher): M, N multiples of 256, K of 64, no 3x3 gather; everything else stays on gemm.hip.
#include "gemm_tiles.h"

namespace {

constexpr int REGION = 16384;          // 128 rows x 128 B (K-contiguous)  or  64 k-rows x 256 B (N-contiguous)
constexpr int BUFB = 4 * REGION;       // A-a0, A-a1, B-b0, B-b1
constexpr int SMEM256 = 2 * BUFB;      // 128 KiB
constexpr unsigned OOB = 0x80000000u;  // per-lane offset beyond num_records: the load returns zeros

template <int N>
struct IC { static constexpr int v = N; };

#ifndef G256_DIAG      // scratch diagnostics only (never defined in the product build): knock out one pipeline component
#define G256_DIAG 0    // bit 0: no MFMA, bit 1: no DMA in the main loop, bit 2: no LDS fragment reads, bit 3: no explicit lgkmcnt(0)
#endif                 // bit 4: s_memtime stamps of waves 0 and 4 of workgroup 0 (K-tiles 8..11) into g256_stamps
#if G256_DIAG & 16
__device__ unsigned long long g256_stamps[2 * 4 * 4 * 8];   // [wave half][K-tile 8..11][phase (2 used)][event]
#define STAMP(ev)                                                                                              \
  do {                                                                                                         \
    if (stamp_on && t >= 8 && t < 12) {                                                                        \
      unsigned long long ts_ = __builtin_amdgcn_s_memtime();                                                   \
      if (lane == 0) g256_stamps[((wr * 4 + (t - 8)) * 4 + ph_) * 8 + (ev)] = ts_;                             \
    }                                                                                                          \
  } while (0)
#else
#define STAMP(ev) do { } while (0)
#endif

template <int FORM, bool BIAS>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmP p) {
  constexpr bool A_KC = FORM != GEMM_TN;   // A tile K-contiguous (rows = output rows)
  constexpr bool B_KC = FORM == GEMM_NT;   // B tile K-contiguous (rows = output columns)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l16 = lane & 15, g = lane >> 4;
  int bx, by;
  // tail mode (1-D grid): workgroups [0, full) own whole tiles of the first tail_n0 tile columns, the others 128 x 256 HALF tiles of the
  // remaining columns -- their waves 4-7 (the tile's rows 128..255) neither read fragments nor multiply, they only stage data and
  // keep the barriers, so the active waves have their SIMD's matrix pipe to themselves and a K-step takes ~0.6 of a full tile's
  bool half = false;
  if (p.tail_n0 > 0) {
    const int gy = p.M / 256, full = p.tail_n0 * gy;
    const int id = blockIdx.x;
    if (id < full) {
      xcd_tile_map_id(p.xcd_px, id, p.tail_n0, gy, bx, by);
      by *= 2;                                          // (by counts 128-row halves below)
    } else {
      half = true;      // (full is a multiple of 8 whenever a band height is set: the launcher checks)
      if (p.xcd_bh > 0) xcd_seq_map_id(p.xcd_bh, id - full, p.N / 256 - p.tail_n0, 2 * gy, bx, by);
      else xcd_tile_map_id(0, id - full, p.N / 256 - p.tail_n0, 2 * gy, bx, by);
      bx += p.tail_n0;
    }
  } else {
    int bz;
    if (p.xcd_bh > 0) xcd_seq_map(p.xcd_bh, bx, by, bz);      // no XCD rectangle fits the grid (16 x 15 tiles): the generic order (one z slice here)
    else xcd_tile_map(p.xcd_px, bx, by);
    by *= 2;
  }
  const int n0 = bx * 256, m0 = by * 128;
  const bool idle = half && wr == 1;
  const bool gmap = p.geglu == 1;
  // first tile column of wave column wcx, quadrant half b (32 columns follow)
  auto cbase = [&](int wcx, int b) { return gmap ? 128 * (wcx >> 1) + 32 * (wcx & 1) + 64 * b : 64 * wcx + 32 * b; };

  // reduction range
  const int ktpt = p.K / 64;
  int split = 0, kt_begin = 0, kt_end = ktpt;
  if (FORM == GEMM_TN && p.splitk > 1) {
    split = blockIdx.z;
    const int chunk = (ktpt + p.splitk - 1) / p.splitk;
    kt_begin = split * chunk;
    kt_end = min(ktpt, kt_begin + chunk);
  }
  const int T = kt_end - kt_begin;

  // ---- LDS-DMA addressing: one per-lane byte offset per operand, scalar offsets per piece ----
  const i32x4 ra = make_srd(p.A, 0x7FFFFFFFu), rb = make_srd(p.B, 0x7FFFFFFFu);   // wave-uniform (kernel arguments only)
  const int lda = (int)p.lda, ldb = (int)p.ldb;
  const unsigned lds_base = lds_addr_of(smem);
  unsigned voA, voB;
  {
    const int kc_row = lane >> 3, kc_vec = (lane & 7) ^ kc_row;           // K-contiguous: 8 rows x 8 vectors per piece
    const int nc_row = lane >> 4;                                          // N-contiguous: 4 k-rows x 16 vectors per piece
    const int F = nc_row | (((wave >> 1) & 1) << 2);                       // swzF(4 (wave + 8h) + nc_row), same for h = 0, 1
    const int c = (((lane & 15) ^ (F << 1)) << 3);                         // logical column inside the 128-column region
    voA = A_KC ? (unsigned)(kc_row * lda + kc_vec * 8) * 2u : (unsigned)(nc_row * lda + 128 * (c >> 6) + (c & 63)) * 2u;
    voB = B_KC ? (unsigned)(kc_row * ldb + kc_vec * 8) * 2u : (unsigned)(nc_row * ldb + 64 * (c >> 5) + (c & 31)) * 2u;
  }
  // piece h (0 / 1) of this wave's share of region R (0: A-a0, 1: A-a1, 2: B-b0, 3: B-b1) of K-tile `tile` (relative to
  // kt_begin) -> buffer buf.  Tiles beyond the range become dummy loads (out-of-range offset: zeros, no memory traffic) so
  // that the vmcnt arithmetic stays uniform.
  auto piece = [&](auto REG, int h, int tile, int buf) {
    constexpr int R = decltype(REG)::v;
    constexpr int ab = R & 1;
    const bool live = tile < T && !(half && R < 2 && h == 1);      // (a half tile has no rows 128..255)
    if ((G256_DIAG & 2) && tile >= 2) return;
    const int k0 = (kt_begin + tile) * 64;
    const int q = wave + 8 * h;
    int so;
    if (R < 2) so = A_KC ? (m0 + 128 * h + 64 * ab + 8 * wave) * lda + k0 : (k0 + 4 * q) * lda + m0 + 64 * ab;
    else so = B_KC ? (n0 + cbase(2 * h + (wave >> 2), ab) + 8 * (wave & 3)) * ldb + k0 : (k0 + 4 * q) * ldb + n0 + 32 * ab;
    const unsigned dst = lds_base + buf * BUFB + R * REGION + q * 1024;   // (a dummy piece overwrites a region no K-tile will read again)
    const unsigned vo = live ? (R < 2 ? voA : voB) : OOB;
    // issued from asm (common.h): the builtin form makes hipcc drain every DMA in flight (vmcnt(0)) before the first
    // transpose read of each phase of the NN / TN forms
    lds_dma16_buffer(R < 2 ? ra : rb, vo, live ? (unsigned)(so * 2) : 0u, dst);
  };
  auto stage = [&](auto REG, int tile, int buf) {
    piece(REG, 0, tile, buf);
    piece(REG, 1, tile, buf);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient (TN): column sums of the A operand = A^T . ones on the matrix pipe, wave column 0 of the n-tile-0 workgroups
  const bool do_bias = BIAS && FORM == GEMM_TN && p.bias_grad != nullptr && bx == 0 && wc == 0;
  f32x4 accb[BIAS ? 8 : 1];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < (BIAS ? 8 : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  bf16x8 fa[4][2], fb[4][2];
  auto read_a = [&](int buf, auto AQ) {
    constexpr int a = decltype(AQ)::v;
    const char* R = smem + buf * BUFB + a * REGION;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (G256_DIAG & 4) fa[i][ks] = ones;
        else if (A_KC) fa[i][ks] = frag_kc<64>(R, 64 * wr + 16 * i + l16, ks * 4 + g);
        else fa[i][ks] = frag_nc<128>(R, ks * 32 + g * 8, 64 * wr + 16 * i, l16);
      }
  };
  auto read_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* R = smem + buf * BUFB + (2 + (j >> 1)) * REGION;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (G256_DIAG & 4) fb[j][ks] = ones;
        else if (B_KC) fb[j][ks] = frag_kc<64>(R, 32 * wc + 16 * (j & 1) + l16, ks * 4 + g);
        else fb[j][ks] = frag_nc<128>(R, ks * 32 + g * 8, 32 * wc + 16 * (j & 1), l16);
      }
    }
  };
  // 32 products of one accumulator half (64 rows x 64 columns of the wave tile); the phase's LDS-DMA pieces are issued
  // between the MFMAs (one after every four), where their issue cost hides under the matrix pipe
  auto mma = [&](auto AQ, int tile_next, auto IDLEC) {
    constexpr int a = decltype(AQ)::v;
    constexpr bool IDLE = decltype(IDLEC)::v != 0;
    const int nbuf = tile_next & 1;
    if (!(G256_DIAG & 8)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (G256_DIAG & 1) acc[4 * a + i][j][0] += (float)fa[i][ks][0] + (float)fb[j][ks][0];
          else if (!IDLE) acc[4 * a + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[4 * a + i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = ks * 4 + i;          // 8 slots of 4 MFMAs
        if (a == 0) {                         // phase A: A-a1 of the next K-tile
          if (slot == 2) piece(IC<1>{}, 0, tile_next, nbuf);
          if (slot == 5) piece(IC<1>{}, 1, tile_next, nbuf);
        } else {                              // phase B: A-a0, B-b0, B-b1 of the K-tile after the next
          if (slot == 0) piece(IC<0>{}, 0, tile_next, nbuf);
          if (slot == 1) piece(IC<0>{}, 1, tile_next, nbuf);
          if (slot == 2) piece(IC<2>{}, 0, tile_next, nbuf);
          if (slot == 3) piece(IC<2>{}, 1, tile_next, nbuf);
          if (slot == 4) piece(IC<3>{}, 0, tile_next, nbuf);
          if (slot == 5) piece(IC<3>{}, 1, tile_next, nbuf);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BIAS && do_bias && !IDLE) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          accb[BIAS ? 4 * a + i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i][ks], accb[BIAS ? 4 * a + i : 0], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto post = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: K-tile 0 complete; A-a0, B-b0, B-b1 of K-tile 1 in flight (complete at the first phase B) ----
  stage(IC<0>{}, 0, 0);
  stage(IC<2>{}, 0, 0);
  stage(IC<3>{}, 0, 0);
  stage(IC<1>{}, 0, 0);
  stage(IC<0>{}, 1, 1);
  stage(IC<2>{}, 1, 1);
  stage(IC<3>{}, 1, 1);
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind

#if G256_DIAG & 16
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0;
#endif
  if (!idle) {
  for (int t = 0; t < T; ++t) {
    const int cb = t & 1;
    int ph_ = 0;
    (void)ph_;
    // phase A: rows 0..63 of the wave tile x all 64 columns; DMA: A-a1 of K-tile t+1
    STAMP(0);
    read_a(cb, IC<0>{});
    read_b(cb);
    STAMP(1);
    wait_vmcnt<6>();                       // A-a1 of this K-tile (issued in phase A of K-tile t-1) has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    mma(IC<0>{}, t + 1, IC<0>{});
    STAMP(3);
    post();
    STAMP(4);
    // phase B: rows 64..127; DMA: A-a0, B-b0, B-b1 of K-tile t+2
    ph_ = 1;
    STAMP(0);
    read_a(cb, IC<1>{});
    STAMP(1);
    wait_vmcnt<2>();                       // A-a0, B-b0, B-b1 of K-tile t+1 (issued in phase B of K-tile t-1) have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    mma(IC<1>{}, t + 2, IC<0>{});
    STAMP(3);
    post();
    STAMP(4);
  }
  } else {
    // waves 4-7 of a half tile: the same DMA pieces, waits and barriers, no fragment reads, no products
    for (int t = 0; t < T; ++t) {
      wait_vmcnt<6>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma(IC<0>{}, t + 1, IC<1>{});
      post();
      wait_vmcnt<2>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mma(IC<1>{}, t + 2, IC<1>{});
      post();
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // both halves execute the same number of barriers
  wait_vmcnt<0>();                              // dummy tail pieces must not outlive the workgroup's LDS allocation

  if (idle) return;
  // ---- epilogue, registers -> global.  Lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile ----
  const int mrow = m0 + 128 * wr + l16;   // + 16 i
  if (FORM == GEMM_TN) {
    if (BIAS && do_bias && g == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) gemm_bias_out(p.bias_grad, p.slab, p.slab_ld, p.splitk, split, p.M, mrow + 16 * i, accb[BIAS ? i : 0][0]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long m = mrow + 16 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + cbase(wc, j >> 1) + 16 * (j & 1) + 4 * g;
        const f32x4 x = acc[i][j];
        if (p.splitk > 1) {
          *(f32x4*)(p.slab + ((long)split * p.M + m) * p.slab_ld + n) = x;
        } else {
          float* c = (float*)p.C + m * p.ldc + n;
          if (p.accumulate) {
            f32x4 a = *(f32x4*)c;
            a[0] += x[0]; a[1] += x[1]; a[2] += x[2]; a[3] += x[3];
            *(f32x4*)c = a;
          } else {
            *(f32x4*)c = x;
          }
        }
      }
    }
    return;
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // 24 wait states: MFMA results -> inline-asm VALU reads below
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long m = mrow + 16 * i;
    float xq[2][8];   // [quadrant half b][8 contiguous columns at cbase(wc, b) + 16 (g & 1) + 8 (g >> 1)]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lo = acc[i][2 * b][r], hi = acc[i][2 * b + 1][r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states: VALU write -> permlane read)
        xq[b][r] = lo;
        xq[b][4 + r] = hi;
      }
    const int cq = 16 * (g & 1) + 8 * (g >> 1);
    if (p.geglu == 1) {   // forward GEGLU: b = 0 value columns, b = 1 the gate columns of the same channels
      const int nv = n0 + cbase(wc, 0) + cq, ng = n0 + cbase(wc, 1) + cq;
      if (p.bias) {
        const bf16x8 bv = *(const bf16x8*)(p.bias + nv), bg = *(const bf16x8*)(p.bias + ng);
#pragma unroll
        for (int e = 0; e < 8; ++e) { xq[0][e] += (float)bv[e]; xq[1][e] += (float)bg[e]; }
      }
      bf16x8 ov, og, oa;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ov[e] = (bf16)xq[0][e];
        og[e] = (bf16)xq[1][e];
        oa[e] = (bf16)((float)ov[e] * gelu_f((float)og[e]));
      }
      *(bf16x8*)((bf16*)p.C + m * p.ldc + nv) = ov;
      *(bf16x8*)((bf16*)p.C + m * p.ldc + ng) = og;
      // channel of packed value column nv: (nv / 2G) * G + nv % G
      *(bf16x8*)(p.aux + m * p.ldaux + (nv / (2 * p.geglu_group)) * p.geglu_group + nv % p.geglu_group) = oa;
    } else if (p.geglu == 2) {   // dgrad of the second feed-forward projection: dG -> dU (value and gate halves)
      const int G = p.geglu_group;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + cbase(wc, b) + cq;
        const long cu = (long)(n / G) * (2 * G) + (n % G);
        const bf16x8 ua = *(const bf16x8*)(p.aux + m * p.ldaux + cu);
        const bf16x8 ut = *(const bf16x8*)(p.aux + m * p.ldaux + cu + G);
        bf16x8 oa, ot;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = (float)(bf16)xq[b][e], tv = (float)ut[e];
          float cdf, pdf;
          gelu_cdf_pdf(tv, &cdf, &pdf);
          oa[e] = (bf16)(dv * tv * cdf);
          ot[e] = (bf16)(dv * (float)ua[e] * fmaf(tv, pdf, cdf));
        }
        *(bf16x8*)((bf16*)p.C + m * p.ldc + cu) = oa;
        *(bf16x8*)((bf16*)p.C + m * p.ldc + cu + G) = ot;
      }
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + cbase(wc, b) + cq;
        float* x = xq[b];
        if (p.bias) {
          const bf16x8 bv = *(const bf16x8*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)bv[e];
        }
        if (p.rowvec) {
          const bf16x8 tv = *(const bf16x8*)(p.rowvec + (m / p.rows_per_batch) * p.ldv + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)tv[e];
        }
        if (p.resid) {
          const bf16x8 rv = *(const bf16x8*)(p.resid + m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)rv[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)x[e];
        *(bf16x8*)((bf16*)p.C + m * p.ldc + n) = o;
      }
    }
  }
}

template <int FORM, bool BIAS>
int launch256(const GemmP& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm256_kernel<FORM, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM256));
    attr_set = true;
  }
  dim3 grid(p.N / 256, p.M / 256, FORM == GEMM_TN ? p.splitk : 1);
  if (p.tail_n0 > 0) grid = dim3(p.tail_n0 * (p.M / 256) + 2 * (p.N / 256 - p.tail_n0) * (p.M / 256), 1, 1);
  hipLaunchKernelGGL((gemm256_kernel<FORM, BIAS>), grid, dim3(512), SMEM256, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// can this problem run on the 256 x 256 kernel?  (p as normalised by launch_gemm: accumulate folded into resid, etc.)
bool gemm256_applicable(const GemmP& p) {
  if (p.taps != 1) return false;
  if (p.M % 256 || p.N % 256 || p.K % 64) return false;
  if (p.lda % 8 || p.ldb % 8) return false;
  if (p.geglu == 1 && p.geglu_group != 64) return false;   // value | gate halves (32 + 32 columns) inside one wave's 64 columns
  if (p.geglu == 2 && p.geglu_group % 8) return false;
  // 32-bit buffer offsets
  const long abytes = 2 * (p.form == GEMM_TN ? (long)p.K * p.lda : (long)p.M * p.lda);
  const long bbytes = 2 * (p.form == GEMM_NT ? (long)p.N * p.ldb : (long)p.K * p.ldb);
  if (abytes >= (1L << 31) || bbytes >= (1L << 31)) return false;
  return true;
}

#if G256_DIAG & 16
extern "C" __attribute__((visibility("default"))) int sdxl_debug_g256_stamps(unsigned long long* out) {   // diagnostics build only
  HIP_CHECK_RET(hipDeviceSynchronize());
  HIP_CHECK_RET(hipMemcpyFromSymbol(out, HIP_SYMBOL(g256_stamps), sizeof(unsigned long long) * 2 * 4 * 4 * 8));
  return 0;
}
#endif

static bool g256_tail_enabled = true;
void gemm256_set_tail(bool on) { g256_tail_enabled = on; }
int launch_gemm256(const GemmP& pin, hipStream_t st) {
  ARG_CHECK(gemm256_applicable(pin), "gemm256: problem %dx%dx%d (form %d) does not fit the 256x256 kernel", pin.M, pin.N, pin.K, pin.form);
  GemmP p = pin;
  {   // px x (8/px) XCD grid over the (n, m) tile grid minimising the per-XCD operand footprint ~ N/px + M/py
    const int gx = p.N / 256, gy = p.M / 256;
    double best = 1e30;
    p.xcd_px = 0;
    for (int px = 1; px <= 8; px *= 2) {
      const int py = 8 / px;
      if (gx % px || gy % py) continue;
      const double cost = (double)p.N / px + (double)p.M / py;
      if (cost < best) { best = cost; p.xcd_px = px; }
    }
  }
  p.xcd_bh = 0;
  if (KNOB(34) != 1 && !(p.form == GEMM_TN && p.splitk > 1)) {
    // the generic compact order (gemm_tiles.h: xcd_seq_map) where no rectangle fits -- or fits badly: 16 x 15 tiles (the QKV projection) only
    // divide as 2 rows x 15 columns per XCD, 4 352 operand rows against 2 816 of a 5 x 6 block
    const int gx = p.N / 256, gy = p.M / 256;
    const int bh = xcd_band_rows(gx, gy, 1, 256, 256);
    const long per = (long)gx * gy / 8 > 0 ? (long)gx * gy / 8 : 1;
    const double seq_cost = 256.0 * (bh + (double)((per + bh - 1) / bh));
    const double rect_cost = p.xcd_px > 0 ? (double)p.N / p.xcd_px + (double)p.M / (8 / p.xcd_px) : 1e30;
    if (seq_cost < 0.85 * rect_cost) p.xcd_bh = bh;
  }
  // tail mode (NT): the tiles beyond the last full round of 256, when they are at most half a round and whole tile columns,
  // go out as twice as many half-height workgroups in the same launch (see the kernel)
  p.tail_n0 = 0;
  if (p.form == GEMM_NT && g256_tail_enabled) {
    const int gx = p.N / 256, gy = p.M / 256;
    const long T = (long)gx * gy, r = T % 256;
    if (T > 256 && r > 0 && r <= 128 && r % gy == 0) {
      p.tail_n0 = gx - (int)(r / gy);
      // XCD map of the full region: it must divide that region's grid
      p.xcd_px = 0;
      double best = 1e30;
      for (int px = 1; px <= 8; px *= 2) {
        const int py = 8 / px;
        if (p.tail_n0 % px || gy % py) continue;
        const double cost = 256.0 * p.tail_n0 / px + (double)p.M / py;
        if (cost < best) { best = cost; p.xcd_px = px; }
      }
      // the half-height workgroups of the tail: compact blocks per XCD too (they ran in identity order: every XCD fetched every panel)
      p.xcd_bh = 0;
      if (KNOB(34) != 1 && (p.tail_n0 * gy) % 8 == 0) p.xcd_bh = xcd_band_rows(gx - p.tail_n0, 2 * gy, 1, 128, 256);
    }
  }
  switch (p.form) {
    case GEMM_NT: return launch256<GEMM_NT, false>(p, st);
    case GEMM_NN: return launch256<GEMM_NN, false>(p, st);
    default: return p.bias_grad ? launch256<GEMM_TN, true>(p, st) : launch256<GEMM_TN, false>(p, st);
  }
}
