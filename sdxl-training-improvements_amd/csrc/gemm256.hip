// bf16 MFMA GEMM for gfx950, 256 x 256 x 64 workgroup tile, 8 waves (2 x 4), wave tile 128 x 64, 8-phase schedule.
//
// Why a second kernel: with 128-row tiles (gemm.hip) every MFMA flop pulls (1/128 + 1/BN) operand elements through the
// L2 -> LDS path, which saturates at ~39 B/clk/CU on this chip -- that stream, not the matrix pipe, set the pace
// (profiles/r01f, r01m: 2.8x the algorithmic bytes fetched).  A 256 x 256 tile halves those bytes per flop and a
// 128 x 64 wave tile needs 0.375 LDS fragment reads per MFMA instead of 0.5, so the LDS read phase of one wave group
// fits inside the MFMA phase of the other.
//
// Structure (per workgroup, one per CU, 129 KiB LDS):
//   * the A tile [256][64] and B tile [256][64] of a K-step live in LDS as four 16 KiB REGIONS: A-a0 / A-a1 (the first /
//     second 64 rows of both wave rows), B-b0 / B-b1 (the first / second 32 columns of all four wave columns); two
//     buffers (even / odd K-tile).
//   * a K-tile is multiplied in four PHASES, one 64 x 32 accumulator quadrant of every wave each (16 MFMAs):
//       P1 (a0,b0)  P2 (a0,b1)  P3 (a1,b1)  P4 (a1,b0)        reads: P1 A-a0 + B-b0, P2 B-b1, P3 A-a1, P4 none
//     phase = { ds_reads of the phase ; 2 LDS-DMA pieces (one region of a later K-tile) ; s_waitcnt vmcnt(8) ; s_barrier ;
//               lgkmcnt(0) ; s_setprio 1 ; 16 MFMA ; s_setprio 0 ; s_barrier }.
//     Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while the other issues
//     its LDS reads / DMA -- the matrix pipe and the LDS path alternate by construction.
//   * region schedule (regions are restaged as soon as they are dead):  P1(t) <- B-b1(t+1), P2(t) <- A-a1(t+1),
//     P3(t) <- A-a0(t+2), P4(t) <- B-b0(t+2).  Hazards, with global barrier b_2p-2 = first barrier of phase p for waves
//     0-3 (b_2p-1 for waves 4-7):  WAR -- a region last read in phase p has been read by every wave before b_2p, the
//     earliest DMA of phase p+2 is issued after b_2p+1;  RAW -- a piece issued in phase q is complete for its issuing wave
//     after the vmcnt(8) of phase q+4 (2 pieces per phase, in-order return), i.e. before b_2(q+4)-1 for every wave, and
//     is first read in phase q+5 or later (after b_2(q+4)-1).  Four phases (= one K-tile of MFMA time) of lead.
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
constexpr int PAD_OFF = 2 * BUFB;      // 1 KiB: destination of the tail's dummy pieces
constexpr int SMEM256 = 2 * BUFB + 1024;
constexpr unsigned OOB = 0x80000000u;  // per-lane offset beyond num_records: the load returns zeros

template <int N>
struct IC { static constexpr int v = N; };

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
  xcd_tile_map(p.xcd_px, bx, by);
  const int n0 = bx * 256, m0 = by * 256;
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
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFFF, 0x00020000);
  const int lda = (int)p.lda, ldb = (int)p.ldb;
  unsigned voA, voB;
  {
    const int kc_row = lane >> 3, kc_vec = (lane & 7) ^ kc_row;           // K-contiguous: 8 rows x 8 vectors per piece
    const int nc_row = lane >> 4;                                          // N-contiguous: 4 k-rows x 16 vectors per piece
    const int F = nc_row | (((wave >> 1) & 1) << 2);                       // swzF(4 (wave + 8h) + nc_row), same for h = 0, 1
    const int c = (((lane & 15) ^ (F << 1)) << 3);                         // logical column inside the 128-column region
    voA = A_KC ? (unsigned)(kc_row * lda + kc_vec * 8) * 2u : (unsigned)(nc_row * lda + 128 * (c >> 6) + (c & 63)) * 2u;
    voB = B_KC ? (unsigned)(kc_row * ldb + kc_vec * 8) * 2u : (unsigned)(nc_row * ldb + 64 * (c >> 5) + (c & 31)) * 2u;
  }
  // region R (0: A-a0, 1: A-a1, 2: B-b0, 3: B-b1) of K-tile `tile` (relative to kt_begin) -> buffer buf; 2 pieces per wave
  auto stage = [&](auto REG, int tile, int buf) {
    constexpr int R = decltype(REG)::v;
    constexpr int ab = R & 1;
    const bool live = tile < T;
    const int k0 = (kt_begin + tile) * 64;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = wave + 8 * h;
      int so;
      if (R < 2) so = A_KC ? (m0 + 128 * h + 64 * ab + 8 * wave) * lda + k0 : (k0 + 4 * q) * lda + m0 + 64 * ab;
      else so = B_KC ? (n0 + cbase(2 * h + (wave >> 2), ab) + 8 * (wave & 3)) * ldb + k0 : (k0 + 4 * q) * ldb + n0 + 32 * ab;
      char* dst = live ? smem + buf * BUFB + R * REGION + q * 1024 : smem + PAD_OFF;
      const unsigned vo = live ? (R < 2 ? voA : voB) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(R < 2 ? ra : rb, (lds_void*)dst, 16, (int)vo, live ? so * 2 : 0, 0, 0);
    }
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

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  auto read_a = [&](int buf, auto AQ) {
    constexpr int a = decltype(AQ)::v;
    const char* R = smem + buf * BUFB + a * REGION;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (A_KC) fa[i][ks] = frag_kc<64>(R, 64 * wr + 16 * i + l16, ks * 4 + g);
        else fa[i][ks] = frag_nc<128>(R, ks * 32 + g * 8, 64 * wr + 16 * i, l16);
      }
  };
  auto read_b = [&](int buf, auto BQ) {
    constexpr int b = decltype(BQ)::v;
    const char* R = smem + buf * BUFB + (2 + b) * REGION;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 f;
        if (B_KC) f = frag_kc<64>(R, 32 * wc + 16 * j + l16, ks * 4 + g);
        else f = frag_nc<128>(R, ks * 32 + g * 8, 32 * wc + 16 * j, l16);
        if (b == 0) fb0[j][ks] = f; else fb1[j][ks] = f;
      }
  };
  auto mma = [&](auto AQ, auto BQ) {
    constexpr int a = decltype(AQ)::v, b = decltype(BQ)::v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * a + i][2 * b + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b == 0 ? fb0[j][ks] : fb1[j][ks], fa[i][ks],
                                                                              acc[4 * a + i][2 * b + j], 0, 0, 0);
      if (BIAS && b == 1 && do_bias) {   // once per A quadrant (phases P2 and P3)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          accb[BIAS ? 4 * a + i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i][ks], accb[BIAS ? 4 * a + i : 0], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // phase boundaries
  auto pre = [&]() {   // reads + DMA of the phase are issued: counted DMA wait, then the first barrier of the phase
    wait_vmcnt<8>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto post = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: K-tile 0 complete, A-a0 / B-b0 of K-tile 1 in flight ----
  stage(IC<0>{}, 0, 0);
  stage(IC<2>{}, 0, 0);
  stage(IC<3>{}, 0, 0);
  stage(IC<1>{}, 0, 0);
  stage(IC<0>{}, 1, 1);
  stage(IC<2>{}, 1, 1);
  wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind

  for (int t = 0; t < T; ++t) {
    const int cb = t & 1, nb = cb ^ 1;
    // P1: quadrant (a0, b0)
    read_a(cb, IC<0>{});
    read_b(cb, IC<0>{});
    stage(IC<3>{}, t + 1, nb);
    pre();
    mma(IC<0>{}, IC<0>{});
    post();
    // P2: (a0, b1)
    read_b(cb, IC<1>{});
    stage(IC<1>{}, t + 1, nb);
    pre();
    mma(IC<0>{}, IC<1>{});
    post();
    // P3: (a1, b1)
    read_a(cb, IC<1>{});
    stage(IC<0>{}, t + 2, cb);
    pre();
    mma(IC<1>{}, IC<1>{});
    post();
    // P4: (a1, b0)
    stage(IC<2>{}, t + 2, cb);
    pre();
    mma(IC<1>{}, IC<0>{});
    post();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // both halves execute the same number of barriers
  wait_vmcnt<0>();                              // dummy tail pieces must not outlive the workgroup's LDS allocation

  // ---- epilogue, registers -> global.  Lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile ----
  const int mrow = m0 + 128 * wr + l16;   // + 16 i
  if (FORM == GEMM_TN) {
    if (BIAS && do_bias && g == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(p.bias_grad + mrow + 16 * i, accb[BIAS ? i : 0][0]);
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
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
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
  if (p.geglu == 1 && (p.geglu_group != 64 && p.geglu_group != 128)) return false;   // value | gate halves inside one wave's 64 columns
  if (p.geglu == 2 && p.geglu_group % 8) return false;
  // 32-bit buffer offsets
  const long abytes = 2 * (p.form == GEMM_TN ? (long)p.K * p.lda : (long)p.M * p.lda);
  const long bbytes = 2 * (p.form == GEMM_NT ? (long)p.N * p.ldb : (long)p.K * p.ldb);
  if (abytes >= (1L << 31) || bbytes >= (1L << 31)) return false;
  return true;
}

int launch_gemm256(const GemmP& p, hipStream_t st) {
  ARG_CHECK(gemm256_applicable(p), "gemm256: problem %dx%dx%d (form %d) does not fit the 256x256 kernel", p.M, p.N, p.K, p.form);
  switch (p.form) {
    case GEMM_NT: return launch256<GEMM_NT, false>(p, st);
    case GEMM_NN: return launch256<GEMM_NN, false>(p, st);
    default: return p.bias_grad ? launch256<GEMM_TN, true>(p, st) : launch256<GEMM_TN, false>(p, st);
  }
}
