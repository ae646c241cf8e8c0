// Co-resident 256-row bf16 MFMA GEMM for gfx950: 256 x BN x 32 workgroup tile (BN = 160 or 128), 8 waves (4 x 2, wave tile
// 64 x BN/2), 3-deep LDS-DMA ring of 32-deep K-steps in <= 78 KiB, accumulators in registers under a 128-VGPR budget, so that TWO
// workgroups (of this kernel, or one of it beside any 4-wave kernel of <= 256 registers) share a CU.
//
// Why: the 128-row kernels of gemm.hip multiply 71 flops per byte they stage (128 x 160 tile) and every route tried on them
// converges at ~0.9 PFLOP/s because the L2 -> LDS stream of a CU delivers ~45-50 KB/us whatever the two co-resident workgroups
// are; the exclusive 256-row kernels (gemm256.hip, wgrad256.hip: 128-157 KiB, 190-250 registers) stage half
// the bytes per flop but evict the other stream of the two-stream backward from their CU.  This tile stages 98 (BN 160) / 85
// (BN 128) flops per byte and keeps the co-residency.  What the step uses it for (cr256_wgrad_cfg, measured in DESIGN.md section 12):
// the weight gradients (TN) of the 4096-token level's linear layers on the side stream, one 8-wave workgroup of this kernel beside one
// 4-wave dgrad workgroup of gemm.hip per CU -- the dgrad / weight-gradient pairs of a transformer block run 8-16 % faster.  The NT / NN
// forms are complete and parity-tested (sdxl_set_gemm_mode(4 * 31 / 32)) but not selected: with two 8-wave workgroups per CU the pair
// stages fewer bytes per microsecond than 4-wave + 8-wave, and N = 1280 outputs at M = 4096 are only 128-160 tiles of 256 rows.
//
// Structure (wgrad256.hip's, thinner and for all three forms):
//   * operand tiles of a K-step go global -> LDS by buffer_load_dwordx4 ... lds through raw buffer descriptors: one constant
//     per-lane 32-bit offset per piece (out-of-range rows / columns: an offset beyond num_records -> zeros, no memory traffic),
//     one running scalar offset per operand; 26 (BN 160) / 24 (BN 128) 1 KiB pieces per K-step = 3 per wave (+1 for waves 0, 1 at
//     BN 160: no padding pieces), issued between the MFMA groups of the step two ahead of them;
//   * counted s_waitcnt vmcnt + ONE raw s_barrier per K-step: step t is read after every wave has waited for its own pieces of
//     step t (those of t+1 may be outstanding); the pieces of t+2 overwrite the slot of t-1, whose reads precede the barrier;
//   * LDS images, bank swizzles and fragment reads (ds_read_b128 for K-contiguous tiles, ds_read_b64_tr_b16 for the N-contiguous
//     ones) are those of gemm_tiles.h; 20 (16) MFMAs v_mfma_f32_16x16x32_bf16 per wave and K-step behind 9 (8) fragment reads;
//   * products with the operands swapped (D^T layout): lane (l16, g) holds C[16 i + l16][16 j + 4 g .. + 3]; bf16 rows leave in
//     16-byte pieces after a v_permlane16_swap of neighbouring fragments, the fp32 weight-gradient tile straight from registers.
// Epilogues: bf16 (NT / NN): + bias + per-sample row vector + residual / accumulate, GEGLU backward (dG -> dU, any packing group);
// fp32 (TN): store / accumulate / split-K slab / bf16 emit (GemmP::Cb), bias gradient by a ones-MFMA (BN 128 only: the 16 extra
// accumulators do not fit beside 80), grouped launches (GemmP::group).
#include "gemm_tiles.h"

#include <type_traits>

namespace {

constexpr int CR_BM = 256, CR_BK = 32, CR_S = 3, CR_S_DEEP = 6;   // ring depth: 3 = the co-resident form (<= 78 KiB), 6 = an exclusive form (156 / 144 KiB, experiment)
constexpr unsigned CR_OOB = 0x80000000u;   // per-lane offset beyond num_records: the load returns zeros

#ifndef SDXL_CR_DIAG      // scratch diagnostics only (never defined in the product build): knock out one pipeline component
#define SDXL_CR_DIAG 0    // bit 0: no MFMA, bit 1: no DMA in the main loop, bit 2: no LDS fragment reads,
#endif                    // bit 3: every workgroup stages tile (0, 0) (all L2 hits after the first touch; results wrong)

template <int BN>
struct CrGeom {
  static constexpr int A_BYTES = CR_BM * CR_BK * 2;     // 16 KiB
  static constexpr int B_BYTES = BN * CR_BK * 2;        // 10 / 8 KiB
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int smem(int S) { return S * STAGE; }  // S = 3: 79 872 / 73 728 bytes
  static constexpr int NCB = B_BYTES / 1024;            // 10 / 8 pieces of the B tile
  static constexpr int NJ = BN / 32;                    // B fragments per wave
};

// PH: the K-step in two phases -- L: issue the DMA pieces of step t + 2, read all fragments of step t; M: the products -- with waves
//     4-7 running one barrier behind waves 0-3 (a SIMD holds one wave of each half): one wave of every SIMD multiplies while the other
//     issues its DMA pieces and fragment reads, instead of all eight doing each at the same time.  128-column tiles only (all nine
//     fragments of a step are live at once: 160 columns do not fit the 128 registers).
template <int FORM, int BN, bool BIASG, int S, bool PH = false>
__global__ __launch_bounds__(512, S != CR_S ? 2 : (PH && (BIASG || BN == 160)) ? 3 : 4) void cr256_kernel(const GemmP pin) {
  using G = CrGeom<BN>;
  constexpr bool A_KC = FORM != GEMM_TN;   // A tile K-contiguous (rows = output rows)
  constexpr bool B_KC = FORM == GEMM_NT;   // B tile K-contiguous (rows = output columns)
  constexpr int NJ = G::NJ;
  constexpr int dbg = SDXL_CR_DIAG;
  static_assert(!BIASG || (FORM == GEMM_TN && (BN == 128 || S > CR_S || PH)), "bias gradient: TN form; 128-column tiles in the lockstep co-resident form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* Ap = pin.A;
  const bf16* Bp = pin.B;
  void* Cp = pin.C;
  float* bias_grad = pin.bias_grad;
  bf16* Cb = pin.Cb;
  int bx, by, bz = blockIdx.z;
  if (pin.xcd_bh > 0) xcd_seq_map(pin.xcd_bh, bx, by, bz);
  else xcd_tile_map(pin.xcd_px, bx, by);
  if (FORM == GEMM_TN && pin.group > 1) {          // grouped launch: this workgroup's problem
    const int gi = bz;
    Ap = pin.gA[gi]; Bp = pin.gB[gi]; Cp = pin.gC[gi]; bias_grad = pin.gbias_grad[gi]; Cb = pin.gCb[gi];
  }
  set_wave_prio(pin.prio);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves, wave tile 64 x BN/2
  const int l16 = lane & 15, g = lane >> 4;
  const int n0 = (dbg & 8) ? 0 : bx * BN, m0 = (dbg & 8) ? 0 : by * CR_BM;
  const int M = pin.M, N = pin.N;
  const int lda = (int)pin.lda, ldb = (int)pin.ldb;

  // reduction range
  const int ktiles = pin.K / CR_BK;
  int split = 0, kt_begin = 0, kt_end = ktiles;
  if (pin.splitk > 1 && pin.group <= 1) {
    split = bz;
    const int chunk = (ktiles + pin.splitk - 1) / pin.splitk;
    kt_begin = split * chunk;
    kt_end = min(ktiles, kt_begin + chunk);
  }
  const int T = kt_end - kt_begin;

  // ---- LDS-DMA addressing: piece 0, 1 = A chunks wave, wave + 8; piece 2, 3 = B chunks wave, wave + 8 (BN 160: waves 0, 1) ----
  const i32x4 ra = make_srd(Ap, 0x7FFFFFFFu), rb = make_srd(Bp, 0x7FFFFFFFu);
  const unsigned lds_base = lds_addr_of(smem);
  constexpr bool HAS_B1 = G::NCB > 8;
  const bool b1 = HAS_B1 && wave + 8 < G::NCB;      // wave-uniform
  unsigned vo[4];                                   // constant per-lane byte offsets (CR_OOB: zeros)
  unsigned soA, soB;                                // running scalar byte offsets of the step being staged
  unsigned dA1, dB1;                                // scalar offset of piece 1 relative to piece 0
  unsigned stepA, stepB;
  {
    // K-contiguous [R][32] image: chunk c = rows 16 c .. 16 c + 15; lane -> row 16 c + lane / 4, physical vector lane % 4
    const int kc_row = lane >> 2, kc_lv = (lane & 3) ^ kc_swz<CR_BK>(kc_row);
    if (A_KC) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = m0 + 16 * (wave + 8 * j) + kc_row;
        vo[j] = row < M ? (unsigned)(kc_row * lda + kc_lv * 8) * 2u : CR_OOB;
      }
      soA = (unsigned)(((long)(m0 + 16 * wave) * lda + (long)kt_begin * CR_BK) * 2);
      dA1 = (unsigned)(128L * lda * 2);
      stepA = CR_BK * 2;
    } else {
      // two [32 k][128 m] N-contiguous sub-tiles: chunk c = sub-tile c >> 3, k-rows 4 (c & 7) ..; lane -> k-row + lane / 16, vector lane % 16
      const int krow = 4 * wave + (lane >> 4);
      const int lv = nc_logical<128>(krow, lane & 15);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + 128 * j + lv * 8;
        vo[j] = m < M ? (unsigned)((lane >> 4) * lda + lv * 8) * 2u : CR_OOB;
      }
      soA = (unsigned)((((long)kt_begin * CR_BK + 4 * wave) * lda + m0) * 2);
      dA1 = 128 * 2;
      stepA = (unsigned)((long)CR_BK * lda * 2);
    }
    if (B_KC) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = n0 + 16 * (wave + 8 * j) + kc_row;
        vo[2 + j] = (row < N && (j == 0 || b1)) ? (unsigned)(kc_row * ldb + kc_lv * 8) * 2u : CR_OOB;
      }
      soB = (unsigned)(((long)(n0 + 16 * wave) * ldb + (long)kt_begin * CR_BK) * 2);
      dB1 = (unsigned)(128L * ldb * 2);
      stepB = CR_BK * 2;
    } else {
      // [32 k][BN n] N-contiguous: chunk c = vectors 64 c .. 64 c + 63 of the [32][BN / 8] vector grid
      constexpr int V = BN / 8;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int krow = q / V, pv = q - krow * V;
        const int n = n0 + (nc_logical<BN>(krow, pv) << 3);
        vo[2 + j] = (n < N && (j == 0 || b1)) ? (unsigned)(krow * ldb + (n - n0)) * 2u : CR_OOB;
      }
      soB = (unsigned)(((long)kt_begin * CR_BK * ldb + n0) * 2);
      dB1 = 0;
      stepB = (unsigned)((long)CR_BK * ldb * 2);
    }
  }
  bool dma_on = true;
  // piece pc (0, 1: A; 2, 3: B) of the step being staged -> ring slot `slot`; live = false: a dummy load (zeros, no traffic)
  auto issue_piece = [&](int pc, int slot, bool live) {
    if (dbg && !dma_on) return;
    const unsigned At = lds_base + slot * G::STAGE, Bt = At + G::A_BYTES;
    if (pc == 0) lds_dma16_buffer(ra, live ? vo[0] : CR_OOB, live ? soA : 0u, At + wave * 1024);
    if (pc == 1) lds_dma16_buffer(ra, live ? vo[1] : CR_OOB, live ? soA + dA1 : 0u, At + (wave + 8) * 1024);
    if (pc == 2) lds_dma16_buffer(rb, live ? vo[2] : CR_OOB, live ? soB : 0u, Bt + wave * 1024);
    if (HAS_B1 && pc == 3 && b1) lds_dma16_buffer(rb, live ? vo[3] : CR_OOB, live ? soB + dB1 : 0u, Bt + (wave + 8) * 1024);
  };
  auto advance = [&]() { soA += stepA; soB += stepB; };
  auto stage = [&](int slot, bool live) {
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) issue_piece(pc, slot, live);
    advance();
  };

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient (TN): column sums of the A operand = A^T . ones on the matrix pipe, wave column 0 of the n-tile-0 workgroups
  const bool do_bias = BIASG && bias_grad != nullptr && bx == 0 && wn == 0;
  f32x4 accb[BIASG ? 4 : 1];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < (BIASG ? 4 : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

#pragma unroll
  for (int d = 0; d < S - 1; ++d) stage(d, d < T);
  if (dbg & 2) dma_on = false;
  int rd = 0, wr = S - 1;
  if (PH) {
    static_assert(!PH || S == 3, "phased loop: 3-deep ring");
    // Barrier sequence X0, X1, ...: half 0 (waves 0-3) runs L(t) in [X_2t, X_2t+1] and M(t) in [X_2t+1, X_2t+2], half 1 one barrier later.
    // RAW: a wave's pieces of step t + 1 are waited for (counted vmcnt: only those of t + 2, just issued, may be outstanding) at the end
    //      of its L(t), i.e. before X_2t+2 at the latest -- the barrier in front of the earliest L(t + 1).
    // WAR: the pieces of step t + 2 overwrite the slot of step t - 1, whose last reads (half 1's L(t - 1), completed by its lgkmcnt(0))
    //      precede X_2t, the barrier in front of the earliest L(t).
    const int half = wave >> 2;
    if (HAS_B1 && b1) wait_vmcnt<4>();
    else wait_vmcnt<3>();
    __builtin_amdgcn_s_barrier();
    if (half == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; ++t) {
      const bool live = t + 2 < T;
      const char* At = smem + rd * G::STAGE;
      const char* Bt = At + G::A_BYTES;
      // ---- L(t) ----
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) issue_piece(pc, wr, live);
      advance();
      bf16x8 fa[4], fb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (dbg & 4) fb[j] = ones;
        else if (B_KC) fb[j] = frag_kc<CR_BK>(Bt, wn * (BN / 2) + j * 16 + l16, g);
        else fb[j] = frag_nc<BN>(Bt, g * 8, wn * (BN / 2) + j * 16, l16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (dbg & 4) fa[i] = ones;
        else if (A_KC) fa[i] = frag_kc<CR_BK>(At, wm * 64 + i * 16 + l16, g);
        else fa[i] = frag_nc<128>(At + (wm >> 1) * (G::A_BYTES / 2), g * 8, (wm & 1) * 64 + i * 16, l16);
      }
      if (HAS_B1 && b1) wait_vmcnt<4>();
      else wait_vmcnt<3>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // ---- M(t) ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (BIASG && do_bias) accb[BIASG ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[BIASG ? i : 0], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (dbg & 1) acc[i][j][0] += (float)fa[i][0] + (float)fb[j][0];
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      rd = rd + 1 == S ? 0 : rd + 1;
      wr = wr + 1 == S ? 0 : wr + 1;
    }
    if (half == 0) __builtin_amdgcn_s_barrier();      // both halves execute the same number of barriers
  } else
  for (int t = 0; t < T; ++t) {
    // this wave's pieces of step t have landed (those of step t + 1 may be outstanding) ...
    if (HAS_B1 && b1) wait_vmcnt<4 * (S - 2)>();
    else wait_vmcnt<3 * (S - 2)>();
    __builtin_amdgcn_s_barrier();           // ... everyone's; and every wave is done reading slot `wr` (step t - 1)
    const bool live = t + S - 1 < T;
    const char* At = smem + rd * G::STAGE;
    const char* Bt = At + G::A_BYTES;
    // fragment reads: all of B and the first two A rows up front, A rows 2 and 3 behind the products of rows 0 and 1 (28 instead
    // of 36 fragment registers live: the 128-register budget)
    bf16x8 fa[4], fb[NJ];
    auto read_a = [&](int i) {
      if (dbg & 4) fa[i] = ones;
      else if (A_KC) fa[i] = frag_kc<CR_BK>(At, wm * 64 + i * 16 + l16, g);
      else fa[i] = frag_nc<128>(At + (wm >> 1) * (G::A_BYTES / 2), g * 8, (wm & 1) * 64 + i * 16, l16);
    };
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (dbg & 4) fb[j] = ones;
      else if (B_KC) fb[j] = frag_kc<CR_BK>(Bt, wn * (BN / 2) + j * 16 + l16, g);
      else fb[j] = frag_nc<BN>(Bt, g * 8, wn * (BN / 2) + j * 16, l16);
    }
    read_a(0);
    read_a(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BIASG && do_bias) accb[BIASG ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[BIASG ? i : 0], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (dbg & 1) acc[i][j][0] += (float)fa[i][0] + (float)fb[j][0];
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
      issue_piece(i, wr, live);             // one DMA piece of step t + 2 behind every group of NJ products
      if (i < 2) { read_a(i + 2); __builtin_amdgcn_sched_barrier(0); }
    }
    advance();
    rd = rd + 1 == S ? 0 : rd + 1;
    wr = wr + 1 == S ? 0 : wr + 1;
  }
  wait_vmcnt<0>();     // dummy tail pieces must not outlive the workgroup's LDS allocation

  // ---- epilogue, registers -> global ----
  if (FORM == GEMM_TN) {
    if (BIASG && do_bias && l16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + i * 16 + g * 4 + r;
          if (m < M) gemm_bias_out(bias_grad, pin.slab, pin.slab_ld, pin.splitk, split, M, m, accb[BIASG ? i : 0][r]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + l16;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
        if (n >= N) continue;
        f32x4 x = acc[i][j];
        float* c = (float*)Cp + (long)m * pin.ldc + n;
        if (pin.splitk > 1) {
          *(f32x4*)(pin.slab + ((long)split * M + m) * pin.slab_ld + n) = x;
        } else if (Cb) {
          if (pin.accumulate) {
            const f32x4 a = *(const f32x4*)c;
            x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3];
          }
          bf16x4 o;
          o[0] = (bf16)(x[0] * pin.cb_scale); o[1] = (bf16)(x[1] * pin.cb_scale);
          o[2] = (bf16)(x[2] * pin.cb_scale); o[3] = (bf16)(x[3] * pin.cb_scale);
          *(bf16x4*)(Cb + (long)m * pin.ldc + n) = o;
        } else if (pin.accumulate) {
          f32x4 a = *(f32x4*)c;
          a[0] += x[0]; a[1] += x[1]; a[2] += x[2]; a[3] += x[3];
          *(f32x4*)c = a;
        } else {
          *(f32x4*)c = x;
        }
      }
    }
    return;
  }
  // bf16 forms.  cnt = 8 or 4 contiguous columns starting at n of row m
  auto store_bf16 = [&](int m, int n, float (&x)[8], auto CNT) {
    constexpr int cnt = decltype(CNT)::value;
    typedef __attribute__((ext_vector_type(cnt))) __bf16 vec;
    if (m >= M || n >= N) return;
    if (pin.geglu == 2) {   // dgrad of the second feed-forward projection: dG -> dU (value and gate halves), see GemmP::geglu
      const int Gp = pin.geglu_group;
      const long cu = (long)(n / Gp) * (2 * Gp) + (n % Gp);
      const vec ua = *(const vec*)(pin.aux + (long)m * pin.ldaux + cu);
      const vec ut = *(const vec*)(pin.aux + (long)m * pin.ldaux + cu + Gp);
      vec oa, ot;
#pragma unroll
      for (int e = 0; e < cnt; ++e) {
        const float dv = (float)(bf16)x[e], tv = (float)ut[e];
        float cdf, pdf;
        gelu_cdf_pdf(tv, &cdf, &pdf);
        oa[e] = (bf16)(dv * tv * cdf);
        ot[e] = (bf16)(dv * (float)ua[e] * fmaf(tv, pdf, cdf));
      }
      *(vec*)((bf16*)Cp + (long)m * pin.ldc + cu) = oa;
      *(vec*)((bf16*)Cp + (long)m * pin.ldc + cu + Gp) = ot;
      return;
    }
    if (pin.bias) {
      const vec bv = *(const vec*)(pin.bias + n);
#pragma unroll
      for (int e = 0; e < cnt; ++e) x[e] += (float)bv[e];
    }
    if (pin.rowvec) {
      const vec tv = *(const vec*)(pin.rowvec + (long)(m / pin.rows_per_batch) * pin.ldv + n);
#pragma unroll
      for (int e = 0; e < cnt; ++e) x[e] += (float)tv[e];
    }
    if (pin.resid) {
      const vec rv = *(const vec*)(pin.resid + (long)m * pin.ldr + n);
#pragma unroll
      for (int e = 0; e < cnt; ++e) x[e] += (float)rv[e];
    }
    vec o;
#pragma unroll
    for (int e = 0; e < cnt; ++e) o[e] = (bf16)x[e];
    *(vec*)((bf16*)Cp + (long)m * pin.ldc + n) = o;
  };
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // 24 wait states: MFMA results -> inline-asm VALU reads below
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l16;
#pragma unroll
    for (int j = 0; j + 1 < NJ; j += 2) {
      float x[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lo = acc[i][j][r], hi = acc[i][j + 1][r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states: VALU write -> permlane read)
        x[r] = lo;
        x[4 + r] = hi;
      }
      store_bf16(m, n0 + wn * (BN / 2) + (j + (g & 1)) * 16 + (g >> 1) * 8, x, std::integral_constant<int, 8>{});
    }
    if (NJ & 1) {   // odd fragment count (BN = 160): the last fragment goes out in 8-byte pieces
      float x[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = acc[i][NJ - 1][r];
      store_bf16(m, n0 + wn * (BN / 2) + (NJ - 1) * 16 + g * 4, x, std::integral_constant<int, 4>{});
    }
  }
}

template <int FORM, int BN, bool BIASG, int S = CR_S, bool PH = false>
int launch_cr(const GemmP& p, hipStream_t st) {
  using G = CrGeom<BN>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)cr256_kernel<FORM, BN, BIASG, S, PH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      G::smem(S) > 100 * 1024 ? G::smem(S) : 100 * 1024));
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, CR_BM), p.group > 1 ? p.group : p.splitk);
  // The co-resident form asks for 84 KiB of LDS, not the 72 - 78 KiB its ring needs: ONE of these workgroups per CU, and 76 KiB + half the
  // registers stay free for a workgroup of the other stream (the 73 KiB dgrad kernel, the norms, attention).  With the ring's own size two of
  // them fit a CU and fill its register file: the caller's stream -- the backward's critical path -- then finds no room there until one of
  // them retires (step -0.1 ... -0.55 ms over four boxes, profiles/r06q_ab_cr_one_per_cu.txt).  knob 38 = 1: the ring's own size, N: N KiB.
  const int kib = KNOB(38) == 0 ? (S == CR_S ? 84 : 0) : KNOB(38) == 1 ? 0 : KNOB(38);
  const int smem = (kib * 1024 > G::smem(S) && kib <= 100) ? kib * 1024 : G::smem(S);
  GEMM_LAUNCH((cr256_kernel<FORM, BN, BIASG, S, PH>), grid, dim3(512), smem, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// can this problem run on the co-resident 256-row kernel?  (p as normalised by launch_gemm: accumulate folded into resid, etc.)
bool cr256_applicable(const GemmP& p) {
  if (p.taps != 1 || p.up2) return false;
  if (p.K % CR_BK || p.M % 8 || p.N % 8 || p.lda % 8 || p.ldb % 8) return false;
  if (p.form == GEMM_TN) {
    if (!p.out_f32 || p.ldc % 4) return false;
  } else {
    if (p.out_f32 || p.splitk > 1 || p.geglu == 1 || p.ldc % 8) return false;
    if (p.geglu == 2 && p.geglu_group % 8) return false;
  }
  // 32-bit buffer offsets
  const long abytes = 2 * (p.form == GEMM_TN ? (long)p.K * p.lda : (long)p.M * p.lda);
  const long bbytes = 2 * (p.form == GEMM_NT ? (long)p.N * p.ldb : (long)p.K * p.ldb);
  if (abytes >= (1L << 31) || bbytes >= (1L << 31)) return false;
  return true;
}

// The plan's choice for the linear weight gradients of the 4096-token level (reduction over 2048 .. 8192 rows; longer reductions go to
// wgrad256.hip): beside the 4-wave dgrad kernels of the caller's stream one 8-wave workgroup of this kernel per CU stages 85-98 flops
// per byte where the 128 x 160 kernel stages 71 -- the dgrad / weight-gradient pairs of a level-2 transformer block run 8-16 % faster
// (profiles/r04a_pair_bench.txt).  A bias gradient forces 128-column tiles (its accumulators).
static int cr256_wgrad_cfg_tiles(int M, int N, long red, bool bias) {
  if (KNOB(16) == 1) return 0;
  if (red < 2048 || red > (KNOB(19) == 1 ? 32768 : 8192) || red % CR_BK || M % 8 || N % 8) return 0;      // (knob 19 = 1, experiment: the 16 384-row level too)
  if ((long)M * N < (red > 8192 ? 640L * 640L : 1280L * 1280L)) return 0;
  if (KNOB(16) == 2) return 32;
  if (KNOB(16) == 3) return (bias || N % 160) ? 32 : 31;
  if (bias || N % 160) return 32;
  // no bias: 160-column tiles for the small outputs that go out grouped (1280 x 1280 three at a time: 120 workgroups), 128-column
  // tiles where those give >= 150 workgroups (3840 x 1280: 150 against 120)
  return (long)cdiv(M, CR_BM) * (N / 160) >= 100 ? 32 : 31;
}
int cr256_wgrad_cfg(int M, int N, long red, bool bias) {
  int cfg = cr256_wgrad_cfg_tiles(M, N, red, bias);
  // 128-column tiles: the LOCKSTEP loop (configuration 32) since the end of round 5 -- same-box, six alternations: 107.24 -> 106.54 ms per step against the
  // phased loop (configuration 35), which had won by 2 ms in round 4: beside the exclusive one-wave-per-SIMD dgrads (gemm_pl.hip) and with its operands
  // fetched once per XCD the phased form's barrier pairs no longer pay (profiles/r05y_ab_cr256_lockstep.txt).  knob 23 = 5: phased everywhere (rounds 4-5),
  // = 2: phased only for >= 256 tiles (106.85 against 106.53).
  if (cfg == 32 && (KNOB(23) == 5 || KNOB(23) == 3 || KNOB(23) == 4 || (KNOB(23) == 2 && (long)cdiv(M, CR_BM) * cdiv(N, 128) >= 256))) cfg = 35;
  // (knob 23 = 3 / 4, experiment: the phased loop on 160-column tiles for outputs of >= 8192 rows / wherever N % 160 == 0)
  if (cfg == 35 && N % 160 == 0 && ((KNOB(23) == 3 && M >= 8192) || KNOB(23) == 4)) cfg = 36;
  return cfg;
}

// split-K factor of a weight gradient on this kernel: whole reductions at the 4096-token level (its tiles fill half the chip beside the
// dgrad); longer reductions with few tiles are cut to ~`target` workgroups of >= 64 K-steps
int cr256_pick_splitk(int M, int N, long red, int cfg) {
  if (red <= 8192 && KNOB(17) <= 0) return 1;      // (knob 17 > 0, experiment: the 4096-row level split to ~that many workgroups too)
  const long tiles = (long)cdiv(M, CR_BM) * cdiv(N, (cfg == 31 || cfg == 33 || cfg == 36) ? 160 : 128);
  const long target = KNOB(17) > 0 ? KNOB(17) : 256;
  long s = (target + tiles / 2) / tiles;
  if (s < 1) s = 1;
  while (s > 1 && red / CR_BK / s < (red <= 8192 ? 32 : 64)) --s;
  return (int)(s > 32 ? 32 : s);
}

// bn = 160 or 128 (0: 160 where N divides and no bias gradient is asked for, else 128)
int launch_cr256(const GemmP& pin, int bn, hipStream_t st, bool deep, bool phased) {
  ARG_CHECK(cr256_applicable(pin), "cr256: problem %dx%dx%d (form %d) does not fit the co-resident 256-row kernel", pin.M, pin.N, pin.K, pin.form);
  GemmP p = pin;
  bool biasg = p.form == GEMM_TN && p.bias_grad != nullptr;
  if (p.form == GEMM_TN && p.group > 1)
    for (int i = 0; i < p.group; ++i) biasg = biasg || p.gbias_grad[i] != nullptr;
  if (bn == 0) bn = (p.N % 160 == 0 && !biasg) ? 160 : 128;
  if (biasg && !deep && !phased) bn = 128;
  {   // px x (8/px) XCD grid over the (n, m) tile grid minimising the per-XCD operand footprint ~ N/px + M/py
    const int gx = cdiv(p.N, bn), gy = cdiv(p.M, CR_BM);
    double best = 1e30;
    p.xcd_px = 0;
    for (int px = 1; px <= 8; px *= 2) {
      const int py = 8 / px;
      if (gx % px || gy % py) continue;
      const double cost = (double)p.N / px + (double)p.M / py;
      if (cost < best) { best = cost; p.xcd_px = px; }
    }
    // no rectangle fits (15 x 10 tiles, grouped 5 x 10 ...): the generic order (knob 34 = 1: identity as before, = 2: generic everywhere)
    const int gz = p.form == GEMM_TN && p.group > 1 ? p.group : (p.splitk > 1 ? p.splitk : 1);
    p.xcd_bh = 0;
    if (KNOB(34) != 1 && (p.xcd_px == 0 || KNOB(34) == 2)) p.xcd_bh = xcd_band_rows(gx, gy, gz, CR_BM, bn);
  }
  if (p.splitk > 1) p.slab_ld = p.N;
#ifdef SDXL_DIAG      // the exclusive 6-deep form (forced configurations 33 / 34): experiment, diagnostics build only
  if (deep) {
    switch (p.form) {
      case GEMM_NT: return bn == 160 ? launch_cr<GEMM_NT, 160, false, CR_S_DEEP>(p, st) : launch_cr<GEMM_NT, 128, false, CR_S_DEEP>(p, st);
      case GEMM_NN: return bn == 160 ? launch_cr<GEMM_NN, 160, false, CR_S_DEEP>(p, st) : launch_cr<GEMM_NN, 128, false, CR_S_DEEP>(p, st);
      default:
        if (biasg) return bn == 160 ? launch_cr<GEMM_TN, 160, true, CR_S_DEEP>(p, st) : launch_cr<GEMM_TN, 128, true, CR_S_DEEP>(p, st);
        return bn == 160 ? launch_cr<GEMM_TN, 160, false, CR_S_DEEP>(p, st) : launch_cr<GEMM_TN, 128, false, CR_S_DEEP>(p, st);
    }
  }
#endif
  if (phased && bn == 128) {
    switch (p.form) {
      case GEMM_NT: return launch_cr<GEMM_NT, 128, false, CR_S, true>(p, st);
      case GEMM_NN: return launch_cr<GEMM_NN, 128, false, CR_S, true>(p, st);
      default: return biasg ? launch_cr<GEMM_TN, 128, true, CR_S, true>(p, st) : launch_cr<GEMM_TN, 128, false, CR_S, true>(p, st);
    }
  }
  if (phased && p.form == GEMM_TN)      // 160-column tiles, phased: the weight-gradient form only (up to 168 registers: three waves per SIMD)
    return biasg ? launch_cr<GEMM_TN, 160, true, CR_S, true>(p, st) : launch_cr<GEMM_TN, 160, false, CR_S, true>(p, st);
  switch (p.form) {
    case GEMM_NT: return bn == 160 ? launch_cr<GEMM_NT, 160, false>(p, st) : launch_cr<GEMM_NT, 128, false>(p, st);
    case GEMM_NN: return bn == 160 ? launch_cr<GEMM_NN, 160, false>(p, st) : launch_cr<GEMM_NN, 128, false>(p, st);
    default:
      if (biasg) return launch_cr<GEMM_TN, 128, true>(p, st);
      return bn == 160 ? launch_cr<GEMM_TN, 160, false>(p, st) : launch_cr<GEMM_TN, 128, false>(p, st);
  }
}
