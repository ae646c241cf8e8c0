// Software-pipelined bf16 MFMA GEMM for gfx950, ONE wave per SIMD: 128 x BN x 64 workgroup tile (BN = 160 or 128), 4 waves (2 x 2, wave
// tile 64 x BN/2), 4-deep LDS-DMA ring (144 / 128 KiB: the workgroup owns its CU), linear NT / NN problems (forward projections, dgrads).
//
// Why (profiles/r05a_*): the lockstep kernels of gemm.hip spend a K-step's three resources one after the other -- the 8-wave 4-deep
// configuration takes 0.75-0.83 us per 64-deep step of a 128 x 160 tile for 0.30 us of products, 0.27 us of L1 -> LDS transfer (64 B/clk: the
// CU's hard floor, profiles/r05a_stage_rate2.txt) and 0.2-0.3 us of fragment reads; two waves per SIMD alternating (configuration 23, the
// co-resident pairs of the backward) recover part of it.  Here ONE wave keeps all three busy: the fragments of the NEXT 32-deep half-step
// are read into a second register set, and the DMA pieces of the step three ahead are issued, BETWEEN the MFMAs of the current half-step --
// at most one LDS read and one DMA piece per MFMA, so that every non-matrix instruction issues in the shadow of a 16-cycle MFMA instead of
// in a block of its own (a wave issues in order: 25 instructions between two groups of MFMAs are 100 cycles of idle matrix pipe --
// the first form of this loop, gemm.hip configuration 5, measured 0.60 us per step).  In cycles the fillers are then nearly free
// (profiles/r05l_mfma_shadow.txt: 676 -> 728 per K-step); in time they are not: the package clocks down under the added LDS / L1 traffic.  The instruction count is cut to fit those shadows:
// buffer_load ... lds through raw descriptors (one constant per-lane offset per piece, one running scalar offset per operand, the LDS
// destination added straight into M0: 3 instructions per piece instead of 12), out-of-range rows / columns and the pieces beyond the last
// K-step are offsets beyond num_records (zeros, no traffic, no select in the live loop).
//
// Schedule (per K-step t; f0 / f1 = the fragment registers of the two 32-deep halves; S = 4 ring slots):
//   phase 0:  MFMAs of (t, half 0) from f0  |  reads of (t, half 1) -> f1      |  first pieces of step t + 3 -> slot (t + 3) % 4
//   -- s_waitcnt vmcnt(NL + NP0): my pieces of step t + 1 have landed; lgkmcnt(0); ONE s_barrier --
//   phase 1:  MFMAs of (t, half 1) from f1  |  reads of (t + 1, half 0) -> f0  |  the other pieces of step t + 3
// Hazards: RAW -- step t + 1 is first read in phase 1 of step t, behind the barrier every wave enters after the counted wait that retires
// its own pieces of that step (pieces land in issue order; outstanding at the wait, oldest first: step t + 1, step t + 2, the NP0 pieces
// of phase 0).  WAR -- the pieces of step t + 3 overwrite the slot of step t - 1, whose last reads (half 1) were issued in phase 0 of step
// t - 1, before that step's barrier, and consumed by the MFMAs of its phase 1: the restaging wave has passed that barrier and a whole
// phase of its own since (the guide's rule: restage a buffer >= 2 phases after its last ds_read), and its pieces land ~1 us after issue.
// The reads of the step after the last one are not skipped: their slot holds zero-fill pieces retired by the same counted wait.
// LDS read addresses: a per-lane base per half-step (loop-invariant) + the slot's scalar base, one VALU per operand and phase; fragment
// rows are immediates.
//
// LDS images, swizzles and fragment reads: gemm_tiles.h (the same as gemm.hip: K-contiguous tiles by ds_read_b128, the N-contiguous
// weight tile of the NN form by ds_read_b64_tr_b16).  Products with the operands swapped (D^T layout): lane (l16, g) holds
// C[16 i + l16][16 j + 4 g .. + 3]; bf16 rows leave in 16-byte pieces after a v_permlane16_swap of neighbouring fragments.
// Epilogue: + bias + residual / accumulate (bf16), operands requested before the products drain.  Everything else (3 x 3 gather, GEGLU, split-K, the Delta
// epilogue, the TN form) stays on gemm.hip / gemm256.hip / gemm_cr256.hip.
#include "gemm_tiles.h"

#include <type_traits>

namespace {

constexpr int PL_BM = 128, PL_BK = 64, PL_S = 4;
constexpr unsigned PL_OOB = 0x80000000u;   // per-lane offset beyond num_records: the load returns zeros

#ifndef SDXL_PL_DIAG      // scratch diagnostics only (never defined in the product build): knock out one pipeline component
#define SDXL_PL_DIAG 0    // bit 0: no MFMA, bit 1: no DMA in the main loop, bit 2: no LDS fragment reads,
#endif                    // bit 3: every workgroup stages tile (0, 0) (all L2 hits after the first touch; results wrong)

template <int BN>
struct PlGeom {
  static constexpr int A_BYTES = PL_BM * PL_BK * 2;     // 16 KiB
  static constexpr int B_BYTES = BN * PL_BK * 2;        // 20 / 16 KiB
  static constexpr int STAGE = A_BYTES + B_BYTES;       // 36 / 32 KiB
  static constexpr int SMEM = PL_S * STAGE;             // 144 / 128 KiB
  static constexpr int ACH = A_BYTES / 1024 / 4;        // 4 pieces of the A tile per wave
  static constexpr int BCH = B_BYTES / 1024 / 4;        // 5 / 4 of the B tile
  static constexpr int NL = ACH + BCH;                  // LDS-DMA instructions per wave and K-step
  static constexpr int NP0 = (NL + 1) / 2;              // ... of them in phase 0
  static constexpr int NJ = BN / 32;                    // B fragments per wave
  static constexpr int MI = 4;                          // A fragments per wave
  static constexpr int NMF = MI * NJ;                   // MFMAs per wave and half-step
};

// one LDS-DMA piece: M0 = slot base + constant, then the load (the s_nop: M0 write -> LDS-DMA needs one wait state).  M0 is written and
// read inside this one statement and NOT restored: it is declared clobbered, so a value hipcc might ever keep in M0 (s_movrel-style indexing
// after some future edit) is not silently lost (hipcc warns that m0 is a reserved register: that warning is the point, -Wno-inline-asm is
// not set).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "clobber list contains reserved registers: m0" -- intended, see above
template <int OFF>
__device__ __forceinline__ void pl_dma(i32x4 srd, unsigned voff, unsigned soff, unsigned slot_base) {
  asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :: "v"(voff), "s"(srd), "s"(soff), "s"(slot_base), "n"(OFF) : "memory", "scc", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ bf16x8 pl_z8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
  return z;
}
template <int NJ>
struct PlFrag { bf16x8 a[4], b[NJ]; };

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{}), in order: a loop whose index is a constant expression in the body
template <typename F, int... I>
__device__ __forceinline__ void pl_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void pl_static_for(F&& f) { pl_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// PF: a FIFTH wave that multiplies nothing: it touches, PL_PF_LEAD K-steps ahead of the step being multiplied, this CU's share of the cache
//     lines the workgroups of its XCD are about to stage (one 4-byte load per 128-byte line, L1 bypassed) -- an L2 prefetch.  Why: inside the
//     step a launch's operands come from HBM, not from the L2 / memory-side cache a 20-launch loop leaves them in, and three K-steps in flight
//     (110 KB per CU) cover ~1.5 us at this kernel's pace: 47 us -> 61 us on NT 4096 x 1280 x 5120 with cold operands
//     (profiles/r05e_pl_insitu.txt).  MEASURED AND NOT USED (launch_pl: off): with any lead (6 / 10 / 20 steps) the cold staging rate stays at
//     0.56 us per K-step (profiles/r05a_stage_rate2.txt) -- the cold penalty is not the latency of the first touch -- and on
//     warm operands the extra wave costs 7 %.  Kept behind configuration 8 so that the measurement can be repeated.  The four compute waves of the 4 (n-tiles) x 8 (m-tiles) workgroups of an XCD request every line 4 or
//     8 times at once; the prefetch waves split the lines among those workgroups (52 lines per CU and K-step: one load instruction).
//     XCD co-location decides only who warms which L2 (speed); the wave joins the workgroup's barriers, nothing else.
constexpr int PL_PF_LEAD = 10;
template <int FORM, int BN, bool PF>
__global__ __launch_bounds__(PF ? 320 : 256, 2) void pl_kernel(const GemmP p) {      // (register budget 256: with 512 hipcc moves accumulators through AGPRs)
  using G = PlGeom<BN>;
  constexpr bool B_KC = FORM == GEMM_NT;
  constexpr int NJ = G::NJ, MI = G::MI, NL = G::NL, NP0 = G::NP0, NMF = G::NMF;
  constexpr int dbg = SDXL_PL_DIAG;
  static_assert(FORM == GEMM_NT || FORM == GEMM_NN, "pipelined kernel: NT / NN");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  set_wave_prio(p.prio);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves, wave tile 64 x BN/2
  const int l16 = lane & 15, g = lane >> 4;
  int bx, by;
  xcd_tile_map(p.xcd_px, bx, by);
  const int n0 = (dbg & 8) ? 0 : bx * BN, m0 = (dbg & 8) ? 0 : by * PL_BM;
  const int M = p.M, N = p.N;
  const int lda = (int)p.lda, ldb = (int)p.ldb;
  const int T = p.K / PL_BK;

  if (PF && wave == 4) {
    // this workgroup's share: the XCD's workgroups are a tn x tm rectangle of tiles (xcd_tile_map); row panels of A are shared by the tn
    // workgroups of a tile row, panels of B by the tm workgroups of a tile column
    int tn = 1, tm = 1, ln = 0, lm = 0;
    if (p.xcd_px > 0 && gridDim.x % p.xcd_px == 0 && gridDim.y % (8 / p.xcd_px) == 0) {
      tn = gridDim.x / p.xcd_px; tm = gridDim.y / (8 / p.xcd_px);
      ln = bx % tn; lm = by % tm;
    }
    const int ra_n = (PL_BM + tn - 1) / tn;                 // rows of the A tile this workgroup warms
    const char* Ab = (const char*)p.A;
    const char* Bb = (const char*)p.B;
    // B: K-contiguous (NT): BN rows of one line per K-step; N-contiguous (NN): 64 k-rows of BN * 2 = 320 bytes = 3 lines each
    constexpr int B_UNITS = B_KC ? BN : PL_BK * ((BN * 2 + 127) / 128);
    const int rb_n = (B_UNITS + tm - 1) / tm;
    unsigned sink = 0;      // every load's destination: ONE register the compiler keeps for it (never read before the final wait -- a fresh
                            // destination per load could be reused for an address while its load is still in flight)
    auto touch = [&](int tp) {
      if (tp >= T) return;
      for (int u = lane; u < ra_n; u += 64) {
        const int row = m0 + ln * ra_n + u;
        if (ln * ra_n + u < PL_BM && row < M) {
          asm volatile("global_load_dword %0, %1, off sc1" : "+v"(sink) : "v"(Ab + ((long)row * lda + (long)tp * PL_BK) * 2) : "memory");
        }
      }
      for (int u = lane; u < rb_n; u += 64) {
        const int q = lm * rb_n + u;
        if (q < B_UNITS) {
          const char* src;
          bool ok;
          if (B_KC) { const int row = n0 + q; ok = row < N; src = Bb + ((long)row * ldb + (long)tp * PL_BK) * 2; }
          else { constexpr int LPR = (BN * 2 + 127) / 128; const int kr = q / LPR, li = q - kr * LPR; const int col = n0 + li * 64; ok = col < N; src = Bb + (((long)tp * PL_BK + kr) * ldb + col) * 2; }
          if (ok) {
            asm volatile("global_load_dword %0, %1, off sc1" : "+v"(sink) : "v"(src) : "memory");
          }
        }
      }
    };
    for (int tp = PL_S - 1; tp < PL_PF_LEAD; ++tp) touch(tp);
    __builtin_amdgcn_s_barrier();                           // (the compute waves' prologue barrier)
    for (int t = 0; t < T; ++t) {
      touch(t + PL_PF_LEAD);
      __builtin_amdgcn_s_barrier();                         // (their barrier of step t)
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) :: "memory");
    return;
  }

  // ---- LDS-DMA addressing: A pieces j = 0..3 = chunks wave + 4 j (8 rows x 128 B each), B pieces likewise ----
  const i32x4 ra = make_srd(p.A, 0x7FFFFFFFu), rb = make_srd(p.B, 0x7FFFFFFFu);
  const unsigned lds_w = lds_addr_of(smem) + (unsigned)wave * 1024u;      // slot 0, this wave's first chunk of the A tile
  unsigned voA[G::ACH], voB[G::BCH];                 // constant per-lane byte offsets (PL_OOB: zeros)
  {
    const int kc_row = lane >> 3, kc_vec = (lane & 7) ^ kc_row;      // K-contiguous [R][64] image: lane -> row 8 c + lane / 8, vector swizzled by the row
#pragma unroll
    for (int j = 0; j < G::ACH; ++j) {
      const int row = m0 + 8 * (wave + 4 * j) + kc_row;
      voA[j] = row < M ? (unsigned)(((long)row * lda + kc_vec * 8) * 2) : PL_OOB;
    }
#pragma unroll
    for (int j = 0; j < G::BCH; ++j) {
      if (B_KC) {
        const int row = n0 + 8 * (wave + 4 * j) + kc_row;
        voB[j] = row < N ? (unsigned)(((long)row * ldb + kc_vec * 8) * 2) : PL_OOB;
      } else {
        // N-contiguous [64 k][BN] image: chunk c = vectors 64 c .. 64 c + 63 of the [64][BN / 8] vector grid
        constexpr int V = BN / 8;
        const int q = (wave + 4 * j) * 64 + lane;
        const int krow = q / V, pv = q - krow * V;
        const int n = n0 + (nc_logical<BN>(krow, pv) << 3);
        voB[j] = n < N ? (unsigned)(((long)krow * ldb + n) * 2) : PL_OOB;
      }
    }
  }
  unsigned soA = 0, soB = 0;                         // running scalar byte offsets of the step being staged
  const unsigned stepA = PL_BK * 2, stepB = B_KC ? PL_BK * 2 : (unsigned)(PL_BK * ldb * 2);
  const unsigned oob = PL_OOB;

  // piece pc (0 .. NL - 1: A chunks first) of the step being staged -> ring slot whose base (+ this wave's chunk) is `sb`; beyond the last
  // K-step the pieces are still issued (the counted waits stay uniform), at offsets beyond num_records: zeros, no traffic
  auto piece = [&](auto PC, auto LIVE, unsigned sb) {
    constexpr int pc = decltype(PC)::value;
    constexpr bool live = decltype(LIVE)::value;
    if constexpr (pc < G::ACH) pl_dma<pc * 4096>(ra, live ? voA[pc] : oob, soA, sb);
    else pl_dma<G::A_BYTES + (pc - G::ACH) * 4096>(rb, live ? voB[pc - G::ACH] : oob, soB, sb);
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  using Frag = PlFrag<NJ>;
  constexpr int NITEM = MI + NJ;                     // fragment reads per half-step: B first (the first MFMAs need all of them), then A
  // Fragment-read addresses: one per-lane LDS byte address per (32-deep half-step[, B fragment of the NN form]) for ring slot 0, computed
  // ONCE and made opaque to the compiler (it would otherwise re-derive them inside the loop: one VALU per read in the MFMA shadows);
  // a phase adds its slot's base once per operand, fragment rows and the +4-row second transpose read are immediates.
  typedef const __attribute__((address_space(3))) bf16x8 lds_bf16x8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
  constexpr int NBB = B_KC ? 1 : NJ;                 // B bases per ks: the N-contiguous image's swizzle is not affine in the fragment index
  unsigned bA[2], bB[2][NBB];
  {
    const unsigned l0 = lds_addr_of(smem);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned x = (unsigned)(((ks * 4 + g) ^ (l16 & 7)) << 4);      // K-contiguous image: vector slot = kv ^ (row & 7); every fragment row has row & 7 = l16 & 7
      bA[ks] = l0 + (unsigned)(wm * 64 + l16) * 128u + x;
      asm volatile("" : "+v"(bA[ks]));
#pragma unroll
      for (int j = 0; j < NBB; ++j) {
        if (B_KC) {
          bB[ks][j] = l0 + G::A_BYTES + (unsigned)(wn * (BN / 2) + l16) * 128u + x;
        } else {      // (frag_nc<BN>, gemm_tiles.h: 8 k-rows from ks * 32 + g * 8, 16 columns from wn * BN/2 + 16 j)
          const int krow = ks * 32 + g * 8 + (l16 >> 2);
          const int v = ((wn * (BN / 2) + j * 16) >> 3) + ((l16 >> 1) & 1);
          bB[ks][j] = l0 + G::A_BYTES + (unsigned)(krow * (BN * 2) + (nc_phys<BN>(krow, v) << 4) + (l16 & 1) * 8);
        }
        asm volatile("" : "+v"(bB[ks][j]));
      }
    }
  }
  // a phase's read addresses: this lane's base of (half-step KS) + the ring slot's byte offset, opaque again (the compiler would fold
  // slot offset + fragment-row immediate into a scalar and spend one VALU per read on the sum)
  struct RAddr { unsigned a, b[NBB]; };
  auto raddr = [&](unsigned rbase, auto KS) {
    constexpr int ks = decltype(KS)::value;
    RAddr r;
    r.a = bA[ks] + rbase;
    asm volatile("" : "+v"(r.a));
#pragma unroll
    for (int j = 0; j < NBB; ++j) { r.b[j] = bB[ks][j] + rbase; asm volatile("" : "+v"(r.b[j])); }
    return r;
  };
  // the fragments at `ra`: item IT of NITEM
  auto read_item = [&](const RAddr& ra_, Frag& f, auto IT) {
    constexpr int it = decltype(IT)::value;
    if constexpr ((dbg & 4) != 0) { if constexpr (it < NJ) f.b[it] = ones; else f.a[it - NJ] = ones; }
    else if constexpr (it < NJ) {
      if constexpr (B_KC) {
        f.b[it] = *(lds_bf16x8*)(size_t)(ra_.b[0] + (unsigned)(it * 2048));
      } else {
        const unsigned a0 = ra_.b[it];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(size_t)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(size_t)(a0 + 4u * (BN * 2)));      // rows + 4: same swizzle
        union { s16x4 s2[2]; bf16x8 v; } u;
        u.s2[0] = lo; u.s2[1] = hi;
        f.b[it] = u.v;
      }
    } else {
      constexpr int i = it - NJ;
      f.a[i] = *(lds_bf16x8*)(size_t)(ra_.a + (unsigned)(i * 2048));
    }
  };
  // one phase (PH = 0 / 1): fc = the fragments multiplied, fn = the set being filled from (slot at `rbase`, RKS); pieces [PH NP0, ...) of
  // the step being staged.  At most ONE read item (behind every second MFMA) or ONE piece (in gaps between) per MFMA: every filler
  // issues in the shadow of a 16-cycle MFMA.  The last fragments read (A rows 1 .. 3) are the last ones the next phase needs.
  auto phase = [&](auto PHC, auto LIVE, const Frag& fc, Frag& fn, unsigned rbase, auto RKS, unsigned sb) {
    constexpr int PH = decltype(PHC)::value;
    constexpr int NPP = PH == 0 ? NP0 : NL - NP0;    // pieces of this phase
    static_assert(2 * (NITEM - 1) < NMF && NPP <= NMF / 4, "filler schedule: reads behind the even MFMAs, pieces behind every fourth odd one");
    const RAddr rad = raddr(rbase, RKS);
    pl_static_for<NMF>([&](auto KC) {
      constexpr int k = decltype(KC)::value, i = k / NJ, j = k % NJ;
      if (dbg & 1) acc[i][j][0] += (float)fc.a[i][0] + (float)fc.b[j][0];
      else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc.b[j], fc.a[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k % 2 == 0 && k / 2 < NITEM) read_item(rad, fn, std::integral_constant<int, k / 2>{});
      if constexpr (!(dbg & 2) && k % 2 == 1) {
        pl_static_for<NPP>([&](auto QC) {
          constexpr int q = decltype(QC)::value;
          if constexpr (k == 1 + 2 * ((q * (NMF / 2)) / NPP)) piece(std::integral_constant<int, PH * NP0 + q>{}, LIVE, sb);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: steps 0 .. 2 in flight, step 0 landed, the first half-step's fragments in f0 ----
  {
    unsigned sb = lds_w;
#pragma unroll
    for (int d = 0; d < PL_S - 1; ++d) {
      if (d < T) pl_static_for<NL>([&](auto PC) { piece(PC, std::true_type{}, sb); });
      else pl_static_for<NL>([&](auto PC) { piece(PC, std::false_type{}, sb); });
      soA += stepA; soB += stepB;
      sb += G::STAGE;
    }
  }
  Frag f0, f1;
  wait_vmcnt<(PL_S - 2) * NL>();
  __builtin_amdgcn_s_barrier();
  {
    const RAddr rad = raddr(0u, std::integral_constant<int, 0>{});
    pl_static_for<NITEM>([&](auto IT) { read_item(rad, f0, IT); });
  }
  int rd = 0, wr = PL_S - 1;
  auto kstep = [&](auto LIVE) {
    const unsigned sb = lds_w + (unsigned)wr * G::STAGE;
    phase(std::integral_constant<int, 0>{}, LIVE, f0, f1, (unsigned)rd * G::STAGE, std::integral_constant<int, 1>{}, sb);
    wait_vmcnt<(PL_S - 3) * NL + NP0>();            // my pieces of the next step have landed
    // WAR on the ring slot, by a count instead of by timing: phase 0 of the NEXT step issues pieces into the slot whose half-1 fragments were
    // read in phase 1 of the step before -- those reads must have RETIRED (not merely issued) in every wave before this barrier releases
    // anyone into that phase.  gfx950 barriers do not wait for LDS traffic by themselves; the reads issued above are consumed right behind
    // the barrier anyway, so the wait is close to free.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    const int rn = rd + 1 == PL_S ? 0 : rd + 1;
    phase(std::integral_constant<int, 1>{}, LIVE, f1, f0, (unsigned)rn * G::STAGE, std::integral_constant<int, 0>{}, sb);
    soA += stepA; soB += stepB;
    rd = rn;
    wr = wr + 1 == PL_S ? 0 : wr + 1;
  };
  const int Tlive = T - (PL_S - 1);                  // steps t < Tlive stage a step t + 3 that exists
  int t = 0;
  for (; t < Tlive; ++t) kstep(std::true_type{});
  for (; t < T; ++t) kstep(std::false_type{});
  // ---- epilogue, registers -> global.  Lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile; after a
  // v_permlane16_swap of neighbouring fragments lane g owns 8 contiguous columns of fragment j + (g & 1) at column 8 (g >> 1).
  // The epilogue's operands (bias, residual) are requested FIRST, all of them, before the products drain: one 4-wave workgroup per CU
  // has nobody to hide a chain of 12 dependent load -> add -> store round trips behind (in the step that chain cost 20 us per launch).
  constexpr int NP = NJ / 2;                         // fragment pairs: 16-byte pieces
  constexpr bool ODD = (NJ & 1) != 0;                // BN = 160: the fifth fragment goes out in 8-byte pieces
  const int mrow = m0 + wm * 64 + l16;               // + 16 i
  const int ncol8 = n0 + wn * (BN / 2) + (g & 1) * 16 + (g >> 1) * 8;      // + 32 jp
  const int ncol4 = n0 + wn * (BN / 2) + (NJ - 1) * 16 + g * 4;
  bf16x8 res8[MI][NP], bias8[NP];
  bf16x4 res4[MI], bias4;
  const bool has_res = p.resid != nullptr, has_bias = p.bias != nullptr;      // (wave-uniform: kernel arguments)
  {   // absent operands are zeros: the store section below is branch-free
    bf16x4 z4; z4[0] = z4[1] = z4[2] = z4[3] = (bf16)0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) res8[i][jp] = pl_z8();
      res4[i] = z4;
    }
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) bias8[jp] = pl_z8();
    bias4 = z4;
  }
  if (has_res) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mrow + 16 * i;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int n = ncol8 + 32 * jp;
        res8[i][jp] = (m < M && n < N) ? *(const bf16x8*)(p.resid + (long)m * p.ldr + n) : pl_z8();
      }
      if (ODD) {
        bf16x4 z; z[0] = z[1] = z[2] = z[3] = (bf16)0.f;
        res4[i] = (m < M && ncol4 < N) ? *(const bf16x4*)(p.resid + (long)m * p.ldr + ncol4) : z;
      }
    }
  }
  if (has_bias) {
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int n = ncol8 + 32 * jp;
      bias8[jp] = n < N ? *(const bf16x8*)(p.bias + n) : pl_z8();
    }
    if (ODD) {
      bf16x4 z; z[0] = z[1] = z[2] = z[3] = (bf16)0.f;
      bias4 = ncol4 < N ? *(const bf16x4*)(p.bias + ncol4) : z;
    }
  }
  wait_vmcnt<0>();                                   // zero-fill tail pieces must not outlive the workgroup's LDS allocation (also retires the loads above)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // 24 wait states: MFMA results -> inline-asm VALU reads below
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + 16 * i;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int j = 2 * jp;
      float x[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lo = acc[i][j][r], hi = acc[i][j + 1][r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states: VALU write -> permlane read)
        x[r] = lo;
        x[4 + r] = hi;
      }
      const int n = ncol8 + 32 * jp;
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] += (float)bias8[jp][e]; x[e] += (float)res8[i][jp][e]; }      // (this order: the lockstep kernels' bits)
      if (m < M && n < N) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)x[e];
        *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + n) = o;
      }
    }
    if (ODD) {
      float x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = acc[i][NJ - 1][r];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] += (float)bias4[e]; x[e] += (float)res4[i][e]; }
      if (m < M && ncol4 < N) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)x[e];
        *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + ncol4) = o;
      }
    }
  }
}

template <int FORM, int BN, bool PF>
int launch_pl_k(const GemmP& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)pl_kernel<FORM, BN, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, PlGeom<BN>::SMEM));
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, PL_BM), 1);
  GEMM_LAUNCH((pl_kernel<FORM, BN, PF>), grid, dim3(PF ? 320 : 256), PlGeom<BN>::SMEM, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// can this problem run on the pipelined kernel?  (p as normalised by launch_gemm: accumulate folded into resid, etc.)
bool pl_applicable(const GemmP& p) {
  if (p.form != GEMM_NT && p.form != GEMM_NN) return false;
  if (p.taps != 1 || p.up2 || p.geglu || p.splitk > 1 || p.group > 1 || p.delta_out || p.ln_x || p.out_f32 || p.Cb || p.rowvec) return false;
  if (p.K % PL_BK || p.N % 8 || p.lda % 8 || p.ldb % 8 || p.ldc % 8) return false;
  // 32-bit buffer offsets
  const long abytes = 2 * (long)p.M * p.lda;
  const long bbytes = 2 * (p.form == GEMM_NT ? (long)p.N * p.ldb : (long)p.K * p.ldb);
  if (abytes >= (1L << 31) || bbytes >= (1L << 31)) return false;
  return true;
}

#ifdef SDXL_DIAG      // experiment (profiles/r05s_l2_prefetch_across_kernels.txt): diagnostics build only
// L2 prefetch of the B operand (the weight) of a COMING one-round launch of pl_kernel, as a kernel of its own for another stream: workgroup i
// runs on XCD i % 8 and reads the part of B that XCD's tiles will stage (the same px x py rectangle as launch_pl), so that the weight is in
// that XCD's L2 when the GEMM starts -- issued while the kernel BEFORE the GEMM runs (an attention kernel leaves the fabric idle), not
// inside the GEMM (the in-kernel prefetch wave competes with the staging stream it wants to help: gemm_pl.hip, configuration 8).
__global__ __launch_bounds__(256) void pl_prefetch_b_kernel(const bf16* __restrict__ B, long ldb, int K, int form, int px, int cols_per_xcd,
                                                            int parts, unsigned* sink) {
  const int i = blockIdx.x, xcd = i & 7, part = i >> 3;
  const int n0 = (xcd % px) * cols_per_xcd;
  const long rows = form == GEMM_NT ? cols_per_xcd : K, rowbytes = 2L * (form == GEMM_NT ? K : cols_per_xcd);
  const char* base = (const char*)(form == GEMM_NT ? B + (long)n0 * ldb : B + n0);
  const long vpr = rowbytes / 16, total = rows * vpr;
  unsigned acc = 0;
  for (long v = (long)part * 256 + threadIdx.x; v < total; v += (long)parts * 256) {
    const long r = v / vpr, c = v - r * vpr;
    const uint4 x = *(const uint4*)(base + r * ldb * 2 + c * 16);
    acc ^= x.x ^ x.y ^ x.z ^ x.w;
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;      // (keeps the loads alive)
}
int launch_pl_prefetch_b(const GemmP& p, int parts, hipStream_t st) {
  const int bn = p.N % 160 == 0 ? 160 : 128;
  const int gx = cdiv(p.N, bn), gy = cdiv(p.M, PL_BM);
  double best = 1e30;
  int bpx = 0;
  for (int px = 1; px <= 8; px *= 2) {
    const int py = 8 / px;
    if (gx % px || gy % py) continue;
    const double cost = (double)p.N / px + (double)p.M / py;
    if (cost < best) { best = cost; bpx = px; }
  }
  if (!bpx) return 0;
  hipLaunchKernelGGL(pl_prefetch_b_kernel, dim3(8 * parts), dim3(256), 0, st, p.B, p.ldb, p.K, p.form, bpx, gx / bpx * bn, parts, (unsigned*)nullptr);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
#endif

// prefetch: the L2 prefetch wave (configuration 8) -- measured, not used (no gain on cold operands, -7 % on warm ones); its instantiations exist in the
// diagnostics build only, the product library carries the four plain kernels and ignores the flag.
int launch_pl(const GemmP& pin, int bn, hipStream_t st, bool prefetch) {
  ARG_CHECK(pl_applicable(pin), "gemm_pl: problem %dx%dx%d (form %d) does not fit the pipelined kernel", pin.M, pin.N, pin.K, pin.form);
  GemmP p = pin;
  if (bn != 128 && bn != 160) bn = p.N % 160 == 0 ? 160 : 128;
  {   // px x (8/px) XCD grid over the (n, m) tile grid minimising the per-XCD operand footprint ~ N/px + M/py
    const int gx = cdiv(p.N, bn), gy = cdiv(p.M, PL_BM);
    double best = 1e30;
    p.xcd_px = 0;
    for (int px = 1; px <= 8; px *= 2) {
      const int py = 8 / px;
      if (gx % px || gy % py) continue;
      const double cost = (double)p.N / px + (double)p.M / py;
      if (cost < best) { best = cost; p.xcd_px = px; }
    }
  }
#ifdef SDXL_DIAG
  if (prefetch && p.K / PL_BK > PL_PF_LEAD) {      // (short reductions: the prologue's three steps are most of it)
    if (p.form == GEMM_NT) return bn == 160 ? launch_pl_k<GEMM_NT, 160, true>(p, st) : launch_pl_k<GEMM_NT, 128, true>(p, st);
    return bn == 160 ? launch_pl_k<GEMM_NN, 160, true>(p, st) : launch_pl_k<GEMM_NN, 128, true>(p, st);
  }
#endif
  (void)prefetch;
  if (p.form == GEMM_NT) return bn == 160 ? launch_pl_k<GEMM_NT, 160, false>(p, st) : launch_pl_k<GEMM_NT, 128, false>(p, st);
  return bn == 160 ? launch_pl_k<GEMM_NN, 160, false>(p, st) : launch_pl_k<GEMM_NN, 128, false>(p, st);
}
