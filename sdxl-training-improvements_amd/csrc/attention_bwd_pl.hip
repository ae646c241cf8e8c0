// Software-pipelined, one-wave-per-SIMD flash-attention BACKWARD (head_dim 64, no mask, softmax scale 1/8) for gfx950: dQ, dK, dV of
// self-attention-shaped problems (Nq, Nk >= 256) in ONE grid, the structure of attention_pl.hip applied to the two bodies of
// attention.hip's fused backward (reference: the backward of the fused attention flow_matching_trainer.py:69-76 asks for, reached through
// loss.backward(), ddpm_trainer.py:271 / flow_matching_trainer.py:252).
//
// Why here and not in the forward: at d = 64 the forward's unit is 16 MFMAs against 40 VALU instructions and does not fit the MFMA shadow
// (attention_pl.hip); the backward's units are 24 MFMAs (dQ: S, dP, dQ) and 32 MFMAs (dK / dV: S, dP, dV, dK) against 40 / 48 VALU
// instructions, and their instruction streams alone run at 70-85 % of the matrix pipe from ONE wave per SIMD
// (profiles/r06i_attn_bwd_shadow.txt) where the tiled kernels reach 36 %.
//
// Structure:
//   * 4-wave workgroups, one per CU (512 registers per wave: gradients in AGPRs, everything the VALU touches in VGPRs), up to 4 blocks of 16
//     rows per wave; work split in the flattened (batch-head pair, 16-row block) space in ranges of 16 blocks, processed pair by pair
//     ("phases"), XCD-contiguous -- as attention_pl.hip.  Key-side workgroups (dK / dV: the longer ones) take the low ids.
//   * dQ body: query blocks stationary (Q^T, dO^T fragments, -LSE, -Delta as accumulator start values), K | V tiles streamed; dK / dV body:
//     key blocks stationary (K, V fragments), Q | dO tiles + LSE | Delta rows streamed.  LDS-DMA through raw buffer descriptors into a
//     4-deep ring, three tiles ahead, counted vmcnt, ONE barrier per 64-row tile.
//   * The pipeline runs over HALF tiles (32 streamed rows): unit u = (half tile, block).  phase(u) issues, in one asm statement per step
//     (4 steps), the S and dP products of unit u + 1, the softmax-backward arithmetic of unit u (P = exp2(S), dS = P (dP - Delta): the
//     subtractions ride in the accumulators' start values) with its consumers one step behind the exponentials, and the gradient products of
//     unit u - 1.  Half tiles keep the streamed operands' fragments at 48 (dQ) / 80 (dK, dV) registers; each fragment register is refilled
//     right behind the last product that read it (one or two LDS reads per step), a whole phase before its next use.
//   * No masks: a ragged last tile's missing rows arrive as zeros (K, V, Q, dO rows and LSE / Delta entries beyond the sequence are
//     out-of-range for the DMA) and contribute exactly nothing -- except P of a padding KEY in the dQ body, 2^-LSE, finite for any
//     LSE > -127 log2 units and multiplied into a zero K row; the launcher routes a ragged Nk whose LSE could be that small nowhere: see
//     bpl_dq_phase (the padding keys' P is forced to zero there, three instructions per unit of the last tile).
#include "attn_tiles.h"

#include <type_traits>
#include <utility>

namespace {

constexpr int BPL_RING = 4;
constexpr int BPL_SLOT_BYTES = 2 * TILE_ELEMS * 2 + 512;      // two [64][64] bf16 tiles + [2][64] fp32 (LSE | Delta of the dK / dV body)
constexpr int BPL_SMEM = BPL_RING * BPL_SLOT_BYTES + 64;

template <typename F, int... I>
__device__ __forceinline__ void bpl_sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void bpl_sfor(F&& f) { bpl_sfor_impl(f, std::make_integer_sequence<int, N>{}); }

typedef unsigned bpl_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 bpl_frag(const bpl_u4& u) { return __builtin_bit_cast(bf16x8, u); }

// stand-alone products (prologue / first and last units): two wait states in front -- hipcc may have copied an operand or moved an AGPR quad
// right before the statement (attention_pl.hip)
// (Tried: the stationary operands -- the wave's own Q^T / dO^T or K / V fragments -- in AGPRs through "a" constraints: hipcc keeps their home in
// VGPRs and copies them into AGPR temporaries in front of every statement, 120 v_accvgpr_write per tile; and a start quad in AGPRs is not encodable
// with a VGPR destination: C and D share one register-file bit.)
template <bool CA>
__device__ __forceinline__ void bpl_mfma_c(f32x4& d, const bf16x8& a, const bf16x8& b, const f32x4& c) {      // d = a . b + c
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void bpl_mfma_v(f32x4& d, const bf16x8& a, const bf16x8& b) {                      // d += a . b (VGPR)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void bpl_mfma_a(f32x4& d, const bf16x8& a, const bf16x8& b) {                      // d += a . b (AGPR)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void bpl_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void bpl_exp2(float& e0, float& e1, const float s0, const float s1) {
  asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(e0), "=v"(e1) : "v"(s0), "v"(s1));
}
__device__ __forceinline__ unsigned bpl_cvt(const float lo, const float hi) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void lds_dma4_buffer(i32x4 srd_uniform, unsigned voffset, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voffset), "s"(srd_uniform), "s"(lds_dst_uniform) : "memory");
}

// ---- dQ body, one step: exponentials of pair k, S(u+1) product, dS = P dP of pair k-1 (in place), dP(u+1) product, conversion, dQ(u-1) product.
// C0: first half of the reduction over d (accumulators start from -LSE / -Delta quads cs / cd).  CONS: a pair k-1 exists.
template <bool C0, bool CONS>
__device__ __forceinline__ void bpl_dq_step(f32x4& s, f32x4& dp, f32x4& dq, float& e0, float& e1, float& pe0, float& pe1, unsigned& pd,
                                            const bf16x8& ka, const bf16x8& qb, const bf16x8& va, const bf16x8& db, const bf16x8& kta, const bf16x8& dsb,
                                            const float x0, const float x1, const float d0, const float d1, const f32x4& cs, const f32x4& cd) {
  // %0 s %1 dp %2 dq %3 e0 %4 e1 %5 pe0 %6 pe1 %7 pd | %8 ka %9 qb %10 va %11 db %12 kta %13 dsb %14 x0 %15 x1 %16 d0 %17 d1 %18 cs %19 cd
#define BQ_EXP "v_exp_f32 %3, %14\n\tv_exp_f32 %4, %15\n\t"
#define BQ_S(C) "v_mfma_f32_16x16x32_bf16 %0, %8, %9, " C "\n\t"
#define BQ_MUL "v_mul_f32 %5, %5, %16\n\tv_mul_f32 %6, %6, %17\n\t"
#define BQ_DP(C) "v_mfma_f32_16x16x32_bf16 %1, %10, %11, " C "\n\t"
#define BQ_CVT "v_cvt_pk_bf16_f32 %7, %5, %6\n\t"
#define BQ_DQ "v_mfma_f32_16x16x32_bf16 %2, %12, %13, %2"
#define BQ_INS : "v"(ka), "v"(qb), "v"(va), "v"(db), "v"(kta), "v"(dsb), "v"(x0), "v"(x1), "v"(d0), "v"(d1), "v"(cs), "v"(cd)
  if (C0 && CONS) asm volatile(BQ_EXP BQ_S("%18") BQ_MUL BQ_DP("%19") BQ_CVT BQ_DQ : "=&v"(s), "=&v"(dp), "+a"(dq), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pd) BQ_INS);
  else if (C0) asm volatile(BQ_EXP BQ_S("%18") BQ_DP("%19") BQ_DQ : "=&v"(s), "=&v"(dp), "+a"(dq), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pd) BQ_INS);
  else if (CONS) asm volatile(BQ_EXP BQ_S("%0") BQ_MUL BQ_DP("%1") BQ_CVT BQ_DQ : "+v"(s), "+v"(dp), "+a"(dq), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pd) BQ_INS);
  else asm volatile(BQ_EXP BQ_S("%0") BQ_DP("%1") BQ_DQ : "+v"(s), "+v"(dp), "+a"(dq), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pd) BQ_INS);
#undef BQ_EXP
#undef BQ_S
#undef BQ_MUL
#undef BQ_DP
#undef BQ_CVT
#undef BQ_DQ
#undef BQ_INS
}
__device__ __forceinline__ void bpl_dq_tail(float& pe0, float& pe1, unsigned& pd, const float d0, const float d1) {      // consumers of a unit's last pair
  asm volatile("v_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %4\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "+v"(pe0), "+v"(pe1), "=&v"(pd) : "v"(d0), "v"(d1));
}

// ---- dK / dV body, one step: exponentials of pair k, S(u+1) product, P conversion of pair k-1, dP(u+1) product, dS = P dP of pair k-1,
// dV(u-1) product, dS conversion, dK(u-1) product.
template <bool C0, bool CONS>
__device__ __forceinline__ void bpl_dkv_step(f32x4& s, f32x4& dp, f32x4& dv, f32x4& dk, float& e0, float& e1, float& pe0, float& pe1, unsigned& pp, unsigned& ps,
                                             const bf16x8& qa, const bf16x8& kb, const bf16x8& da, const bf16x8& vb, const bf16x8& dota, const bf16x8& pfb,
                                             const bf16x8& qta, const bf16x8& dsb, const float x0, const float x1, const float d0, const float d1,
                                             const f32x4& cs, const f32x4& cd) {
  // %0 s %1 dp %2 dv %3 dk %4 e0 %5 e1 %6 pe0 %7 pe1 %8 pp %9 ps | %10 qa %11 kb %12 da %13 vb %14 dota %15 pfb %16 qta %17 dsb %18 x0 %19 x1 %20 d0 %21 d1 %22 cs %23 cd
#define BK_EXP "v_exp_f32 %4, %18\n\tv_exp_f32 %5, %19\n\t"
#define BK_S(C) "v_mfma_f32_16x16x32_bf16 %0, %10, %11, " C "\n\t"
#define BK_CVTP "v_cvt_pk_bf16_f32 %8, %6, %7\n\t"
#define BK_DP(C) "v_mfma_f32_16x16x32_bf16 %1, %12, %13, " C "\n\t"
#define BK_MUL "v_mul_f32 %6, %6, %20\n\tv_mul_f32 %7, %7, %21\n\t"
#define BK_DV "v_mfma_f32_16x16x32_bf16 %2, %14, %15, %2\n\t"
#define BK_CVTS "v_cvt_pk_bf16_f32 %9, %6, %7\n\t"
#define BK_DK "v_mfma_f32_16x16x32_bf16 %3, %16, %17, %3"
#define BK_INS : "v"(qa), "v"(kb), "v"(da), "v"(vb), "v"(dota), "v"(pfb), "v"(qta), "v"(dsb), "v"(x0), "v"(x1), "v"(d0), "v"(d1), "v"(cs), "v"(cd)
  if (C0 && CONS) asm volatile(BK_EXP BK_S("%22") BK_CVTP BK_DP("%23") BK_MUL BK_DV BK_CVTS BK_DK
                               : "=&v"(s), "=&v"(dp), "+a"(dv), "+a"(dk), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pp), "=&v"(ps) BK_INS);
  else if (C0) asm volatile(BK_EXP BK_S("%22") BK_DP("%23") BK_DV BK_DK
                            : "=&v"(s), "=&v"(dp), "+a"(dv), "+a"(dk), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pp), "=&v"(ps) BK_INS);
  else if (CONS) asm volatile(BK_EXP BK_S("%0") BK_CVTP BK_DP("%1") BK_MUL BK_DV BK_CVTS BK_DK
                              : "+v"(s), "+v"(dp), "+a"(dv), "+a"(dk), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pp), "=&v"(ps) BK_INS);
  else asm volatile(BK_EXP BK_S("%0") BK_DP("%1") BK_DV BK_DK
                    : "+v"(s), "+v"(dp), "+a"(dv), "+a"(dk), "=&v"(e0), "=&v"(e1), "+v"(pe0), "+v"(pe1), "=&v"(pp), "=&v"(ps) BK_INS);
#undef BK_EXP
#undef BK_S
#undef BK_CVTP
#undef BK_DP
#undef BK_MUL
#undef BK_DV
#undef BK_CVTS
#undef BK_DK
#undef BK_INS
}
__device__ __forceinline__ void bpl_dkv_tail(float& pe0, float& pe1, unsigned& pp, unsigned& ps, const float d0, const float d1) {
  asm volatile("v_cvt_pk_bf16_f32 %2, %0, %1\n\tv_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %3, %0, %1"
               : "+v"(pe0), "+v"(pe1), "=&v"(pp), "=&v"(ps) : "v"(d0), "v"(d1));
}

// tile staging shared by both bodies: two [64][64] bf16 tiles (A | B: K | V or Q | dO) of rows [64 t, 64 t + 64) -> ring slot t % 4, 2 + 2 pieces of
// 1 KiB per wave; rows beyond nrows and whole tiles beyond the last are out of range: zeros
struct BplStage {
  i32x4 asrd, bsrd;
  unsigned avo[2], bvo[2], atile, btile, lds0;
  int wave;
  __device__ __forceinline__ void init(const bf16* A, long lda, const bf16* B, long ldb, int nrows, bf16* sm, int wave_, int lane) {
    asrd = make_srd(A, (unsigned)((((long)nrows - 1) * lda + HD) * 2));
    bsrd = make_srd(B, (unsigned)((((long)nrows - 1) * ldb + HD) * 2));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave_ * 2 + j) * 8 + (lane >> 3);
      avo[j] = (unsigned)((r * lda + (((lane & 7) ^ (r & 7)) << 3)) * 2);
      bvo[j] = (unsigned)((r * ldb + (((lane & 7) ^ (r & 7)) << 3)) * 2);
    }
    atile = (unsigned)(64 * lda * 2);
    btile = (unsigned)(64 * ldb * 2);
    lds0 = lds_addr_of(sm);
    wave = wave_;
  }
  __device__ __forceinline__ void tile(int t) const {
    const unsigned slot = lds0 + (unsigned)(t & (BPL_RING - 1)) * BPL_SLOT_BYTES + (unsigned)wave * 2048u;
#pragma unroll
    for (int j = 0; j < 2; ++j) lds_dma16_buffer(asrd, avo[j] + (unsigned)t * atile, 0u, slot + j * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) lds_dma16_buffer(bsrd, bvo[j] + (unsigned)t * btile, 0u, slot + TILE_ELEMS * 2 + j * 1024);
  }
};
__device__ __forceinline__ const bf16* bpl_tile_a(const bf16* sm, int t) { return (const bf16*)((const char*)sm + (t & (BPL_RING - 1)) * BPL_SLOT_BYTES); }
__device__ __forceinline__ const bf16* bpl_tile_b(const bf16* sm, int t) { return bpl_tile_a(sm, t) + TILE_ELEMS; }
__device__ __forceinline__ const float* bpl_stats(const bf16* sm, int t) { return (const float*)(bpl_tile_a(sm, t) + 2 * TILE_ELEMS); }

// ================================================================================================================================
// dQ: this wave's N query blocks qb0, qb0 + 4, ... of pair bh against all key tiles.  S^T[key][q] = K . Q^T (a lane owns one query column).
// ================================================================================================================================
template <int N>
__device__ __forceinline__ void bpl_dq_phase(const AttnP& p, bf16* sm, const int bh, const int qb0, const int lane, const int wave) {
  const int l16 = lane & 15, g = lane >> 4;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qb = p.Q + (long)b * p.Nq * p.ldq + h * HD;
  const bf16* dOb = p.dO + (long)b * p.Nq * p.lddo + h * HD;
  const bf16* Kb = p.K + (long)b * p.Nk * p.ldk + h * HD;
  const bf16* Vb = p.V + (long)b * p.Nk * p.ldv + h * HD;
  const int ntiles = (p.Nk + 63) >> 6, nhalf = 2 * ntiles;      // ntiles >= 4 (launcher)
  BplStage st;
  st.init(Kb, p.ldk, Vb, p.ldv, p.Nk, sm, wave, lane);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // a previous phase's stores (reads and writes return out of order relative to each other)
  __builtin_amdgcn_s_barrier();
  st.tile(0);
  st.tile(1);
  st.tile(2);
  constexpr int NN = N > 0 ? N : 1;
  bf16x8 qf[NN][2], df[NN][2];
  f32x4 negl[NN], negd[NN], dq[NN][4];
  if constexpr (N > 0) {
    const float c = SCALE * LOG2E;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int q = (qb0 + 4 * i) * 16 + l16;
      const bool ok = q < p.Nq;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[i][ks] = ok ? *(const bf16x8*)(Qb + (long)q * p.ldq + ks * 32 + g * 8) : z8();
        df[i][ks] = ok ? *(const bf16x8*)(dOb + (long)q * p.lddo + ks * 32 + g * 8) : z8();
      }
      const float l2 = ok ? p.LSE[(long)bh * p.Nq + q] * LOG2E : 0.f;
      const float dl = ok ? p.Delta[(long)bh * p.Nq + q] : 0.f;
      negl[i] = (f32x4){-l2, -l2, -l2, -l2};
      negd[i] = (f32x4){-dl, -dl, -dl, -dl};
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[i][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      qf[i][0] = scale8(qf[i][0], c); qf[i][1] = scale8(qf[i][1], c);      // scores in the log2 domain
      landed(qf[i][0]); landed(qf[i][1]); landed(df[i][0]); landed(df[i][1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // fragments of the streamed half tile (32 keys): K rows [kbl][k half], V rows, K^T [d block]
  bf16x8 kf[2][2], vf[2][2], ktr[4];
  f32x4 Sn[2], Dn[2];
  bpl_u4 dsp;
  auto rd_kv = [&](int hh, int kbl, int ks) {
    const int t = hh >> 1, hb = hh & 1;
    kf[kbl][ks] = ld_frag(bpl_tile_a(sm, t), (2 * hb + kbl) * 16 + l16, ks * 32 + g * 8);
    vf[kbl][ks] = ld_frag(bpl_tile_b(sm, t), (2 * hb + kbl) * 16 + l16, ks * 32 + g * 8);
  };
  auto rd_kt = [&](int hh, int db) { ktr[db] = tr_frag(bpl_tile_a(sm, hh >> 1), hh & 1, db * 16, l16, g); };
  if constexpr (N > 0) {
#pragma unroll
    for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) rd_kv(0, kbl, ks);
    // unit (0, 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kbl = k & 1, ks = k >> 1;
      if (ks == 0) { bpl_mfma_c<true>(Sn[kbl], kf[kbl][0], qf[0][0], negl[0]); bpl_mfma_c<true>(Dn[kbl], vf[kbl][0], df[0][0], negd[0]); }
      else { bpl_mfma_v(Sn[kbl], kf[kbl][1], qf[0][1]); bpl_mfma_v(Dn[kbl], vf[kbl][1], df[0][1]); }
    }
  }
  const int nk_last = p.Nk - (ntiles - 1) * 64;      // valid keys of the last tile (64: not ragged)
  // one half tile: N phases.  FIRST: hh = 0 (no unit precedes its first); LAST: no half follows; MASK: keys beyond Nk in this half
  auto half = [&](const int hh, auto HBC, auto FIRSTC, auto LASTC, auto MASKC) {
    constexpr bool FIRST = decltype(FIRSTC)::value, LAST = decltype(LASTC)::value, MASK = decltype(MASKC)::value;
    if (decltype(HBC)::value == 0) {      // top of a 64-key tile
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // my pieces of tile t + 1 have landed, my LDS reads have retired
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      st.tile((hh >> 1) + 3);
    }
    if constexpr (N > 0) {
      bpl_sfor<N>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
        constexpr bool NEXT_IN_HALF = i + 1 < N;
        constexpr int inext = NEXT_IN_HALF ? i + 1 : 0;
        constexpr bool HAS_NEXT = NEXT_IN_HALF || !LAST;
        constexpr bool HAS_PREV = i > 0 || !FIRST;
        constexpr int iprev = i > 0 ? i - 1 : N - 1;
        constexpr bool STEADY = HAS_NEXT && HAS_PREV && !MASK;
        f32x4 Sc[2], Dc[2];
        Sc[0] = Sn[0]; Sc[1] = Sn[1]; Dc[0] = Dn[0]; Dc[1] = Dn[1];
        bpl_u4 dsc;
        if (N == 1 && !LAST) {      // one block per wave: the next half's K / V fragments are needed right here (exposed; boundary phases only)
#pragma unroll
          for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rd_kv(hh + 1, kbl, ks);
        }
        float pe0 = 0.f, pe1 = 0.f;
        bpl_sfor<4>([&](auto KC) {
          constexpr int k = decltype(KC)::value;
          constexpr int kbl = k & 1, ks = k >> 1;                // S / dP products of unit u + 1: key block kbl of the half, k half ks
          constexpr int kq = k >> 1, hp = k & 1;                 // score pair k of unit u: block kq, registers 2 hp, 2 hp + 1
          constexpr int kp = k > 0 ? k - 1 : 0, kqp = kp >> 1, hpp = kp & 1;      // the pair whose consumers run in this step
          if constexpr (STEADY) {
            float ne0, ne1;
            unsigned pd = 0u;
            if (k == 0) bpl_dq_step<true, false>(Sn[kbl], Dn[kbl], dq[iprev][k], ne0, ne1, pe0, pe1, pd, kf[kbl][0], qf[inext][0], vf[kbl][0], df[inext][0], ktr[k],
                                                 bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], 0.f, 0.f, negl[inext], negd[inext]);
            else if (ks == 0) bpl_dq_step<true, true>(Sn[kbl], Dn[kbl], dq[iprev][k], ne0, ne1, pe0, pe1, pd, kf[kbl][0], qf[inext][0], vf[kbl][0], df[inext][0], ktr[k],
                                                      bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], Dc[kqp][2 * hpp], Dc[kqp][2 * hpp + 1], negl[inext], negd[inext]);
            else bpl_dq_step<false, true>(Sn[kbl], Dn[kbl], dq[iprev][k], ne0, ne1, pe0, pe1, pd, kf[kbl][1], qf[inext][1], vf[kbl][1], df[inext][1], ktr[k],
                                          bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], Dc[kqp][2 * hpp], Dc[kqp][2 * hpp + 1], negl[inext], negd[inext]);
            if (k > 0) dsc[kp] = pd;
            pe0 = ne0;
            pe1 = ne1;
            if (k == 3) {
              unsigned pl;
              bpl_dq_tail(pe0, pe1, pl, Dc[1][2], Dc[1][3]);
              dsc[3] = pl;
            }
            // refill the fragment registers this step's products were the last to read
            if (N >= 2 && i == N - 2 && !LAST) rd_kv(hh + 1, kbl, ks);      // K / V of the next half: S(u + 1) of this half's last phase
            if (i == 0) rd_kt(hh, k);                                       // K^T of this half: dQ(u - 1) from the next phase on
          } else {
            // boundary forms (first / last unit of the phase, masked half): separate statements
            if (HAS_NEXT) {
              if (ks == 0) { bpl_mfma_c<true>(Sn[kbl], kf[kbl][0], qf[inext][0], negl[inext]); bpl_mfma_c<true>(Dn[kbl], vf[kbl][0], df[inext][0], negd[inext]); }
              else { bpl_mfma_v(Sn[kbl], kf[kbl][1], qf[inext][1]); bpl_mfma_v(Dn[kbl], vf[kbl][1], df[inext][1]); }
            }
            float e0, e1;
            bpl_exp2(e0, e1, Sc[kq][2 * hp], Sc[kq][2 * hp + 1]);
            if (HAS_PREV) bpl_mfma_a(dq[iprev][k], ktr[k], bpl_frag(dsp));
            if (MASK) {      // a padding key's P = 2^-LSE is finite only while LSE > -127: force it to zero (its K row is zero, 0 * inf would not be)
              const int key = (hh & 1) * 32 + kq * 16 + g * 4 + 2 * hp;
              if (key >= nk_last) e0 = 0.f;
              if (key + 1 >= nk_last) e1 = 0.f;
            }
            dsc[k] = bpl_cvt(e0 * Dc[kq][2 * hp], e1 * Dc[kq][2 * hp + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if (!STEADY) {
          if (N >= 2 && i == N - 2 && !LAST) {
#pragma unroll
            for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) rd_kv(hh + 1, kbl, ks);
          }
          if (i == 0) {
#pragma unroll
            for (int db = 0; db < 4; ++db) rd_kt(hh, db);
          }
        }
        dsp = dsc;
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };
  const bool ragged = nk_last < 64;
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  half(0, H0{}, std::true_type{}, std::false_type{}, std::false_type{});
  half(1, H1{}, std::false_type{}, std::false_type{}, std::false_type{});
  for (int t = 1; t < ntiles - 1; ++t) {      // (a whole tile per iteration: 2 N units, the S / dP / dS register pairs end where they started)
    half(2 * t, H0{}, std::false_type{}, std::false_type{}, std::false_type{});
    half(2 * t + 1, H1{}, std::false_type{}, std::false_type{}, std::false_type{});
  }
  if (ragged) {
    half(nhalf - 2, H0{}, std::false_type{}, std::false_type{}, std::true_type{});
    half(nhalf - 1, H1{}, std::false_type{}, std::true_type{}, std::true_type{});
  } else {
    half(nhalf - 2, H0{}, std::false_type{}, std::false_type{}, std::false_type{});
    half(nhalf - 1, H1{}, std::false_type{}, std::true_type{}, std::false_type{});
  }
  if constexpr (N > 0) {
    // drain: dQ of the last unit
#pragma unroll
    for (int db = 0; db < 4; ++db) bpl_mfma_a(dq[N - 1][db], ktr[db], bpl_frag(dsp));
    bpl_drain();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int q = (qb0 + 4 * i) * 16 + l16;
      if (q < p.Nq) {
        bf16* row = p.dQ + ((long)b * p.Nq + q) * p.lddq + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          bf16x4 o;
          o[0] = (bf16)(dq[i][db][0] * SCALE); o[1] = (bf16)(dq[i][db][1] * SCALE);
          o[2] = (bf16)(dq[i][db][2] * SCALE); o[3] = (bf16)(dq[i][db][3] * SCALE);
          *(bf16x4*)(row + db * 16 + g * 4) = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the dummy tail pieces must not outlive the phase
}

// ================================================================================================================================
// dK / dV: this wave's N key blocks kb0, kb0 + 4, ... of pair bh against all query tiles.  S[q][key] = Q . K^T (a lane owns one key column).
// ================================================================================================================================
template <int N>
__device__ __forceinline__ void bpl_dkv_phase(const AttnP& p, bf16* sm, const int bh, const int kb0, const int lane, const int wave) {
  const int l16 = lane & 15, g = lane >> 4;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qb = p.Q + (long)b * p.Nq * p.ldq + h * HD;
  const bf16* dOb = p.dO + (long)b * p.Nq * p.lddo + h * HD;
  const bf16* Kb = p.K + (long)b * p.Nk * p.ldk + h * HD;
  const bf16* Vb = p.V + (long)b * p.Nk * p.ldv + h * HD;
  const int ntiles = (p.Nq + 63) >> 6, nhalf = 2 * ntiles;      // ntiles >= 4 (launcher)
  BplStage st;
  st.init(Qb, p.ldq, dOb, p.lddo, p.Nq, sm, wave, lane);
  // LSE | Delta rows of a query tile: 64 floats each, one 256-byte piece per wave (waves 0 / 1; waves 2 / 3 repeat them -- the same bytes to the
  // same place -- so that every wave counts five pieces per tile); entries beyond Nq are out of range: zeros
  const i32x4 lsrd = make_srd(p.LSE + (long)bh * p.Nq, (unsigned)(p.Nq * 4)), dsrd = make_srd(p.Delta + (long)bh * p.Nq, (unsigned)(p.Nq * 4));
  auto stage = [&](int t) {
    st.tile(t);
    const unsigned dst = st.lds0 + (unsigned)(t & (BPL_RING - 1)) * BPL_SLOT_BYTES + 2 * TILE_ELEMS * 2 + (unsigned)(wave & 1) * 256u;
    const unsigned vo = (unsigned)((t * 64 + lane) * 4);
    if (wave & 1) lds_dma4_buffer(dsrd, vo, dst);
    else lds_dma4_buffer(lsrd, vo, dst);
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stage(0);
  stage(1);
  stage(2);
  constexpr int NN = N > 0 ? N : 1;
  bf16x8 kfr[NN][2], vfr[NN][2];      // B operands [d][key]
  f32x4 dk[NN][4], dv[NN][4];
  if constexpr (N > 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int key = (kb0 + 4 * i) * 16 + l16;
      const bool ok = key < p.Nk;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kfr[i][ks] = ok ? *(const bf16x8*)(Kb + (long)key * p.ldk + ks * 32 + g * 8) : z8();
        vfr[i][ks] = ok ? *(const bf16x8*)(Vb + (long)key * p.ldv + ks * 32 + g * 8) : z8();
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) { dk[i][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      kfr[i][0] = scale8(kfr[i][0], SCALE * LOG2E); kfr[i][1] = scale8(kfr[i][1], SCALE * LOG2E);      // scores in the log2 domain
      landed(kfr[i][0]); landed(kfr[i][1]); landed(vfr[i][0]); landed(vfr[i][1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // fragments of the streamed half tile (32 queries): Q rows [qbl][k half], dO rows, dO^T / Q^T [d block], -LSE log2 e / -Delta of the lane's rows
  bf16x8 qa[2][2], da[2][2], dot[4], qt[4];
  f32x4 nl[2], nd[2];
  f32x4 Sn[2], Dn[2];
  bpl_u4 pfp, dsp;
  auto rd_qd = [&](int hh, int qbl, int ks) {
    const int t = hh >> 1, hb = hh & 1;
    qa[qbl][ks] = ld_frag(bpl_tile_a(sm, t), (2 * hb + qbl) * 16 + l16, ks * 32 + g * 8);
    da[qbl][ks] = ld_frag(bpl_tile_b(sm, t), (2 * hb + qbl) * 16 + l16, ks * 32 + g * 8);
  };
  auto rd_tr = [&](int hh, int db) {
    dot[db] = tr_frag(bpl_tile_b(sm, hh >> 1), hh & 1, db * 16, l16, g);
    qt[db] = tr_frag(bpl_tile_a(sm, hh >> 1), hh & 1, db * 16, l16, g);
  };
  auto rd_st = [&](int hh, int qbl) {
    const float* s = bpl_stats(sm, hh >> 1) + (hh & 1) * 32 + qbl * 16 + g * 4;
    const f32x4 l = *(const f32x4*)s, d = *(const f32x4*)(s + 64);
    nl[qbl] = l * (-LOG2E);
    nd[qbl] = (f32x4){-d[0], -d[1], -d[2], -d[3]};
  };
  if constexpr (N > 0) {
#pragma unroll
    for (int qbl = 0; qbl < 2; ++qbl) {
      rd_st(0, qbl);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) rd_qd(0, qbl, ks);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qbl = k & 1, ks = k >> 1;
      if (ks == 0) { bpl_mfma_c<false>(Sn[qbl], qa[qbl][0], kfr[0][0], nl[qbl]); bpl_mfma_c<false>(Dn[qbl], da[qbl][0], vfr[0][0], nd[qbl]); }
      else { bpl_mfma_v(Sn[qbl], qa[qbl][1], kfr[0][1]); bpl_mfma_v(Dn[qbl], da[qbl][1], vfr[0][1]); }
    }
  }
  auto half = [&](const int hh, auto HBC, auto FIRSTC, auto LASTC) {
    constexpr bool FIRST = decltype(FIRSTC)::value, LAST = decltype(LASTC)::value;
    if (decltype(HBC)::value == 0) {
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      stage((hh >> 1) + 3);
    }
    if constexpr (N > 0) {
      bpl_sfor<N>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
        constexpr bool NEXT_IN_HALF = i + 1 < N;
        constexpr int inext = NEXT_IN_HALF ? i + 1 : 0;
        constexpr bool HAS_NEXT = NEXT_IN_HALF || !LAST;
        constexpr bool HAS_PREV = i > 0 || !FIRST;
        constexpr int iprev = i > 0 ? i - 1 : N - 1;
        constexpr bool STEADY = HAS_NEXT && HAS_PREV;
        f32x4 Sc[2], Dc[2];
        Sc[0] = Sn[0]; Sc[1] = Sn[1]; Dc[0] = Dn[0]; Dc[1] = Dn[1];
        bpl_u4 pfc, dsc;
        if (N == 1 && !LAST) {
#pragma unroll
          for (int qbl = 0; qbl < 2; ++qbl) {
            rd_st(hh + 1, qbl);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rd_qd(hh + 1, qbl, ks);
          }
        }
        float pe0 = 0.f, pe1 = 0.f;
        bpl_sfor<4>([&](auto KC) {
          constexpr int k = decltype(KC)::value;
          constexpr int qbl = k & 1, ks = k >> 1;
          constexpr int kq = k >> 1, hp = k & 1;
          constexpr int kp = k > 0 ? k - 1 : 0, kqp = kp >> 1, hpp = kp & 1;
          if constexpr (STEADY) {
            float ne0, ne1;
            unsigned pp = 0u, ps = 0u;
            if (k == 0) bpl_dkv_step<true, false>(Sn[qbl], Dn[qbl], dv[iprev][k], dk[iprev][k], ne0, ne1, pe0, pe1, pp, ps, qa[qbl][0], kfr[inext][0], da[qbl][0], vfr[inext][0],
                                                  dot[k], bpl_frag(pfp), qt[k], bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], 0.f, 0.f, nl[qbl], nd[qbl]);
            else if (ks == 0) bpl_dkv_step<true, true>(Sn[qbl], Dn[qbl], dv[iprev][k], dk[iprev][k], ne0, ne1, pe0, pe1, pp, ps, qa[qbl][0], kfr[inext][0], da[qbl][0], vfr[inext][0],
                                                       dot[k], bpl_frag(pfp), qt[k], bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], Dc[kqp][2 * hpp], Dc[kqp][2 * hpp + 1],
                                                       nl[qbl], nd[qbl]);
            else bpl_dkv_step<false, true>(Sn[qbl], Dn[qbl], dv[iprev][k], dk[iprev][k], ne0, ne1, pe0, pe1, pp, ps, qa[qbl][1], kfr[inext][1], da[qbl][1], vfr[inext][1],
                                           dot[k], bpl_frag(pfp), qt[k], bpl_frag(dsp), Sc[kq][2 * hp], Sc[kq][2 * hp + 1], Dc[kqp][2 * hpp], Dc[kqp][2 * hpp + 1],
                                           nl[qbl], nd[qbl]);
            if (k > 0) { pfc[kp] = pp; dsc[kp] = ps; }
            pe0 = ne0;
            pe1 = ne1;
            if (k == 3) {
              unsigned pl, sl;
              bpl_dkv_tail(pe0, pe1, pl, sl, Dc[1][2], Dc[1][3]);
              pfc[3] = pl;
              dsc[3] = sl;
            }
            if (N >= 2 && i == N - 2 && !LAST) {      // Q / dO rows (and, behind the last step that starts from them, the statistics) of the next half
              rd_qd(hh + 1, qbl, ks);
              if (k == 1) { rd_st(hh + 1, 0); rd_st(hh + 1, 1); }
            }
            if (i == 0) rd_tr(hh, k);                 // dO^T / Q^T of this half: dV / dK (u - 1) from the next phase on
          } else {
            if (HAS_NEXT) {
              if (ks == 0) { bpl_mfma_c<false>(Sn[qbl], qa[qbl][0], kfr[inext][0], nl[qbl]); bpl_mfma_c<false>(Dn[qbl], da[qbl][0], vfr[inext][0], nd[qbl]); }
              else { bpl_mfma_v(Sn[qbl], qa[qbl][1], kfr[inext][1]); bpl_mfma_v(Dn[qbl], da[qbl][1], vfr[inext][1]); }
            }
            float e0, e1;
            bpl_exp2(e0, e1, Sc[kq][2 * hp], Sc[kq][2 * hp + 1]);
            if (HAS_PREV) { bpl_mfma_a(dv[iprev][k], dot[k], bpl_frag(pfp)); bpl_mfma_a(dk[iprev][k], qt[k], bpl_frag(dsp)); }
            pfc[k] = bpl_cvt(e0, e1);
            dsc[k] = bpl_cvt(e0 * Dc[kq][2 * hp], e1 * Dc[kq][2 * hp + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if (!STEADY) {
          if (N >= 2 && i == N - 2 && !LAST) {
#pragma unroll
            for (int qbl = 0; qbl < 2; ++qbl) {
              rd_st(hh + 1, qbl);
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) rd_qd(hh + 1, qbl, ks);
            }
          }
          if (i == 0) {
#pragma unroll
            for (int db = 0; db < 4; ++db) rd_tr(hh, db);
          }
        }
        pfp = pfc;
        dsp = dsc;
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  half(0, H0{}, std::true_type{}, std::false_type{});
  half(1, H1{}, std::false_type{}, std::false_type{});
  for (int t = 1; t < ntiles - 1; ++t) {
    half(2 * t, H0{}, std::false_type{}, std::false_type{});
    half(2 * t + 1, H1{}, std::false_type{}, std::false_type{});
  }
  half(nhalf - 2, H0{}, std::false_type{}, std::false_type{});
  half(nhalf - 1, H1{}, std::false_type{}, std::true_type{});
  if constexpr (N > 0) {
#pragma unroll
    for (int db = 0; db < 4; ++db) { bpl_mfma_a(dv[N - 1][db], dot[db], bpl_frag(pfp)); bpl_mfma_a(dk[N - 1][db], qt[db], bpl_frag(dsp)); }
    bpl_drain();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int key = (kb0 + 4 * i) * 16 + l16;
      if (key < p.Nk) {
        bf16* kr = p.dK + ((long)b * p.Nk + key) * p.lddk + h * HD;
        bf16* vr = p.dV + ((long)b * p.Nk + key) * p.lddv + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          bf16x4 a, c2;
#pragma unroll
          for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[i][db][r] * SCALE); c2[r] = (bf16)dv[i][db][r]; }
          *(bf16x4*)(kr + db * 16 + g * 4) = a;
          *(bf16x4*)(vr + db * 16 + g * 4) = c2;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ranges of G blocks in the flattened (pair, block) space, as attention_pl.hip: this wave's blocks f = f_lo + 4 j + wave of pair pr
template <int G, typename PH>
__device__ __forceinline__ void bpl_walk(const int c, const int BP, const int F, const int wave, PH&& phase) {
  const int f_lo = c * G, f_hi = min(F, f_lo + G);
  if (f_lo >= f_hi) return;
  const int pA = f_lo / BP, pB = (f_hi - 1) / BP;
  for (int pr = pA; pr <= pB; ++pr) {
    const int lo = max(f_lo, pr * BP), hi = min(f_hi, (pr + 1) * BP);
    int j0 = lo - f_lo - wave, j1 = hi - f_lo - wave;
    j0 = j0 > 0 ? (j0 + 3) >> 2 : 0;
    j1 = j1 > 0 ? (j1 + 3) >> 2 : 0;
    phase(pr, j1 - j0, f_lo + 4 * j0 + wave - pr * BP);
  }
}
__device__ __forceinline__ int bpl_xcd_chunk(const int i, const int n) {      // XCD x owns a contiguous range of the n chunks (bijective for any n)
  const int x = i & 7, q = n >> 3, r = n & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (i >> 3);
}

// one grid: ids [0, nkv8) = key-side ranges (nkv of them, padded to a multiple of 8 so that both parts see the same id -> XCD map), then the query side
template <int G>
__global__ __launch_bounds__(256, 1) void attn_bwd_pl_kernel(const AttnP p, const int KBP, const int FK, const int nkv, const int nkv8, const int QBP,
                                                              const int FQ, const int nq) {
  extern __shared__ __attribute__((aligned(16))) char bpl_smem[];
  bf16* sm = (bf16*)bpl_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = blockIdx.x;
  if (id < nkv8) {
    const int c = p.xcd ? bpl_xcd_chunk(id, nkv8) : id;
    if (c >= nkv) return;
    bpl_walk<G>(c, KBP, FK, wave, [&](int pr, int n, int blk0) {
      switch (n) {
        case 0: bpl_dkv_phase<0>(p, sm, pr, blk0, lane, wave); break;
        case 1: bpl_dkv_phase<1>(p, sm, pr, blk0, lane, wave); break;
        case 2: bpl_dkv_phase<2>(p, sm, pr, blk0, lane, wave); break;
        case 3: bpl_dkv_phase<3>(p, sm, pr, blk0, lane, wave); break;
        default: bpl_dkv_phase<4>(p, sm, pr, blk0, lane, wave); break;
      }
    });
  } else {
    id -= nkv8;
    const int c = p.xcd ? bpl_xcd_chunk(id, nq) : id;
    bpl_walk<G>(c, QBP, FQ, wave, [&](int pr, int n, int blk0) {
      switch (n) {
        case 0: bpl_dq_phase<0>(p, sm, pr, blk0, lane, wave); break;
        case 1: bpl_dq_phase<1>(p, sm, pr, blk0, lane, wave); break;
        case 2: bpl_dq_phase<2>(p, sm, pr, blk0, lane, wave); break;
        case 3: bpl_dq_phase<3>(p, sm, pr, blk0, lane, wave); break;
        default: bpl_dq_phase<4>(p, sm, pr, blk0, lane, wave); break;
      }
    });
  }
}

}  // namespace

bool attn_bwd_pl_applicable(const AttnP& p) {
  return p.Nk >= 256 && p.Nq >= 256 && p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.lddo % 8 == 0 && p.lddq % 4 == 0 && p.lddk % 4 == 0 &&
         p.lddv % 4 == 0 && !p.accumulate && p.LSE && p.Delta && p.dO && p.dQ && p.dK && p.dV;
}

// dQ, dK, dV of one attention problem; Delta must be in place (attn_delta_kernel or the producing GEMM's epilogue: launch_attn_bwd_fused handles that)
int launch_attn_bwd_pl(const AttnP& pin, hipStream_t st) {
  ARG_CHECK(attn_bwd_pl_applicable(pin), "attention (pipelined backward): Nq=%d Nk=%d does not fit", pin.Nq, pin.Nk);
  AttnP p = pin;
  p.xcd = KNOB(32) != 1;
  const int nbh = p.B * p.H;
  const int KBP = cdiv(p.Nk, 16), QBP = cdiv(p.Nq, 16), FK = KBP * nbh, FQ = QBP * nbh;
  // Ranges of 16 blocks: 4 per wave.  (Ranges of 20 = 5 per wave make 40 x 4096 rows whole rounds of 256 CUs, but 5 blocks per wave do not fit the
  // register file: hipcc spills 24 registers to scratch, and scratch stores are vector-memory operations that complete out of order with the
  // LDS-DMA loads the kernel's counted s_waitcnt vmcnt(N) stand for -- the dQ body then read tiles that had not landed: wrong dQ on every shape,
  // deterministically.  With 4 blocks nothing is spilled, and 640 + 640 workgroups = 2.5 + 2.5 rounds still measure 506 us against 512.)
  constexpr int G = 16;
  const int nkv = cdiv(FK, G), nkv8 = (nkv + 7) & ~7, nq = cdiv(FQ, G);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_pl_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, BPL_SMEM));
    attr_set = true;
  }
  // (a persistent form -- 256 workgroups walking the ranges -- measured 6 % slower: the loop around the phases costs the register allocation
  //  40 more spilled registers)
  hipLaunchKernelGGL(attn_bwd_pl_kernel<G>, dim3(nkv8 + nq), dim3(256), BPL_SMEM, st, p, KBP, FK, nkv, nkv8, QBP, FQ, nq);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
