// Flash attention (head_dim 64, no mask, softmax scale 1/8) forward + backward for gfx950.
//
// Orientation trick (wave64 MFMA 16x16x32): scores are produced TRANSPOSED, S^T = K . Q^T, so each
// lane owns one query column (lane & 15) and 4 consecutive keys per 16-key block.  Row max / sum
// are then an in-lane reduction plus two cross-lane steps (xor 16, 32), the bf16 P^T fragments feed
// the second MFMA (O^T = V^T . P^T) as its B operand without any data movement, and the running
// (m, l) live in the same lanes as the O^T columns they rescale.  V^T (and K^T / Q^T / dO^T in the
// backward) are read from row-major LDS tiles with the gfx950 transpose read ds_read_b64_tr_b16.
//
// Slot convention for a 32-deep MFMA step built from two 16-row score blocks (2t, 2t+1):
//   slot (g, j<4)  <-> row 32t + 4g + j         slot (g, j>=4) <-> row 32t + 16 + 4g + (j-4)
// used identically by the register operand (packed scores) and the transpose-read operand.
//
// Kernels: attn_fwd (O, LSE) ; attn_bwd_dq (also Delta = rowsum dO*O) ; attn_bwd_dkv (+ its split-query reduce).
#include "attn_tiles.h"

#include <type_traits>

// ------------------------------------------------------------------------------------------------
// forward: block = 128 queries (4 waves x 32), loop over 64-key tiles.
// Per tile and wave: 8 K fragments (ds_read_b128) -> 16 MFMAs (S^T) -> the 8 V^T fragments are requested (16 transpose
// reads, 32 VGPRs) BEFORE the softmax arithmetic so that their LDS latency hides under it -> 16 MFMAs (O^T) back to back.
// (The first version read each V^T fragment right before its two MFMAs: ~100 cycles of exposed LDS latency eight times
// per tile; PMC: 45 % of the wave cycles issue-stalled, profiles/r02d.)
// ------------------------------------------------------------------------------------------------
#if defined(ATTN_DIAG) && (ATTN_DIAG & 256)
__device__ unsigned long long g_attn_wg[4096 * 3];
extern "C" __attribute__((visibility("default"))) int sdxl_debug_attn_wg(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_wg), sizeof(unsigned long long) * 4096 * 3) == hipSuccess ? 0 : 2;
}
#endif
#if defined(ATTN_DIAG) && (ATTN_DIAG & 128)
__device__ unsigned long long g_attn_stamps[4 * 4 * 6];
extern "C" __attribute__((visibility("default"))) int sdxl_debug_attn_stamps(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : 2;
}
#endif
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnP p) {
#if defined(ATTN_DIAG) && (ATTN_DIAG & 256)
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * TILE_ELEMS];  // K0 V0 K1 V1
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, g = lane >> 4;
  int qblk, bh;
  attn_xcd_map(p, blockIdx.x, (p.Nq + 127) >> 7, qblk, bh);
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qblk * 128 + wave * 32;
  const bf16* Qb = p.Q + (long)b * p.Nq * p.ldq + h * HD;
  const bf16* Kb = p.K + (long)b * p.Nk * p.ldk + h * HD;
  const bf16* Vb = p.V + (long)b * p.Nk * p.ldv + h * HD;

  // Q^T B-operand fragments: [qb][ks], lane = query column, 8 contiguous d
  bf16x8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int q = q0 + qb * 16 + l16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[qb][ks] = q < p.Nq ? *(const bf16x8*)(Qb + (long)q * p.ldq + ks * 32 + g * 8) : z8();
  }
  f32x4 ot[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // running reference mrow, in the scaled log2 domain (s * c): a tile's score accumulators start from -mrow, so exp2's
  // argument is the MFMA result itself; 0 until the first tile sets it
  float mrow[2] = {0.f, 0.f}, lrow[2] = {0.f, 0.f};
  const float c = SCALE * LOG2E;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) { qf[qb][0] = scale8(qf[qb][0], c); qf[qb][1] = scale8(qf[qb][1], c); }   // Q^T * (scale * log2 e), once
#ifdef ATTN_DIAG   // diagnostics only (never defined in the product build; profiles/tools/build_diag_attn.sh): knock out one component of
  constexpr int dbg = ATTN_DIAG;   // the tile loop -- 1: no exp, 2: no S MFMAs, 4: no PV MFMAs, 8: no max / rescale logic, 32: no K fragment
#else                              // reads, 64: no V^T fragment reads (tile-dependent stand-ins: nothing becomes loop-invariant);
  constexpr int dbg = 0;           // 128: s_memtime per phase (attn_stamps.py); 256: start / end / placement per workgroup (attn_wg_trace.py)
#endif
#if defined(ATTN_DIAG) && (ATTN_DIAG & 128)
  unsigned long long acc_ph[6] = {0, 0, 0, 0, 0, 0}, ts_prev = 0;
#define ATTN_STAMP(ph) { unsigned long long ts_ = __builtin_amdgcn_s_memtime(); acc_ph[ph] += ts_ - ts_prev; ts_prev = ts_; }
#else
#define ATTN_STAMP(ph)
#endif

  const int ntiles = (p.Nk + 63) / 64;
  const TileSrc ksrc = tile_src(Kb, p.ldk, wave, lane), vsrc = tile_src(Vb, p.ldv, wave, lane);
  tile_dma(ksrc, 0, p.Nk, sm, wave);
  tile_dma(vsrc, 0, p.Nk, sm + TILE_ELEMS, wave);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) { landed(qf[qb][0]); landed(qf[qb][1]); }
  int buf = 0;
  // one key tile; MASK: the tile holds keys beyond Nk (only the last tile of a ragged sequence)
  // FIXED (t > 0 of the first pass): the reference stays where the first tile put it -- no per-tile maximum, no rescale test.
  // bf16 / fp32 carry 8 exponent bits, so P up to 2^100 loses nothing; what can go wrong is overflow (a later score more than
  // ~100 log2 units above the first tile's maximum), which shows in the row sums after the loop and sends the whole workgroup
  // through a second pass with the tracking form (rare; attention knock-outs: the maximum / rescale logic was 13 % of the tile).
  auto tile = [&](int t, auto MASKC, auto FIXEDC) {
    constexpr bool MASK = decltype(MASKC)::value;
    constexpr bool FIXED = decltype(FIXEDC)::value;
    ATTN_STAMP(5)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of tile t have landed ...
    ATTN_STAMP(0)
    __syncthreads();                                     // ... everyone's have; everyone is done with the other buffer
    ATTN_STAMP(1)
    if (t + 1 < ntiles) {
      tile_dma(ksrc, t + 1, p.Nk, sm + ((buf ^ 1) * 2) * TILE_ELEMS, wave);
      tile_dma(vsrc, t + 1, p.Nk, sm + ((buf ^ 1) * 2 + 1) * TILE_ELEMS, wave);
    }
    const bf16* Kt = sm + (buf * 2) * TILE_ELEMS;
    const bf16* Vt = Kt + TILE_ELEMS;
    // S^T[key][q] = K . Q^T
    bf16x8 kf[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (dbg & 32) {
        kf[kb][0] = qf[0][0]; kf[kb][1] = qf[1][1];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(unsigned*)&kf[kb][0]) : "s"(t));
        continue;
      }
      kf[kb][0] = ld_frag(Kt, kb * 16 + l16, g * 8);
      kf[kb][1] = ld_frag(Kt, kb * 16 + l16, 32 + g * 8);
    }
    f32x4 st[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x4 a = (f32x4){-mrow[qb], -mrow[qb], -mrow[qb], -mrow[qb]};
        if (dbg & 2) { a[0] = (float)kf[kb][0][0]; a[1] = (float)qf[qb][1][1]; a[2] = a[0] + 1.f; a[3] = a[1] - 1.f; }
        else {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][0], qf[qb][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][1], qf[qb][1], a, 0, 0, 0);
        }
        st[kb][qb] = a;
      }
    // V^T fragments for the second product: requested now, consumed after the softmax arithmetic
    bf16x8 vfr[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (dbg & 64) { vfr[db][t2] = qf[t2][db & 1]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(unsigned*)&vfr[db][t2]) : "s"(t)); }
        else vfr[db][t2] = tr_frag(Vt, t2, db * 16, l16, g);
      }
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(2)
    if (MASK) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = t * 64 + kb * 16 + g * 4 + r;
          if (key >= p.Nk) { st[kb][0][r] = -1e30f; st[kb][1][r] = -1e30f; }
        }
    }
    // online softmax per query column
    bf16x8 pf[2][2];  // [step t2][qb]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (!FIXED) {
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (!(dbg & 8)) mx = fmaxf(mx, st[kb][qb][r]); }
      if (!(dbg & 8)) mx = max_over_g(mx);
      // st holds s * c - mrow.  Lazy rescaling: the reference only moves when some query's tile maximum exceeds it by more
      // than 2^8 (p <= 256 until then: exact in fp32 sums, same relative precision in bf16) -- and in the first tile, which
      // sets it to that tile's maximum; afterwards that is rare, and the rescale (alpha on the accumulators, the shift of
      // this tile's scores) is skipped wave-uniformly
      if (t == 0 || __builtin_amdgcn_ballot_w64(mx > 8.f) != 0) {
        const float shift = t == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha2 = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-shift);
        mrow[qb] += shift;
        lrow[qb] *= alpha2;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          ot[db][qb][0] *= alpha2; ot[db][qb][1] *= alpha2; ot[db][qb][2] *= alpha2; ot[db][qb][3] *= alpha2;
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { st[kb][qb][0] -= shift; st[kb][qb][1] -= shift; st[kb][qb][2] -= shift; st[kb][qb][3] -= shift; }
      }
      }
      f32x2 ls2 = (f32x2){0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {      // two-wide fp32 arithmetic on the register pairs of the MFMA results
          const f32x2 arg = (f32x2){st[kb][qb][2 * hh], st[kb][qb][2 * hh + 1]};
          const f32x2 e = (dbg & 1) ? arg : (f32x2){__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
          st[kb][qb][2 * hh] = e[0]; st[kb][qb][2 * hh + 1] = e[1];
          ls2 += e;
        }
      const float ls = ls2[0] + ls2[1];
      lrow[qb] += ls;  // per-lane partial (reduced over g at the end)
      pf[0][qb] = pack8(st[0][qb], st[1][qb]);
      pf[1][qb] = pack8(st[2][qb], st[3][qb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(3)
    // O^T[d][q] += V^T . P^T
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          if (dbg & 4) ot[db][qb][0] += (float)vfr[db][t2][0] + (float)pf[t2][qb][1];
          else ot[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[db][t2], pf[t2][qb], ot[db][qb], 0, 0, 0);
    buf ^= 1;
  };
  const int nfull = p.Nk / 64;
#if defined(ATTN_DIAG) && (ATTN_DIAG & 128)
  ts_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz
#endif
  // first pass: tile 0 sets the reference (tracking form), the others keep it
  if (nfull > 0) tile(0, std::false_type{}, std::false_type{});
  for (int t = 1; t < nfull; ++t) tile(t, std::false_type{}, std::true_type{});
  if (nfull < ntiles) {
    if (nfull == 0) tile(0, std::true_type{}, std::false_type{});
    else tile(nfull, std::true_type{}, std::true_type{});
  }
  {
    // overflow check (NaN-safe): every per-lane partial row sum must be a moderate finite number
    const bool bad = !(lrow[0] < 1e30f) || !(lrow[1] < 1e30f);
    if (__syncthreads_or(bad ? 1 : 0)) {       // second pass, tracking form everywhere (all waves: the tiles are shared)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ot[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mrow[0] = mrow[1] = 0.f;
      lrow[0] = lrow[1] = 0.f;
      buf = 0;
      tile_dma(ksrc, 0, p.Nk, sm, wave);
      tile_dma(vsrc, 0, p.Nk, sm + TILE_ELEMS, wave);
      for (int t = 0; t < nfull; ++t) tile(t, std::false_type{}, std::false_type{});
      if (nfull < ntiles) tile(nfull, std::true_type{}, std::false_type{});
    }
  }
#if defined(ATTN_DIAG) && (ATTN_DIAG & 128)
  acc_ph[4] = __builtin_amdgcn_s_memrealtime() - rt0;
  if (lane == 0 && qblk < 4 && bh == 5)
    for (int i = 0; i < 6; ++i) g_attn_stamps[(qblk * 4 + wave) * 6 + i] = acc_ph[i];
#endif
  // finalize
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = sum_over_g(lrow[qb]);
    float inv = 1.f / l;
    int q = q0 + qb * 16 + l16;
    if (q < p.Nq) {
      bf16* orow = p.O + ((long)b * p.Nq + q) * p.ldo + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x4 o;
        o[0] = (bf16)(ot[db][qb][0] * inv); o[1] = (bf16)(ot[db][qb][1] * inv);
        o[2] = (bf16)(ot[db][qb][2] * inv); o[3] = (bf16)(ot[db][qb][3] * inv);
        *(bf16x4*)(orow + db * 16 + g * 4) = o;
      }
      if (g == 0 && p.LSE) p.LSE[(long)bh * p.Nq + q] = mrow[qb] * (1.f / LOG2E) + logf(l);      // natural log: mrow is in log2 units
    }
  }
#if defined(ATTN_DIAG) && (ATTN_DIAG & 256)
  if (threadIdx.x == 0) {
    const int id = blockIdx.x;
    if (id < 4096) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_attn_wg[id * 3] = wg_t0; g_attn_wg[id * 3 + 1] = __builtin_amdgcn_s_memrealtime(); g_attn_wg[id * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// dQ: same orientation as forward.  dQ^T[d][q] = scale * K^T . dS^T,  dS^T = P^T o (dP^T - Delta)
// Delta[bh][q] = sum_d dO[q][d] * O[q][d] is computed here from the dO fragments the kernel holds anyway (each lane
// has 16 of the row's 64 d; two cross-lane steps finish the sum) and stored for the dK/dV kernel that follows.
// ------------------------------------------------------------------------------------------------
// (bx, bh): the workgroup's 128-query block and (batch, head); sm: 4 tiles of LDS.  The body of attn_bwd_dq_kernel and of the dQ part of
// attn_bwd_fused_kernel.
// WRITE_DELTA: store Delta for a dK / dV kernel launched behind this one; false in the fused grid, where attn_delta_kernel wrote it before the
// launch and a second writer inside the launch would race with the dK / dV workgroups reading it (different summation order: different bits).
template <bool WRITE_DELTA>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnP& p, bf16* sm, const int bx, const int bh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, g = lane >> 4;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = bx * 128 + wave * 32;
  const bf16* Qb = p.Q + (long)b * p.Nq * p.ldq + h * HD;
  const bf16* dOb = p.dO + (long)b * p.Nq * p.lddo + h * HD;
  const bf16* Ob = p.O + (long)b * p.Nq * p.ldo + h * HD;
  const bf16* Kb = p.K + (long)b * p.Nk * p.ldk + h * HD;
  const bf16* Vb = p.V + (long)b * p.Nk * p.ldv + h * HD;

  bf16x8 qf[2][2], df[2][2];
  float lse2[2], delta[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int q = q0 + qb * 16 + l16;
    bool ok = q < p.Nq;
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qb][ks] = ok ? *(const bf16x8*)(Qb + (long)q * p.ldq + ks * 32 + g * 8) : z8();
      df[qb][ks] = ok ? *(const bf16x8*)(dOb + (long)q * p.lddo + ks * 32 + g * 8) : z8();
      bf16x8 of = ok ? *(const bf16x8*)(Ob + (long)q * p.ldo + ks * 32 + g * 8) : z8();
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)df[qb][ks][e] * (float)of[e];
    }
    dl = sum_over_g(dl);
    lse2[qb] = ok ? p.LSE[(long)bh * p.Nq + q] * LOG2E : 0.f;
    delta[qb] = dl;
    if (WRITE_DELTA && ok && g == 0) p.Delta[(long)bh * p.Nq + q] = dl;
  }
  f32x4 dq[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) dq[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float c = SCALE * LOG2E;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) { qf[qb][0] = scale8(qf[qb][0], c); qf[qb][1] = scale8(qf[qb][1], c); }   // as the forward: scores in the log2 domain

  const int ntiles = (p.Nk + 63) / 64;
  const TileSrc ksrc = tile_src(Kb, p.ldk, wave, lane), vsrc = tile_src(Vb, p.ldv, wave, lane);
  tile_dma(ksrc, 0, p.Nk, sm, wave);
  tile_dma(vsrc, 0, p.Nk, sm + TILE_ELEMS, wave);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) { landed(qf[qb][0]); landed(qf[qb][1]); landed(df[qb][0]); landed(df[qb][1]); landed(lse2[qb]); landed(delta[qb]); }
  int buf = 0;
  // one key tile.  Order: K, V fragments (16 b128 reads) -> 32 MFMAs (S^T, dP^T) -> the K^T transpose-read fragments of the third
  // product are requested BEFORE the exp / dS arithmetic (their LDS latency hides under it) -> 16 MFMAs (dQ^T) back to back.
  auto tile = [&](int t, auto MASKC) {
    constexpr bool MASK = decltype(MASKC)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) {
      tile_dma(ksrc, t + 1, p.Nk, sm + ((buf ^ 1) * 2) * TILE_ELEMS, wave);
      tile_dma(vsrc, t + 1, p.Nk, sm + ((buf ^ 1) * 2 + 1) * TILE_ELEMS, wave);
    }
    const bf16* Kt = sm + (buf * 2) * TILE_ELEMS;
    const bf16* Vt = Kt + TILE_ELEMS;
    bf16x8 kf[4][2], vf[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      kf[kb][0] = ld_frag(Kt, kb * 16 + l16, g * 8); kf[kb][1] = ld_frag(Kt, kb * 16 + l16, 32 + g * 8);
      vf[kb][0] = ld_frag(Vt, kb * 16 + l16, g * 8); vf[kb][1] = ld_frag(Vt, kb * 16 + l16, 32 + g * 8);
    }
    f32x4 st[4][2], dp[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x4 a = (f32x4){-lse2[qb], -lse2[qb], -lse2[qb], -lse2[qb]};      // exp2's argument s * c - LSE * log2 e straight from the matrix pipe
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][0], qf[qb][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][1], qf[qb][1], a, 0, 0, 0);
        st[kb][qb] = a;
        f32x4 d = (f32x4){-delta[qb], -delta[qb], -delta[qb], -delta[qb]};    // dP - Delta for free: the accumulator starts at -Delta
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb][0], df[qb][0], d, 0, 0, 0);      // (a lane's four scores share its query column)
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb][1], df[qb][1], d, 0, 0, 0);
        dp[kb][qb] = d;
      }
    bf16x8 ktr[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) ktr[db][t2] = tr_frag(Kt, t2, db * 16, l16, g);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 dsf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // two-wide fp32 arithmetic (v_pk_fma / v_pk_add / v_pk_mul) on the register pairs of the MFMA results
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x2 arg = (f32x2){st[kb][qb][2 * hh], st[kb][qb][2 * hh + 1]};
          f32x2 pr = (f32x2){__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
          if (MASK) {
            if (t * 64 + kb * 16 + g * 4 + 2 * hh >= p.Nk) pr[0] = 0.f;
            if (t * 64 + kb * 16 + g * 4 + 2 * hh + 1 >= p.Nk) pr[1] = 0.f;
          }
          const f32x2 ds = pr * (f32x2){dp[kb][qb][2 * hh], dp[kb][qb][2 * hh + 1]};   // (dp holds dP - Delta; softmax scale: once, on dQ, at the end)
          st[kb][qb][2 * hh] = ds[0]; st[kb][qb][2 * hh + 1] = ds[1];
        }
      dsf[0][qb] = pack8(st[0][qb], st[1][qb]);
      dsf[1][qb] = pack8(st[2][qb], st[3][qb]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          dq[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr[db][t2], dsf[t2][qb], dq[db][qb], 0, 0, 0);
    buf ^= 1;
  };
  const int nfull = p.Nk / 64;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < ntiles) tile(nfull, std::true_type{});
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int q = q0 + qb * 16 + l16;
    if (q < p.Nq) {
      bf16* row = p.dQ + ((long)b * p.Nq + q) * p.lddq + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x4 o;
        o[0] = (bf16)(dq[db][qb][0] * SCALE); o[1] = (bf16)(dq[db][qb][1] * SCALE);
        o[2] = (bf16)(dq[db][qb][2] * SCALE); o[3] = (bf16)(dq[db][qb][3] * SCALE);
        *(bf16x4*)(row + db * 16 + g * 4) = o;
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnP p) {
  set_wave_prio(p.prio);
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * TILE_ELEMS];
  int qblk, bh;
  attn_xcd_map(p, blockIdx.x, (p.Nq + 127) >> 7, qblk, bh);
  attn_bwd_dq_body<true>(p, sm, qblk, bh);
}

// ------------------------------------------------------------------------------------------------
// dK, dV: block = 64 keys (4 waves x 16), loop over 64-query tiles.  Scores un-transposed here:
// S[q][key] = Q . K^T so that each lane owns one key column; P / dS feed dV^T = dO^T . P and
// dK^T = Q^T . dS as B operands, dO^T / Q^T come from transpose reads of the row-major tiles.
// ------------------------------------------------------------------------------------------------
// KB = 16-key column blocks per wave: 2 for self attention (block = 128 keys; every Q / dO fragment read from LDS
// feeds two key blocks, halving the LDS traffic per MFMA), 1 for the split-query cross-attention form.
// (bx, bh, bz): the workgroup's key block, (batch, head) and query split; nbh = B * H; sm: 4 tiles + 512 bf16 of LDS.  The body of
// attn_bwd_dkv_kernel and of the dK / dV part of attn_bwd_fused_kernel.
template <int KB>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnP& p, bf16* sm, const int bx, const int bh, const int bz, const int nbh) {
  float (*sstat)[2][64] = (float (*)[2][64])(sm + 4 * TILE_ELEMS);                                   // [buf][lse2|delta][q]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, g = lane >> 4;
  const int b = bh / p.H, h = bh - b * p.H;
  const int key0 = bx * (64 * KB) + wave * (16 * KB) + l16;  // this lane's key column of block kb: key0 + 16 kb
  const bf16* Qb = p.Q + (long)b * p.Nq * p.ldq + h * HD;
  const bf16* dOb = p.dO + (long)b * p.Nq * p.lddo + h * HD;
  const bf16* Kb = p.K + (long)b * p.Nk * p.ldk + h * HD;
  const bf16* Vb = p.V + (long)b * p.Nk * p.ldv + h * HD;
  bool kok[KB];
  bf16x8 kf[KB][2], vf[KB][2];  // B operands: [k=d][col=key]
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = key0 + kb * 16;
    kok[kb] = key < p.Nk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[kb][ks] = kok[kb] ? scale8(*(const bf16x8*)(Kb + (long)key * p.ldk + ks * 32 + g * 8), SCALE * LOG2E) : z8();   // K * (scale * log2 e): scores in the log2 domain
      vf[kb][ks] = kok[kb] ? *(const bf16x8*)(Vb + (long)key * p.ldv + ks * 32 + g * 8) : z8();
    }
  }
  f32x4 dk[KB][4], dv[KB][4];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk[kb][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[kb][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  const int ntiles_all = (p.Nq + 63) / 64;
  const int per = (ntiles_all + p.qsplit - 1) / p.qsplit;
  const int t_begin = bz * per;
  const int ntiles = min(ntiles_all, t_begin + per);
  // LSE / Delta of a query tile go to LDS by DMA as well (waves 0 / 1, 4 bytes per lane): no VGPR-destination load is in
  // flight inside the tile loop, whose compiler-placed s_waitcnt vmcnt(0) would drain the tile DMAs
  auto load_stats = [&](int t, int bufi) {
    if (wave < 2) {
      const int q = t * 64 + lane;
      const float* src = (wave == 0 ? p.LSE : p.Delta) + (long)bh * p.Nq + q;
      lds_dma4_global(q < p.Nq ? (const void*)src : (const void*)g_attn_zero16, lds_addr_of(&sstat[bufi][wave][0]));
    }
  };
  const TileSrc qsrc = tile_src(Qb, p.ldq, wave, lane), dsrc = tile_src(dOb, p.lddo, wave, lane);
  tile_dma(qsrc, t_begin, p.Nq, sm, wave);
  tile_dma(dsrc, t_begin, p.Nq, sm + TILE_ELEMS, wave);
  load_stats(t_begin, 0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) { landed(kf[kb][0]); landed(kf[kb][1]); landed(vf[kb][0]); landed(vf[kb][1]); }
  int buf = 0;
  // one query tile.  Order: Q, dO fragments (16 b128 reads) -> S, dP (16 KB MFMAs) -> the dO^T / Q^T transpose-read fragments
  // of the first two d-blocks are requested BEFORE the exp / dS arithmetic -> dV, dK of those d-blocks while the fragments of
  // the other two d-blocks arrive (their registers are the ones S / dP just vacated).
  auto tile = [&](int t, auto MASKC) {
    constexpr bool MASK = decltype(MASKC)::value;       // the tile holds query rows beyond Nq
    const bool more = t + 1 < ntiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile t (DMA) and its statistics landed; everyone done with the other buffer
    if (more) {
      tile_dma(qsrc, t + 1, p.Nq, sm + ((buf ^ 1) * 2) * TILE_ELEMS, wave);
      tile_dma(dsrc, t + 1, p.Nq, sm + ((buf ^ 1) * 2 + 1) * TILE_ELEMS, wave);
      load_stats(t + 1, buf ^ 1);
    }
    const bf16* Qt = sm + (buf * 2) * TILE_ELEMS;
    const bf16* Dt = Qt + TILE_ELEMS;
    bf16x8 qa[4][2], da[4][2];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      qa[qb][0] = ld_frag(Qt, qb * 16 + l16, g * 8); qa[qb][1] = ld_frag(Qt, qb * 16 + l16, 32 + g * 8);
      da[qb][0] = ld_frag(Dt, qb * 16 + l16, g * 8); da[qb][1] = ld_frag(Dt, qb * 16 + l16, 32 + g * 8);
    }
    f32x4 s[KB][4], dp[KB][4];
    f32x4 ndl4[4], nl4[4];               // -Delta, -LSE * log2 e of the lane's four query rows per query block (the tile's statistics are in LDS)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const f32x4 dl = *(const f32x4*)&sstat[buf][1][qb * 16 + g * 4];
      const f32x4 l2 = *(const f32x4*)&sstat[buf][0][qb * 16 + g * 4];      // natural-log LSE
      ndl4[qb] = (f32x4){-dl[0], -dl[1], -dl[2], -dl[3]};
      nl4[qb] = l2 * (-LOG2E);
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        f32x4 a = nl4[qb];                                                  // exp2's argument s * c - LSE * log2 e straight from the matrix pipe
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qb][0], kf[kb][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qb][1], kf[kb][1], a, 0, 0, 0);
        s[kb][qb] = a;
        f32x4 d = ndl4[qb];                                                 // dP - Delta for free: the accumulator starts at -Delta
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[qb][0], vf[kb][0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[qb][1], vf[kb][1], d, 0, 0, 0);
        dp[kb][qb] = d;
      }
    bf16x8 dot[2][2], qt[2][2];       // [d-block within the half][t2]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) { dot[db][t2] = tr_frag(Dt, t2, db * 16, l16, g); qt[db][t2] = tr_frag(Qt, t2, db * 16, l16, g); }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 pf[KB][2], dsf[KB][2];
    // P = exp2(c S - log2e LSE), dS = P (dP - Delta): two-wide fp32 operations (v_pk_fma / v_pk_add / v_pk_mul) on the
    // register pairs the MFMA results arrive in.  Keys beyond Nk need no masking here: their K / V fragments are zero and
    // their P / dS ROWS only feed their own dK / dV rows, which are never stored.
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      f32x4 pr[4];
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x2 sv = (f32x2){s[kb][qb][2 * hh], s[kb][qb][2 * hh + 1]};
          f32x2 e = (f32x2){__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
          if (MASK) {
            if (t * 64 + qb * 16 + g * 4 + 2 * hh >= p.Nq) e[0] = 0.f;
            if (t * 64 + qb * 16 + g * 4 + 2 * hh + 1 >= p.Nq) e[1] = 0.f;
          }
          const f32x2 dpv = (f32x2){dp[kb][qb][2 * hh], dp[kb][qb][2 * hh + 1]};
          const f32x2 dsv = e * dpv;                          // (dpv = dP - Delta; the softmax scale is applied once, to dK, at the end)
          pr[qb][2 * hh] = e[0]; pr[qb][2 * hh + 1] = e[1];
          s[kb][qb][2 * hh] = dsv[0]; s[kb][qb][2 * hh + 1] = dsv[1];
        }
      }
      pf[kb][0] = pack8(pr[0], pr[1]);
      pf[kb][1] = pack8(pr[2], pr[3]);
      dsf[kb][0] = pack8(s[kb][0], s[kb][1]);
      dsf[kb][1] = pack8(s[kb][2], s[kb][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 dot2[2][2], qt2[2][2];     // d-blocks 2, 3: requested now, used after the first half's products
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) { dot2[db][t2] = tr_frag(Dt, t2, (db + 2) * 16, l16, g); qt2[db][t2] = tr_frag(Qt, t2, (db + 2) * 16, l16, g); }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          dv[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[db][t2], pf[kb][t2], dv[kb][db], 0, 0, 0);
          dk[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[db][t2], dsf[kb][t2], dk[kb][db], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          dv[kb][db + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot2[db][t2], pf[kb][t2], dv[kb][db + 2], 0, 0, 0);
          dk[kb][db + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt2[db][t2], dsf[kb][t2], dk[kb][db + 2], 0, 0, 0);
        }
    buf ^= 1;
  };
  {
    const int nfull_q = p.Nq / 64;                      // tiles before this index hold 64 valid query rows
    for (int t = t_begin; t < ntiles; ++t) {
      if (t < nfull_q) tile(t, std::false_type{});
      else tile(t, std::true_type{});
    }
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = key0 + kb * 16;
    if (p.qsplit > 1) {   // fp32 partials: part[z][bh][key][2][64]
      const int kvt = (p.Nk + 63) / 64;
      float* base = p.part + ((((long)bz * nbh + bh) * kvt * 64 + key) * 2) * 64;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        *(f32x4*)(base + db * 16 + g * 4) = dk[kb][db] * SCALE;
        *(f32x4*)(base + 64 + db * 16 + g * 4) = dv[kb][db];
      }
    } else if (kok[kb]) {
      bf16* kr = p.dK + ((long)b * p.Nk + key) * p.lddk + h * HD;
      bf16* vr = p.dV + ((long)b * p.Nk + key) * p.lddv + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x4 a, c2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = (bf16)(dk[kb][db][r] * SCALE); c2[r] = (bf16)dv[kb][db][r]; }
        *(bf16x4*)(kr + db * 16 + g * 4) = a;
        *(bf16x4*)(vr + db * 16 + g * 4) = c2;
      }
    }
  }
}

template <int KB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnP p) {
  set_wave_prio(p.prio);
  // ONE LDS object (a second one makes hipcc drain the LDS-DMA before every ds_read): Q0 dO0 Q1 dO1 | stats
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * TILE_ELEMS + 512];
  attn_bwd_dkv_body<KB>(p, sm, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// Delta[bh][q] = sum_d dO[q][d] * O[q][d] as a pass of its own (8 lanes per (row, head), 16-byte loads): what the fused backward
// launch needs before it starts, because there the dK / dV workgroups no longer run behind the dQ workgroups that used to produce it.
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnP p) {
  const long i = (long)blockIdx.x * 32 + (threadIdx.x >> 3);           // (b, q, h) triple, h fastest
  const int c = threadIdx.x & 7;
  const long total = (long)p.B * p.Nq * p.H;
  float s = 0.f;
  long bh = 0; int q = 0;
  if (i < total) {
    const int h = (int)(i % p.H);
    const long bq = i / p.H;
    q = (int)(bq % p.Nq);
    const int b = (int)(bq / p.Nq);
    bh = (long)b * p.H + h;
    const bf16x8 d = *(const bf16x8*)(p.dO + bq * p.lddo + h * HD + c * 8);
    const bf16x8 o = *(const bf16x8*)(p.O + bq * p.ldo + h * HD + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)d[e] * (float)o[e];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (i < total && c == 0) p.Delta[bh * p.Nq + q] = s;
}

// Self-attention backward as ONE grid: the dK / dV workgroups (128 keys each, the longer ones: first) and the dQ workgroups (128 queries
// each) of a launch fill the 512 slots together.  As two launches each left its last round partly empty -- 640 workgroups per kernel at
// N = 1024 (1.25 rounds), 1 280 at N = 4096 (2.5) -- and the second waited for the first; Delta comes from attn_delta_kernel.
template <int KB>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused_kernel(const AttnP p, const int ndkv, const int nkb, const int nqb) {
  set_wave_prio(p.prio);
  __shared__ __attribute__((aligned(16))) bf16 sm[4 * TILE_ELEMS + 512];
  const int id = blockIdx.x;
  int blk, bh;
  if (id < ndkv) {
    attn_xcd_map(p, id, nkb, blk, bh);
    attn_bwd_dkv_body<KB>(p, sm, blk, bh, 0, p.B * p.H);
  } else {
    attn_xcd_map(p, id - ndkv, nqb, blk, bh);      // (ndkv is a multiple of 8 whenever the map is on)
    attn_bwd_dq_body<false>(p, sm, blk, bh);
  }
}

// dK/dV = sum over query splits of the fp32 partials (fixed order), cast to bf16
__global__ void attn_dkv_reduce_kernel(const AttnP p) {
  const int kvt = (p.Nk + 63) / 64;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H * Nk * 2 * 16 (float4 groups)
  const long total = (long)p.B * p.H * p.Nk * 32;
  if (i >= total) return;
  const int v4 = (int)(i & 15), which = (int)((i >> 4) & 1);
  const long rk = i >> 5;
  const int key = (int)(rk % p.Nk);
  const long bh = rk / p.Nk;
  const int b = (int)(bh / p.H), h = (int)(bh - (long)b * p.H);
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.qsplit; ++z) {
    const float* src = p.part + ((((long)z * p.B * p.H + bh) * kvt * 64 + key) * 2 + which) * 64 + v4 * 4;
    f32x4 v = *(const f32x4*)src;
    a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
  }
  bf16x4 o;
  o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
  bf16* dst = which == 0 ? p.dK + ((long)b * p.Nk + key) * p.lddk + h * HD : p.dV + ((long)b * p.Nk + key) * p.lddv + h * HD;
  *(bf16x4*)(dst + v4 * 4) = o;
}

size_t attn_part_floats(int B, int H, int Nk, int qsplit) {
  return qsplit > 1 ? (size_t)qsplit * B * H * ((Nk + 63) / 64) * 64 * 2 * 64 : 0;
}
int attn_pick_qsplit(int B, int H, int Nq, int Nk) {
  long blocks = (long)((Nk + 63) / 64) * B * H;
  int qt = (Nq + 63) / 64;
  // ~160 workgroups (knob 36: the target, A/B runs; 640 until round 5): the 4096-query level-2 prompt attention then runs UNSPLIT -- 160 workgroups write
  // bf16 dK | dV directly, no fp32 partials (42 MB per attention), no reduce launch -- and the 16 384-query level splits in two; step -0.5 ms
  // (profiles/r05v_ab_cross_dkv_split.txt)
  int s = (int)((KNOB(36) > 0 ? KNOB(36) : 160) / blocks);
  if (s < 1) s = 1;
  if (s > qt / 2) s = qt / 2 > 0 ? qt / 2 : 1;
  if (s > 16) s = 16;
  return s;
}

static int check_attn(const AttnP& p) {
  ARG_CHECK(p.B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "attention: empty problem");
  ARG_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0,
            "attention: row strides must be multiples of 8 elements");
  return 0;
}

int launch_attn_fwd(const AttnP& p, hipStream_t st) {
  if (int e = check_attn(p)) return e;
#ifdef SDXL_DIAG      // the software-pipelined forward (attention_pl.hip): built, parity-green, measured at parity with this kernel -- diagnostics build only
  if (KNOB(33) == 2 && attn_fwd_pl_applicable(p)) return launch_attn_fwd_pl(p, st);
#endif
  return launch_attn_fwd_tiled(p, st);
}
int launch_attn_fwd_tiled(const AttnP& p, hipStream_t st) {
  if (int e = check_attn(p)) return e;
  if (FILE* f = launch_log()) { fprintf(f, "A,0,%d,%d,%d,%d\n", p.B, p.H, p.Nq, p.Nk); fflush(f); }
  AttnP q = p;
  q.xcd = (p.B * p.H) % 8 == 0 && KNOB(32) != 1;      // (knob 32 = 1: plain (block, pair) order, A/B runs)
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(p.Nq, 128) * p.B * p.H), dim3(256), 0, st, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int check_attn_bwd(const AttnP& p) {
  if (int e = check_attn(p)) return e;
  ARG_CHECK(p.O && p.dO && p.dQ && p.dK && p.dV && p.LSE && p.Delta, "attention bwd: missing buffers");
  ARG_CHECK(!p.accumulate, "attention bwd: accumulate not implemented");
  return 0;
}
// dQ (+ Delta, which the dK / dV kernel reads: launch this one first, on a stream the other is ordered behind)
int launch_attn_bwd_dq(const AttnP& p, hipStream_t st) {
  if (int e = check_attn_bwd(p)) return e;
  if (FILE* f = launch_log()) { fprintf(f, "A,1,%d,%d,%d,%d\n", p.B, p.H, p.Nq, p.Nk); fflush(f); }
  AttnP q = p;
  q.xcd = (p.B * p.H) % 8 == 0 && KNOB(32) != 1;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(p.Nq, 128) * p.B * p.H), dim3(256), 0, st, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_attn_bwd_dkv(const AttnP& p, hipStream_t st) {
  if (int e = check_attn_bwd(p)) return e;
  if (FILE* f = launch_log()) { fprintf(f, "A,2,%d,%d,%d,%d\n", p.B, p.H, p.Nq, p.Nk); fflush(f); }
  AttnP q = p;
  if (q.qsplit < 1 || !q.part) q.qsplit = 1;
  // two key blocks per wave (232 VGPRs, 2 workgroups per CU) when the 128-key workgroups fill the chip four times over; at
  // N = 1024 (640 of them: 1.25 rounds of 512 slots) the one-block variant (143 VGPRs, 3 per CU, 1 280 workgroups) is 4 % faster
  if (q.qsplit == 1 && (long)cdiv(p.Nk, 128) * p.B * p.H >= 1024)
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, dim3(cdiv(p.Nk, 128), p.B * p.H, 1), dim3(256), 0, st, q);
  else
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, dim3(cdiv(p.Nk, 64), p.B * p.H, q.qsplit), dim3(256), 0, st, q);
  if (q.qsplit > 1) {
    long total = (long)p.B * p.H * p.Nk * 32;
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, q);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// self-attention: Delta pass + one grid of dK / dV and dQ workgroups (see attn_bwd_fused_kernel)
int launch_attn_bwd_fused(const AttnP& p, hipStream_t st) {
  if (int e = check_attn_bwd(p)) return e;
  ARG_CHECK(p.lddo % 8 == 0 && p.ldo % 8 == 0, "attention bwd (fused): dO / O row strides must be multiples of 8 elements");
  if (FILE* f = launch_log()) { fprintf(f, "A,1,%d,%d,%d,%d\nA,2,%d,%d,%d,%d\n", p.B, p.H, p.Nq, p.Nk, p.B, p.H, p.Nq, p.Nk); fflush(f); }
  AttnP q = p;
  q.qsplit = 1;
  q.xcd = (p.B * p.H) % 8 == 0 && KNOB(32) != 1;
  const long total = (long)p.B * p.Nq * p.H;
  if (!p.delta_ready) hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(total, 32)), dim3(256), 0, st, q);
  // the software-pipelined backward (attention_bwd_pl.hip) for long sequences: 4096 x 4096 x 40 pairs 546 -> 506 us on identical buffers
  // (profiles/r06l_attn_bwd_ab.txt); at 1024 tokens (16 key tiles per range: its prologue and epilogue are a seventh of a workgroup's life,
  // and the two co-resident workgroups of this kernel hide theirs) it is 3 % behind and not used.  knob 35 = 1: never, = 2: wherever it applies.
  if (KNOB(35) != 1 && ((p.Nq >= 2048 && p.Nk >= 2048) || KNOB(35) == 2) && attn_bwd_pl_applicable(q)) return launch_attn_bwd_pl(q, st);
  const int nkb = cdiv(p.Nk, 128), nqb = cdiv(p.Nq, 128), nbh = p.B * p.H;
  hipLaunchKernelGGL(attn_bwd_fused_kernel<2>, dim3(nkb * nbh + nqb * nbh), dim3(256), 0, st, q, nkb * nbh, nkb, nqb);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// dQ, dK, dV.  Self-attention-shaped problems whose dK / dV kernel needs no query split go out as the fused grid (knob 20 = 1,
// diagnostics build: the two-launch form, A/B runs); short key sequences (cross attention) keep dQ, then the split dK / dV kernel + reduce.
int launch_attn_bwd(const AttnP& p, hipStream_t st) {
  const bool fused = (KNOB(20) != 1 || p.delta_ready) && (p.qsplit <= 1 || !p.part) && p.Nk >= 256 && p.lddo % 8 == 0 && p.ldo % 8 == 0;
  if (fused) return launch_attn_bwd_fused(p, st);
  if (int e = launch_attn_bwd_dq(p, st)) return e;
  return launch_attn_bwd_dkv(p, st);
}
