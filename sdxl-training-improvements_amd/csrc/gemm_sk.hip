// Persistent stream-K bf16 MFMA GEMM for gfx950: 256 x 256 x 64 workgroup tiles (the 8-wave, two-phase main loop of
// gemm256.hip), ONE workgroup per CU for the whole launch, a work list of up to SK_MAX_PROB problems.
//
// Why: at B = 4, 1024^2 the level-2 linears are M x N = 4096 x 1280 outputs -- 80 tiles of 256 x 256 for 256 CUs, or
// 640 tiles (2.5 rounds) for the GEGLU projection.  A tile-per-workgroup launch either starves the chip (big tiles) or
// halves the arithmetic intensity per CU (128 x 160 tiles, one per CU: the L2 -> LDS stream then sets the pace).  Here
// the unit of work is one K-step of one tile ("iteration"): the launch's iterations -- over ALL its problems, e.g. the
// dgrad and the wgrad of one layer, which share dY -- are cut into `workers` equal contiguous ranges, so every CU
// multiplies the same number of 256 x 256 x 64 steps whatever the tile counts are, and *the launcher* decides what shares
// a CU (nothing does: 128 KiB LDS, 2 x 256 VGPRs per SIMD).
//
// A worker's range = [tail of a tile] [whole tiles ...] [head of a tile].  The worker that multiplies K-step 0 of a tile
// OWNS it (it does so at the END of its range); the workers that hold the rest of the tile do so at the START of theirs,
// write their fp32 partial accumulators write-through (sc1) to a per-worker 256 KiB slot and raise a flag; the owner
// polls the flags (one lane, relaxed, bounded), adds the partials IN WORKER ORDER (fixed partition => bit-reproducible)
// and runs the epilogue.  Protocol = the guide's publish / consume recipe R1 (sc1 payload + every wave drains + one flag
// store; consumer: relaxed poll, then sc1 loads).  Flags are reset by their (single) consumer, so back-to-back launches on
// one stream need no memset; every stream has its own slots (sk_workspace).
//
// Applicability: M, N multiples of 256, K of 64, no 3x3 gather, operands < 2 GiB (32-bit buffer offsets).
#include "gemm_tiles.h"

#include <map>

#define SK_MAX_PROB 4
#define SK_MAX_WORKERS 256

struct SkProb {
  const bf16* A;
  const bf16* B;
  void* C;
  const bf16* bias;
  const bf16* resid;
  const bf16* rowvec;
  bf16* aux;
  bf16* Cb;
  float* bias_grad;
  long ldc, ldr, ldv, ldaux;
  int form, M, N, K;
  int lda, ldb;
  int rows_per_batch, geglu, geglu_group, accumulate;
  float cb_scale;
  int tiles_m, tiles_n, ktiles;
  int it_begin;      // first iteration of this problem in the launch's iteration space
  int strip;         // tile order: strips of `strip` tile columns, n fastest inside a strip, then m
};
struct SkP {
  SkProb pr[SK_MAX_PROB];
  int nprob, total_its, workers;
  int its_base, its_rem;   // worker w multiplies iterations [w * base + min(w, rem), ... + base + (w < rem))
  float* ws;          // workers x 65536 floats
  unsigned* flags;    // workers words (0 = empty, 1 = partial published) + 1 error word at [SK_MAX_WORKERS]
};

namespace {

constexpr int REGION = 16384;          // 128 rows x 128 B (K-contiguous)  or  64 k-rows x 256 B (N-contiguous)
constexpr int BUFB = 4 * REGION;       // A-a0, A-a1, B-b0, B-b1
constexpr int SMEMSK = 2 * BUFB;       // 128 KiB
constexpr unsigned OOB = 0x80000000u;  // per-lane offset beyond num_records: the load returns zeros
constexpr int SLOT_FLOATS = 256 * 256;

template <int N>
struct IC { static constexpr int v = N; };

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(4))) SkP CSkP;
typedef const __attribute__((address_space(4))) SkProb CSkProb;
typedef __attribute__((address_space(1))) unsigned gu32;

// one segment: K-steps [k0, k1) of tile `tile` of problem pr, by logical worker w; it0 = the segment's first iteration
template <int FORM, bool BIAS>
__device__ __forceinline__ void sk_segment(CSkP* kp, CSkProb* pp, char* smem, int w, int tile, int k0, int k1) {
  // (fields of the problem are scalar-loaded from the kernarg segment where they are used: the main loop keeps only the
  //  operand descriptors live, the epilogue fetches its own)
  struct { const bf16 *A, *B; int lda, ldb, geglu, tiles_m, tiles_n, strip; float* bias_grad; } p;
  p.A = pp->A; p.B = pp->B; p.lda = pp->lda; p.ldb = pp->ldb; p.geglu = pp->geglu;
  p.tiles_m = pp->tiles_m; p.tiles_n = pp->tiles_n; p.strip = pp->strip;
  p.bias_grad = BIAS ? pp->bias_grad : nullptr;
  constexpr bool A_KC = FORM != GEMM_TN;   // A tile K-contiguous (rows = output rows)
  constexpr bool B_KC = FORM == GEMM_NT;   // B tile K-contiguous (rows = output columns)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l16 = lane & 15, g = lane >> 4;
  // tile -> (tm, tn)
  int tm, tn;
  {
    const int sw = p.strip, per = sw * p.tiles_m, full = (p.tiles_n / sw) * per;
    if (tile < full) {
      const int s = tile / per, r = tile - s * per;
      tm = r / sw;
      tn = s * sw + (r - tm * sw);
    } else {
      const int rw = p.tiles_n - (p.tiles_n / sw) * sw, r = tile - full;
      tm = r / rw;
      tn = (p.tiles_n / sw) * sw + (r - tm * rw);
    }
  }
  const int n0 = __builtin_amdgcn_readfirstlane(tn * 256), m0 = __builtin_amdgcn_readfirstlane(tm * 256);
  const bool gmap = p.geglu == 1;
  auto cbase = [&](int wcx, int b) { return gmap ? 128 * (wcx >> 1) + 32 * (wcx & 1) + 64 * b : 64 * wcx + 32 * b; };
  const int T = k1 - k0;

  // ---- LDS-DMA addressing: one per-lane byte offset per operand, scalar offsets per piece ----
  const i32x4 ra = make_srd(p.A, 0x7FFFFFFFu), rb = make_srd(p.B, 0x7FFFFFFFu);
  const int lda = p.lda, ldb = p.ldb;
  const unsigned lds_base = lds_addr_of(smem);
  unsigned voA, voB;
  {
    const int kc_row = lane >> 3, kc_vec = (lane & 7) ^ kc_row;
    const int nc_row = lane >> 4;
    const int F = nc_row | (((wave >> 1) & 1) << 2);
    const int c = (((lane & 15) ^ (F << 1)) << 3);
    voA = A_KC ? (unsigned)(kc_row * lda + kc_vec * 8) * 2u : (unsigned)(nc_row * lda + 128 * (c >> 6) + (c & 63)) * 2u;
    voB = B_KC ? (unsigned)(kc_row * ldb + kc_vec * 8) * 2u : (unsigned)(nc_row * ldb + 64 * (c >> 5) + (c & 31)) * 2u;
  }
  auto piece = [&](auto REG, int h, int t, int buf) {
    constexpr int R = decltype(REG)::v;
    constexpr int ab = R & 1;
    const bool live = t < T;
    const int kk = (k0 + t) * 64;
    const int q = wave + 8 * h;
    int so;
    if (R < 2) so = A_KC ? (m0 + 128 * h + 64 * ab + 8 * wave) * lda + kk : (kk + 4 * q) * lda + m0 + 64 * ab;
    else so = B_KC ? (n0 + cbase(2 * h + (wave >> 2), ab) + 8 * (wave & 3)) * ldb + kk : (kk + 4 * q) * ldb + n0 + 32 * ab;
    const unsigned dst = lds_base + buf * BUFB + R * REGION + q * 1024;
    const unsigned vo = live ? (R < 2 ? voA : voB) : OOB;
    lds_dma16_buffer(R < 2 ? ra : rb, vo, live ? (unsigned)(so * 2) : 0u, dst);
  };
  auto stage = [&](auto REG, int t, int buf) {
    piece(REG, 0, t, buf);
    piece(REG, 1, t, buf);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = BIAS && FORM == GEMM_TN && p.bias_grad != nullptr && tn == 0 && wc == 0;
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
        if (A_KC) fa[i][ks] = frag_kc<64>(R, 64 * wr + 16 * i + l16, ks * 4 + g);
        else fa[i][ks] = frag_nc<128>(R, ks * 32 + g * 8, 64 * wr + 16 * i, l16);
      }
  };
  auto read_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* R = smem + buf * BUFB + (2 + (j >> 1)) * REGION;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (B_KC) fb[j][ks] = frag_kc<64>(R, 32 * wc + 16 * (j & 1) + l16, ks * 4 + g);
        else fb[j][ks] = frag_nc<128>(R, ks * 32 + g * 8, 32 * wc + 16 * (j & 1), l16);
      }
    }
  };
  auto mma = [&](auto AQ, int t_next) {
    constexpr int a = decltype(AQ)::v;
    const int nbuf = t_next & 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[4 * a + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[4 * a + i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = ks * 4 + i;
        if (a == 0) {
          if (slot == 2) piece(IC<1>{}, 0, t_next, nbuf);
          if (slot == 5) piece(IC<1>{}, 1, t_next, nbuf);
        } else {
          if (slot == 0) piece(IC<0>{}, 0, t_next, nbuf);
          if (slot == 1) piece(IC<0>{}, 1, t_next, nbuf);
          if (slot == 2) piece(IC<2>{}, 0, t_next, nbuf);
          if (slot == 3) piece(IC<2>{}, 1, t_next, nbuf);
          if (slot == 4) piece(IC<3>{}, 0, t_next, nbuf);
          if (slot == 5) piece(IC<3>{}, 1, t_next, nbuf);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BIAS && do_bias) {
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

  // ---- prologue (every wave of the workgroup is past the previous segment's last LDS read: see the caller) ----
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

  for (int t = 0; t < T; ++t) {
    const int cb = t & 1;
    read_a(cb, IC<0>{});
    read_b(cb);
    wait_vmcnt<6>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(IC<0>{}, t + 1);
    post();
    read_a(cb, IC<1>{});
    wait_vmcnt<2>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(IC<1>{}, t + 2);
    post();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // both halves execute the same number of barriers
  wait_vmcnt<0>();

  const int kt = pp->ktiles;
  const int P = kp->workers;
  // bias gradient: column sums of this segment's K range (fp32 atomics, as the other wgrad kernels)
  if (BIAS && do_bias && g == 0) {
    const int mrow = m0 + 128 * wr + l16;
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(p.bias_grad + mrow + 16 * i, accb[BIAS ? i : 0][0]);
  }
  __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)kp->ws, 0, 0x7FFFFFFF, 0x00020000);
  if (k0 != 0) {
    // ---- contributor: fp32 partial tile -> this worker's slot (write-through), then the flag ----
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rws, tid * 16, w * (SLOT_FLOATS * 4) + (i * 4 + j) * 8192, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains
    __syncthreads();
    if (tid == 0) __hip_atomic_store((gu32*)kp->flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (k1 < kt) {
    // ---- owner: add the partials of the workers that hold the rest of this tile, in worker order ----
    const int itb = pp->it_begin;
    const int tile_end = itb + (tile + 1) * kt;
    int covered = itb + tile * kt + k1;
    const int base = kp->its_base, rem = kp->its_rem;
    for (int c = w + 1; covered < tile_end && c < P; ++c) {
      const int ce_ = covered + base + (c < rem ? 1 : 0);     // worker c's range starts where the previous one ended
      if (ce_ <= covered) continue;
      if (tid == 0) {
        gu32* f = (gu32*)kp->flags + c;
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > (1u << 21)) {     // ~2 s: give up loudly instead of hanging the device
            __hip_atomic_store((gu32*)kp->flags + SK_MAX_WORKERS, 1u + (unsigned)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rws, tid * 16, c * (SLOT_FLOATS * 4) + (i * 4 + j) * 8192, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 x = __builtin_bit_cast(f32x4, v[j]);
          acc[i][j][0] += x[0]; acc[i][j][1] += x[1]; acc[i][j][2] += x[2]; acc[i][j][3] += x[3];
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);     // at most two rows of loads in flight (registers)
      }
      __syncthreads();      // every wave's loads of the slot have returned (the adds above consumed them)
      if (tid == 0) __hip_atomic_store((gu32*)kp->flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      covered = ce_ < tile_end ? ce_ : tile_end;
    }
  }

  // ---- epilogue, registers -> global.  Lane (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3] of its wave tile ----
  const int mrow = m0 + 128 * wr + l16;   // + 16 i
  if (FORM == GEMM_TN) {
    struct { void* C; bf16* Cb; long ldc; int accumulate; float cb_scale; } p;
    p.C = pp->C; p.Cb = pp->Cb; p.ldc = pp->ldc; p.accumulate = pp->accumulate; p.cb_scale = pp->cb_scale;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long m = mrow + 16 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + cbase(wc, j >> 1) + 16 * (j & 1) + 4 * g;
        f32x4 x = acc[i][j];
        float* c = (float*)p.C + m * p.ldc + n;
        if (p.Cb) {           // final value straight to the bf16 exchange arena (C is read for the sum, not written)
          if (p.accumulate) {
            const f32x4 a = *(const f32x4*)c;
            x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3];
          }
          bf16x4 o;
          o[0] = (bf16)(x[0] * p.cb_scale); o[1] = (bf16)(x[1] * p.cb_scale);
          o[2] = (bf16)(x[2] * p.cb_scale); o[3] = (bf16)(x[3] * p.cb_scale);
          *(bf16x4*)(p.Cb + m * p.ldc + n) = o;
        } else if (p.accumulate) {
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
  struct { void* C; const bf16 *bias, *resid, *rowvec; bf16* aux; long ldc, ldr, ldv, ldaux; int geglu, geglu_group, rows_per_batch; } e;
  e.C = pp->C; e.bias = pp->bias; e.resid = pp->resid; e.rowvec = pp->rowvec; e.aux = pp->aux; e.ldc = pp->ldc; e.ldr = pp->ldr;
  e.ldv = pp->ldv; e.ldaux = pp->ldaux; e.geglu = pp->geglu; e.geglu_group = pp->geglu_group; e.rows_per_batch = pp->rows_per_batch;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results -> inline-asm VALU reads below
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
    if (e.geglu == 1) {   // forward GEGLU: b = 0 value columns, b = 1 the gate columns of the same channels
      const int nv = n0 + cbase(wc, 0) + cq, ng = n0 + cbase(wc, 1) + cq;
      if (e.bias) {
        const bf16x8 bv = *(const bf16x8*)(e.bias + nv), bg = *(const bf16x8*)(e.bias + ng);
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
      *(bf16x8*)((bf16*)e.C + m * e.ldc + nv) = ov;
      *(bf16x8*)((bf16*)e.C + m * e.ldc + ng) = og;
      *(bf16x8*)(e.aux + m * e.ldaux + (nv / (2 * e.geglu_group)) * e.geglu_group + nv % e.geglu_group) = oa;
    } else if (e.geglu == 2) {   // dgrad of the second feed-forward projection: dG -> dU (value and gate halves)
      const int G = e.geglu_group;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + cbase(wc, b) + cq;
        const long cu = (long)(n / G) * (2 * G) + (n % G);
        const bf16x8 ua = *(const bf16x8*)(e.aux + m * e.ldaux + cu);
        const bf16x8 ut = *(const bf16x8*)(e.aux + m * e.ldaux + cu + G);
        bf16x8 oa, ot;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = (float)(bf16)xq[b][e], tv = (float)ut[e];
          float cdf, pdf;
          gelu_cdf_pdf(tv, &cdf, &pdf);
          oa[e] = (bf16)(dv * tv * cdf);
          ot[e] = (bf16)(dv * (float)ua[e] * fmaf(tv, pdf, cdf));
        }
        *(bf16x8*)((bf16*)e.C + m * e.ldc + cu) = oa;
        *(bf16x8*)((bf16*)e.C + m * e.ldc + cu + G) = ot;
      }
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + cbase(wc, b) + cq;
        float* x = xq[b];
        if (e.bias) {
          const bf16x8 bv = *(const bf16x8*)(e.bias + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)bv[e];
        }
        if (e.rowvec) {
          const bf16x8 tv = *(const bf16x8*)(e.rowvec + (m / e.rows_per_batch) * e.ldv + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)tv[e];
        }
        if (e.resid) {
          const bf16x8 rv = *(const bf16x8*)(e.resid + m * e.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)rv[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)x[e];
        *(bf16x8*)((bf16*)e.C + m * e.ldc + n) = o;
      }
    }
  }
}

__global__ __launch_bounds__(512, 2) void gemm_sk_kernel(const SkP pin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the problem list is indexed dynamically: read it from the kernarg segment (scalar loads), not from a private copy
  CSkP* kp = (CSkP*)__builtin_amdgcn_kernarg_segment_ptr();
  const int P = kp->workers;
  const int b = blockIdx.x;
  if (b >= P) return;
  // logical worker: workgroup b runs on XCD b % 8 (observed; speed only) -- give every XCD a contiguous eighth of the
  // iteration space, so that neighbouring tiles (shared operand panels) and most hand-offs stay inside one L2
  const int w = (P & 7) == 0 ? (b & 7) * (P >> 3) + (b >> 3) : b;
  const int base = kp->its_base, rem = kp->its_rem;
  int it = w * base + (w < rem ? w : rem);
  const int it_end = it + base + (w < rem ? 1 : 0);
  const int nprob = kp->nprob;
  bool first = true;
  while (it < it_end) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < SK_MAX_PROB; ++i)
      if (i < nprob && it >= kp->pr[i].it_begin) q = i;
    CSkProb* pp = &kp->pr[q];
    const int kt = pp->ktiles;
    const int local = it - pp->it_begin;
    const int form = pp->form;
    const bool has_bias_grad = pp->bias_grad != nullptr;
    const int tile = __builtin_amdgcn_readfirstlane(local / kt);
    const int k0 = local - tile * kt;
    int k1 = k0 + (it_end - it);
    if (k1 > kt) k1 = kt;
    if (!first) __syncthreads();       // every wave is past the previous segment's LDS reads before the ring is restaged
    first = false;
    if (form == GEMM_NT) sk_segment<GEMM_NT, false>(kp, pp, smem, w, tile, k0, k1);
    else if (form == GEMM_NN) sk_segment<GEMM_NN, false>(kp, pp, smem, w, tile, k0, k1);
    else if (has_bias_grad) sk_segment<GEMM_TN, true>(kp, pp, smem, w, tile, k0, k1);
    else sk_segment<GEMM_TN, false>(kp, pp, smem, w, tile, k0, k1);
    it += k1 - k0;
  }
}

struct SkWs {
  float* ws = nullptr;
  unsigned* flags = nullptr;
};
std::map<hipStream_t, SkWs> g_skws;
int g_sk_workers = 0;      // > 0: forced worker count (microbenchmarks)
int g_sk_cus = 0;

int sk_workspace(hipStream_t st, SkWs* out) {
  auto it = g_skws.find(st);
  if (it == g_skws.end()) {
    SkWs w;
    HIP_CHECK_RET(hipMalloc((void**)&w.ws, (size_t)SK_MAX_WORKERS * SLOT_FLOATS * sizeof(float)));
    HIP_CHECK_RET(hipMalloc((void**)&w.flags, (SK_MAX_WORKERS + 64) * sizeof(unsigned)));
    HIP_CHECK_RET(hipMemset(w.flags, 0, (SK_MAX_WORKERS + 64) * sizeof(unsigned)));
    it = g_skws.emplace(st, w).first;
  }
  *out = it->second;
  return 0;
}

}  // namespace

void gemm_sk_set_workers(int n) { g_sk_workers = n; }

bool gemm_sk_applicable(const GemmP& p) {
  if (p.taps != 1 || p.group > 1) return false;
  if (p.M % 256 || p.N % 256 || p.K % 64) return false;
  if (p.lda % 8 || p.ldb % 8) return false;
  if (p.form == GEMM_TN) { if (!p.out_f32 || p.ldc % 4) return false; }
  else if (p.out_f32 || p.ldc % 8) return false;
  if (p.geglu == 1 && (p.geglu_group != 64 || p.form != GEMM_NT)) return false;
  if (p.geglu == 2 && (p.geglu_group % 8 || p.form != GEMM_NN)) return false;
  if (p.lda >= (1L << 30) || p.ldb >= (1L << 30)) return false;
  const long abytes = 2 * (p.form == GEMM_TN ? (long)p.K * p.lda : (long)p.M * p.lda);
  const long bbytes = 2 * (p.form == GEMM_NT ? (long)p.N * p.ldb : (long)p.K * p.ldb);
  if (abytes >= (1L << 31) || bbytes >= (1L << 31)) return false;
  return true;
}

// worker count for a launch of `total` iterations whose shortest tile has `kt_min` K-steps and `tiles` tiles in all:
// every CU unless the ranges would get shorter than a few K-steps (the hand-off then costs more than it balances)
int gemm_sk_pick_workers(long total, int tiles, int kt_min, int cus) {
  if (g_sk_workers > 0) return g_sk_workers < total ? g_sk_workers : (int)total;
  long P = cus;
  if (tiles <= P && (long)tiles * 2 > P) {
    // a whole tile per worker fills more than half the chip: compare one tile per worker with an even split
    const double dp = (double)kt_min + 6.0, sk = (double)total / P + 10.0;      // K-steps + prologue / hand-off, in K-step units
    if (dp <= sk) return tiles;
  }
  while (P > 1 && total / P < 4) P /= 2;
  if (P > total) P = total;
  return (int)P;
}

int launch_gemm_sk(const GemmP* gs, int n, hipStream_t st) {
  ARG_CHECK(n >= 1 && n <= SK_MAX_PROB, "gemm_sk: %d problems (1..%d)", n, SK_MAX_PROB);
  if (!g_sk_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipGetDeviceProperties(&prop, dev));
    g_sk_cus = prop.multiProcessorCount > SK_MAX_WORKERS ? SK_MAX_WORKERS : prop.multiProcessorCount;
  }
  SkP s;
  memset(&s, 0, sizeof(s));
  long total = 0;
  int tiles_all = 0, kt_min = 1 << 30;
  for (int i = 0; i < n; ++i) {
    GemmP p = gs[i];
    ARG_CHECK(gemm_sk_applicable(p), "gemm_sk: problem %dx%dx%d (form %d) does not fit the stream-K kernel", p.M, p.N, p.K, p.form);
    if (p.form != GEMM_TN && p.accumulate) { p.resid = (const bf16*)p.C; p.ldr = p.ldc; }
    SkProb& q = s.pr[i];
    q.A = p.A; q.B = p.B; q.C = p.C;
    q.bias = p.bias; q.resid = p.resid; q.rowvec = p.rowvec; q.aux = p.aux; q.Cb = p.Cb; q.bias_grad = p.bias_grad;
    q.ldc = p.ldc; q.ldr = p.ldr; q.ldv = p.ldv; q.ldaux = p.ldaux;
    q.form = p.form; q.M = p.M; q.N = p.N; q.K = p.K;
    q.lda = (int)p.lda; q.ldb = (int)p.ldb;
    q.rows_per_batch = p.rows_per_batch > 0 ? p.rows_per_batch : 1;
    q.geglu = p.geglu; q.geglu_group = p.geglu_group ? p.geglu_group : 64; q.accumulate = p.accumulate;
    q.cb_scale = p.cb_scale;
    q.tiles_m = p.M / 256; q.tiles_n = p.N / 256; q.ktiles = p.K / 64;
    q.it_begin = (int)total;
    q.strip = q.tiles_n < 8 ? q.tiles_n : 8;
    total += (long)q.tiles_m * q.tiles_n * q.ktiles;
    tiles_all += q.tiles_m * q.tiles_n;
    if (q.ktiles < kt_min) kt_min = q.ktiles;
  }
  ARG_CHECK(total < (1L << 30), "gemm_sk: too many iterations");
  s.nprob = n;
  s.total_its = (int)total;
  s.workers = gemm_sk_pick_workers(total, tiles_all, kt_min, g_sk_cus);
  s.its_base = (int)(total / s.workers);
  s.its_rem = (int)(total % s.workers);
  SkWs ws;
  { int rc = sk_workspace(st, &ws); if (rc) return rc; }
  s.ws = ws.ws;
  s.flags = ws.flags;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEMSK));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_sk_kernel, dim3(s.workers), dim3(512), SMEMSK, st, s);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// error word of the stream's hand-off state: 0 = no owner ever gave up waiting for a partial tile
int gemm_sk_error(hipStream_t st, unsigned* out) {
  *out = 0;
  auto it = g_skws.find(st);
  if (it == g_skws.end()) return 0;
  HIP_CHECK_RET(hipMemcpy(out, it->second.flags + SK_MAX_WORKERS, sizeof(unsigned), hipMemcpyDeviceToHost));
  return 0;
}
