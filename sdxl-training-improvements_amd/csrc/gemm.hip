// bf16 MFMA GEMM family for gfx950 (MI355X): Linear / conv3x3 forward, dgrad and wgrad.
//
// One workgroup = 256 threads = 4 waves (2x2), block tile 128x128, K-step 64.
// Each wave owns a 64x64 sub-tile = 4x4 fragments of v_mfma_f32_16x16x32_bf16.
// Operand tiles are staged global -> VGPR -> LDS (double-buffered LDS, next tile's global loads in
// flight under the current tile's MFMAs).  An operand whose reduction dimension is the
// contiguous one is read from LDS with ds_read_b128; an operand whose reduction dimension is the
// strided one (dgrad's weights, both wgrad operands) is read with the gfx950 transpose read
// ds_read_b64_tr_b16, so no transposed copies of weights or activations are ever materialised.
// The 3x3 convolution is an implicit GEMM: the gathered operand's rows are pixels and each K-step
// is (tap, channel chunk) with zero fill at the image border.
//
// MFMA operand slots: lane l holds slot (g = l>>4, j = 0..7) of row/col (l & 15).  The hardware
// pairs A slot (g,j) with B slot (g,j); both operands map slot (g,j) to k = 8g + j of the 32-deep
// step, so the two read paths (b128 / transpose) agree by construction.
// C/D layout: col = l & 15, row = 4*(l>>4) + reg.
#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define LDK 72    // K-contiguous tile:  [128][72] bf16 (144 B rows: 16-B aligned, de-phased banks)
#define LDN 136   // N-contiguous tile:  [64][136] bf16 (272 B rows)
#define TILE_BYTES 18432
#define LDC 132   // fp32 staging of the output tile [128][132]
#define GEMM_SMEM (4 * TILE_BYTES)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

// fragment read, K-contiguous tile: row r, k offset
__device__ __forceinline__ bf16x8 frag_kc(const bf16* tile, int r, int kofs) {
  return *(const bf16x8*)(tile + r * LDK + kofs);
}
// fragment read, N-contiguous tile via transpose read: 8 k-rows starting at krow, 16 columns at col0.
// Lane i of a 16-lane group supplies the address of 4 contiguous bf16 of row (i>>2), cols 4*(i&3)..;
// it receives column i of the 4x16 block (rows krow..krow+3), then the same for rows krow+4..+7.
__device__ __forceinline__ bf16x8 frag_nc(const bf16* tile, int krow, int col0, int lane16) {
  const bf16* p0 = tile + (krow + (lane16 >> 2)) * LDN + col0 + (lane16 & 3) * 4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LDN));
  union { s16x4 s[2]; bf16x8 v; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.v;
}

struct PixRow {  // decoded pixel of a gathered row
  int b, y, x, ok;
};

__device__ __forceinline__ PixRow decode_pix(int m, int Mlimit, int Hm, int Wm) {
  PixRow r;
  r.ok = m < Mlimit;
  int hw = Hm * Wm;
  r.b = m / hw;
  int rem = m - r.b * hw;
  r.y = rem / Wm;
  r.x = rem - r.y * Wm;
  return r;
}

// source pixel index (in pixels) for tap (dy,dx) or -1
__device__ __forceinline__ long gather_src(const PixRow& r, int dy, int dx, const GemmP& p) {
  int ys = r.y * p.sm + dy - 1;
  int xs = r.x * p.sm + dx - 1;
  if (!r.ok || ys < 0 || xs < 0) return -1;
  if (p.sd > 1) {
    if ((ys % p.sd) | (xs % p.sd)) return -1;
    ys /= p.sd;
    xs /= p.sd;
  }
  if (ys >= p.Hs || xs >= p.Ws) return -1;
  return ((long)r.b * p.Hs + ys) * p.Ws + xs;
}

template <int FORM, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;

  // reduction schedule
  int tap_fixed = 0, split = 0;
  if (FORM == GEMM_TN) {
    tap_fixed = blockIdx.z / p.splitk;
    split = blockIdx.z - tap_fixed * p.splitk;
  }
  const int ktiles_per_tap = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end;
  if (FORM == GEMM_TN) {
    int chunk = (ktiles_per_tap + p.splitk - 1) / p.splitk;
    kt_begin = split * chunk;
    kt_end = min(ktiles_per_tap, kt_begin + chunk);
  } else {
    kt_end = ktiles_per_tap * p.taps;
  }

  // ---- per-thread load descriptors (4 x 16-byte vectors per operand per K-step) ----
  // K-contiguous tile: vector id = v*256+tid -> row id>>3, k offset (id&7)*8
  // N-contiguous tile: vector id -> k row id>>4, column offset (id&15)*8
  PixRow arow[4];
  if (CONV && FORM != GEMM_TN) {
#pragma unroll
    for (int v = 0; v < 4; ++v) arow[v] = decode_pix(m0 + ((v * 256 + tid) >> 3), p.M, p.Hm, p.Wm);
  }

  bf16x8 ra[4], rb[4];

  auto load_tiles = [&](int kt) {
    int tap = 0, c0;
    if (FORM == GEMM_TN) {
      tap = tap_fixed;
      c0 = kt * BK;
    } else {
      tap = kt / ktiles_per_tap;
      c0 = (kt - tap * ktiles_per_tap) * BK;
    }
    const int dy = tap / 3, dx = tap - dy * 3;
    // ---------------- A ----------------
    if (FORM == GEMM_TN) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int id = v * 256 + tid;
        int kk = c0 + (id >> 4);
        int m = m0 + (id & 15) * 8;
        bool ok = kk < p.K && m < p.M;
        ra[v] = ok ? *(const bf16x8*)(p.A + (long)kk * p.lda + m) : zero8();
      }
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int id = v * 256 + tid;
        int c = c0 + (id & 7) * 8;
        long rowoff;
        bool ok;
        if (CONV) {
          long src = gather_src(arow[v], dy, dx, p);
          ok = src >= 0 && c < p.K;
          rowoff = src * p.lda;
        } else {
          int m = m0 + (id >> 3);
          ok = m < p.M && c < p.K;
          rowoff = (long)m * p.lda;
        }
        ra[v] = ok ? *(const bf16x8*)(p.A + rowoff + c) : zero8();
      }
    }
    // ---------------- B ----------------
    if (FORM == GEMM_NT) {
      const int wtap = p.flip ? (p.taps - 1 - tap) : tap;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int id = v * 256 + tid;
        int n = n0 + (id >> 3);
        int c = c0 + (id & 7) * 8;
        bool ok = n < p.N && c < p.K;
        rb[v] = ok ? *(const bf16x8*)(p.B + (long)n * p.ldb + (long)wtap * p.b_tap_stride + c) : zero8();
      }
    } else if (FORM == GEMM_NN) {
      const int wtap = p.flip ? (p.taps - 1 - tap) : tap;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int id = v * 256 + tid;
        int kk = c0 + (id >> 4);
        int n = n0 + (id & 15) * 8;
        bool ok = kk < p.K && n < p.N;
        rb[v] = ok ? *(const bf16x8*)(p.B + (long)kk * p.ldb + (long)wtap * p.b_tap_stride + n) : zero8();
      }
    } else {  // TN: rows are reduction pixels, optionally gathered
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int id = v * 256 + tid;
        int kk = c0 + (id >> 4);
        int n = n0 + (id & 15) * 8;
        long src;
        if (CONV) {
          PixRow r = decode_pix(kk, p.K, p.Hm, p.Wm);
          src = gather_src(r, dy, dx, p);
        } else {
          src = kk < p.K ? kk : -1;
        }
        bool ok = src >= 0 && n < p.N;
        rb[v] = ok ? *(const bf16x8*)(p.B + src * p.ldb + n) : zero8();
      }
    }
  };

  auto store_tiles = [&](int buf) {
    bf16* At = (bf16*)(smem + (buf * 2 + 0) * TILE_BYTES);
    bf16* Bt = (bf16*)(smem + (buf * 2 + 1) * TILE_BYTES);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      int id = v * 256 + tid;
      if (FORM == GEMM_TN)
        *(bf16x8*)(At + (id >> 4) * LDN + (id & 15) * 8) = ra[v];
      else
        *(bf16x8*)(At + (id >> 3) * LDK + (id & 7) * 8) = ra[v];
      if (FORM == GEMM_NT)
        *(bf16x8*)(Bt + (id >> 3) * LDK + (id & 7) * 8) = rb[v];
      else
        *(bf16x8*)(Bt + (id >> 4) * LDN + (id & 15) * 8) = rb[v];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (kt_begin < kt_end) {
    load_tiles(kt_begin);
    store_tiles(0);
  }
  __syncthreads();

  int buf = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) load_tiles(kt + 1);
    const bf16* At = (const bf16*)(smem + (buf * 2 + 0) * TILE_BYTES);
    const bf16* Bt = (const bf16*)(smem + (buf * 2 + 1) * TILE_BYTES);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (FORM == GEMM_TN)
          af[i] = frag_nc(At, ks * 32 + g * 8, wm * 64 + i * 16, l16);
        else
          af[i] = frag_kc(At, wm * 64 + i * 16 + l16, ks * 32 + g * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (FORM == GEMM_NT)
          bfr[j] = frag_kc(Bt, wn * 64 + j * 16 + l16, ks * 32 + g * 8);
        else
          bfr[j] = frag_nc(Bt, ks * 32 + g * 8, wn * 64 + j * 16, l16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: stage fp32 tile in LDS, then row-contiguous 16-byte stores ----
  float* Cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wm * 64 + i * 16 + g * 4 + r) * LDC + wn * 64 + j * 16 + l16] = acc[i][j][r];
  __syncthreads();

#pragma unroll
  for (int v = 0; v < 8; ++v) {
    int id = v * 256 + tid;
    int row = id >> 4, col = (id & 15) * 8;
    int m = m0 + row, n = n0 + col;
    if (m >= p.M || n >= p.N) continue;
    float x[8];
    {
      f32x4 a = *(const f32x4*)(Cs + row * LDC + col);
      f32x4 b = *(const f32x4*)(Cs + row * LDC + col + 4);
      x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3];
      x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
    }
    if (p.out_f32) {
      float* c = (float*)p.C + (long)m * p.ldc + (long)tap_fixed * p.c_tap_stride + n;
      if (p.splitk > 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(c + e, x[e]);
      } else if (p.accumulate) {
        f32x4 a = *(f32x4*)c, b = *(f32x4*)(c + 4);
        a[0] += x[0]; a[1] += x[1]; a[2] += x[2]; a[3] += x[3];
        b[0] += x[4]; b[1] += x[5]; b[2] += x[6]; b[3] += x[7];
        *(f32x4*)c = a;
        *(f32x4*)(c + 4) = b;
      } else {
        *(f32x4*)c = (f32x4){x[0], x[1], x[2], x[3]};
        *(f32x4*)(c + 4) = (f32x4){x[4], x[5], x[6], x[7]};
      }
    } else {
      if (p.bias) {
        bf16x8 bv = *(const bf16x8*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)bv[e];
      }
      if (p.rowvec) {
        bf16x8 tv = *(const bf16x8*)(p.rowvec + (long)(m / p.rows_per_batch) * p.ldv + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)tv[e];
      }
      if (p.resid) {
        bf16x8 rv = *(const bf16x8*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)rv[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)x[e];
      *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + n) = o;
    }
  }
}

void gemm_defaults(GemmP* p) {
  memset(p, 0, sizeof(*p));
  p->taps = 1;
  p->sm = 1;
  p->sd = 1;
  p->splitk = 1;
  p->rows_per_batch = 1;
}

template <int FORM, bool CONV>
static int launch_one(const GemmP& p, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_kernel<FORM, CONV>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SMEM));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<FORM, CONV>), grid, dim3(256), GEMM_SMEM, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---- optional per-launch timing (bench.py roofline): HIP events on the launch stream around every GEMM launch ----
#include <vector>
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  std::vector<double> flops;
};
static GemmProf g_prof;
int gemm_profile_begin() {
  g_prof.on = true;
  g_prof.used = 0;
  g_prof.flops.clear();
  return 0;
}
int gemm_profile_end(double* flops, double* ms, int* launches) {
  g_prof.on = false;
  double f = 0, t = 0;
  HIP_CHECK_RET(hipDeviceSynchronize());
  for (size_t i = 0; i < g_prof.flops.size(); ++i) {
    float e = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&e, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    t += e;
    f += g_prof.flops[i];
  }
  if (flops) *flops = f;
  if (ms) *ms = t;
  if (launches) *launches = (int)g_prof.flops.size();
  return 0;
}
static int launch_gemm_impl(const GemmP& pin, hipStream_t st);
int launch_gemm(const GemmP& p, hipStream_t st) {
  if (!g_prof.on) return launch_gemm_impl(p, st);
  while (g_prof.ev.size() < g_prof.used + 2) {
    hipEvent_t e;
    HIP_CHECK_RET(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  HIP_CHECK_RET(hipEventRecord(g_prof.ev[g_prof.used], st));
  int rc = launch_gemm_impl(p, st);
  HIP_CHECK_RET(hipEventRecord(g_prof.ev[g_prof.used + 1], st));
  g_prof.used += 2;
  g_prof.flops.push_back(2.0 * (double)p.M * (double)p.N * (double)p.K * (double)p.taps);
  return rc;
}

static int launch_gemm_impl(const GemmP& pin, hipStream_t st) {
  GemmP p = pin;
  ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  ARG_CHECK(p.N % 8 == 0, "gemm: N=%d must be a multiple of 8", p.N);
  ARG_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, "gemm: lda=%ld ldb=%ld must be multiples of 8", p.lda, p.ldb);
  ARG_CHECK(p.taps == 1 || p.taps == 9, "gemm: taps=%d", p.taps);
  ARG_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0,
            "gemm: operands must be 16-byte aligned");
  if (p.form == GEMM_TN) {
    ARG_CHECK(p.M % 8 == 0, "gemm TN: M=%d must be a multiple of 8", p.M);
    ARG_CHECK(p.out_f32, "gemm TN: only fp32 output is implemented");
    ARG_CHECK(p.ldc % 4 == 0, "gemm TN: ldc=%ld must be a multiple of 4", p.ldc);
  } else {
    ARG_CHECK(p.K % 8 == 0, "gemm: K=%d must be a multiple of 8", p.K);
    ARG_CHECK(!p.out_f32 && p.splitk == 1, "gemm NT/NN: bf16 output, no split-K");
    ARG_CHECK(p.ldc % 8 == 0, "gemm: ldc=%ld must be a multiple of 8", p.ldc);
    if (p.accumulate) { p.resid = (const bf16*)p.C; p.ldr = p.ldc; }
  }
  if (p.splitk < 1) p.splitk = 1;
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), p.form == GEMM_TN ? p.taps * p.splitk : 1);
  const bool conv = p.taps == 9;
  switch (p.form) {
    case GEMM_NT: return conv ? launch_one<GEMM_NT, true>(p, grid, st) : launch_one<GEMM_NT, false>(p, grid, st);
    case GEMM_NN: return conv ? launch_one<GEMM_NN, true>(p, grid, st) : launch_one<GEMM_NN, false>(p, grid, st);
    case GEMM_TN: return conv ? launch_one<GEMM_TN, true>(p, grid, st) : launch_one<GEMM_TN, false>(p, grid, st);
  }
  ARG_CHECK(false, "gemm: unknown form %d", p.form);
}
