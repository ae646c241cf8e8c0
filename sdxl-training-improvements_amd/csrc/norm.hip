// GroupNorm(+SiLU) and LayerNorm, forward and backward, for token-major bf16 activations on gfx950.
// HBM-bound kernels: 16-byte (8 x bf16) loads, every thread owns a fixed 8-channel vector so the
// per-channel coefficients stay in registers; fp32 statistics with a per-group shift (first element
// of the group) so E[(x-K)^2]-E[x-K]^2 does not cancel when |mean| >> std (sigma up to 2e4 inputs).
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// GroupNorm
// ------------------------------------------------------------------------------------------------
// launch geometry shared by the row-streaming kernels: blockDim = vpr * rpi threads
struct GnGeom {
  int vpr;   // 8-channel vectors per row = C/8
  int rpi;   // rows per iteration in a block
  int threads;
  int chunks;  // grid.x : row chunks per sample
  int rows_per_chunk;
};
#define GN_MAX_CHUNKS 256
static GnGeom gn_geom(int HW, int C) {
  GnGeom g;
  g.vpr = C / 8;
  g.rpi = g.vpr >= 256 ? 1 : 256 / g.vpr;
  g.threads = g.vpr * g.rpi;
  int chunks = HW / (g.rpi * 8);
  if (chunks < 1) chunks = 1;
  if (chunks > GN_MAX_CHUNKS) chunks = GN_MAX_CHUNKS;
  g.rows_per_chunk = (HW + chunks - 1) / chunks;
  g.chunks = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
  return g;
}
size_t groupnorm_ws_floats(int B, int C, int G) {
  size_t fwd = (size_t)GN_MAX_CHUNKS * B * G * 2 + (size_t)B * C * 2;
  size_t bwd = (size_t)GN_MAX_CHUNKS * B * C * 2 + (size_t)B * C * 5;
  return fwd > bwd ? fwd : bwd;
}

// All reductions below are order-deterministic (no atomics on the activation path): a block reduces its rows
// through LDS in a fixed order and writes one partial per chunk; the finalize kernel sums chunks in order.
// Bitwise-reproducible statistics keep the bf16 activations (and so the loss) reproducible run to run.

// partial sums of (x-K), (x-K)^2 per (chunk, b, group) -> part[chunk][b][G][2]
__global__ void gn_stats_kernel(const bf16* __restrict__ x, float* __restrict__ part, int HW, int C, int G,
                                int vpr, int rpi, int rows_per_chunk) {
  extern __shared__ float sred[];  // [2][rpi][C]
  const int b = blockIdx.y, B = gridDim.y;
  const int vec = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int cpg = C / G;
  const bf16* xb = x + (long)b * HW * C;
  float K[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) K[e] = (float)xb[((vec * 8 + e) / cpg) * cpg];
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += rpi) {
    bf16x8 v = *(const bf16x8*)(xb + (long)r * C + vec * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float d = (float)v[e] - K[e];
      s1[e] += d;
      s2[e] += d * d;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sred[(long)rsub * C + vec * 8 + e] = s1[e];
    sred[(long)(rpi + rsub) * C + vec * 8 + e] = s2[e];
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c)
      for (int r = 0; r < rpi; ++r) { a += sred[(long)r * C + c]; q += sred[(long)(rpi + r) * C + c]; }
    float* o = part + (((long)blockIdx.x * B + b) * G + g) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// stats[b][g] = (mean, rstd).  One 64-lane block per (g, b): lane k sums chunks k, k+64, ... and a fixed-order shuffle
// tree finishes (order-deterministic, ~2 us instead of a serial walk over up to 256 chunk partials)
__global__ void gn_finalize_kernel(const bf16* __restrict__ x, const float* __restrict__ part, int chunks,
                                   float* __restrict__ stats, int HW, int C, int G, float eps) {
  const int g = blockIdx.x, b = blockIdx.y, B = gridDim.y;
  const int cpg = C / G;
  float a = 0.f, q = 0.f;
  for (int k = threadIdx.x; k < chunks; k += 64) {
    const float* pp = part + (((long)k * B + b) * G + g) * 2;
    a += pp[0];
    q += pp[1];
  }
  a = wave_sum(a);
  q = wave_sum(q);
  if (threadIdx.x == 0) {
    const float n = (float)HW * (float)cpg;
    const float K = (float)x[(long)b * HW * C + g * cpg];
    const float m1 = a / n, m2 = q / n;
    const float var = fmaxf(m2 - m1 * m1, 0.f);
    stats[((long)b * G + g) * 2] = K + m1;
    stats[((long)b * G + g) * 2 + 1] = rsqrtf(var + eps);
  }
}

// y = act(a*x + s) with a = rstd*gamma, s = beta - mean*a rebuilt per thread from the 2 statistics of its channels' group
template <bool SILU>
__global__ void gn_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ stats,
                                const bf16* __restrict__ gamma, const bf16* __restrict__ beta, int HW, int C, int G,
                                int vpr, int rpi, int rows_per_chunk) {
  const int b = blockIdx.y;
  const int vec = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int cpg = C / G;
  float a[8], s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vec * 8 + e, g = c / cpg;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    a[e] = rstd * (float)gamma[c];
    s[e] = (float)beta[c] - mean * a[e];
  }
  const bf16* xb = x + (long)b * HW * C;
  bf16* yb = y + (long)b * HW * C;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += rpi) {
    bf16x8 v = *(const bf16x8*)(xb + (long)r * C + vec * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float n = (float)v[e] * a[e] + s[e];
      o[e] = (bf16)(SILU ? silu_f(n) : n);
    }
    *(bf16x8*)(yb + (long)r * C + vec * 8) = o;
  }
}

int launch_groupnorm_fwd(const bf16* x, bf16* y, const bf16* gamma, const bf16* beta, float* stats, float* ws,
                         int B, int HW, int C, int G, float eps, int silu, hipStream_t st) {
  ARG_CHECK(C % 8 == 0 && C % G == 0 && G <= 64, "groupnorm: C=%d G=%d unsupported", C, G);
  ARG_CHECK(C / 8 <= 512, "groupnorm: C=%d too wide", C);
  GnGeom g = gn_geom(HW, C);
  float* part = ws;                                      // [chunks][B][G][2]
  size_t sh = sizeof(float) * 2 * g.rpi * C;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(g.chunks, B), dim3(g.threads), sh, st, x, part, HW, C, G, g.vpr, g.rpi,
                     g.rows_per_chunk);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(64), 0, st, x, part, g.chunks, stats, HW, C, G, eps);
  if (silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(g.chunks, B), dim3(g.threads), 0, st, x, y, stats, gamma, beta, HW, C,
                       G, g.vpr, g.rpi, g.rows_per_chunk);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(g.chunks, B), dim3(g.threads), 0, st, x, y, stats, gamma, beta, HW, C,
                       G, g.vpr, g.rpi, g.rows_per_chunk);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// backward pass 1: per (chunk,b,c)  A = sum dn, Bs = sum dn*xhat  -> part[chunk][b][C][2]
template <bool SILU>
__global__ void gn_bwd_reduce_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                     const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                     const float* __restrict__ stats, float* __restrict__ part, int HW, int C, int G,
                                     int vpr, int rpi, int rows_per_chunk) {
  extern __shared__ float sred[];  // [2][rpi][C]
  const int b = blockIdx.y, B = gridDim.y;
  const int vec = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int cpg = C / G;
  float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = vec * 8 + e, g = c / cpg;
    mean[e] = stats[((long)b * G + g) * 2];
    rstd[e] = stats[((long)b * G + g) * 2 + 1];
    ga[e] = (float)gamma[c];
    be[e] = (float)beta[c];
  }
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
  const bf16* xb = x + (long)b * HW * C;
  const bf16* db = dy + (long)b * HW * C;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += rpi) {
    bf16x8 v = *(const bf16x8*)(xb + (long)r * C + vec * 8);
    bf16x8 d = *(const bf16x8*)(db + (long)r * C + vec * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xh = ((float)v[e] - mean[e]) * rstd[e];
      float dn = (float)d[e];
      if (SILU) dn *= silu_grad_f(xh * ga[e] + be[e]);
      sa[e] += dn;
      sb[e] += dn * xh;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sred[(long)rsub * C + vec * 8 + e] = sa[e];
    sred[(long)(rpi + rsub) * C + vec * 8 + e] = sb[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < rpi; ++r) { a += sred[(long)r * C + c]; q += sred[(long)(rpi + r) * C + c]; }
    float* o = part + (((long)blockIdx.x * B + b) * C + c) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// The same partial sums WITHOUT LDS: the backward runs this kernel beside the side stream's weight-gradient workgroups, whose
// exclusive forms (wgrad256, conv_wgrad3: 157 KiB) leave no room for the 15 KiB reduction buffer of the kernel above -- it waited
// for CUs the GEMMs did not hold (57 us per launch in the step against 23 alone).  Thread = (8-channel vector, row lane) with the
// RPI row lanes of a vector ADJACENT in the wave (RPI a power of two <= 8): a wave's load covers RPI rows x 64 / RPI vectors (whole
// 128-byte lines at RPI = 8), the row lanes are folded by a fixed-order xor tree, lane 0 of each vector stores the chunk's partial.
template <bool SILU, int RPI>
__global__ void gn_bwd_reduce_nolds_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                           const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                           const float* __restrict__ stats, float* __restrict__ part, int HW, int C, int G,
                                           int rows_per_chunk) {
  const int b = blockIdx.y, B = gridDim.y;
  const int vec = threadIdx.x / RPI, rsub = threadIdx.x % RPI;
  const int cpg = C / G;
  float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = vec * 8 + e, g = c / cpg;
    mean[e] = stats[((long)b * G + g) * 2];
    rstd[e] = stats[((long)b * G + g) * 2 + 1];
    ga[e] = (float)gamma[c];
    be[e] = (float)beta[c];
  }
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
  const bf16* xb = x + (long)b * HW * C + vec * 8;
  const bf16* db = dy + (long)b * HW * C + vec * 8;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
#pragma unroll 2
  for (int r = r0 + rsub; r < r1; r += RPI) {
    bf16x8 v = *(const bf16x8*)(xb + (long)r * C);
    bf16x8 d = *(const bf16x8*)(db + (long)r * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xh = ((float)v[e] - mean[e]) * rstd[e];
      float dn = (float)d[e];
      if (SILU) dn *= silu_grad_f(xh * ga[e] + be[e]);
      sa[e] += dn;
      sb[e] += dn * xh;
    }
  }
#pragma unroll
  for (int off = 1; off < RPI; off <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sa[e] += __shfl_xor(sa[e], off, 64);
      sb[e] += __shfl_xor(sb[e], off, 64);
    }
  if (rsub == 0) {
    float* o = part + (((long)blockIdx.x * B + b) * C + vec * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) *(f32x4*)(o + 2 * e) = (f32x4){sa[e], sb[e], sa[e + 1], sb[e + 1]};
  }
}

// One block per (group, sample): fixed-order sum of the group's channels over the chunk partials, then -- all of a group's
// channels being here -- dgamma / dbeta, the two group sums and the per-(b, c) coefficients of dx = c1*dn + c3*x + c2.
// (Was two launches: a chunk sum over 64-channel blocks and a per-sample finalize; 46 of each per step.)
__global__ __launch_bounds__(1024) void gn_bwd_group_kernel(const float* __restrict__ part, int chunks, const bf16* __restrict__ gamma,
                                                            const float* __restrict__ stats, float* __restrict__ coef,
                                                            float* __restrict__ prow, int HW, int C, int G) {
  __shared__ float ps[1024][2];
  __shared__ float wa[128], wb[128];
  const int g = blockIdx.x, b = blockIdx.y, B = gridDim.y;
  const int cpg = C / G;                        // <= 128 (launcher check)
  const int nsl = 1024 / cpg;                   // chunk slices per channel
  const int cl = threadIdx.x % cpg, sl = threadIdx.x / cpg;
  const int c = g * cpg + cl;
  float a = 0.f, q = 0.f;
  if (sl < nsl)
    for (int k = sl; k < chunks; k += nsl) {
      const float* pp = part + (((long)k * B + b) * C + c) * 2;
      a += pp[0];
      q += pp[1];
    }
  if (sl < nsl) { ps[sl * cpg + cl][0] = a; ps[sl * cpg + cl][1] = q; }
  __syncthreads();
  float A = 0.f, Bs = 0.f, ga = 0.f;
  if (sl == 0) {
    for (int k = 0; k < nsl; ++k) { A += ps[k * cpg + cl][0]; Bs += ps[k * cpg + cl][1]; }
    // parameter gradients: this sample's row of partial sums [b][dgamma | dbeta][C], plain stores; the samples are added in a fixed order by
    // ln_param_reduce_kernel (bitwise reproducible; until round 6: B fp32 atomics per channel, whose order the hardware chose)
    prow[((long)b * 2) * C + c] = Bs;
    prow[((long)b * 2 + 1) * C + c] = A;
    ga = (float)gamma[c];
    wa[cl] = ga * A;
    wb[cl] = ga * Bs;
  }
  __syncthreads();
  if (sl == 0) {
    float S1 = 0.f, S2 = 0.f;
    for (int k = 0; k < cpg; ++k) { S1 += wa[k]; S2 += wb[k]; }
    const float n = (float)HW * (float)cpg;
    S1 /= n; S2 /= n;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    const float c1 = rstd * ga;
    const float c3 = -rstd * rstd * S2;
    const float c2 = -rstd * S1 - mean * c3;
    coef[((long)b * C + c) * 3] = c1;
    coef[((long)b * C + c) * 3 + 1] = c2;
    coef[((long)b * C + c) * 3 + 2] = c3;
  }
}

template <bool SILU, bool ACC>
__global__ void gn_bwd_apply_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                    const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                    const float* __restrict__ stats, const float* __restrict__ coef,
                                    bf16* dx, const bf16* addend, int HW, int C, int G, int vpr, int rpi,
                                    int rows_per_chunk) {
  const int b = blockIdx.y;
  const int vec = threadIdx.x % vpr, rsub = threadIdx.x / vpr;
  const int cpg = C / G;
  float a[8], s[8], c1[8], c2[8], c3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = vec * 8 + e, g = c / cpg;
    float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    a[e] = rstd * (float)gamma[c];
    s[e] = (float)beta[c] - mean * a[e];
    c1[e] = coef[((long)b * C + c) * 3];
    c2[e] = coef[((long)b * C + c) * 3 + 1];
    c3[e] = coef[((long)b * C + c) * 3 + 2];
  }
  const bf16* xb = x + (long)b * HW * C;
  const bf16* db = dy + (long)b * HW * C;
  bf16* ob = dx + (long)b * HW * C;
  const bf16* ab = addend + (long)b * HW * C;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += rpi) {
    bf16x8 v = *(const bf16x8*)(xb + (long)r * C + vec * 8);
    bf16x8 d = *(const bf16x8*)(db + (long)r * C + vec * 8);
    bf16x8 o;
    if (ACC) o = *(const bf16x8*)(ab + (long)r * C + vec * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xv = (float)v[e];
      float dn = (float)d[e];
      if (SILU) dn *= silu_grad_f(xv * a[e] + s[e]);
      float g = c1[e] * dn + c3[e] * xv + c2[e];
      if (ACC) g += (float)o[e];
      o[e] = (bf16)g;
    }
    *(bf16x8*)(ob + (long)r * C + vec * 8) = o;
  }
}

int launch_groupnorm_bwd(const bf16* x, const bf16* dy, const bf16* gamma, const bf16* beta, const float* stats,
                         bf16* dx, const bf16* addend, float* dgamma, float* dbeta, float* ws, int B, int HW, int C,
                         int G, int silu, hipStream_t st, float* prow) {
  const int accumulate = addend != nullptr;
  ARG_CHECK(C % 8 == 0 && C % G == 0 && G <= 64 && C / G <= 128, "groupnorm bwd: C=%d G=%d unsupported", C, G);
  GnGeom g = gn_geom(HW, C);
  float* part = ws;                                      // [chunks][B][C][2]
  float* coef = ws + (size_t)GN_MAX_CHUNKS * B * C * 2;  // [B][C][3]
  int red_chunks = g.chunks;
  if (KNOB(24) == 1) {
    // LDS-free reduce: RPI row lanes per vector (<= 512 threads), ~8 rows per thread but >= 512 blocks where the sample allows
    const int rpi = g.vpr * 8 <= 512 ? 8 : g.vpr * 4 <= 512 ? 4 : g.vpr * 2 <= 512 ? 2 : 1;
    int rows = 8 * rpi;
    while (rows > rpi && (long)cdiv(HW, rows) * B < 512) rows >>= 1;
    if (cdiv(HW, rows) > GN_MAX_CHUNKS) rows = cdiv(HW, GN_MAX_CHUNKS);
    red_chunks = cdiv(HW, rows);
#define GN_RED(S, R) hipLaunchKernelGGL((gn_bwd_reduce_nolds_kernel<S, R>), dim3(red_chunks, B), dim3(g.vpr * R), 0, st, x, dy, \
                                        gamma, beta, stats, part, HW, C, G, rows)
#define GN_RED_R(S) { if (rpi == 8) GN_RED(S, 8); else if (rpi == 4) GN_RED(S, 4); else if (rpi == 2) GN_RED(S, 2); else GN_RED(S, 1); }
    if (silu) GN_RED_R(true) else GN_RED_R(false)
#undef GN_RED_R
#undef GN_RED
  } else {
    size_t sh = sizeof(float) * 2 * g.rpi * C;
    if (silu)
      hipLaunchKernelGGL(gn_bwd_reduce_kernel<true>, dim3(g.chunks, B), dim3(g.threads), sh, st, x, dy, gamma, beta,
                         stats, part, HW, C, G, g.vpr, g.rpi, g.rows_per_chunk);
    else
      hipLaunchKernelGGL(gn_bwd_reduce_kernel<false>, dim3(g.chunks, B), dim3(g.threads), sh, st, x, dy, gamma, beta,
                         stats, part, HW, C, G, g.vpr, g.rpi, g.rows_per_chunk);
  }
  // per-sample partial rows of dgamma | dbeta: the caller's buffer (it folds them later: Engine::flush_ln_params), or the tail of ws + a reduce right here
  float* rows = prow ? prow : coef + (size_t)B * C * 3;
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(G, B), dim3(1024), 0, st, part, red_chunks, gamma, stats, coef, rows, HW, C, G);
  if (!prow) {
    LnRedBatch rb;
    rb.n = 1;
    rb.e[0].part = rows; rb.e[0].dgamma = dgamma; rb.e[0].dbeta = dbeta; rb.e[0].nblk = B; rb.e[0].C = C;
    if (int e = launch_ln_param_reduce(rb, st)) return e;
  }
#define GN_BWD_APPLY(S, A)                                                                                          \
  hipLaunchKernelGGL((gn_bwd_apply_kernel<S, A>), dim3(g.chunks, B), dim3(g.threads), 0, st, x, dy, gamma, beta, stats, \
                     coef, dx, addend, HW, C, G, g.vpr, g.rpi, g.rows_per_chunk)
  if (silu) { if (accumulate) GN_BWD_APPLY(true, true); else GN_BWD_APPLY(true, false); }
  else      { if (accumulate) GN_BWD_APPLY(false, true); else GN_BWD_APPLY(false, false); }
#undef GN_BWD_APPLY
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: 16 lanes per row, NV 16-byte vectors per lane (C = 128*NV), whole row in registers,
// exact two-pass mean/variance.
// ------------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  if (LPR == 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// LPR lanes per row (16, or 32 for wide rows: twice the waves in flight for the same bytes), NV vectors per lane
template <int NV, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const bf16* __restrict__ gamma,
                              const bf16* __restrict__ beta, float* __restrict__ stats, int M, float eps) {
  constexpr int C = NV * LPR * 8;
  const int sub = threadIdx.x & (LPR - 1);
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  if (row >= M) return;  // whole 16-lane group exits together
  bf16x8 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *(const bf16x8*)(x + (long)row * C + (i * LPR + sub) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
  }
  const float mean = group_sum<LPR>(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { float d = (float)v[i][e] - mean; sq += d * d; }
  const float rstd = rsqrtf(group_sum<LPR>(sq) * (1.f / C) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    bf16x8 gv = *(const bf16x8*)(gamma + (i * LPR + sub) * 8);
    bf16x8 bv = *(const bf16x8*)(beta + (i * LPR + sub) * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)(((float)v[i][e] - mean) * rstd * (float)gv[e] + (float)bv[e]);
    *(bf16x8*)(y + (long)row * C + (i * LPR + sub) * 8) = o;
  }
  if (sub == 0) { stats[(long)row * 2] = mean; stats[(long)row * 2 + 1] = rstd; }
}

int launch_layernorm_fwd(const bf16* x, bf16* y, const bf16* gamma, const bf16* beta, float* stats, int M, int C,
                         float eps, hipStream_t st) {
  ARG_CHECK(C % 128 == 0, "layernorm: C=%d must be a multiple of 128", C);
  dim3 blk(256);
  if (C % 256 == 0 && C >= 1024) {      // wide rows: 32 lanes per row
    dim3 grid(cdiv(M, 8));
    switch (C / 256) {
#define LN_CASE(NV) case NV: hipLaunchKernelGGL((ln_fwd_kernel<NV, 32>), grid, blk, 0, st, x, y, gamma, beta, stats, M, eps); break;
      LN_CASE(4) LN_CASE(5) LN_CASE(8) LN_CASE(10)
#undef LN_CASE
      default: ARG_CHECK(false, "layernorm: C=%d not instantiated", C);
    }
  } else {
    dim3 grid(cdiv(M, 16));
    switch (C / 128) {
#define LN_CASE(NV) case NV: hipLaunchKernelGGL((ln_fwd_kernel<NV, 16>), grid, blk, 0, st, x, y, gamma, beta, stats, M, eps); break;
      LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7)
#undef LN_CASE
      default: ARG_CHECK(false, "layernorm: C=%d not instantiated", C);
    }
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// backward: dx (same geometry as forward, whole row in registers) and -- PARAMS -- the dgamma / dbeta column sums of the
// rows this block handles, from the values it already holds (the separate column-sum pass re-read x and dy: 2.8 ms per
// step).  A block walks ITER groups of 256 / LPR rows; per 8-column vector the per-row products are summed over the rows of
// a wave with lane shuffles, over the 4 waves through LDS, and the block stores its 2 x C partial sums to part[block]
// (plain coalesced stores).  ln_param_reduce_kernel folds the partials of many LayerNorms into dgamma / dbeta later, off
// the critical path.  (One fp32 atomicAdd per column and block straight to dgamma / dbeta -- 655k device-scope atomics per
// launch at M = 4096, C = 1280, each a fabric transaction -- cost more than the whole dx pass: 16.3 us vs 7.0 us, r02
// profiles/tools/ln_bench.py.)
template <int NV, int LPR, bool ACC, bool PARAMS>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                 const bf16* __restrict__ gamma, const float* __restrict__ stats,
                                 bf16* dx, const bf16* addend, float* __restrict__ part, int M, int iters) {
  constexpr int C = NV * LPR * 8;
  constexpr int RPB = 256 / LPR;                     // rows per block and iteration
  __shared__ float red[PARAMS ? 4 : 1][2][PARAMS ? C : 8];
  const int sub = threadIdx.x & (LPR - 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (PARAMS) {   // each wave accumulates into its own LDS slice (lanes 0 .. LPR-1 own fixed columns: no conflicts, no barrier)
    if (lane < LPR) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          *(f32x4*)&red[wave][q][(i * LPR + sub) * 8] = (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f32x4*)&red[wave][q][(i * LPR + sub) * 8 + 4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  }
  for (int it = 0; it < iters; ++it) {
    const int row = (blockIdx.x * iters + it) * RPB + threadIdx.x / LPR;
    const bool ok = row < M;                      // (rows beyond M contribute zeros; the wave stays converged for the shuffles)
    const long ro = ok ? (long)row * C : 0;
    const float mean = ok ? stats[(long)row * 2] : 0.f, rstd = ok ? stats[(long)row * 2 + 1] : 0.f;
    bf16x8 v[NV], d[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = *(const bf16x8*)(x + ro + (i * LPR + sub) * 8);
      d[i] = *(const bf16x8*)(dy + ro + (i * LPR + sub) * 8);
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[i][e] = (bf16)0.f;
      }
      bf16x8 gv = *(const bf16x8*)(gamma + (i * LPR + sub) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float h = ((float)v[i][e] - mean) * rstd;
        float t = (float)d[i][e] * (float)gv[e];
        s1 += t;
        s2 += t * h;
      }
    }
    s1 = group_sum<LPR>(s1) * (1.f / C);
    s2 = group_sum<LPR>(s2) * (1.f / C);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      bf16x8 gv = *(const bf16x8*)(gamma + (i * LPR + sub) * 8);
      bf16x8 o;
      if (ACC) o = *(const bf16x8*)(addend + ro + (i * LPR + sub) * 8);
      float pg[8], pb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float h = ((float)v[i][e] - mean) * rstd;
        float t = (float)d[i][e] * (float)gv[e];
        float g = rstd * (t - s1 - h * s2);
        if (ACC) g += (float)o[e];
        o[e] = (bf16)g;
        pg[e] = (float)d[i][e] * h;
        pb[e] = (float)d[i][e];
      }
      if (ok) *(bf16x8*)(dx + ro + (i * LPR + sub) * 8) = o;
      if (PARAMS) {   // the rows of this wave (64 / LPR) -> lanes 0 .. LPR-1 -> the wave's LDS accumulators
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int off = LPR; off < 64; off <<= 1) {
            pg[e] += __shfl_xor(pg[e], off, 64);
            pb[e] += __shfl_xor(pb[e], off, 64);
          }
        }
        if (lane < LPR) {
          float* rg = &red[wave][0][(i * LPR + sub) * 8];
          float* rb = &red[wave][1][(i * LPR + sub) * 8];
          f32x4 a0 = *(f32x4*)rg, a1 = *(f32x4*)(rg + 4), b0 = *(f32x4*)rb, b1 = *(f32x4*)(rb + 4);
          a0[0] += pg[0]; a0[1] += pg[1]; a0[2] += pg[2]; a0[3] += pg[3];
          a1[0] += pg[4]; a1[1] += pg[5]; a1[2] += pg[6]; a1[3] += pg[7];
          b0[0] += pb[0]; b0[1] += pb[1]; b0[2] += pb[2]; b0[3] += pb[3];
          b1[0] += pb[4]; b1[1] += pb[5]; b1[2] += pb[6]; b1[3] += pb[7];
          *(f32x4*)rg = a0; *(f32x4*)(rg + 4) = a1; *(f32x4*)rb = b0; *(f32x4*)(rb + 4) = b1;
        }
      }
    }
  }
  if (!PARAMS) return;
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * 2 * C;          // [dgamma partial | dbeta partial]
  const float* r0 = &red[0][0][0];
  for (int c = threadIdx.x * 4; c < 2 * C; c += 1024) {
    f32x4 a = *(const f32x4*)(r0 + c), b = *(const f32x4*)(r0 + 2 * C + c), cc = *(const f32x4*)(r0 + 4 * C + c),
          d = *(const f32x4*)(r0 + 6 * C + c);
    *(f32x4*)(out + c) = (a + b) + (cc + d);
  }
}

// The critical path's lean form of the dx pass (no parameter sums, no LDS): the packed row stays in registers (8 NV VGPRs for x
// and dy), everything else is transient and the per-vector steps are fenced so that hipcc does not hoist all of their operand
// loads -- ~90 VGPRs instead of 166, so that two or three blocks fit on a CU beside a weight-gradient workgroup of the side
// stream (224 VGPRs, 73 KiB LDS) where the fused form (180 VGPRs + 40 KiB) fits once.
template <int NV, int LPR, bool ACC>
__global__ __launch_bounds__(256) void ln_bwd_lean_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                   const bf16* __restrict__ gamma, const float* __restrict__ stats,
                                   bf16* dx, const bf16* addend, int M, int prio) {
  constexpr int C = NV * LPR * 8;
  set_wave_prio(prio);
  const int sub = threadIdx.x & (LPR - 1);
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const bool ok = row < M;                      // (rows beyond M: zeros; the wave stays converged for the shuffles)
  const long ro = ok ? (long)row * C : 0;
  const float mean = ok ? stats[(long)row * 2] : 0.f, rstd = ok ? stats[(long)row * 2 + 1] : 0.f;
  bf16x8 v[NV], d[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *(const bf16x8*)(x + ro + (i * LPR + sub) * 8);
    d[i] = *(const bf16x8*)(dy + ro + (i * LPR + sub) * 8);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16x8 gv = *(const bf16x8*)(gamma + (i * LPR + sub) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float h = ((float)v[i][e] - mean) * rstd;
      const float t = ok ? (float)d[i][e] * (float)gv[e] : 0.f;
      s1 += t;
      s2 += t * h;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  s1 = group_sum<LPR>(s1) * (1.f / C);
  s2 = group_sum<LPR>(s2) * (1.f / C);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16x8 gv = *(const bf16x8*)(gamma + (i * LPR + sub) * 8);
    bf16x8 o;
    if (ACC) o = *(const bf16x8*)(addend + ro + (i * LPR + sub) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float h = ((float)v[i][e] - mean) * rstd;
      const float t = (float)d[i][e] * (float)gv[e];
      float g = rstd * (t - s1 - h * s2);
      if (ACC) g += (float)o[e];
      o[e] = (bf16)g;
    }
    if (ok) *(bf16x8*)(dx + ro + (i * LPR + sub) * 8) = o;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dgamma / dbeta += the block partials of up to LN_RED_MAX LayerNorm backward launches.  grid (64-column chunk of [dgamma | dbeta], row chunk,
// entry), block = 64 columns x 4 row slices: a thread sums the rows r = slice (mod 4) of its chunk of one column, the four slices are added in a
// fixed order through LDS, and ONE value goes into the gradient.  With one row chunk (the launcher's choice up to 1024 partial rows) the sum is
// bitwise reproducible; more chunks: one atomic each.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const LnRedBatch b) {
  __shared__ float sl[4][64];
  const LnRedEntry e = b.e[blockIdx.z];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  const int per = (e.nblk + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(e.nblk, r0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (c < 2 * e.C) {
    const float* p = e.part + (size_t)(r0 + slice) * 2 * e.C + c;
    int r = r0 + slice;
    for (; r + 4 < r1; r += 8, p += (size_t)16 * e.C) { s0 += p[0]; s1 += p[(size_t)8 * e.C]; }
    if (r < r1) s0 += p[0];
  }
  sl[slice][col] = s0 + s1;
  __syncthreads();
  if (slice == 0 && c < 2 * e.C && r1 > r0)
    atomicAdd((c < e.C ? e.dgamma : e.dbeta - e.C) + c, (sl[0][col] + sl[1][col]) + (sl[2][col] + sl[3][col]));
}

// backward, dgamma/dbeta: column-oriented (thread = fixed 8-column vector, 8 row lanes per block,
// 256 columns per block); the re-read of x/dy right after the dx kernel is served by L2/MALL.
// With x == nullptr this is a plain column sum (bias gradients): out_b[n] += sum_m dy[m][n].
__global__ __launch_bounds__(256) void col_reduce_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, long ld,
                                  const float* __restrict__ stats, float* __restrict__ out_g,
                                  float* __restrict__ out_b, int M, int C, int rows_per_chunk, long batch_stride,
                                  long out_ld, float* __restrict__ part) {
  __shared__ float sred[2][8][256 + 8];
  dy += blockIdx.z * batch_stride;       // batched plain column sums (x == nullptr): batch z = rows [z*M, z*M + M) -> out_b + z*out_ld
  out_b += blockIdx.z * out_ld;
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cv * 8;
  const bool cok = c0 < C;
  float sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  if (cok) {
    for (int r = r0 + rl; r < r1; r += 8) {
      bf16x8 d = *(const bf16x8*)(dy + (long)r * ld + c0);
      if (x) {
        bf16x8 v = *(const bf16x8*)(x + (long)r * ld + c0);
        const float mean = stats[(long)r * 2], rstd = stats[(long)r * 2 + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float dv = (float)d[e];
          sg[e] += dv * ((float)v[e] - mean) * rstd;
          sb[e] += dv;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[e] += (float)d[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sred[0][rl][cv * 8 + e] = sg[e]; sred[1][rl][cv * 8 + e] = sb[e]; }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    float g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { g += sred[0][k][threadIdx.x]; b += sred[1][k][threadIdx.x]; }
    if (part) {      // plain column sums as partial rows part[batch][chunk][C]: folded in a fixed order by ln_param_reduce_kernel (launch_colsum_f32_batched)
      part[((long)blockIdx.z * gridDim.y + blockIdx.y) * C + c] = b;
      return;
    }
    if (x) atomicAdd(&out_g[c], g);
    atomicAdd(&out_b[c], b);
  }
}

// dgamma | dbeta partial sums of one LayerNorm as PLAIN STORES: part[chunk][2][C], folded by ln_param_reduce_kernel for a whole batch of
// LayerNorms at the segment's end.  (The column-sum form above adds every block's 2 C sums with device-scope atomics: 128 chunks x 2 560
// columns = 327 k fabric atomics per LayerNorm, 30 us for a 21 MB pass that reads at 7 us -- profiles/r05m_ln_params.txt.)
// CV = 16-byte column vectors per block row: 32 (256 columns x 8 row lanes) or 16 (128 columns x 16 row lanes: C = 640 is 5 whole blocks).
template <int CV>
__global__ __launch_bounds__(256) void ln_param_partials_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                const float* __restrict__ stats, float* __restrict__ part, int M, int C,
                                                                int rows_per_chunk) {
  constexpr int RL = 256 / CV, W = CV * 8;
  __shared__ float sred[2][RL][W + 8];
  const int cv = threadIdx.x % CV, rl = threadIdx.x / CV;
  const int c0 = blockIdx.x * W + cv * 8;
  float sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  if (c0 < C) {
    for (int r = r0 + rl; r < r1; r += 4 * RL) {      // four rows in flight per thread
      bf16x8 d[4], v[4];
      float mean[4], rstd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = min(r + k * RL, M - 1);
        d[k] = *(const bf16x8*)(dy + (long)rr * C + c0);
        v[k] = *(const bf16x8*)(x + (long)rr * C + c0);
        mean[k] = stats[(long)rr * 2];
        rstd[k] = stats[(long)rr * 2 + 1];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (r + k * RL >= r1) break;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = (float)d[k][e];
          sg[e] += dv * ((float)v[k][e] - mean[k]) * rstd[k];
          sb[e] += dv;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sred[0][rl][cv * 8 + e] = sg[e]; sred[1][rl][cv * 8 + e] = sb[e]; }
  __syncthreads();
  const int c = blockIdx.x * W + threadIdx.x;
  if (threadIdx.x < W && c < C) {
    float g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < RL; ++k) { g += sred[0][k][threadIdx.x]; b += sred[1][k][threadIdx.x]; }
    float* o = part + (size_t)blockIdx.y * 2 * C;
    o[c] = g;
    o[C + c] = b;
  }
}
// chunks (= partial rows) of launch_layernorm_param_partials: ~1024 blocks, whole unrolled iterations per chunk, at most 256 chunks
static void ln_partials_geom(int M, int C, int* cv, int* colblocks, int* rows_per_chunk, int* chunks) {
  *cv = C % 256 == 0 ? 32 : 16;
  const int W = *cv * 8, quantum = 4 * (256 / *cv);
  *colblocks = cdiv(C, W);
  int want = 1024 / *colblocks;
  want = want < 1 ? 1 : (want > 256 ? 256 : want);
  int rpc = cdiv(M, want);
  rpc = cdiv(rpc, quantum) * quantum;
  *rows_per_chunk = rpc;
  *chunks = cdiv(M, rpc);
}
int layernorm_param_partial_rows(int M, int C) {
  int cv, cb, rpc, chunks;
  ln_partials_geom(M, C, &cv, &cb, &rpc, &chunks);
  return chunks;
}

static void col_reduce_geom(int M, int C, dim3* grid, int* rows_per_chunk) {
  int colblocks = cdiv(C, 256);
  int chunks = 1024 / colblocks;
  if (chunks < 1) chunks = 1;
  int maxchunks = cdiv(M, 32);
  if (chunks > maxchunks) chunks = maxchunks;
  *rows_per_chunk = cdiv(M, chunks);
  *grid = dim3(colblocks, cdiv(M, *rows_per_chunk));
}

// row groups of 256 / LPR rows per block: >= 512 blocks where the tensor allows (the partial sums are cheap to store)
static void ln_bwd_geometry(int M, int C, int* lpr, int* iters, int* nblk) {
  *lpr = (C % 256 == 0 && C >= 1024) ? 32 : 16;
  const int rpb = 256 / *lpr;
  int it = M / (rpb * 512);
  it = it < 1 ? 1 : (it > 4 ? 4 : it);
  *iters = it;
  *nblk = cdiv(M, rpb * it);
}
size_t layernorm_bwd_part_floats(int M, int C) {
  int lpr, iters, nblk;
  ln_bwd_geometry(M, C, &lpr, &iters, &nblk);
  const int rows = layernorm_param_partial_rows(M, C);      // (the parameter-gradient leaf pass writes its partial rows into the same buffer)
  return (size_t)(nblk > rows ? nblk : rows) * 2 * C;
}
// dx, and with `part` the per-block dgamma | dbeta partial sums (part[nblk][2][C], *nblk returned) in the same pass
int launch_layernorm_bwd(const bf16* x, const bf16* dy, const bf16* gamma, const float* stats, bf16* dx, const bf16* addend,
                         float* part, int* nblk_out, int M, int C, hipStream_t st) {
  ARG_CHECK(C % 128 == 0, "layernorm bwd: C=%d must be a multiple of 128", C);
  const bool accumulate = addend != nullptr, params = part != nullptr;
  int lpr, iters, nblk;
  ln_bwd_geometry(M, C, &lpr, &iters, &nblk);
  if (!params) { iters = 1; nblk = cdiv(M, 256 / lpr); }
  if (nblk_out) *nblk_out = nblk;
  dim3 blk(256), grid(nblk);
#define LN_LAUNCH(NV, LPR)                                                                                               \
  {                                                                                                                      \
    if (!params && KNOB(11) != 1) {                                                                                  \
      if (accumulate) hipLaunchKernelGGL((ln_bwd_lean_kernel<NV, LPR, true>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, M, KNOB(28));  \
      else hipLaunchKernelGGL((ln_bwd_lean_kernel<NV, LPR, false>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, M, KNOB(28));            \
    } else if (params) {                                                                                                        \
      if (accumulate) hipLaunchKernelGGL((ln_bwd_dx_kernel<NV, LPR, true, true>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, part, M, iters); \
      else hipLaunchKernelGGL((ln_bwd_dx_kernel<NV, LPR, false, true>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, part, M, iters);          \
    } else {                                                                                                             \
      if (accumulate) hipLaunchKernelGGL((ln_bwd_dx_kernel<NV, LPR, true, false>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, part, M, iters); \
      else hipLaunchKernelGGL((ln_bwd_dx_kernel<NV, LPR, false, false>), grid, blk, 0, st, x, dy, gamma, stats, dx, addend, part, M, iters);          \
    }                                                                                                                    \
  }
  if (lpr == 32) {
    switch (C / 256) {
      case 4: LN_LAUNCH(4, 32) break;
      case 5: LN_LAUNCH(5, 32) break;
      case 8: LN_LAUNCH(8, 32) break;
      case 10: LN_LAUNCH(10, 32) break;
      default: ARG_CHECK(false, "layernorm bwd: C=%d not instantiated", C);
    }
  } else {
    switch (C / 128) {
      case 1: LN_LAUNCH(1, 16) break;
      case 2: LN_LAUNCH(2, 16) break;
      case 3: LN_LAUNCH(3, 16) break;
      case 4: LN_LAUNCH(4, 16) break;
      case 5: LN_LAUNCH(5, 16) break;
      case 6: LN_LAUNCH(6, 16) break;
      case 7: LN_LAUNCH(7, 16) break;
      default: ARG_CHECK(false, "layernorm bwd: C=%d not instantiated", C);
    }
  }
#undef LN_LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_ln_param_reduce(const LnRedBatch& b, hipStream_t st) {
  if (b.n <= 0) return 0;
  ARG_CHECK(b.n <= LN_RED_MAX, "ln_param_reduce: %d entries", b.n);
  int cmax = 0, nmax = 0;
  for (int i = 0; i < b.n; ++i) {
    cmax = b.e[i].C > cmax ? b.e[i].C : cmax;
    nmax = b.e[i].nblk > nmax ? b.e[i].nblk : nmax;
  }
  // up to 1024 partial rows (the one-pass dx + partials form writes 512): one thread walks a column's rows in order and adds ONE value to the
  // gradient (bitwise reproducible); more: 16 row chunks, one atomic each
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(2 * cmax, 64), nmax <= 1024 ? 1 : 16, b.n), dim3(256), 0, st, b);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// dgamma[c] += sum_m dy[m][c] * xhat[m][c], dbeta[c] += sum_m dy[m][c] as a pass of its own (re-reads x, dy: a leaf, for the side stream)
int launch_layernorm_param_grads(const bf16* x, const bf16* dy, const float* stats, float* dgamma, float* dbeta, int M, int C, hipStream_t st) {
  ARG_CHECK(C % 8 == 0, "layernorm param grads: C=%d must be a multiple of 8", C);
  dim3 g2; int rpc;
  col_reduce_geom(M, C, &g2, &rpc);
  hipLaunchKernelGGL(col_reduce_kernel, g2, dim3(256), 0, st, x, dy, (long)C, stats, dgamma, dbeta, M, C, rpc, 0L, 0L, (float*)nullptr);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// the same sums as partial rows part[layernorm_param_partial_rows(M, C)][2][C] (plain stores; launch_ln_param_reduce folds them)
int launch_layernorm_param_partials(const bf16* x, const bf16* dy, const float* stats, float* part, int M, int C, hipStream_t st) {
  ARG_CHECK(C % 128 == 0, "layernorm param partials: C=%d must be a multiple of 128", C);
  int cv, cb, rpc, chunks;
  ln_partials_geom(M, C, &cv, &cb, &rpc, &chunks);
  if (cv == 32) hipLaunchKernelGGL(ln_param_partials_kernel<32>, dim3(cb, chunks), dim3(256), 0, st, x, dy, stats, part, M, C, rpc);
  else hipLaunchKernelGGL(ln_param_partials_kernel<16>, dim3(cb, chunks), dim3(256), 0, st, x, dy, stats, part, M, C, rpc);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
static void colsum_geom(int batches, int rows, int N, dim3* g2, int* rpc) {
  col_reduce_geom(rows, N, g2, rpc);
  if (batches > 1 && g2->y >= 2u * batches) {          // keep the total number of blocks
    *rpc *= batches;
    g2->y = cdiv(rows, *rpc);
  }
  g2->z = batches;
}
// floats of the partial rows launch_colsum_f32_batched wants for these sizes
size_t colsum_part_floats(int batches, int rows, int N) {
  dim3 g2; int rpc;
  colsum_geom(batches, rows, N, &g2, &rpc);
  return (size_t)batches * g2.y * N;
}
// out[b][n] += sum over the rows of batch b of x[b*rows + r][n], all batches in one launch (out row stride ld_out).  Bitwise reproducible:
// the blocks store partial rows part[b][chunk][N] (colsum_part_floats) and a launch of the fixed-order reduce adds them (an entry per batch:
// the two column halves play dgamma | dbeta) -- no atomics.  launch_colsum_partials leaves the reduce to the caller (`entries`[batches], to be
// handed to launch_ln_param_reduce on the same stream, any number of column sums per launch); launch_colsum_f32_batched reduces right away.
int launch_colsum_partials(const bf16* x, float* out, int batches, int rows, int N, long ldx, long ld_out, float* part, LnRedEntry* entries,
                           hipStream_t st) {
  ARG_CHECK(N % 8 == 0 && ldx % 8 == 0 && part != nullptr, "colsum: N=%d ld=%ld must be multiples of 8 (and the partial rows given)", N, ldx);
  dim3 g2; int rpc;
  colsum_geom(batches, rows, N, &g2, &rpc);
  hipLaunchKernelGGL(col_reduce_kernel, g2, dim3(256), 0, st, (const bf16*)nullptr, x, ldx, (const float*)nullptr,
                     (float*)nullptr, out, rows, N, rpc, (long)rows * ldx, ld_out, part);
  HIP_CHECK_RET(hipGetLastError());
  for (int b = 0; b < batches; ++b) {
    LnRedEntry& e = entries[b];
    e.part = part + (size_t)b * g2.y * N;
    e.dgamma = out + (size_t)b * ld_out;
    e.dbeta = e.dgamma + N / 2;
    e.nblk = (int)g2.y;
    e.C = N / 2;
  }
  return 0;
}
int launch_colsum_f32_batched(const bf16* x, float* out, int batches, int rows, int N, long ldx, long ld_out, float* part, hipStream_t st) {
  ARG_CHECK(batches >= 1 && batches <= LN_RED_MAX, "colsum: %d batches", batches);
  LnRedBatch rb;
  rb.n = batches;
  if (int e = launch_colsum_partials(x, out, batches, rows, N, ldx, ld_out, part, rb.e, st)) return e;
  return launch_ln_param_reduce(rb, st);
}
