// Elementwise / data-movement kernels (HBM-bound): 16-byte vectors, grid-stride, fp32 math.
#include "kernels.h"

#define EW_BLOCK 256
static inline int ew_grid(long nvec) {
  long g = (nvec + EW_BLOCK - 1) / EW_BLOCK;
  if (g > 4096) g = 4096;  // 16 blocks per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}
#define VEC_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ SiLU / add (flat, n % 8 == 0)
__global__ void silu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long nvec) {
  VEC_LOOP(i, nvec) {
    bf16x8 v = *(const bf16x8*)(x + i * 8), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)silu_f((float)v[e]);
    *(bf16x8*)(y + i * 8) = o;
  }
}
template <bool ACC>
__global__ void silu_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* dx, const bf16* addend,
                                long nvec) {
  VEC_LOOP(i, nvec) {
    bf16x8 v = *(const bf16x8*)(x + i * 8), d = *(const bf16x8*)(dy + i * 8), o;
    if (ACC) o = *(const bf16x8*)(addend + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float g = (float)d[e] * silu_grad_f((float)v[e]);
      if (ACC) g += (float)o[e];
      o[e] = (bf16)g;
    }
    *(bf16x8*)(dx + i * 8) = o;
  }
}
__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o, long nvec) {
  VEC_LOOP(i, nvec) {
    bf16x8 x = *(const bf16x8*)(a + i * 8), y = *(const bf16x8*)(b + i * 8), r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)x[e] + (float)y[e]);
    *(bf16x8*)(o + i * 8) = r;
  }
}
int launch_silu_fwd(const bf16* x, bf16* y, long n, hipStream_t st) {
  ARG_CHECK(n % 8 == 0, "silu: n=%ld", n);
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, st, x, y, n / 8);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_silu_bwd(const bf16* x, const bf16* dy, bf16* dx, const bf16* addend, long n, hipStream_t st) {
  ARG_CHECK(n % 8 == 0, "silu: n=%ld", n);
  if (addend) hipLaunchKernelGGL(silu_bwd_kernel<true>, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, st, x, dy, dx, addend, n / 8);
  else hipLaunchKernelGGL(silu_bwd_kernel<false>, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, st, x, dy, dx, addend, n / 8);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_add(const bf16* a, const bf16* b, bf16* o, long n, hipStream_t st) {
  ARG_CHECK(n % 8 == 0, "add: n=%ld", n);
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8)), dim3(EW_BLOCK), 0, st, a, b, o, n / 8);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ concat / split along channels
__global__ void concat_kernel(const bf16* __restrict__ a, int Ca, const bf16* __restrict__ b, int Cb,
                              bf16* __restrict__ o, long rows) {
  const int va = Ca / 8, vt = (Ca + Cb) / 8;
  VEC_LOOP(i, rows * vt) {
    long r = i / vt;
    int v = (int)(i - r * vt);
    bf16x8 x = v < va ? *(const bf16x8*)(a + r * Ca + v * 8) : *(const bf16x8*)(b + r * Cb + (v - va) * 8);
    *(bf16x8*)(o + i * 8) = x;
  }
}
__global__ void split_add_kernel(const bf16* __restrict__ g, bf16* ga, int Ca, const bf16* add_a, bf16* gb, int Cb,
                                 const bf16* add_b, long rows) {
  const int va = Ca / 8, vt = (Ca + Cb) / 8;
  VEC_LOOP(i, rows * vt) {
    long r = i / vt;
    int v = (int)(i - r * vt);
    bf16x8 x = *(const bf16x8*)(g + i * 8);
    const long off = v < va ? r * Ca + v * 8 : r * Cb + (v - va) * 8;
    bf16* dst = (v < va ? ga : gb) + off;
    const bf16* add = v < va ? add_a : add_b;
    if (add) {
      bf16x8 y = *(const bf16x8*)(add + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (bf16)((float)x[e] + (float)y[e]);
    }
    *(bf16x8*)dst = x;
  }
}
int launch_concat(const bf16* a, int Ca, const bf16* b, int Cb, bf16* o, long rows, hipStream_t st) {
  ARG_CHECK(Ca % 8 == 0 && Cb % 8 == 0, "concat: Ca=%d Cb=%d", Ca, Cb);
  long nv = rows * ((Ca + Cb) / 8);
  hipLaunchKernelGGL(concat_kernel, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, a, Ca, b, Cb, o, rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_split_add(const bf16* g, bf16* ga, int Ca, const bf16* add_a, bf16* gb, int Cb, const bf16* add_b, long rows,
                     hipStream_t st) {
  ARG_CHECK(Ca % 8 == 0 && Cb % 8 == 0, "split: Ca=%d Cb=%d", Ca, Cb);
  long nv = rows * ((Ca + Cb) / 8);
  hipLaunchKernelGGL(split_add_kernel, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, g, ga, Ca, add_a, gb, Cb, add_b, rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ nearest 2x upsample (token-major)
__global__ void upsample2x_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int H, int W, int C) {
  const int vpr = C / 8;
  const long n = (long)B * (2 * H) * (2 * W) * vpr;
  VEC_LOOP(i, n) {
    long pix = i / vpr;
    int v = (int)(i - pix * vpr);
    int xo = (int)(pix % (2 * W));
    long t = pix / (2 * W);
    int yo = (int)(t % (2 * H));
    int b = (int)(t / (2 * H));
    long src = ((long)b * H + (yo >> 1)) * W + (xo >> 1);
    *(bf16x8*)(y + i * 8) = *(const bf16x8*)(x + src * C + v * 8);
  }
}
template <bool ACC>
__global__ void upsample2x_bwd_kernel(const bf16* __restrict__ dy, bf16* dx, const bf16* addend, int B, int H, int W, int C) {
  const int vpr = C / 8;
  const long n = (long)B * H * W * vpr;
  VEC_LOOP(i, n) {
    long pix = i / vpr;
    int v = (int)(i - pix * vpr);
    int xi = (int)(pix % W);
    long t = pix / W;
    int yi = (int)(t % H);
    int b = (int)(t / H);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        long src = ((long)b * 2 * H + 2 * yi + dyy) * 2 * W + 2 * xi + dxx;
        bf16x8 g = *(const bf16x8*)(dy + src * C + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)g[e];
      }
    bf16x8 o;
    if (ACC) o = *(const bf16x8*)(addend + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)(ACC ? s[e] + (float)o[e] : s[e]);
    *(bf16x8*)(dx + i * 8) = o;
  }
}
// ------------------------------------------------------------------ nearest-2x upsampling folded into the following 3x3 convolution
// (GemmP::up2).  Weight entry e = 4 (2a + b) + 2u + v of output phase (a, b), stencil position (u, v): the sum of the taps (dy, dx) whose
// high-resolution source row 2i + a + dy falls into low-resolution row i + u - (1 - a) (columns alike):
//   a = 0: u = 0 <- {dy = -1}, u = 1 <- {0, +1};   a = 1: u = 0 <- {-1, 0}, u = 1 <- {+1}
__global__ void upconv_fold_weights_kernel(const bf16* __restrict__ w, bf16* __restrict__ we, int Cout, int Cin) {
  const int vpr = Cin / 8;
  const long n = (long)Cout * 16 * vpr;
  VEC_LOOP(i, n) {
    const int v = (int)(i % vpr);
    const long t = i / vpr;
    const int e = (int)(t & 15);
    const long co = t >> 4;
    const int a = (e >> 3) & 1, b = (e >> 2) & 1, u = (e >> 1) & 1, vv = e & 1;
    // rows / columns of the 3x3 kernel that fold onto (u | a) and (vv | b): index 0..2 = dy + 1
    const int r0 = a == 0 ? (u == 0 ? 0 : 1) : (u == 0 ? 0 : 2), r1 = a == 0 ? (u == 0 ? 0 : 2) : (u == 0 ? 1 : 2);
    const int c0 = b == 0 ? (vv == 0 ? 0 : 1) : (vv == 0 ? 0 : 2), c1 = b == 0 ? (vv == 0 ? 0 : 2) : (vv == 0 ? 1 : 2);
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int r = r0; r <= r1; ++r)
      for (int c = c0; c <= c1; ++c) {
        const bf16x8 g = *(const bf16x8*)(w + ((co * 9 + r * 3 + c) * Cin) + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += (float)g[k];
      }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)s[k];
    *(bf16x8*)(we + i * 8) = o;
  }
}
int launch_upconv_fold_weights(const bf16* w, bf16* weff, int Cout, int Cin, hipStream_t st) {
  ARG_CHECK(Cin % 8 == 0, "upconv: Cin=%d", Cin);
  const long nv = (long)Cout * 16 * (Cin / 8);
  hipLaunchKernelGGL(upconv_fold_weights_kernel, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, w, weff, Cout, Cin);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// the transpose of the fold: dW[co][r][c][ci] (=, +=) sum over the four phases of dWeff[co][phase][u(a, r)][v(b, c)][ci]; optionally the
// bf16 image of the result x scale (the data-parallel exchange arena, as the weight-gradient GEMM epilogues write it)
__global__ void upconv_unfold_grads_kernel(const float* __restrict__ de, float* dw, bf16* emit, float scale, int accumulate, int Cout, int Cin) {
  const int vpr = Cin / 4;
  const long n = (long)Cout * 9 * vpr;
  VEC_LOOP(i, n) {
    const int v = (int)(i % vpr);
    const long t = i / vpr;
    const int tap = (int)(t % 9);
    const long co = t / 9;
    const int r = tap / 3, c = tap - r * 3;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int a = ph >> 1, b = ph & 1;
      const int u = a == 0 ? (r == 0 ? 0 : 1) : (r == 2 ? 1 : 0), vv = b == 0 ? (c == 0 ? 0 : 1) : (c == 2 ? 1 : 0);
      const f32x4 g = *(const f32x4*)(de + ((co * 16 + ph * 4 + u * 2 + vv) * Cin) + v * 4);
      s[0] += g[0]; s[1] += g[1]; s[2] += g[2]; s[3] += g[3];
    }
    float* d = dw + i * 4;
    if (accumulate) { const f32x4 o = *(const f32x4*)d; s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3]; }
    *(f32x4*)d = s;
    if (emit) {
      bf16x4 o;
      o[0] = (bf16)(s[0] * scale); o[1] = (bf16)(s[1] * scale); o[2] = (bf16)(s[2] * scale); o[3] = (bf16)(s[3] * scale);
      *(bf16x4*)(emit + i * 4) = o;
    }
  }
}
int launch_upconv_unfold_grads(const float* dweff, float* dw, bf16* emit, float emit_scale, int accumulate, int Cout, int Cin, hipStream_t st) {
  ARG_CHECK(Cin % 4 == 0, "upconv: Cin=%d", Cin);
  const long nv = (long)Cout * 9 * (Cin / 4);
  hipLaunchKernelGGL(upconv_unfold_grads_kernel, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, dweff, dw, emit, emit_scale, accumulate, Cout, Cin);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// planar phase-major [4][plane rows >= B H W][C] <-> token-major high resolution [B][2H][2W][C] (TO_HI: planar -> high resolution)
template <bool TO_HI>
__global__ void pixel_shuffle2_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int H, int W, int C, long plane, const bf16* addend) {
  const int vpr = C / 8;
  const long n = (long)B * 4 * H * W * vpr;
  VEC_LOOP(i, n) {                                   // i runs over the high-resolution tensor
    const long pix = i / vpr;
    const int v = (int)(i - pix * vpr);
    const int xo = (int)(pix % (2 * W));
    const long t = pix / (2 * W);
    const int yo = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const int ph = (yo & 1) * 2 + (xo & 1);
    const long pl = (long)ph * plane + ((long)b * H + (yo >> 1)) * W + (xo >> 1);
    if (TO_HI) {
      bf16x8 o = *(const bf16x8*)(src + pl * C + v * 8);
      if (addend) {
        const bf16x8 ad = *(const bf16x8*)(addend + i * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)((float)o[k] + (float)ad[k]);
      }
      *(bf16x8*)(dst + i * 8) = o;
    }
    else *(bf16x8*)(dst + pl * C + v * 8) = *(const bf16x8*)(src + i * 8);
  }
}
int launch_pixel_shuffle2(const bf16* src, bf16* dst, int B, int H, int W, int C, int to_hi, hipStream_t st, const bf16* addend) {
  ARG_CHECK(C % 8 == 0, "pixel shuffle: C=%d", C);
  const long nv = (long)B * 4 * H * W * (C / 8);
  if (to_hi) hipLaunchKernelGGL(pixel_shuffle2_kernel<true>, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, src, dst, B, H, W, C, upconv_plane_rows(B, H, W), addend);
  else hipLaunchKernelGGL(pixel_shuffle2_kernel<false>, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, src, dst, B, H, W, C, upconv_plane_rows(B, H, W), addend);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_upsample2x(const bf16* x, bf16* y, int B, int H, int W, int C, hipStream_t st) {
  ARG_CHECK(C % 8 == 0, "upsample: C=%d", C);
  long nv = (long)B * 4 * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, x, y, B, H, W, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_upsample2x_bwd(const bf16* dy, bf16* dx, const bf16* addend, int B, int H, int W, int C, hipStream_t st) {
  ARG_CHECK(C % 8 == 0, "upsample: C=%d", C);
  long nv = (long)B * H * W * (C / 8);
  if (addend) hipLaunchKernelGGL(upsample2x_bwd_kernel<true>, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, dy, dx, addend, B, H, W, C);
  else hipLaunchKernelGGL(upsample2x_bwd_kernel<false>, dim3(ew_grid(nv)), dim3(EW_BLOCK), 0, st, dy, dx, addend, B, H, W, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ sinusoidal embedding (cos first)
__global__ void sincos_kernel(const float* __restrict__ t, bf16* __restrict__ out, int rows, int dim, long ldo) {
  const int half = dim / 2;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * half) return;
  int r = i / half, k = i - r * half;
  float f = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
  float a = t[r] * f;
  out[(long)r * ldo + k] = (bf16)cosf(a);
  out[(long)r * ldo + half + k] = (bf16)sinf(a);
}
int launch_sincos(const float* t, bf16* out, int rows, int dim, long ldo, hipStream_t st) {
  ARG_CHECK(dim % 2 == 0, "sincos: dim=%d", dim);
  int n = rows * (dim / 2);
  hipLaunchKernelGGL(sincos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, t, out, rows, dim, ldo);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ small utilities
__global__ void copy_cols_kernel(const bf16* __restrict__ src, long lds_, bf16* __restrict__ dst, long ldd, int rows,
                                 int cols) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int r = i / cols, c = i - r * cols;
  dst[(long)r * ldd + c] = src[(long)r * lds_ + c];
}
int launch_copy_cols(const bf16* src, long lds_, bf16* dst, long ldd, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv((long)rows * cols, 256)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n, float scale) {
  VEC_LOOP(i, n) y[i] = (bf16)(x[i] * scale);
}
__global__ void bf16_to_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long n) {
  VEC_LOOP(i, n) y[i] = (float)x[i];
}
int launch_f32_to_bf16(const float* x, bf16* y, long n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, n, scale);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_bf16_to_f32(const bf16* x, float* y, long n, hipStream_t st) {
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, x, y, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// sum of squares of an fp32 or bf16 array (gradient norm): 16-byte vectors; fixed summation order (per-thread stride loop ->
// wave shuffle tree -> waves of a block in order -> one partial per block -> a second single-block pass over the partials in
// index order), so the norm and the clip coefficient derived from it are bit-reproducible run to run and rank to rank.
#define SUMSQ_MAX_BLOCKS 4096
__device__ float g_sumsq_partials[SUMSQ_MAX_BLOCKS];   // scratch of the (stream-ordered) two-pass reduction
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void sumsq_kernel(const T* __restrict__ x, long n) {
  __shared__ float wsum[EW_BLOCK / 64];
  constexpr int V = 16 / sizeof(T);
  float s = 0.f;
  const long nv = n / V;
  VEC_LOOP(i, nv) {
    if (sizeof(T) == 4) {
      const f32x4 v = *(const f32x4*)((const float*)x + i * 4);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      const bf16x8 v = *(const bf16x8*)((const bf16*)x + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nv * V) { const float t = (float)x[nv * V + threadIdx.x]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < EW_BLOCK / 64; ++w) t += wsum[w];
    g_sumsq_partials[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(int nblocks, float* __restrict__ out) {
  __shared__ float wsum[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += g_sumsq_partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
template <typename T>
static int sumsq_launch(const T* x, long n, float* out, hipStream_t st) {
  int g = ew_grid(n / (16 / (long)sizeof(T)) + 1);
  if (g > SUMSQ_MAX_BLOCKS) g = SUMSQ_MAX_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel<T>, dim3(g), dim3(EW_BLOCK), 0, st, x, n);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, g, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_sumsq_f32(const float* x, long n, float* out, hipStream_t st) { return sumsq_launch(x, n, out, st); }
int launch_sumsq_bf16(const bf16* x, long n, float* out, hipStream_t st) { return sumsq_launch(x, n, out, st); }
// torch.nn.utils.clip_grad_norm_'s coefficient from the squared norm, on the device (no host round trip):
// coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef) {
  const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
  *coef = c < 1.f ? c : 1.f;
}
int launch_clip_coef(const float* sumsq, float max_norm, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, st, sumsq, max_norm, coef);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
__global__ void scale_kernel(float* __restrict__ x, long n, const float* __restrict__ scale) {
  const float s = *scale;
  VEC_LOOP(i, n) x[i] *= s;
}
int launch_scale_f32(float* x, long n, const float* scale_dev, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_BLOCK), 0, st, x, n, scale_dev);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---- exchange shadow (measurement hook, include/sdxlstep_diag.h part 1) ----
// Stand-in for a collective's device kernel on a single GPU: `workgroups` x 256 threads with `lds_bytes` of LDS each (RCCL runs one such
// workgroup per channel) read `buf` once and write it back (16-byte loads and stores, values unchanged), paced evenly over `busy_us`
// microseconds of the 100 MHz real-time counter.  What it costs the step that runs beside it = the co-residency price of the gradient exchange's kernels,
// before any multi-GPU node is available (bench.py --exchange-shadow).
__global__ __launch_bounds__(256) void exchange_shadow_kernel(i32x4* __restrict__ buf, size_t n16, unsigned busy_ticks) {
  extern __shared__ char shadow_lds[];
  if (threadIdx.x == 0) shadow_lds[0] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
  const size_t nchunks = hi > lo ? (hi - lo + 255) / 256 : 1;
  // PACED: the slice is read and written back exactly once, spread evenly over busy_ticks (a collective's kernel moves its bucket at the
  // links' rate, not at the HBM's): chunk c may start once c / nchunks of the time has passed
  size_t c = 0;
  for (size_t i = lo; i < hi; i += 256, ++c) {
    const unsigned long long due = (unsigned long long)((double)busy_ticks * (double)c / (double)nchunks);
    while (__builtin_amdgcn_s_memrealtime() - t0 < due) __builtin_amdgcn_s_sleep(8);
    if (i + threadIdx.x < hi) {
      i32x4 v = __builtin_nontemporal_load(buf + i + threadIdx.x);
      __builtin_nontemporal_store(v, buf + i + threadIdx.x);
    }
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(8);
}
int launch_exchange_shadow(void* buf, size_t bytes, int workgroups, int lds_bytes, float busy_us, hipStream_t st) {
  ARG_CHECK(buf && bytes >= 4096 && workgroups >= 1 && workgroups <= 256 && lds_bytes >= 16 && lds_bytes <= 160 * 1024 && busy_us > 0.f,
            "exchange shadow: bad arguments");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)exchange_shadow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(exchange_shadow_kernel, dim3(workgroups), dim3(256), lds_bytes, st, (i32x4*)buf, bytes / 16, (unsigned)(busy_us * 100.f));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
