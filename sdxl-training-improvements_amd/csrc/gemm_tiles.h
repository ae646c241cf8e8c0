// Pieces shared by the bf16 MFMA GEMM kernels (gemm.hip: 128-row tiles; gemm256.hip: 256x256 tiles, 8-phase schedule):
// LDS image conventions of the LDS-DMA written operand tiles, fragment reads, gathered-pixel decoding, XCD-aware tile order.
#pragma once
#include "kernels.h"

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// 16 zero bytes in global memory: source of every out-of-range / padded 16-byte vector of the LDS-DMA loads
static __device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// ---- LDS images -----------------------------------------------------------------------------------------
// Tiles are written by global_load_lds_dwordx4 (64 lanes x 16 B = one contiguous 1 KiB chunk per wave instruction,
// no VGPR round trip, no ds_write).  The DMA destination is lane-linear, so the bank swizzle is applied on the
// per-lane SOURCE address and again on the fragment read (same permutation on both sides).
//  K-contiguous tile [R][BK]: BK = 64 -> 128-B rows, 8 vectors: slot = kv ^ (r & 7)
//                             BK = 32 ->  64-B rows, 4 vectors: slot = kv ^ P[(r >> 2) & 3], P = {0,2,3,1}
//     (conflict-free for the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31})
//  N-contiguous tile [BK][W] (W = 128 or 160 columns, V = W/8 vectors per k-row):
//     W = 128: vector v of k-row k sits at slot v ^ (F(k) << 1),  F(k) = (k & 3) | (((k >> 3) & 1) << 2)
//     W = 160: 320-B rows already spread 4 consecutive k-rows over disjoint banks; rows k and k+8 would collide, so
//              rows with bit 3 set are rotated by 2 vectors: slot = (v + 2*((k>>3)&1)) % 20
//     (conflict-free for ds_read_b64_tr_b16)
__device__ __forceinline__ int swzF(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
template <int W>
__device__ __forceinline__ int nc_phys(int k, int v) {
  if (W == 128) return v ^ (swzF(k) << 1);
  int q = v + 2 * ((k >> 3) & 1);
  return q >= 20 ? q - 20 : q;
}
template <int W>
__device__ __forceinline__ int nc_logical(int k, int pv) {
  if (W == 128) return pv ^ (swzF(k) << 1);
  int q = pv - 2 * ((k >> 3) & 1);
  return q < 0 ? q + 20 : q;
}
template <int BKT>
__device__ __forceinline__ int kc_swz(int r) {
  if (BKT == 64) return r & 7;
  return (0x78 >> (((r >> 2) & 3) * 2)) & 3;  // 0b01_11_10_00 -> {0,2,3,1}
}
template <int BKT>
__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int r, int kv) {
  return *(const bf16x8*)(tile + r * (BKT * 2) + ((kv ^ kc_swz<BKT>(r)) << 4));
}
// 8 k-rows starting at kb (multiple of 8), 16 columns starting at col0 (multiple of 16): lane i of each 16-lane
// group supplies the address of 4 contiguous bf16 of row (i>>2), columns 4*(i&3).., and receives column i.
template <int W>
__device__ __forceinline__ bf16x8 frag_nc(const char* tile, int kb, int col0, int l16) {
  const int krow = kb + (l16 >> 2);
  const int v = (col0 >> 3) + ((l16 >> 1) & 1);
  const char* p0 = tile + krow * (W * 2) + (nc_phys<W>(krow, v) << 4) + (l16 & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * (W * 2)));  // rows +4: same swizzle
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
// GemmP::up2: row / column offset of stencil entry e = 4 (2a + b) + 2u + v on the low-resolution image
__device__ __forceinline__ int up2_dy(int e) { return ((e >> 1) & 1) - (1 - ((e >> 3) & 1)); }
__device__ __forceinline__ int up2_dx(int e) { return (e & 1) - (1 - ((e >> 2) & 1)); }
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// XCD-aware tile order: hardware workgroup id i runs on XCD i % 8 (observed; used for speed only).  The 8 XCDs are laid
// out as a px x py grid over the (n, m) tile grid (px chosen by the launcher to minimise the A-panel + B-panel footprint
// per XCD) so that each private 4 MiB L2 sees a compact rectangle of output tiles; inside a rectangle tiles go n-fastest
// in strips of 8 columns.  Identity when the grid does not divide.
// the same map for a linear workgroup id over a gx x gy tile grid (regions of a 1-D launch)
__device__ __forceinline__ void xcd_tile_map_id(int xcd_px, int id, int gx, int gy, int& bx, int& by) {
  bx = id % gx;
  by = id / gx;
  if (xcd_px > 0 && gx % xcd_px == 0 && gy % (8 / xcd_px) == 0) {
    const int px = xcd_px, py = 8 / px;
    const int tn = gx / px, tm = gy / py;
    const int xcd = id & 7, li = id >> 3;
    const int sw = tn < 8 ? tn : 8;
    const int full = (tn / sw) * sw * tm;
    int ln, lm;
    if (li < full) {
      const int strip = li / (sw * tm), w = li - strip * (sw * tm);
      lm = w / sw;
      ln = strip * sw + (w - lm * sw);
    } else {
      const int rw = tn - (tn / sw) * sw, w = li - full;
      lm = w / rw;
      ln = (tn / sw) * sw + (w - lm * rw);
    }
    bx = (xcd % px) * tn + ln;
    by = (xcd / px) * tm + lm;
  }
}
// Locality order for ANY 3-D grid (no divisibility needed): the launch's linear workgroup id i (x fastest, then y, then z; consecutive ids
// go round the 8 XCDs) -> position s in the sequence "for z: for band of bh tile rows: for tile column: for row of the band".  XCD x owns a
// contiguous range of that sequence (sizes differ by at most one), handed out in order, so what an XCD runs at any time is a compact
// block of tiles of one z: its operand panels cross the fabric once and are shared in that XCD's L2.  (Weight gradients whose tile grid
// does not divide over 8 XCDs -- 15 x 10 tiles, 3 tiles x 12 splits -- ran in identity order: every XCD fetched nearly every panel,
// 3 - 8 x the algorithmic bytes: profiles/r05q_fetch_by_kernel.txt.)  bh <= 0: identity.
// The z slices come in two kinds: OUTER ones share nothing (reduction splits, the problems of a grouped launch) and are kept apart -- the
// sequence runs through one outer slice's tiles before the next --; INNER ones (zin of them: the filter taps / stencil rows of a
// convolution's weight gradient, which read the same dY panel and shifted copies of the same pixels) sit next to each other at every tile.
// Returns the inner index in zi and the outer in zo (gridDim.z = zin x outer count).
__device__ __forceinline__ void xcd_seq_map(int bh, int zin, int& bx, int& by, int& zi, int& zo) {
  const int gx = gridDim.x, gy = gridDim.y, T = gx * gy, W = T * gridDim.z;
  const int i = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  const int q = W >> 3, rem = W & 7, x = i & 7;
  const int s = x * q + min(x, rem) + (i >> 3);
  const int r0 = s / zin;
  zi = s - r0 * zin;
  zo = r0 / T;
  const int t = r0 - zo * T;
  const int band_tiles = bh * gx, band = t / band_tiles, r = t - band * band_tiles;
  const int h = min(bh, gy - band * bh);
  bx = r / h;
  by = band * bh + (r - bx * h);
}
__device__ __forceinline__ void xcd_seq_map(int bh, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (bh <= 0) return;
  int zi;
  xcd_seq_map(bh, 1, bx, by, zi, bz);
}
// the same order for a linear workgroup id over a gx x gy tile grid (regions of a 1-D launch whose first id is a multiple of 8)
__device__ __forceinline__ void xcd_seq_map_id(int bh, int i, int gx, int gy, int& bx, int& by) {
  const int T = gx * gy;
  const int q = T >> 3, rem = T & 7, x = i & 7;
  const int t = x * q + min(x, rem) + (i >> 3);
  const int band_tiles = bh * gx, band = t / band_tiles, r = t - band * band_tiles;
  const int h = min(bh, gy - band * bh);
  bx = r / h;
  by = band * bh + (r - bx * h);
}
// band height (tile rows) of xcd_seq_map for a gx x gy grid of bm x bn tiles with gz slices: the block an XCD holds is ~square in elements
__host__ __device__ inline int xcd_band_rows(int gx, int gy, int gz, int bm, int bn, int zin = 1) {
  const long T = (long)gx * gy, W = T * gz;
  long per = W / 8 / zin < T ? W / 8 / zin : T;
  if (per < 1) per = 1;
  int bh = 1;
  while ((long)(bh + 1) * (bh + 1) * bm <= per * bn && bh + 1 <= gy) ++bh;      // bh ~ sqrt(per * bn / bm)
  return bh;
}
__device__ __forceinline__ void xcd_tile_map(int xcd_px, int& bx, int& by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  if (xcd_px > 0 && gridDim.x % xcd_px == 0 && gridDim.y % (8 / xcd_px) == 0) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int px = xcd_px, py = 8 / px;
    const int tn = gx / px, tm = gy / py;
    const int id = blockIdx.y * gx + blockIdx.x;
    const int xcd = id & 7, li = id >> 3;
    const int sw = tn < 8 ? tn : 8;
    const int full = (tn / sw) * sw * tm;
    int ln, lm;
    if (li < full) {
      const int strip = li / (sw * tm), w = li - strip * (sw * tm);
      lm = w / sw;
      ln = strip * sw + (w - lm * sw);
    } else {
      const int rw = tn - (tn / sw) * sw, w = li - full;
      lm = w / rw;
      ln = (tn / sw) * sw + (w - lm * rw);
    }
    bx = (xcd % px) * tn + ln;
    by = (xcd / px) * tm + lm;
  }
}
