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
