// Device helpers shared by the attention kernels (attention.hip: the multi-wave-per-SIMD forms; attention_pl.hip: the software-pipelined
// one-wave-per-SIMD forward): the [64][64] bf16 LDS tile image and its LDS-DMA staging, MFMA fragment reads, cross-row lane exchanges.
#pragma once
#include "kernels.h"

#define HD 64
#define SCALE 0.125f
#define LOG2E 1.4426950408889634f

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

static __device__ __attribute__((aligned(16))) const unsigned int g_attn_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ bf16x8 z8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}
// LDS image of every [64 rows][64 cols] bf16 tile: 128-byte rows, no padding, 16-byte vector v of row r stored at
// slot v ^ (r & 7).  Written by LDS-DMA (global_load_lds_dwordx4: the swizzle is applied on the per-lane source
// address), read conflict-free both as rows (ds_read_b128 fragments) and transposed (ds_read_b64_tr_b16).
// b128 fragment: row r, 8 contiguous columns starting at c (multiple of 8)
__device__ __forceinline__ bf16x8 ld_frag(const bf16* t, int r, int c) {
  return *(const bf16x8*)(t + r * 64 + ((((c >> 3) ^ (r & 7))) << 3));
}
// transpose-read fragment for one 32-deep step t over tile rows (slot convention above), 16 columns at col0
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int t, int col0, int lane16, int g) {
  const int r = t * 32 + g * 4 + (lane16 >> 2);
  const int v = (col0 >> 3) + ((lane16 >> 1) & 1);
  const bf16* p0 = tile + r * 64 + ((v ^ (r & 7)) << 3) + (lane16 & 1) * 4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 16 * 64));   // rows +16: same r & 7
  union { s16x4 s[2]; bf16x8 v; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
  bf16x8 o;
  o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
  o[4] = (bf16)b[0]; o[5] = (bf16)b[1]; o[6] = (bf16)b[2]; o[7] = (bf16)b[3];
  return o;
}
// LDS-DMA of [64][64] bf16 tiles (zero beyond nrows): 8 chunks of 1 KiB = 8 rows each, 2 per wave.  The per-lane source
// pointers of tile 0 are computed once; a tile is then one 64-bit add per piece (the first version recomputed row * ld with
// a 64-bit multiply and an out-of-range select for every piece of every tile: ~30 of the ~190 VALU instructions per tile of
// kernels whose VALU pipe, not the matrix pipe, sets the pace).
struct TileSrc {
  const bf16* p[2];   // this lane's 16-byte vector of pieces 0 / 1 in tile 0
  long step;          // elements per tile (64 rows)
  int r[2];           // row of the lane's vector inside a tile
};
__device__ __forceinline__ TileSrc tile_src(const bf16* base, long ld, int wave, int lane) {
  TileSrc s;
  s.step = 64 * ld;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + (lane >> 3);
    s.r[j] = r;
    s.p[j] = base + (long)r * ld + (((lane & 7) ^ (r & 7)) << 3);
  }
  return s;
}
__device__ __forceinline__ void tile_dma(const TileSrc& s, int t, int nrows, bf16* tile, int wave) {
  const bool full = (t + 1) * 64 <= nrows;   // wave-uniform
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16* src = s.p[j] + (long)t * s.step;
    if (!full && t * 64 + s.r[j] >= nrows) src = (const bf16*)g_attn_zero16;
    lds_dma16_global(src, lds_addr_of(tile + (wave * 2 + j) * 512));   // asm: see common.h
  }
}

#define TILE_ELEMS (64 * 64)

// hipcc places the s_waitcnt vmcnt(0) for registers loaded before a loop at their first use INSIDE the loop body, where it
// runs every iteration and drains the LDS-DMA pieces in flight.  Touching the registers here makes it wait here, once.
__device__ __forceinline__ void landed(bf16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void landed(float& v) { asm volatile("" : "+v"(v)); }

// XCD-aware (block, pair) of a grid of n_inner x nbh workgroups from the linear workgroup id: consecutive ids go round the 8 XCDs, and all
// workgroups that stream the SAME K / V (or Q / dO) tiles -- the blocks of one (batch, head) pair -- should meet in ONE XCD's L2.  With the
// plain (block, pair) order every XCD fetched every pair's tiles through the fabric: 55.7 GB per step for ~13 GB of algorithmic attention
// traffic (profiles/r05_pmc_step_summary.json).
__device__ __forceinline__ void attn_xcd_map(const AttnP& p, int i, int n_inner, int& inner, int& bh) {
  if (p.xcd) {
    const int slot = i >> 3, grp = slot / n_inner;
    bh = grp * 8 + (i & 7);
    inner = slot - grp * n_inner;
  } else {
    bh = i / n_inner;
    inner = i - bh * n_inner;
  }
}
// cross-row lane exchanges without the LDS crossbar (ds_bpermute + lgkmcnt wait): gfx950 v_permlane16_swap / v_permlane32_swap
// trade 16-lane rows (odd rows of the first operand <-> even rows of the second) / wave halves between two registers; with
// both operands holding x the pair afterwards holds x and x from the partner row / half in every lane.
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float max_over_g(float x) {   // max over the four lanes l16 + 16 g, g = 0..3 (same value in all four)
  float a = x, b = x;
  swap16(a, b);
  x = fmaxf(a, b);
  a = x; b = x;
  swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float sum_over_g(float x) {
  float a = x, b = x;
  swap16(a, b);
  x = a + b;
  a = x; b = x;
  swap32(a, b);
  return a + b;
}

// x * c rounded back to bf16, element by element (operand prescale: the scores then come out of the matrix pipe in the log2
// domain, exp2's argument needs no multiply)
__device__ __forceinline__ bf16x8 scale8(bf16x8 v, float c) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)v[e] * c);
  return o;
}
