// bf16 MFMA GEMM family for gfx950 (MI355X): Linear / conv3x3 forward, dgrad and wgrad.
//
// One workgroup = NW waves (4 or 8) in a (NW/2) x 2 grid over a 128 x BN block tile (BN = 128 or 160), K-step BK
// (64 or 32).  Each wave owns a (256/NW) x (BN/2) sub-tile of v_mfma_f32_16x16x32_bf16 fragments.
// Operand tiles go global -> LDS directly (global_load_lds_dwordx4, LDS-DMA) into an S-deep ring: S-1 K-steps of DMA
// are in flight under the current step's MFMAs (counted s_waitcnt vmcnt + raw s_barrier, one barrier per K-step).
// An operand whose reduction dimension is the contiguous one is read from LDS with ds_read_b128; an operand whose
// reduction dimension is the strided one (dgrad's weights, both wgrad operands) is read with the gfx950 transpose
// read ds_read_b64_tr_b16, so no transposed copies of weights or activations are ever materialised.
// The 3x3 convolution is an implicit GEMM: the gathered operand's rows are pixels and each K-step is
// (tap, channel chunk) with zero fill at the image border.
//
// MFMA operand slots: lane l holds slot (g = l>>4, j = 0..7) of row/col (l & 15).  The hardware pairs A slot (g,j)
// with B slot (g,j); both operands map slot (g,j) to k = 8g + j of the 32-deep step, so the two read paths
// (b128 / transpose) agree by construction.  C/D layout: col = l & 15, row = 4*(l>>4) + reg.
#include "gemm_tiles.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#define BM 128
#ifndef GEMM_DIRECT_EPILOGUE
#define GEMM_DIRECT_EPILOGUE 1
#endif

static int g_mode256 = 1;     // 0: 128-row kernel only, 1: selection policy (gemm_use256), 2: 256 x 256 kernel wherever applicable
static int g_force_cfg = 0;   // > 0: force that configuration of the 128-row kernel (microbenchmarks; sdxl_set_gemm_mode)
#ifdef SDXL_DIAG
static int g_sk_mode = 0;     // stream-K kernel (gemm_sk.hip, diagnostics build only): 0 never, 1 policy (gemm_use_sk), 2 wherever applicable
#endif

static constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }
// LayerNorm-backward epilogue (diagnostics build): set when a workgroup gave up waiting for the other column tiles of its row block (bit 0: their
// ready flags, bit 1: a granule) -- the results of that launch are invalid; read and cleared by gemm_ln_error (sdxl_ln_error)
__device__ unsigned g_ln_spin_error = 0;
// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), valid in the row's lane 15: four row_shr steps on the VALU, zeros shifted in
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ bf16x8 z8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
  return z;
}
static constexpr int gemm_smem_bytes(int BN, int S, int BK, int NW = 4) {
  // + 1 KiB that absorbs the padding DMA pieces, where the tiles' 1 KiB chunks do not divide evenly among the waves
  const bool pad = ((BM * BK * 2 / 1024) % NW) != 0 || ((BN * BK * 2 / 1024) % NW) != 0;
  int ring = S * (BM + BN) * BK * 2 + (pad ? 1024 : 0), stg = BK == 32 ? 0 : 64 * (BN + 4) * 4;   // (GEGLU staging: BK 64 only)
  return ring > stg ? ring : stg;
}
// workgroups per CU the register budget is sized for: what the 160 KiB of LDS admits, at most 3 (4-wave) / 1 (8-wave)
static constexpr int gemm_occupancy(int BN, int S, int BK, int NW) {
  int byl = (160 * 1024) / gemm_smem_bytes(BN, S, BK, NW);
  return NW == 8 ? 1 : (byl > 3 ? 3 : (byl < 1 ? 1 : byl));
}

// FORM : GEMM_NT / NN / TN        CONV : implicit-GEMM 3x3 gather
// BN   : 128 or 160 output columns per workgroup          S : LDS ring depth (S-1 K-steps of DMA in flight)
// BK   : 64 or 32 reduction elements per K-step           NW: waves per workgroup (4: 2x2, 8: 4x2)
// FAST : reduction length a multiple of BK and either no gather or a same-size stride-1 3x3 gather: every lane's DMA
//        source is a running pointer (conv: fixed base + uniform offset + border predicate) advanced by a
//        per-lane constant each K-step (0 for out-of-range rows, which keep pointing at the zero vector), and the DMA
//        pieces are issued between groups of MFMAs so their issue cost hides under the matrix pipe.
// KSP  : 8 waves as two GROUPS of 2 x 2 waves with 64 x (BN/2) wave tiles; group g multiplies the g-th 32-deep half of
//        every K-step (intra-workgroup split-K) and runs one barrier behind group 0, so that on every SIMD one wave
//        is in its MFMA section (20 MFMAs + the step's DMA pieces) while the other issues its 9 fragment reads and
//        waits -- the matrix pipe and the LDS path alternate by construction (the 4 x 2 layout of the same tile has all
//        eight waves read, wait and multiply in lockstep: 0.73 us per K-step of a 128x160 tile against 0.3 at the MFMA
//        rate).  The two partial accumulator tiles are exchanged through LDS once, after the main loop.
// PL   : four waves, ONE per SIMD, software-pipelined (configurations 5 / 6): the fragments of the NEXT 32-deep half-step are read
//        (and the DMA pieces of step t + S - 1 issued) between the MFMA groups of the current one, into a second register set -- the
//        matrix pipe, the LDS read path and the staging stream run side by side inside ONE wave instead of taking turns between two
//        waves of a SIMD (configuration 3: 0.75 us per K-step for 0.30 of products, 0.43 of staging and 0.3 of fragment reads).
//        64 x (BN/2) wave tiles: 72 KB of fragment reads per K-step instead of 112 with eight waves; one barrier per K-step, in the
//        middle of it; the workgroup owns its CU (S x 36 KiB of LDS, up to 512 registers per lane).
template <int FORM, bool CONV, int BN, int S, int BK, bool FAST, int NW, bool KSP = false, bool PL = false>
// (PL: the register budget of TWO waves per SIMD, 256 -- with the 512 of one wave hipcc picks the AGPR form of the MFMAs and then moves all 80
//  accumulators between the two halves of the file every K-step; the loop needs ~210)
__global__ __launch_bounds__(NW * 64, PL ? 2 : gemm_occupancy(BN, S, BK, NW)) void gemm_kernel(const GemmP pin) {
  GemmP p = pin;
  set_wave_prio(pin.prio);
  int bx, by, bz = blockIdx.z, ztap = 0;
  if (FORM == GEMM_TN && pin.xcd_bh > 0) {      // generic XCD-aware order (weight gradients whose grid fits no rectangle): taps inner, splits / problems outer
    const int tz = pin.group > 1 ? 1 : (int)gridDim.z / pin.splitk;
    xcd_seq_map(pin.xcd_bh, tz, bx, by, ztap, bz);
  } else xcd_tile_map(pin.xcd_px, bx, by);                                           // XCD-aware tile order (gemm_tiles.h)
  if (FORM == GEMM_TN && pin.group > 1) {          // grouped launch: this workgroup's problem
    const int gi = bz;
    p.A = pin.gA[gi]; p.B = pin.gB[gi]; p.C = pin.gC[gi]; p.bias_grad = pin.gbias_grad[gi]; p.Cb = pin.gCb[gi];
  }
  static_assert(!KSP || (NW == 8 && BK == 64 && FAST && S >= 3 && FORM != GEMM_TN), "split-K groups: 8 waves, BK 64, FAST staging, ring >= 3");
  static_assert(!PL || (NW == 4 && BK == 64 && FAST && S >= 3 && FORM != GEMM_TN && !KSP), "pipelined loop: 4 waves, BK 64, FAST staging, ring >= 3, NT / NN");
  constexpr int BMT = BM;
  constexpr int A_TILE_BYTES = BMT * BK * 2;      // [BMT][BK] or [BK][128] bf16
  constexpr int B_TILE_BYTES = BN * BK * 2;       // [BN][BK] or [BK][BN] bf16
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int RING_BYTES = S * STAGE_BYTES;     // followed by 1 KiB that absorbs the padding DMA pieces
  constexpr int WGM = KSP ? 2 : NW / 2;           // wave grid WGM x 2 (x 2 K-groups)
  constexpr int MI = BMT / (WGM * 16);            // A fragments per wave (4 or 2)
  constexpr int NJ = BN / 32;                     // B fragments per wave (wave tile (16 MI) x BN/2)
  constexpr int NCA = A_TILE_BYTES / 1024;        // 1 KiB DMA chunks of the A tile
  constexpr int NCB = B_TILE_BYTES / 1024;
  constexpr int ACH = cdiv_c(NCA, NW);            // chunks per wave (chunk id = wave + j*NW; ids >= NC* are padding
  constexpr int BCH = cdiv_c(NCB, NW);            //  pieces so that every wave issues the same number of loads)
  constexpr int NL = ACH + BCH;                   // LDS-DMA instructions per wave per K-step
  constexpr int VR = BK / 8;                      // vectors per K-contiguous row
  constexpr int KC_ROWS = 64 / VR;                // rows of a K-contiguous tile per 1 KiB chunk
  constexpr int LDC = BN + 4;
  constexpr int NT = NW * 64;
  constexpr bool PIPE = true;
  constexpr bool DIRECT = GEMM_DIRECT_EPILOGUE;   // operand-swapped products (D^T layout) + register -> global epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KSP ? wave >> 2 : 0;             // K-group
  const int wm = KSP ? (wave >> 1) & 1 : wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, g = lane >> 4;
#ifdef SDXL_GEMM_DIAG   // scratch diagnostics only (never defined in the product build): knock out one pipeline component
  constexpr int dbg = SDXL_GEMM_DIAG;       // bit 0: no MFMA, bit 1: no DMA in the main loop, bit 2: no fragment reads,
#else                                       // bit 3: every workgroup works on tile (0, 0) (all L2 hits after the first touch)
  constexpr int dbg = 0;
#endif
  const int n0 = (dbg & 8) ? 0 : bx * BN;
  const int m0 = (dbg & 8) ? 0 : by * BMT;

  // reduction schedule
  int tap_fixed = 0, split = 0;
  if (FORM == GEMM_TN && p.group <= 1) {
    if (pin.xcd_bh > 0) {
      split = bz;
      tap_fixed = ztap;
    } else {
      tap_fixed = bz / p.splitk;
      split = bz - tap_fixed * p.splitk;
    }
  }
  // NT / NN with split-K (linear FAST staging only, launcher-checked: small-M problems -- a 256-row batch is 16 tiles for 256 CUs
  // and a 160-step reduction per tile): split `blockIdx.z` multiplies its K range, the fp32 partial tile goes to its slab,
  // splitk_epilogue_kernel sums the slabs in a fixed order and applies the bf16 epilogue
  constexpr bool NSPLIT = FORM != GEMM_TN && FAST && !KSP;      // (linear, or the same-size stride-1 3x3 gather: K range = taps x channels)
  if (NSPLIT && p.splitk > 1) split = bz;
  const int ktiles_per_tap = (p.K + BK - 1) / BK;
  int kt_begin = 0, kt_end;
  if (FORM == GEMM_TN) {
    int chunk = (ktiles_per_tap + p.splitk - 1) / p.splitk;
    kt_begin = split * chunk;
    kt_end = min(ktiles_per_tap, kt_begin + chunk);
  } else if (NSPLIT && p.splitk > 1) {
    const int total = ktiles_per_tap * p.taps;
    int chunk = (total + p.splitk - 1) / p.splitk;
    kt_begin = split * chunk;
    kt_end = min(total, kt_begin + chunk);
  } else {
    kt_end = ktiles_per_tap * p.taps;
  }
  // up2 == 2 (NN: input gradient of a stride-2 3x3 convolution by output phase): this workgroup's phase has 1, 2, 2 or 4 taps
  const int dn_phase = (FORM == GEMM_NN && CONV && p.up2 == 2) ? m0 / p.up_plane : 0;
  const int dn_nb = (dn_phase & 1) ? 2 : 1;
  if (FORM == GEMM_NN && CONV && p.up2 == 2) kt_end = ktiles_per_tap * ((dn_phase >> 1) ? 2 : 1) * dn_nb;

  // ---- per-lane LDS-DMA descriptors ----
  // K-contiguous tile: chunk c = rows KC_ROWS*c .. ; lane -> row KC_ROWS*c + lane/VR, physical vector lane%VR
  // N-contiguous tile of width Wd: chunk c = vectors 64c..64c+63 of the [BK][Wd/8] vector grid
  const int kc_rowl = lane / VR, kc_pv = lane % VR;
  PixRow arow[ACH];
  if (CONV && !FAST && FORM != GEMM_TN) {
#pragma unroll
    for (int j = 0; j < ACH; ++j) arow[j] = decode_pix(m0 + (wave + j * NW) * KC_ROWS + kc_rowl, p.M, p.Hm, p.Wm);
  }
  const bf16* zsrc = (const bf16*)g_zero16;
  char* const pad_dst = smem + RING_BYTES;   // 1 KiB: destination of the padding pieces

  // FAST path state.  Linear: running source pointers + per-lane step (elements) for this wave's chunks.
  // Conv (same-size stride-1 3x3, Cin a multiple of BK): fixed per-lane base pointers + a wave-uniform offset that
  // follows (tap, channel step) [NT / NN] or the pixel step [TN]; the border zero-fill is a per-lane 9-bit tap mask
  // [NT / NN: the gathered rows are this workgroup's fixed output pixels] or an incrementally tracked (y, x) of the
  // gathered pixel [TN: the reduction runs over pixels, the tap is fixed per workgroup].
  constexpr bool CF = FAST && CONV;
  const bf16* pa[ACH];
  const bf16* pb[BCH];
  long sa[ACH], sb[BCH];
  int amask[ACH];            // CF: NT / NN tap-validity mask of the A rows; TN: static validity of the dY columns
  int bok[BCH];              // CF: static validity of the B lanes (column in range)
  int by_[BCH], bx_[BCH];    // CF, TN: (y, x) of the pixel this lane gathers at the step being staged
  long ua = 0, ub = 0;       // CF: uniform element offsets of the step being staged
  int s_tap = 0, s_c = 0;    // CF, NT / NN: (tap, channel step) of the step being staged
  // CF, TN: this workgroup's tap.  (up2 == 3: stride-2 gather from the four phase planes of the input: kernel row 0 reads plane row i - 1)
  const int tdy = (SDXL_UP2_3 && p.up2 == 3) ? (tap_fixed / 3 == 0 ? -1 : 0) : p.up2 ? up2_dy(tap_fixed) : tap_fixed / 3 - 1;
  const int tdx = (SDXL_UP2_3 && p.up2 == 3) ? (tap_fixed % 3 == 0 ? -1 : 0) : p.up2 ? up2_dx(tap_fixed) : tap_fixed - (tap_fixed / 3) * 3 - 1;
  const long s3_plane = (SDXL_UP2_3 && p.up2 == 3) ? (long)(((tap_fixed / 3 != 1) ? 2 : 0) + ((tap_fixed % 3 != 1) ? 1 : 0)) * p.up_plane : 0;      // TN: phase plane of this tap's source pixels
  const int up_phase = (CF && FORM == GEMM_NT && p.up2) ? m0 / p.up_plane : dn_phase;      // up2, NT (and NN of mode 2): this workgroup's output phase
  // CF, NT / NN: uniform element offsets of (tap s_tap, channel step s_c)
  auto tap_offsets = [&]() {
    int dy, dx, wtap;
    long plane = 0;
    if (FORM == GEMM_NT && (SDXL_UP2_3 && p.up2 == 3)) {         // stride-2 forward on the four phase planes of the input: kernel row 0 / 1 / 2 = plane row i - 1 (a = 1) / i (0) / i (1)
      const int ky = s_tap / 3, kx = s_tap - ky * 3;
      dy = ky == 0 ? -1 : 0; dx = kx == 0 ? -1 : 0;
      wtap = s_tap;
      plane = (long)((ky != 1 ? 2 : 0) + (kx != 1 ? 1 : 0)) * p.up_plane * p.lda;
    } else
    if (FORM == GEMM_NN && p.up2 == 2) {         // tap s_tap of phase (a, b): (ry, rx) over the rows / columns that reach this phase
      const int a = dn_phase >> 1, b = dn_phase & 1;
      const int ry = s_tap / dn_nb, rx = s_tap - ry * dn_nb;
      const int ky = a ? (ry == 0 ? 0 : 2) : 1, kx = b ? (rx == 0 ? 0 : 2) : 1;       // kernel row / column
      dy = a ? (ry == 0 ? 1 : 0) : 0; dx = b ? (rx == 0 ? 1 : 0) : 0;                   // offset on the low-resolution gradient image
      wtap = ky * 3 + kx;
    } else
    if (p.up2) {
      const int e = FORM == GEMM_NT ? up_phase * 4 + s_tap : s_tap;
      dy = up2_dy(e); dx = up2_dx(e); wtap = e;
      if (FORM == GEMM_NN) { dy = -dy; dx = -dx; plane = (long)(e >> 2) * p.up_plane * p.lda; }
    } else {
      dy = s_tap / 3 - 1; dx = s_tap - (s_tap / 3) * 3 - 1;
      wtap = p.flip ? p.taps - 1 - s_tap : s_tap;
    }
    ua = plane + (long)(dy * p.Wm + dx) * p.lda + (long)s_c * BK;
    if (FORM == GEMM_NT) ub = (long)wtap * p.b_tap_stride + (long)s_c * BK;
    else ub = (long)wtap * p.b_tap_stride + (long)s_c * BK * p.ldb;
  };
  if (FAST && !CONV) {
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
      const int c = wave + j * NW;
      bool ok = c < NCA;
      if (FORM == GEMM_TN) {
        const int krow = c * 4 + (lane >> 4);
        const int m = m0 + (nc_logical<128>(krow, lane & 15) << 3);
        ok = ok && m < p.M;
        pa[j] = ok ? p.A + ((long)kt_begin * BK + krow) * p.lda + m : zsrc;
        sa[j] = ok ? (long)BK * p.lda : 0;
      } else {
        const int row = c * KC_ROWS + kc_rowl;
        const int m = m0 + row;
        ok = ok && m < p.M;
        pa[j] = ok ? p.A + (long)m * p.lda + (long)kt_begin * BK + ((kc_pv ^ kc_swz<BK>(row)) << 3) : zsrc;
        sa[j] = ok ? BK : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int c = wave + j * NW;
      bool ok = c < NCB;
      if (FORM == GEMM_NT) {
        const int row = c * KC_ROWS + kc_rowl;
        const int n = n0 + row;
        ok = ok && n < p.N;
        pb[j] = ok ? p.B + (long)n * p.ldb + (long)kt_begin * BK + ((kc_pv ^ kc_swz<BK>(row)) << 3) : zsrc;
        sb[j] = ok ? BK : 0;
      } else {
        constexpr int V = BN / 8;
        const int q = c * 64 + lane;
        const int krow = q / V, pv = q - krow * V;
        const int n = n0 + (nc_logical<BN>(krow, pv) << 3);
        ok = ok && n < p.N;
        pb[j] = ok ? p.B + ((long)kt_begin * BK + krow) * p.ldb + n : zsrc;
        sb[j] = ok ? (long)BK * p.ldb : 0;
      }
    }
  }
  if (CF) {
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
      const int c = wave + j * NW;
      if (FORM == GEMM_TN) {
        const int krow = c * 4 + (lane >> 4);
        const int m = m0 + (nc_logical<128>(krow, lane & 15) << 3);
        amask[j] = (c < NCA && m < p.M) ? 1 : 0;
        pa[j] = p.A + ((long)kt_begin * BK + krow + (p.up2 == 1 ? (long)(tap_fixed >> 2) * p.up_plane : 0)) * p.lda + m;      // (up2 == 1: the phase plane of dY)
      } else {
        const int row = c * KC_ROWS + kc_rowl;
        const int m = m0 + row;
        int mask = 0, ml = m;
        if (p.up2) {                               // (tap bit t = stencil entry: NT 4 up_phase + t, NN t)
          ml = m - up_phase * p.up_plane;
          const PixRow r = decode_pix(ml, p.up_rows, p.Hm, p.Wm);
          if (c < NCA && r.ok) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              if (t >= p.taps) break;
              int ys, xs;
              if (FORM == GEMM_NT && (SDXL_UP2_3 && p.up2 == 3)) {
                ys = r.y + (t / 3 == 0 ? -1 : 0);
                xs = r.x + (t % 3 == 0 ? -1 : 0);
              } else if (FORM == GEMM_NN && p.up2 == 2) {
                const int ry = t / dn_nb, rx = t - ry * dn_nb;
                ys = r.y + ((dn_phase >> 1) ? (ry == 0 ? 1 : 0) : 0);
                xs = r.x + ((dn_phase & 1) ? (rx == 0 ? 1 : 0) : 0);
              } else {
                const int e = FORM == GEMM_NT ? up_phase * 4 + t : t;
                ys = FORM == GEMM_NT ? r.y + up2_dy(e) : r.y - up2_dy(e);
                xs = FORM == GEMM_NT ? r.x + up2_dx(e) : r.x - up2_dx(e);
              }
              if (ys >= 0 && ys < p.Hm && xs >= 0 && xs < p.Wm) mask |= 1 << t;
            }
          }
        } else {
        const PixRow r = decode_pix(m, p.M, p.Hm, p.Wm);
        if (c < NCA && r.ok) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ys = r.y + t / 3 - 1, xs = r.x + t % 3 - 1;
            if (ys >= 0 && ys < p.Hm && xs >= 0 && xs < p.Wm) mask |= 1 << t;
          }
        }
        }
        amask[j] = mask;
        pa[j] = p.A + (long)ml * p.lda + ((kc_pv ^ kc_swz<BK>(row)) << 3);
      }
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int c = wave + j * NW;
      if (FORM == GEMM_NT) {
        const int row = c * KC_ROWS + kc_rowl;
        const int n = n0 + row;
        bok[j] = (c < NCB && n < p.N) ? 1 : 0;
        pb[j] = p.B + (long)n * p.ldb + ((kc_pv ^ kc_swz<BK>(row)) << 3);
      } else {
        constexpr int V = BN / 8;
        const int q = c * 64 + lane;
        const int krow = q / V, pv = q - krow * V;
        const int n = n0 + (nc_logical<BN>(krow, pv) << 3);
        bok[j] = (c < NCB && n < p.N) ? 1 : 0;
        if (FORM == GEMM_NN) {
          pb[j] = p.B + (long)krow * p.ldb + n;
        } else {
          const long kk0 = (long)kt_begin * BK + krow;           // first gathered pixel of this lane
          const PixRow r = decode_pix((int)kk0, p.K, p.Hm, p.Wm);
          by_[j] = r.y;
          bx_[j] = r.x;
          pb[j] = p.B + (s3_plane + kk0 + tdy * p.Wm + tdx) * p.ldb + n;    // only dereferenced where the tap is in bounds
        }
      }
    }
    if (FORM != GEMM_TN) {                                       // (tap, channel step) of this workgroup's first K-step
      s_tap = kt_begin / ktiles_per_tap;
      s_c = kt_begin - s_tap * ktiles_per_tap;
      tap_offsets();
    }
  }
  // CF: move the uniform state to the next K-step (called once all pieces of a step have been issued)
  auto advance = [&]() {
    if (!CF) return;
    if (FORM == GEMM_TN) {
      ua += (long)BK * p.lda;
      ub += (long)BK * p.ldb;
      const int dxs = BK % p.Wm, dys = BK / p.Wm;
#pragma unroll
      for (int j = 0; j < BCH; ++j) {
        bx_[j] += dxs;
        if (bx_[j] >= p.Wm) { bx_[j] -= p.Wm; by_[j] += 1; }
        by_[j] += dys;
        if (by_[j] >= p.Hm) by_[j] -= p.Hm;
      }
    } else {
      if (++s_c == ktiles_per_tap) { s_c = 0; ++s_tap; }
      tap_offsets();
    }
  };
  // FAST: DMA piece `pc` (A chunks first, then B chunks) of the step being staged into ring slot `buf`;
  // live = false re-targets the load at the zero vector (branch-free tail).
  bool dma_on = true;
  auto issue_piece = [&](int pc, int buf, bool live) {
    if (dbg && !dma_on) return;
    char* At = smem + buf * STAGE_BYTES;
    char* Bt = At + A_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < ACH; ++j)
      if (pc == j) {
        const int c = wave + j * NW;
        const bf16* src;
        if (CF) {
          const bool v = live && (FORM == GEMM_TN ? amask[j] != 0 : ((amask[j] >> s_tap) & 1) != 0);
          src = v ? pa[j] + ua : zsrc;
        } else {
          src = live ? pa[j] : zsrc;
        }
        char* dst = (NCA % NW == 0 || c < NCA) ? At + c * 1024 : pad_dst;
        if (PL) lds_dma16_global(src, lds_addr_of(dst));      // (from asm: the builtin makes hipcc drain the ring before LDS reads that follow it)
        else __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        if (!CF) pa[j] += sa[j];
      }
#pragma unroll
    for (int j = 0; j < BCH; ++j)
      if (pc == ACH + j) {
        const int c = wave + j * NW;
        const bf16* src;
        if (CF) {
          bool v = live && bok[j] != 0;
          if (FORM == GEMM_TN) {
            const int ys = by_[j] + tdy, xs = bx_[j] + tdx;
            v = v && ys >= 0 && ys < p.Hm && xs >= 0 && xs < p.Wm;
          }
          src = v ? pb[j] + ub : zsrc;
        } else {
          src = live ? pb[j] : zsrc;
        }
        char* dst = (NCB % NW == 0 || c < NCB) ? Bt + c * 1024 : pad_dst;
        if (PL) lds_dma16_global(src, lds_addr_of(dst));
        else __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        if (!CF) pb[j] += sb[j];
      }
  };

  // general path: addresses recomputed per K-step (tap arithmetic, border / tail predicates)
  auto stage_gen = [&](int kt, int buf) {
    int tap = 0, c0;
    if (FORM == GEMM_TN) {
      tap = tap_fixed;
      c0 = kt * BK;
    } else {
      tap = kt / ktiles_per_tap;
      c0 = (kt - tap * ktiles_per_tap) * BK;
    }
    const int dy = tap / 3, dx = tap - dy * 3;
    char* At = smem + buf * STAGE_BYTES;
    char* Bt = At + A_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
      const int c = wave + j * NW;
      const bool cok = NCA % NW == 0 || c < NCA;
      const bf16* src;
      if (FORM == GEMM_TN) {
        const int krow = c * 4 + (lane >> 4);
        const int kk = c0 + krow;
        const int m = m0 + (nc_logical<128>(krow, lane & 15) << 3);
        src = (cok && kk < p.K && m < p.M) ? p.A + (long)kk * p.lda + m : zsrc;
      } else {
        const int row = c * KC_ROWS + kc_rowl;
        const int kofs = c0 + ((kc_pv ^ kc_swz<BK>(row)) << 3);
        if (CONV) {
          long s = gather_src(arow[j], dy, dx, p);
          src = (cok && s >= 0 && kofs < p.K) ? p.A + s * p.lda + kofs : zsrc;
        } else {
          const int m = m0 + row;
          src = (cok && m < p.M && kofs < p.K) ? p.A + (long)m * p.lda + kofs : zsrc;
        }
      }
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(cok ? At + c * 1024 : pad_dst), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int c = wave + j * NW;
      const bool cok = NCB % NW == 0 || c < NCB;
      const bf16* src;
      if (FORM == GEMM_NT) {
        const int wtap = p.flip ? (p.taps - 1 - tap) : tap;
        const int row = c * KC_ROWS + kc_rowl;
        const int n = n0 + row;
        const int kofs = c0 + ((kc_pv ^ kc_swz<BK>(row)) << 3);
        src = (cok && n < p.N && kofs < p.K) ? p.B + (long)n * p.ldb + (long)wtap * p.b_tap_stride + kofs : zsrc;
      } else {
        constexpr int V = BN / 8;
        const int q = c * 64 + lane;
        const int krow = q / V, pv = q - krow * V;
        const int kk = c0 + krow;
        const int n = n0 + (nc_logical<BN>(krow, pv) << 3);
        if (FORM == GEMM_NN) {
          const int wtap = p.flip ? (p.taps - 1 - tap) : tap;
          src = (cok && kk < p.K && n < p.N) ? p.B + (long)kk * p.ldb + (long)wtap * p.b_tap_stride + n : zsrc;
        } else {
          long s;
          if (CONV) {
            PixRow r = decode_pix(kk, p.K, p.Hm, p.Wm);
            s = gather_src(r, dy, dx, p);
          } else {
            s = kk < p.K ? kk : -1;
          }
          src = (cok && s >= 0 && n < p.N) ? p.B + s * p.ldb + n : zsrc;
        }
      }
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(cok ? Bt + c * 1024 : pad_dst), 16, 0, 0);
    }
  };
  auto stage = [&](int kt, int buf) {
    if (FAST) {
#pragma unroll
      for (int pc = 0; pc < NL; ++pc) issue_piece(pc, buf, true);   // steps are staged in increasing order
      advance();
    } else {
      stage_gen(kt, buf);
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient (TN): column sums of the A operand = A^T . ones, on the matrix pipe, by one wave column of the
  // workgroups that own n-tile 0 (and tap 0)
  const bool do_bias = FORM == GEMM_TN && p.bias_grad != nullptr && bx == 0 && (p.up2 == 1 ? (tap_fixed & 3) == 0 : tap_fixed == 0) && wn == 0;      // (up2 == 1: once per phase plane)
  f32x4 accb[MI];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < MI; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // ---- S-deep ring: K-steps t+1 .. t+S-1 are in flight (LDS-DMA) while step t is multiplied ----
  const int T = kt_end - kt_begin;
  constexpr int U = 1;
#pragma unroll
  for (int d = 0; d < S - U; ++d) {
    if (d < T) stage(kt_begin + d, d);
    else if (FAST) {                       // keep the vmcnt arithmetic uniform: same number of loads every step
#pragma unroll
      for (int pc = 0; pc < NL; ++pc) issue_piece(pc, d, false);
      advance();
    }
  }
  int rd = 0, wr = S - U;  // ring slots: read slot of step t, write slot of step t + S - U
  if (dbg & 2) dma_on = false;
  constexpr int KS = BK / 32;
  struct FragK { bf16x8 a[MI], b[NJ]; };          // the fragments of one 32-deep half of a K-step
  // every fragment read of a (K-step, ks) is issued in one go (the compiler's counted lgkmcnt waits then let the
  // products start as soon as their operands land while later reads are still in flight)
  auto read_ks = [&](int slot, int ks, FragK& f) {
    const char* At = smem + slot * STAGE_BYTES;
    const char* Bt = At + A_TILE_BYTES;
    if (dbg & 4) {
#pragma unroll
      for (int i = 0; i < MI; ++i) f.a[i] = ones;
#pragma unroll
      for (int j = 0; j < NJ; ++j) f.b[j] = ones;
      return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (FORM == GEMM_TN)
        f.a[i] = frag_nc<128>(At, ks * 32 + g * 8, wm * (MI * 16) + i * 16, l16);
      else
        f.a[i] = frag_kc<BK>(At, wm * (MI * 16) + i * 16 + l16, ks * 4 + g);
      if (i == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (FORM == GEMM_NT)
            f.b[j] = frag_kc<BK>(Bt, wn * (BN / 2) + j * 16 + l16, ks * 4 + g);
          else
            f.b[j] = frag_nc<BN>(Bt, ks * 32 + g * 8, wn * (BN / 2) + j * 16, l16);
        }
      }
    }
  };
  // one ks worth of products; FAST: the DMA pieces of step t + S - 1 (ring slot `wslot`) ride between the MFMA groups
  auto mfma_ks = [&](const FragK& f, int ks, int wslot, bool live) {
    if (FORM == GEMM_TN && do_bias) {
#pragma unroll
      for (int i = 0; i < MI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i], ones, accb[i], 0, 0, 0);
    }
    if (!FAST) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (dbg & 1) acc[i][j][0] += (float)f.a[i][0] + (float)f.b[j][0];
        else if (DIRECT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[j], f.a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
      }
      if (FAST) {  // one DMA piece per MFMA group
        constexpr int NSLOT = KSP ? MI : KS * MI;
        const int slot = KSP ? i : ks * MI + i;
#pragma unroll
        for (int pc = 0; pc < NL; ++pc)
          if (pc % NSLOT == slot) issue_piece(pc, wslot, live);
      }
    }
    if (!FAST) __builtin_amdgcn_s_setprio(0);
  };
  if constexpr (PL) {
    // Software-pipelined loop, one wave per SIMD.  Register sets f0 / f1 hold the fragments of the two 32-deep halves of a K-step.
    //   phase 0 of step t:  products of (t, half 0) from f0  |  reads of (t, half 1) -> f1        |  first DMA pieces of step t + S - 1
    //   -- counted vmcnt (my pieces of step t + 1 have landed), lgkmcnt(0), ONE barrier --
    //   phase 1 of step t:  products of (t, half 1) from f1  |  reads of (t + 1, half 0) -> f0    |  the other DMA pieces
    // Hazards.  RAW: step t + 1 is first read in phase 1 of step t, behind the barrier that every wave enters after the counted wait
    // that retires its own pieces of that step (pieces land in issue order; outstanding at the wait, oldest first: step t + 1, the
    // steps t + 2 .. t + S - 2, the NP0 pieces issued in phase 0).  WAR: the pieces of step t + S - 1 go to the ring slot of step
    // t - 1, whose last reads (half 1, phase 0 of step t - 1) were retired by that wave's lgkmcnt(0) before the barrier of step t - 1,
    // which every wave has left before any wave reaches step t.
    constexpr int NG = MI;                      // MFMA groups per phase (one fragment row of A each)
    constexpr int NITEM = MI + NJ;              // fragment reads per half-step: B first (every product of a phase's first group needs them), then A
    constexpr int RG = 3;                       // groups that carry reads (the last group of a phase carries none: its reads would be waited for at once)
    auto np0 = []() constexpr { int n = 0; for (int pc = 0; pc < NL; ++pc) if (pc % (2 * NG) < NG) ++n; return n; };      // pieces issued in phase 0
    auto read_item = [&](int slot, int ks, FragK& f, int it) {
      const char* At = smem + slot * STAGE_BYTES;
      const char* Bt = At + A_TILE_BYTES;
      if (dbg & 4) { if (it < NJ) f.b[it] = ones; else f.a[it - NJ] = ones; return; }
      if (it < NJ) {
        const int j = it;
        if (FORM == GEMM_NT) f.b[j] = frag_kc<BK>(Bt, wn * (BN / 2) + j * 16 + l16, ks * 4 + g);
        else f.b[j] = frag_nc<BN>(Bt, ks * 32 + g * 8, wn * (BN / 2) + j * 16, l16);
      } else {
        const int i = it - NJ;
        f.a[i] = frag_kc<BK>(At, wm * (MI * 16) + i * 16 + l16, ks * 4 + g);
      }
    };
    // one phase: PH = 0 / 1; fc = the fragments multiplied, fn = the set being filled from (rslot, rks)
    // (the reads of the step after the last one are not skipped: their slot holds zero-fill pieces that the same counted wait retires)
    auto phase = [&](auto PHC, const FragK& fc, FragK& fn, int rslot, int rks, int wslot, bool live) {
      constexpr int PH = decltype(PHC)::value;
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (dbg & 1) acc[gi][j][0] += (float)fc.a[gi][0] + (float)fc.b[j][0];
          else acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc.b[j], fc.a[gi], acc[gi][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gi < RG) {
#pragma unroll
          for (int it = 0; it < NITEM; ++it)
            if (it * RG / NITEM == gi) read_item(rslot, rks, fn, it);
        }
#pragma unroll
        for (int pc = 0; pc < NL; ++pc)
          if (pc % (2 * NG) == PH * NG + gi) issue_piece(pc, wslot, live);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    FragK f0, f1;
    wait_vmcnt<(S - 2) * NL>();                       // step 0 landed (mine)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int it = 0; it < NITEM; ++it) read_item(rd, 0, f0, it);
    for (int t = 0; t < T; ++t) {
      const bool live = t + S - 1 < T;
      phase(std::integral_constant<int, 0>{}, f0, f1, rd, 1, wr, live);
      wait_vmcnt<(S - 3) * NL + np0()>();             // my pieces of step t + 1 have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      const int rn = rd + 1 == S ? 0 : rd + 1;
      phase(std::integral_constant<int, 1>{}, f1, f0, rn, 0, wr, live);
      advance();
      rd = rn;
      wr = wr + 1 == S ? 0 : wr + 1;
    }
  } else
  if (KSP) {
    // Global barrier sequence b_0, b_1, ...; group 0 reads its half of step t in the interval before b_2t and multiplies it
    // between b_2t and b_2t+1, group 1 one interval later.  DMA of step t+S-1 rides in the MFMA section of step t (after
    // that section's first barrier): its ring slot held step t-1, whose last fragment reads (group 1) were consumed before
    // b_2t.  A wave's counted vmcnt before its first barrier of step t guarantees its pieces of step t+1 (at most the
    // steps t+2 .. t+S-2 stay outstanding); every wave has passed such a wait before any wave reads step t+1.
    wait_vmcnt<(S - 2) * NL>();                       // step 0 landed (mine)
    __builtin_amdgcn_s_barrier();
    if (kg == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind
    for (int t = 0; t < T; ++t) {
      FragK f;
      read_ks(rd, kg, f);
      wait_vmcnt<(S - 3) * NL>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      mfma_ks(f, kg, wr, t + S - 1 < T);
      __builtin_amdgcn_s_setprio(0);
      advance();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      rd = rd + 1 == S ? 0 : rd + 1;
      wr = wr + 1 == S ? 0 : wr + 1;
    }
    if (kg == 0) __builtin_amdgcn_s_barrier();        // both groups execute the same number of barriers
  } else
  // (A register-prefetch variant -- fragments of step t + 1 read while the products of step t run, one ring slot
  // fewer in flight -- was measured on the S >= 3 configurations: 5-20 % slower, so the loop stays as it is.)
  for (int t = 0; t < T; ++t) {
    // step t has landed once at most the S-2 later steps' DMAs of this wave are still outstanding ...
    if (FAST) wait_vmcnt<(S - 2) * NL>();                           // FAST issues NL loads for every step, live or not
    else if (S >= 3 && t + S - 2 < T) wait_vmcnt<(S - 2) * NL>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave has finished reading slot `wr` (step t-1)
    const bool live = t + S - 1 < T;
    if (!FAST && live) stage(kt_begin + t + S - 1, wr);
    FragK f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) read_ks(rd, ks, f[ks]);
    if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) mfma_ks(f[ks], ks, wr, live);
    if (FAST) advance();   // every piece of step t + S - 1 has been issued
    rd = rd + 1 == S ? 0 : rd + 1;
    wr = wr + 1 == S ? 0 : wr + 1;
  }
  wait_vmcnt<0>();   // padding / tail DMA pieces must not outlive the workgroup's LDS allocation
  if (KSP) {
    // the two K-groups hold partial sums of the same 64 x (BN/2) wave tiles: each hands the partner the half (two of the
    // four 16-row fragment rows) the partner finalises, through the ring (now idle) as an fp32 [128][BN + 4] tile
    __syncthreads();
    float* X = (float*)smem;
    // (static fragment indices + a wave-uniform predicate: a runtime index into acc[][] would put the accumulators in scratch)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if ((i >> 1) == kg) continue;            // rows the partner finalises: give them away
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *(f32x4*)(X + (wm * 64 + i * 16 + l16) * LDC + wn * (BN / 2) + j * 16 + g * 4) = acc[i][j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if ((i >> 1) != kg) continue;            // rows this group keeps: add the partner's partial sums
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 o = *(const f32x4*)(X + (wm * 64 + i * 16 + l16) * LDC + wn * (BN / 2) + j * 16 + g * 4);
        acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
      }
    }
  }
  if (FORM == GEMM_TN && do_bias && l16 == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (MI * 16) + i * 16 + g * 4 + r;
        if (m < p.M) {
          if (p.up2 == 1) atomicAdd(p.bias_grad + m, accb[i][r]);      // (four phase planes add: the plan takes the up-sampler's bias gradient from its own column-sum pass)
          else gemm_bias_out(p.bias_grad, p.slab, p.slab_ld, p.splitk, split, p.M, m, accb[i][r]);
        }
      }
  }
  // ---- epilogue, registers -> global.  The products were issued with the operands swapped (D^T layout), so lane
  // (l16, g) holds C[m = 16 i + l16][n = 16 j + 4 g .. + 3]: no LDS staging, no workgroup barrier, a wave retires as
  // soon as its own tile is done.
  //   fp32 (wgrad): one 16-byte store per fragment.
  //   bf16: two neighbouring fragments (j, j+1) trade halves with v_permlane16_swap (odd 16-lane rows of one register
  //   <-> even rows of the other), after which lane g owns 8 contiguous columns of fragment j + (g & 1) at column
  //   8 (g >> 1): one 16-byte store, 64 contiguous bytes per output row and instruction.
  auto store_bf16 = [&](int m, int n, float (&x)[8], int cnt) {   // cnt = 8 or 4 columns
    if (m >= p.M || n >= p.N) return;
    if (cnt == 8) {
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
    } else {
      if (p.bias) {
        bf16x4 bv = *(const bf16x4*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += (float)bv[e];
      }
      if (p.rowvec) {
        bf16x4 tv = *(const bf16x4*)(p.rowvec + (long)(m / p.rows_per_batch) * p.ldv + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += (float)tv[e];
      }
      if (p.resid) {
        bf16x4 rv = *(const bf16x4*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += (float)rv[e];
      }
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)x[e];
      *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = o;
    }
  };
  if (NSPLIT && p.splitk > 1) {      // fp32 partial tile -> this split's slab (same fragment layout as the TN epilogue)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * (MI * 16) + i * 16 + l16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
        if (n >= p.N) continue;
        *(f32x4*)(p.slab + ((long)split * p.M + m) * p.slab_ld + n) = acc[i][j];
      }
    }
    return;
  }
  // ---- LayerNorm backward in the epilogue of the dgrad that produces its dy (GemmP::ln_x; kernels.h) ----
  constexpr bool LN_OK = SDXL_LN_EPILOGUE && FORM == GEMM_NN && !CONV && BN == 128 && NW == 4 && !KSP && FAST && BK == 64 && S == 2;      // (160-column tiles: 80 accumulators + this epilogue spill)
  if constexpr (LN_OK) if (p.ln_x) {
    constexpr int NP = NJ / 2;            // fragment pairs = 8-column chunks per lane and fragment row
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    __syncthreads();                      // every wave is done with the ring: it becomes the exchange area
    float* rowpart = (float*)smem;                 // [2 wave columns][128 rows][2]
    float* rowtot = rowpart + 2 * 128 * 2;         // [2 gather halves][128][2]  S1, S2 of the whole rows
    float* colpart = rowtot + 2 * 128 * 2;         // [2 wave rows][2][BN]
    const int wcol = n0 + wn * (BN / 2);
    const int mrow = m0 + wm * (MI * 16) + l16;    // fragment row i: + 16 i
    float s1[MI], s2[MI], mean[MI], rstd[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const bool mok = mrow + i * 16 < p.M;
      mean[i] = mok ? p.ln_stats[(long)(mrow + i * 16) * 2] : 0.f;
      rstd[i] = mok ? p.ln_stats[(long)(mrow + i * 16) * 2 + 1] : 0.f;
      s1[i] = 0.f; s2[i] = 0.f;
    }
    // pass 1, column chunk by column chunk: dy (bf16-rounded, as a stored dy would be read back; the accumulators keep the exchanged,
    // rounded values for pass 2), the row sums of this wave's columns, the column sums of its rows.  (Predicated loads, chunk by chunk
    // behind scheduling fences: 233 VGPRs; the straight-line form with clamped addresses takes 253, and above 240 a workgroup of this
    // kernel no longer shares a CU with the bias-gradient instance of the co-resident weight-gradient kernel.)
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int j = 2 * jp;
      const int nc = wcol + (j + (g & 1)) * 16 + (g >> 1) * 8;
      const bool nok = nc < p.N;
      bf16x8 gv = z8();
      if (nok) gv = *(const bf16x8*)(p.ln_gamma + nc);
      float pg[8], pb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { pg[e] = 0.f; pb[e] = 0.f; }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow + i * 16;
        const bool ok = nok && m < p.M;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float lo = acc[i][j][r], hi = acc[i][j + 1][r];
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states: VALU write -> permlane read)
          lo = (float)(bf16)lo; hi = (float)(bf16)hi;
          acc[i][j][r] = lo; acc[i][j + 1][r] = hi;
          v[r] = lo; v[4 + r] = hi;
        }
        if (p.C && ok) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + nc) = o;
        }
        bf16x8 xv = z8();
        if (ok) xv = *(const bf16x8*)(p.ln_x + (long)m * p.ln_ldx + nc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = ((float)xv[e] - mean[i]) * rstd[i];
          const float t = v[e] * (float)gv[e];
          s1[i] += t; s2[i] += t * xh;
          pg[e] += v[e] * xh; pb[e] += v[e];
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps hipcc from hoisting every chunk's loads to the top: registers)
      }
      if (p.ln_pcol) {      // column sums over this wave's 64 rows: over the fragment rows in the lane (done), then over the 16 lanes of a row group
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sg = row16_sum(pg[e]), sb = row16_sum(pb[e]);
          if (l16 == 15) {
            colpart[(wm * 2 + 0) * BN + (nc - n0) + e] = sg;
            colpart[(wm * 2 + 1) * BN + (nc - n0) + e] = sb;
          }
        }
      }
    }
    static_assert(!LN_OK || (NJ & 1) == 0, "LayerNorm-backward epilogue: whole fragment pairs");
    // a row's columns of this wave sit in the four lanes l16 + 16 g: two exchange steps finish its sums (all four lanes get them)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float a = s1[i], b2 = s1[i];
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
      float t = a + b2; a = t; b2 = t;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
      const float r1 = a + b2;
      a = s2[i]; b2 = s2[i];
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
      t = a + b2; a = t; b2 = t;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
      const float r2 = a + b2;
      if (g == 0) {
        float* rp = rowpart + ((wn * 128) + wm * (MI * 16) + i * 16 + l16) * 2;
        rp[0] = r1; rp[1] = r2;
      }
    }
    __syncthreads();
    const int ncolt = gridDim.x;
    const long Mpad = (long)gridDim.y * BMT;
    typedef unsigned long long u64;
    u64* const slots = (u64*)p.ln_part;      // [column tile][row]{(S1, tag), (S2, tag)}: 8-byte granules, tag = this launch's epoch
    const u64 tag = (u64)(unsigned)p.ln_epoch << 32;
    if (tid < BMT) {     // this workgroup's partial sums of row tid: write-through (agent-scope) 8-byte stores, each carrying the tag
      u64* dst = slots + ((long)bx * Mpad + m0 + tid) * 2;
      const float q1 = rowpart[tid * 2] + rowpart[(128 + tid) * 2], q2 = rowpart[tid * 2 + 1] + rowpart[(128 + tid) * 2 + 1];
      __hip_atomic_store(dst, tag | (u64)__float_as_uint(q1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, tag | (u64)__float_as_uint(q2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.ln_pcol && tid < BN && n0 + tid < p.N) {
      float* dst = p.ln_pcol + (long)by * 2 * p.N + n0 + tid;
      dst[0] = colpart[tid] + colpart[2 * BN + tid];
      dst[p.N] = colpart[BN + tid] + colpart[3 * BN + tid];
    }
    // meet the other column tiles of the row block.  ONE lane per column tile polls that tile's ready flag (set behind its granules: stores
    // drained, workgroup barrier, then the flag) -- 320 workgroups x 256 threads polling the granules themselves is a 50 TB/s request storm
    // on the fabric whenever the workgroups of a row block do not finish together, which is exactly when the others need the fabric
    // (fused step 110.1 -> measured again below) -- then every thread gathers once.
    unsigned* const flags = (unsigned*)(slots + (long)ncolt * Mpad * 2);      // [column tile][row block]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + (long)bx * gridDim.y + by, (unsigned)p.ln_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      int spins = 0;
      bool ok;
      do {
        unsigned f = (unsigned)p.ln_epoch;
        if (tid < ncolt) f = __hip_atomic_load(flags + (long)tid * gridDim.y + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = __all(f == (unsigned)p.ln_epoch);
        if (!ok) __builtin_amdgcn_s_sleep(4);
      } while (!ok && ++spins < (1 << 22));
      if (!ok && tid == 0) atomicOr(&g_ln_spin_error, 1u);      // gave up: surfaced by sdxl_ln_error, like the stream-K kernel's error word
    }
    __syncthreads();
    {   // thread (row, h) gathers the tiles c = 2 k + h (the tags are still checked: a granule that is not there yet is polled for)
      constexpr int KC = 5;      // <= 10 column tiles (launcher-checked)
      const int row = tid & (BMT - 1), h = tid >> 7;
      u64 a[KC], b[KC];
      int spins = 0;
      bool ok;
      do {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int c = 2 * k + h;
          a[k] = tag; b[k] = tag;
          if (c < ncolt) {
            const u64* src = slots + ((long)c * Mpad + m0 + row) * 2;
            a[k] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b[k] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        ok = true;
#pragma unroll
        for (int k = 0; k < KC; ++k) ok = ok && (a[k] >> 32) == (tag >> 32) && (b[k] >> 32) == (tag >> 32);
        if (!ok) __builtin_amdgcn_s_sleep(8);
      } while (!ok && ++spins < (1 << 20));
      if (!ok) atomicOr(&g_ln_spin_error, 2u);
      float S1 = 0.f, S2 = 0.f;      // fixed order: the same sums in every workgroup of the row block, run to run
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (2 * k + h < ncolt) { S1 += __uint_as_float((unsigned)a[k]); S2 += __uint_as_float((unsigned)b[k]); }
      rowtot[(h * BMT + row) * 2] = S1;
      rowtot[(h * BMT + row) * 2 + 1] = S2;
    }
    __syncthreads();
    // pass 2: dx = rstd (dy g - S1 / N - xhat S2 / N) (+ addend); x is read again (L2)
    float S1[MI], S2[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * (MI * 16) + i * 16 + l16;
      const float invn = 1.f / (float)p.N;
      S1[i] = (rowtot[row * 2] + rowtot[(BMT + row) * 2]) * invn; S2[i] = (rowtot[row * 2 + 1] + rowtot[(BMT + row) * 2 + 1]) * invn;
    }
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int j = 2 * jp;
      const int nc = wcol + (j + (g & 1)) * 16 + (g >> 1) * 8;
      if (nc >= p.N) continue;
      const bf16x8 gv = *(const bf16x8*)(p.ln_gamma + nc);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow + i * 16;
        if (m >= p.M) continue;
        const bf16x8 xv = *(const bf16x8*)(p.ln_x + (long)m * p.ln_ldx + nc);
        bf16x8 o;
        if (p.ln_addend) o = *(const bf16x8*)(p.ln_addend + (long)m * p.ln_ldo + nc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = e < 4 ? acc[i][j][e & 3] : acc[i][j + 1][e & 3];
          const float xh = ((float)xv[e] - mean[i]) * rstd[i];
          float d = rstd[i] * (v * (float)gv[e] - S1[i] - xh * S2[i]);
          if (p.ln_addend) d += (float)o[e];
          o[e] = (bf16)d;
        }
        *(bf16x8*)(p.ln_dx + (long)m * p.ln_ldo + nc) = o;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
  if (FORM == GEMM_TN || !p.geglu) {
    if (FORM != GEMM_TN) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // 24 wait states: MFMA results -> inline-asm VALU reads below (XDL write -> VALU read needs up to 18)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (KSP && (i >> 1) != kg) continue;      // split-K groups: each group finalises two of the four fragment rows
      const int m = m0 + wm * (MI * 16) + i * 16 + l16;
      if (FORM == GEMM_TN) {
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
          if (n >= p.N) continue;
          f32x4 x = acc[i][j];
          float* c = (float*)p.C + (long)m * p.ldc + (long)tap_fixed * p.c_tap_stride + n;
          if (p.splitk > 1) {   // deterministic split-K: partial tile to this split's slab, summed by splitk_reduce_kernel
            *(f32x4*)(p.slab + ((long)split * p.M + m) * p.slab_ld + (long)tap_fixed * p.c_tap_stride + n) = x;
          } else if (p.Cb) {    // final value straight to the bf16 exchange arena (C is read for the sum, not written)
            if (p.accumulate) {
              const f32x4 a = *(const f32x4*)c;
              x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3];
            }
            bf16x4 o;
            o[0] = (bf16)(x[0] * p.cb_scale); o[1] = (bf16)(x[1] * p.cb_scale);
            o[2] = (bf16)(x[2] * p.cb_scale); o[3] = (bf16)(x[3] * p.cb_scale);
            *(bf16x4*)(p.Cb + (long)m * p.ldc + (long)tap_fixed * p.c_tap_stride + n) = o;
          } else if (p.accumulate) {
            f32x4 a = *(f32x4*)c;
            a[0] += x[0]; a[1] += x[1]; a[2] += x[2]; a[3] += x[3];
            *(f32x4*)c = a;
          } else {
            *(f32x4*)c = x;
          }
        }
      } else {
        constexpr bool DELTA_OK = FORM == GEMM_NN && BN == 128 && NW == 4 && !KSP;     // wave tile 64 x 64 = one attention head per wave
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j + 1 < NJ; j += 2) {
          float x[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (inline asm: hipcc 7.2 folds four __builtin_amdgcn_permlane16_swap calls on the elements of a vector
            //  into one and reuses its result for all of them)
            float lo = acc[i][j][r], hi = acc[i][j + 1][r];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states: VALU write -> permlane read)
            x[r] = lo;
            x[4 + r] = hi;
          }
          const int n = n0 + wn * (BN / 2) + (j + (g & 1)) * 16 + (g >> 1) * 8;
          store_bf16(m, n, x, 8);
          if (DELTA_OK && p.delta_out && m < p.M && n < p.N) {      // (x now holds the pre-rounding sums: the same bf16 as stored)
            const bf16x8 ov = *(const bf16x8*)(p.delta_o + (long)m * p.delta_ldo + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)(bf16)x[e] * (float)ov[e];
          }
        }
        if (DELTA_OK && p.delta_out) {      // the row's 64 columns sit in the four lanes l16 + 16 g: two exchange steps finish the sum
          float a = dl, b2 = dl;
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
          dl = a + b2;
          a = dl; b2 = dl;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2));
          dl = a + b2;
          if (g == 0 && m < p.M) {
            const int bb = m / p.delta_nq, qq = m - bb * p.delta_nq, hh = (n0 + wn * (BN / 2)) >> 6;
            if (hh < p.delta_heads) p.delta_out[((long)bb * p.delta_heads + hh) * p.delta_nq + qq] = dl;
          }
        }
        if (NJ & 1) {   // odd fragment count (BN = 160): the last fragment goes out in 8-byte pieces
          float x[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = acc[i][NJ - 1][r];
          store_bf16(m, n0 + wn * (BN / 2) + (NJ - 1) * 16 + g * 4, x, 4);
        }
      }
    }
    return;
  }
  __syncthreads();  // all fragment reads (and padding DMA) done before the ring is reused as the fp32 staging tile
  // ---- GEGLU epilogues (value and gate of a channel sit in different waves): the fp32 tile is staged in LDS 64 rows
  // at a time (34 KiB), then each thread handles 8 channels of one row ----
  float* Cs = (float*)smem;
  constexpr int VPR = BN / 8;  // 8-column vectors per tile row
#pragma unroll 1
  for (int half = 0; half < BMT / 64; ++half) {
    const int wrow = wm * (MI * 16);           // this wave's first row in the tile
    if (wrow / 64 == half) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (KSP && (i >> 1) != kg) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          *(f32x4*)(Cs + (wrow - half * 64 + i * 16 + l16) * LDC + wn * (BN / 2) + j * 16 + g * 4) = acc[i][j];
      }
    }
    __syncthreads();
    if (FORM != GEMM_TN && (BN == 128 || BN == 160)) {
      // GEGLU fused epilogues (see GemmP): both operate on the bf16-rounded values, exactly like a separate pass would
      constexpr int G = BN / 2;   // channels per packing group = half a tile
      for (int id = tid; id < 64 * VPR; id += NT) {
        const int row = id / VPR, col = (id - row * VPR) * 8;
        const int m = m0 + half * 64 + row;
        if (m >= p.M) continue;
        if (FORM == GEMM_NT) {          // forward: this tile = value | gate halves of 64 channels
          if (col >= G || n0 + col >= p.N) continue;
          const int na = n0 + col, nt = na + G;
          float xa[8], xt[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { xa[e] = Cs[row * LDC + col + e]; xt[e] = Cs[row * LDC + G + col + e]; }
          if (p.bias) {
            bf16x8 ba = *(const bf16x8*)(p.bias + na), bt = *(const bf16x8*)(p.bias + nt);
#pragma unroll
            for (int e = 0; e < 8; ++e) { xa[e] += (float)ba[e]; xt[e] += (float)bt[e]; }
          }
          bf16x8 oa, ot, og;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            oa[e] = (bf16)xa[e];
            ot[e] = (bf16)xt[e];
            og[e] = (bf16)((float)oa[e] * gelu_f((float)ot[e]));
          }
          *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + na) = oa;
          *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + nt) = ot;
          *(bf16x8*)(p.aux + (long)m * p.ldaux + (n0 >> 1) + col) = og;
        } else {                        // dgrad of the second projection: dG tile -> dU (value and gate halves); any packing group
          const int n = n0 + col;       // (rows are read and written in whole 16-byte vectors along the row: coalesced, which the
          if (n >= p.N) continue;       //  register epilogue's 64-byte pieces per row are not -- measured 81 vs 75 us at level 2)
          const int Gp = p.geglu_group;
          const long cu = (long)(n / Gp) * (2 * Gp) + (n % Gp);
          bf16x8 ua = *(const bf16x8*)(p.aux + (long)m * p.ldaux + cu);
          bf16x8 ut = *(const bf16x8*)(p.aux + (long)m * p.ldaux + cu + Gp);
          bf16x8 oa, ot;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dv = (float)(bf16)Cs[row * LDC + col + e], tv = (float)ut[e];
            float cdf, pdf;
            gelu_cdf_pdf(tv, &cdf, &pdf);
            oa[e] = (bf16)(dv * tv * cdf);
            ot[e] = (bf16)(dv * (float)ua[e] * fmaf(tv, pdf, cdf));
          }
          *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + cu) = oa;
          *(bf16x8*)((bf16*)p.C + (long)m * p.ldc + cu + Gp) = ot;
        }
      }
    }
    __syncthreads();
  }
}

// C[m][0..cols) (+)= sum_s slab[s][m][0..cols)   (fixed summation order)
// bias_grad (optional): bias_grad[m] += sum_s tail[s][m], the splits' bias-gradient partials behind the tiles (gemm_bias_out), same fixed order
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int cols, long ldc,
                                     long slab_ld, int splitk, int accumulate, bf16* __restrict__ Cb, float cb_scale,
                                     float* __restrict__ bias_grad) {
  const int vpr = cols / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * vpr; i += (long)gridDim.x * blockDim.x) {
    long m = i / vpr;
    int c = (int)(i - m * vpr) * 4;
    if (bias_grad && c == 0) {
      const float* tail = slab + (long)splitk * M * slab_ld + m;
      float b = 0.f;
      for (int s = 0; s < splitk; ++s) b += tail[(long)s * M];
      bias_grad[m] += b;
    }
    f32x4 a = accumulate ? *(const f32x4*)(C + m * ldc + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splitk; ++s) {
      f32x4 v = *(const f32x4*)(slab + ((long)s * M + m) * slab_ld + c);
      a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    if (Cb) {     // GemmP::Cb: the final value as bf16 to the exchange arena, C untouched
      bf16x4 o;
      o[0] = (bf16)(a[0] * cb_scale); o[1] = (bf16)(a[1] * cb_scale); o[2] = (bf16)(a[2] * cb_scale); o[3] = (bf16)(a[3] * cb_scale);
      *(bf16x4*)(Cb + m * ldc + c) = o;
    } else {
      *(f32x4*)(C + m * ldc + c) = a;
    }
  }
}

// split-K of the bf16-output forms: C[m][n] = bf16(sum_s slab[s][m][n] + bias[n] + rowvec[m / rpb][n] + resid[m][n]), fixed order
__global__ void splitk_epilogue_kernel(const float* __restrict__ slab, bf16* __restrict__ C, int M, int N, long ldc, long slab_ld,
                                       int splitk, const bf16* __restrict__ bias, const bf16* __restrict__ resid, long ldr,
                                       const bf16* __restrict__ rowvec, long ldv, int rows_per_batch) {
  const int vpr = N / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * vpr; i += (long)gridDim.x * blockDim.x) {
    const long m = i / vpr;
    const int n = (int)(i - m * vpr) * 8;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splitk; ++s) {
      const float* sp = slab + ((long)s * M + m) * slab_ld + n;
      const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
      x[0] += a[0]; x[1] += a[1]; x[2] += a[2]; x[3] += a[3]; x[4] += b[0]; x[5] += b[1]; x[6] += b[2]; x[7] += b[3];
    }
    if (bias) {
      const bf16x8 v = *(const bf16x8*)(bias + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += (float)v[e];
    }
    if (rowvec) {
      const bf16x8 v = *(const bf16x8*)(rowvec + (m / rows_per_batch) * ldv + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += (float)v[e];
    }
    if (resid) {
      const bf16x8 v = *(const bf16x8*)(resid + m * ldr + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += (float)v[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)x[e];
    *(bf16x8*)(C + m * ldc + n) = o;
  }
}

// Configuration (1: 128 x 128 tiles) of an NN launch whose epilogue runs a LayerNorm backward (GemmP::ln_x), 0: it cannot -- all workgroups
// of the launch must fit the chip at once (2 per CU: the column tiles of a row block wait for each other), and the epilogue exists on
// 128-column tiles only (with the 80 accumulators of a 160-column tile it spills).
int gemm_ln_cfg(int M, int N, int K) {
  if (K % 64 || N % 8 || M <= 0) return 0;
  const long t128 = (long)cdiv(M, BM) * cdiv(N, 128);
  return t128 <= 512 && cdiv(N, 128) <= 10 ? 1 : 0;
}
size_t gemm_ln_part_floats(int M, int N) {      // GemmP::ln_part: two 8-byte granules per (column tile, row) + a ready flag per (column tile, row block)
  return (size_t)cdiv(N, 128) * cdiv(M, BM) * BM * 4 + (size_t)cdiv(N, 128) * cdiv(M, BM) + 4;
}
size_t gemm_ln_pcol_floats(int M, int N) { return (size_t)cdiv(M, BM) * 2 * N; }                      // GemmP::ln_pcol
int gemm_ln_rowblocks(int M) { return cdiv(M, BM); }

#ifdef SDXL_DIAG
int gemm_ln_error(unsigned* out) {      // != 0: a LayerNorm-backward epilogue gave up its in-launch meeting since the last call (results invalid); clears the word
  HIP_CHECK_RET(hipDeviceSynchronize());
  HIP_CHECK_RET(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ln_spin_error), sizeof(unsigned)));
  const unsigned z = 0;
  HIP_CHECK_RET(hipMemcpyToSymbol(HIP_SYMBOL(g_ln_spin_error), &z, sizeof(unsigned)));
  return 0;
}
#endif
size_t gemm_slab_floats(int M, int N, int taps, int splitk) { return splitk > 1 ? (size_t)splitk * M * N * taps + (size_t)splitk * M : 0; }   // tiles + bias tail

void gemm_defaults(GemmP* p) {
  memset(p, 0, sizeof(*p));
  p->taps = 1;
  p->sm = 1;
  p->sd = 1;
  p->splitk = 1;
  p->rows_per_batch = 1;
}

template <int FORM, bool CONV, int BN, int S, int BK, bool FAST, int NW, bool KSP = false, bool PL = false>
static int launch_k(const GemmP& p, hipStream_t st) {
  static bool attr_set = false;
  constexpr int smem0 = gemm_smem_bytes(BN, S, BK, NW);
  // knob 39 (experiment) = a + 1000 b: a KiB of LDS per workgroup for the NN (dgrad) launches, b KiB for the TN ones -- at most one of them per CU
  const int kreq = FORM == GEMM_NN ? KNOB(39) % 1000 : FORM == GEMM_TN ? KNOB(39) / 1000 : 0;
  const int smem = (kreq * 1024 > smem0 && kreq <= 100 && smem0 <= 80 * 1024) ? kreq * 1024 : smem0;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_kernel<FORM, CONV, BN, S, BK, FAST, NW, KSP, PL>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem0 > 100 * 1024 ? smem0 : 100 * 1024));
    attr_set = true;
  }
  dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), FORM == GEMM_TN ? (p.group > 1 ? p.group : p.taps * p.splitk) : p.splitk);
  if (FORM == GEMM_TN) {      // weight gradients whose tile grid fits no px x py rectangle of XCDs: the generic compact order (gemm_tiles.h)
    GemmP q = p;
    const bool rect = p.xcd_px > 0 && grid.x % p.xcd_px == 0 && grid.y % (8 / p.xcd_px) == 0;
    q.xcd_bh = (!rect && KNOB(34) != 1) ? xcd_band_rows(grid.x, grid.y, grid.z, BM, BN, p.group > 1 ? 1 : p.taps) : 0;
    GEMM_LAUNCH((gemm_kernel<FORM, CONV, BN, S, BK, FAST, NW, KSP, PL>), grid, dim3(NW * 64), smem, st, q);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  GEMM_LAUNCH((gemm_kernel<FORM, CONV, BN, S, BK, FAST, NW, KSP, PL>), grid, dim3(NW * 64), smem, st, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// the FAST staging forms only (reduction a multiple of BK; linear, or the same-size stride-1 3x3 gather): can this problem take them?
static bool gemm_fast_ok(const GemmP& p, bool conv, int BK) {
  if (p.K % BK) return false;
  if (!conv) return true;
  return (p.taps == 9 || p.up2) && p.sm == 1 && p.sd == 1 && p.Hm == p.Hs && p.Wm == p.Ws;
}
template <int FORM, bool CONV, int BN, int S, int BK, int NW, bool KSP = false, bool PL = false>
static int launch_cfg(const GemmP& p, hipStream_t st) {
  constexpr bool KS_OK = KSP && FORM != GEMM_TN;     // (the split-K groups exist for the FAST staging of the NT / NN forms)
  constexpr bool PL_OK = PL && FORM != GEMM_TN;      // (the pipelined loop likewise; launch_one routes only FAST problems to it)
  if (!CONV && p.K % BK == 0) return launch_k<FORM, false, BN, S, BK, true, NW, KS_OK, PL_OK>(p, st);
  // same-size stride-1 3x3 convolutions (all but the two downsamplers, their transposed dgrads and conv_in)
  if (CONV && (p.taps == 9 || p.up2) && p.sm == 1 && p.sd == 1 && p.Hm == p.Hs && p.Wm == p.Ws && p.K % BK == 0)
    return launch_k<FORM, true, BN, S, BK, true, NW, KS_OK, PL_OK>(p, st);
  if constexpr (PL) { ARG_CHECK(false, "gemm: the pipelined configurations take FAST-staging problems only"); }
  else return launch_k<FORM, CONV, BN, S, BK, false, NW>(p, st);
}

// Tile / pipeline selection of the 128-row kernel.  Four configurations:
//   1: 128x128, BK 64, 2-deep ring, 4 waves  (65 KiB LDS, 2 workgroups per CU)          -- default
//   2: 128x128, BK 32, 2-deep ring, 4 waves  (34 KiB LDS, 3-4 per CU)                   -- large-grid dgrad
//   3: 128x160, BK 64, 4-deep ring, 8 waves  (145 KiB LDS, 1 per CU, 3 K-steps of DMA in flight)
//  13: 128x160, BK 64, 2-deep ring, 4 waves x (64x80)  (73 KiB, 2 per CU)               -- where 160-wide tiles fill the rounds
// (Measured and dropped: 4-deep / 3-deep 128x128 rings, 6-deep BK 32 rings, a 64x160 tile, a 256x128 ping-pong schedule --
//  DESIGN.md section 10.)
template <int FORM, bool CONV>
static int launch_one(const GemmP& p, hipStream_t st) {
  int cfg = 1;
  // Tile-count quantisation decides most of it on this model's shapes: a launch of t workgroups runs in
  // ceil(t / 512) rounds of (256 CUs x 2 resident workgroups), so e.g. an N = 640 output at M = 16384 is 640 tiles of
  // 128x128 (1.25 rounds, 62 % of the slots used) but exactly 512 tiles of 128x160.
  const long zmul = FORM == GEMM_TN ? p.taps * p.splitk : 1;
  const long blocks = (long)cdiv(p.M, BM) * cdiv(p.N, 128) * zmul;
  const bool g80 = p.geglu == 1 && p.geglu_group == 80;   // forward GEGLU packed for 160-column tiles: must run in a BN = 160 configuration
  const bool n160 = p.N % 160 == 0 && (p.geglu != 1 || g80);
  const long t160 = n160 ? (long)cdiv(p.M, BM) * (p.N / 160) * zmul : 0;
  auto fill = [](long t) { const long cap = 512; return (double)t / (double)(((t + cap - 1) / cap) * cap); };
  // the transpose-read forms (dgrad) spend twice the LDS-read issue slots per K-step: the BK = 32 variant with 3-4
  // workgroups per CU hides that better on large grids
  if (FORM == GEMM_NN) cfg = blocks >= 700 ? 2 : 1;
  // 128x160 tiles, 4 waves x (64x80), 2-deep, 73 KiB (2 per CU): where they fill the rounds at least as well as 128x128
  // they also move 10 % fewer operand bytes per flop (measured +5..45 %: 16384x640x2560 641 -> 931 TFLOP/s)
  if (n160 && t160 > 256 && fill(t160) * 1.05 >= fill(blocks)) cfg = 13;
  // wgrad (side stream): always the wider tile where the output allows it -- 20 % fewer workgroups competing with the
  // dgrad chain for CU slots is worth more at step level (-0.7 ms) than the tile's own speed
  if (n160 && FORM == GEMM_TN) cfg = 13;
  // forward problems that fit one round of one 128x160 workgroup per CU (N = 1280 outputs at M = 4096, the 1280-channel
  // convs): the 8-wave, 4-deep-ring configuration keeps 3 K-steps of DMA in flight and wins +15..30 % there.  Not for
  // dgrad / wgrad: a one-per-CU workgroup on one stream starves the other stream's kernels of LDS (168 vs 152 ms/step).
  if (FORM == GEMM_NT && n160 && t160 <= 256) cfg = 3;
  // the same tile with the eight waves as two staggered split-K groups (configuration 23): 10-15 % less time per K-step,
  // +1.6 us per launch for the exchange of the partial tiles -- long reductions that fit one round
  // (forward only: in the backward a 145 KiB workgroup evicts the other stream from its CU -- dgrads on it: GEMM family
  //  -2 ms, step +2.4 ms, profiles/r02c)
  if (FORM == GEMM_NT && n160 && t160 <= 256 && (long)p.K * p.taps >= 2560) cfg = 23;
  // the pipelined one-wave-per-SIMD loop on the same tiles (configuration 5; 6 = 128-column tiles): knob 30 (kernels.h) selects where
  {
    const int k30 = KNOB(30);
    const bool one_round = (n160 ? t160 : blocks) <= 256 && (n160 ? t160 : blocks) > 128;
    const bool fastp = FORM != GEMM_TN && gemm_fast_ok(p, CONV, 64) && p.splitk <= 1;
    if (fastp && one_round) {
      const bool fwd_conv = FORM == GEMM_NT && CONV, dg_conv = FORM == GEMM_NN && CONV;      // (linear problems: gemm_pl.hip, launch_gemm_impl)
      if ((fwd_conv && (k30 & 4)) || (dg_conv && (k30 & 8))) cfg = n160 ? 5 : 6;
    }
  }
  if (p.cfg > 0) cfg = p.cfg;
  if (g_force_cfg > 0) cfg = g_force_cfg;
  if (p.delta_out) cfg = 1;      // the epilogue that also writes an attention layer's Delta exists on 128 x 128 tiles only (whatever is forced)
  if (p.ln_x) cfg = 1;      // the LayerNorm-backward epilogue exists on 128 x 128 tiles only (gemm_ln_cfg, launcher-checked)
  if (FORM != GEMM_TN && p.splitk > 1) cfg = n160 ? 13 : 1;     // split-K of the bf16-output forms: the 4-wave FAST configurations
  if (p.geglu == 1 && !g80 && (cfg == 3 || cfg == 13 || cfg == 23 || cfg == 43)) cfg = 1;   // forward, group-64 packing: 128-column tiles
  if (g80 && cfg != 3 && cfg != 5 && cfg != 13 && cfg != 23 && cfg != 43) cfg = 13;                 // group-80 packing needs 160-column tiles
  // 5 / 6: the pipelined one-wave-per-SIMD loop on 128 x 160 / 128 x 128 tiles (4-deep ring, the workgroup owns its CU): NT / NN problems with
  // FAST staging only
  if (cfg == 5 && !n160) cfg = 6;
  if (cfg == 6 && g80) cfg = 5;
  if ((cfg == 5 || cfg == 6) && (FORM == GEMM_TN || !gemm_fast_ok(p, CONV, 64))) cfg = cfg == 5 ? 13 : 1;
  if ((cfg == 3 || cfg == 13 || cfg == 23 || cfg == 43) && p.N % 160 != 0) cfg = 1;
  if (cfg == 23 && FORM == GEMM_TN) cfg = 13;
  if (p.geglu == 1 && cfg == 2) cfg = 1;                      // the BK = 32 configuration has no room for the GEGLU staging tile
  switch (cfg) {
    case 2: return launch_cfg<FORM, CONV, 128, 2, 32, 4>(p, st);
    case 3: return launch_cfg<FORM, CONV, 160, 4, 64, 8>(p, st);
    case 23: return launch_cfg<FORM, CONV, 160, 4, 64, 8, true>(p, st);   // split-K wave groups
    case 13: return launch_cfg<FORM, CONV, 160, 2, 64, 4>(p, st);
    case 5: return launch_cfg<FORM, CONV, 160, 4, 64, 4, false, true>(p, st);      // pipelined, one wave per SIMD, 145 KiB
    case 6: return launch_cfg<FORM, CONV, 128, 4, 64, 4, false, true>(p, st);      // the same on 128 x 128 tiles, 128 KiB
#ifdef SDXL_DIAG
    case 43: return launch_cfg<FORM, CONV, 160, 4, 64, 4>(p, st);      // the 4-deep ring with FOUR waves (64 x 80 wave tiles: 74 KB of fragment reads per K-step against 115 with eight): 73.9 vs 67.3 us on NT 4096 x 1280 x 5120, not selected
#endif
    default: return launch_cfg<FORM, CONV, 128, 2, 64, 4>(p, st);
  }
}

// ---- optional per-launch timing (bench.py roofline): HIP events on the launch stream around every GEMM launch ----
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  std::vector<double> flops;
  struct Rec { int form, taps, M, N, K, splitk; };
  std::vector<Rec> recs;
};
static GemmProf g_prof;
bool gemm_profiling() { return g_prof.on; }
int gemm_profile_begin() {
  g_prof.on = true;
  g_prof.used = 0;
  g_prof.flops.clear();
  g_prof.recs.clear();
  return 0;
}
int gemm_profile_end(double* flops, double* ms, int* launches) {
  g_prof.on = false;
  double f = 0, t = 0;
  HIP_CHECK_RET(hipDeviceSynchronize());
  FILE* dump = nullptr;   // SDXL_GEMM_PROF_DUMP=<path>: per-launch records (appended), for shape-level analysis
  if (const char* dp = getenv("SDXL_GEMM_PROF_DUMP")) dump = fopen(dp, "a");
  for (size_t i = 0; i < g_prof.flops.size(); ++i) {
    float e = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&e, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    t += e;
    f += g_prof.flops[i];
    if (dump) {
      const GemmProf::Rec& r = g_prof.recs[i];
      fprintf(dump, "%d,%d,%d,%d,%d,%d,%.4f\n", r.form, r.taps, r.M, r.N, r.K, r.splitk, e);
    }
  }
  if (dump) fclose(dump);
  if (flops) *flops = f;
  if (ms) *ms = t;
  if (launches) *launches = (int)g_prof.flops.size();
  return 0;
}
// Selection between the two kernels (measured on MI355X, profiles/r02b_gemm_bench.txt, r02c_g256_insitu.txt).  The 256 x 256
// kernel runs one 128 KiB-LDS workgroup per CU, so what counts is how well its tiles fill rounds of 256 CUs and how many
// K-tiles amortise its ~12 us of unoverlapped prologue + epilogue per tile (with 2-3 workgroups per CU the 128-row kernel
// overlaps those by itself).  In isolation: wgrad (TN, reduction over 4096+ rows) +25..35 % wherever tiles x splits reaches
// ~3/4 of a round; forward / dgrad only where the tiles fill their rounds (>= 90 %): 4096x3840x1280 (240 tiles) +5..20 %,
// 16384x5120x640 (5 rounds) +12 %; 640 tiles (2.5 rounds) or 320 tiles lose to the 128-row kernel.
// In the training step the backward runs wgrad (side stream) and dgrad / attention / norms (caller's stream) CONCURRENTLY,
// and that overlap lives on LDS co-residency (two 65 KiB workgroups per CU): a 128 KiB workgroup on either stream evicts the
// other stream from its CU -- with the wgrads on the 256 x 256 kernel the GEMM family itself got 3.4 ms faster and the step
// 1.7-3 ms slower; shrinking the dgrad kernels to 32 KiB (BK = 32) so that they fit beside a 128 KiB workgroup: +7 ms.
// Policy: the 256 x 256 kernel in the forward pass only (NT form; nothing else competes for the CUs there).
bool gemm_use256(int form, int M, int N, int K, int splitk) {
  if (M % 256 || N % 256 || K % 64) return false;
  if (form != GEMM_NT) return false;
  const long wgs = (long)(M / 256) * (N / 256);
  const double fill = (double)wgs / (double)(((wgs + 255) / 256) * 256);
  return wgs >= 192 && (fill >= 0.9 || wgs >= 1024);
}
// split-K factor of a wgrad GEMM [M][N*taps] (+)= A^T . B over `red` rows
int gemm_pick_splitk(int M, int N, int taps, long red) {
  if (taps == 1 && g_mode256 == 2 && M % 256 == 0 && N % 256 == 0 && red % 64 == 0) {   // forced 256 x 256 kernel: one round of tiles
    const long t256 = (long)(M / 256) * (N / 256);
    if (t256 >= 48) {
      long s = (224 + t256 / 2) / t256;
      if (s < 1) s = 1;
      while (s > 1 && red / 64 / s < 8) --s;
      return (int)s;
    }
  }
  // 128-row kernel: enough workgroups to fill 256 CUs x 2 (tiles x splits ~ 384), at least 8 K-steps per split
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 128) * taps;
  const long ktiles = cdiv(red, 64);
  long s = 384 / tiles;       // (100 / 200 / 256 / 512 / 768 measured in the step: +12 / +2.5 / +0.6 / +0.9 / +4 ms)
  if (s < 1) s = 1;
  long maxs = ktiles / 8;
  if (maxs < 1) maxs = 1;
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  return (int)s;
}
// split-K factor for the bf16-output forms (forward, dgrad) of SMALL problems: a 512^2 image at batch 1 has 256 rows at the
// 1280-channel level -- 16 tiles of 128x160 for 256 CUs, each walking up to 160 K-steps at ~1 us: 58 us per dgrad launch,
// 22 ms of a 59 ms step.  Enough splits to put ~one workgroup on every CU, at least 4 K-steps each; 1 when the tiles alone
// fill half the chip (every problem of the B = 4, 1024^2 step).
int gemm_pick_splitk_small(int M, int N, int K, int kind) {     // K = the whole reduction length (taps x channels for a convolution)
  // (rows < 64: the batch-sized time-embedding projections -- only the long ones: the dgrad of the 17 resnets' concatenated
  //  time_emb_proj reduces over 13 760 columns with 4 rows: 8 workgroups x 215 K-steps = 0.3 ms at the very end of the backward)
  if (K % 64 || (M < 64 && K < 1024)) return 1;
  const long tiles = (long)cdiv(M, 128) * cdiv(N, N % 160 == 0 ? 160 : 128);
  // 3x3 convolutions (kind 0 forward, 1 dgrad) whose tiles fill one round of one workgroup per CU (the 1280-channel level at B = 4,
  // 1024^2: 256 tiles x 180-360 K-steps): two halves of the reduction as 512 co-resident 4-wave workgroups + the fixed-order
  // sum -- 4 x 32 x 32, 1280 -> 1280: forward 170 (8-wave split-K groups) / 139 (8-wave 4 x 2) -> 117 us = 1 030 TFLOP/s, dgrad
  // 171 -> 120; step -0.8 ms.  Not for the linear layers (kinds 2, 3; knob 2 bits, experiment): the forward FF2 projection loses
  // 0.9 ms of the step to the slab round trip, the long dgrads are neutral beside the weight-gradient stream.
  const bool conv_kind = (kind == 0 || kind == 1) && KNOB(2) != 1;
  const bool lin_kind = (kind == 2 || kind == 3) && KNOB(2) > 1 && ((KNOB(2) >> kind) & 1);
  if (conv_kind && (KNOB(30) & (kind == 0 ? 4 : 8)) && tiles > 128 && tiles <= 256) return 1;      // experiment: unsplit on the pipelined loop (launch_one)
  if ((conv_kind || lin_kind) && tiles > 128 && tiles <= 256 && K >= 5120) return 2;
  if (tiles >= 128) return 1;
  long s = 256 / tiles;
  const long maxs = K / 64 / 4;
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}
// weight gradients whose tile count fills a third or a quarter of the chip are launched 3 or 4 at a time (GemmP::group)
// instead of alone with split-K slabs + a reduce pass each: 1280 x 1280 x 4096 (192 per step at B=4 1024^2) took
// 43 us + 6 us of reduce each, three of them in one grid take the time of one 3840 x 1280 x 4096 product (~90 us)
int gemm_pick_group(int M, int N, int taps, long red, int splitk) {
  if (taps != 1 || splitk < 2 || red > 8192 || red % 64) return 1;
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 160);
  if (tiles < 64 || tiles > 128) return 1;
  const int g = KNOB(37) > 0 ? KNOB(37) : (int)(256 / tiles);      // (knob 37: problems per grouped launch, A/B runs)
  return g < 2 ? 1 : (g > GEMM_MAX_GROUP ? GEMM_MAX_GROUP : g);
}
void gemm_set_mode(int mode) { g_mode256 = mode & 3; g_force_cfg = mode >> 2; }
static int launch_gemm_impl(const GemmP& pin, hipStream_t st);
#ifdef SDXL_DIAG
void gemm_set_sk_mode(int mode) { g_sk_mode = mode; }
int gemm_sk_mode() { return g_sk_mode; }
// Policy of the stream-K kernel for a single problem (measured, profiles/r03*): where 256 x 256 tiles do not fill whole rounds
// of 256 CUs -- few tiles with a long reduction (N = 1280 outputs at M = 4096: 80 tiles), or 2.5 rounds (the GEGLU projection).
bool gemm_use_sk(const GemmP& p) {
  if (!gemm_sk_applicable(p)) return false;
  const long tiles = (long)(p.M / 256) * (p.N / 256), kt = p.K / 64;
  if (tiles * kt < 1024) return false;                 // less than 4 K-steps per CU
  return true;
}
// several problems in ONE stream-K launch (a layer's dgrad + wgrad); every problem must satisfy gemm_sk_applicable
int launch_gemm_multi(const GemmP* ps, int n, hipStream_t st) {
  if (!g_prof.on) return launch_gemm_sk(ps, n, st);
  while (g_prof.ev.size() < g_prof.used + 2) {
    hipEvent_t e;
    HIP_CHECK_RET(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  HIP_CHECK_RET(hipEventRecord(g_prof.ev[g_prof.used], st));
  int rc = launch_gemm_sk(ps, n, st);
  HIP_CHECK_RET(hipEventRecord(g_prof.ev[g_prof.used + 1], st));
  g_prof.used += 2;
  double f = 0;
  for (int i = 0; i < n; ++i) f += 2.0 * (double)ps[i].M * (double)ps[i].N * (double)ps[i].K;
  g_prof.flops.push_back(f);
  g_prof.recs.push_back({n > 1 ? 3 : ps[0].form, n, ps[0].M, ps[0].N, ps[0].K, 1});
  return rc;
}
#endif
// SDXL_LAUNCH_LOG=<path>: one line per GEMM / attention launch in host launch order (= rocprofv3's Dispatch_Id order), so that a
// kernel trace can be joined with the problems' shapes (profiles/tools/phase_rate.py)
FILE* launch_log() {
  static FILE* f = []() -> FILE* { const char* p = getenv("SDXL_LAUNCH_LOG"); return p ? fopen(p, "w") : nullptr; }();
  return f;
}
int launch_gemm(const GemmP& p, hipStream_t st) {
  if (FILE* f = launch_log()) { fprintf(f, "G,%d,%d,%d,%d,%d,%d,%d\n", p.form, p.taps, p.M, p.N, p.K, p.splitk, p.group > 1 ? p.group : 1); fflush(f); }
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
  g_prof.flops.push_back(2.0 * (double)p.M * (double)p.N * (double)p.K * (p.up2 == 2 ? 2.25 : (double)p.taps) * (p.group > 1 ? p.group : 1));      // (up2 == 2: 1 + 2 + 2 + 4 taps over four planes)
  g_prof.recs.push_back({p.form, p.taps, p.M, p.N, p.K, p.splitk});
  return rc;
}

static int launch_gemm_impl(const GemmP& pin, hipStream_t st) {
  GemmP p = pin;
  ARG_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  ARG_CHECK(p.N % 8 == 0, "gemm: N=%d must be a multiple of 8", p.N);
  ARG_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, "gemm: lda=%ld ldb=%ld must be multiples of 8", p.lda, p.ldb);
  if (p.up2) {
    ARG_CHECK(p.taps == ((SDXL_UP2_3 && p.up2 == 3) ? 9 : p.form == GEMM_NT || p.up2 == 2 ? 4 : 16) && p.sm == 1 && p.sd == 1 && p.Hm == p.Hs && p.Wm == p.Ws &&
              p.K % 64 == 0 && !p.geglu && p.group <= 1 && p.up_plane % 128 == 0 && p.up_rows <= p.up_plane && p.up_rows % (p.Hm * p.Wm) == 0 &&
              ((SDXL_UP2_3 && p.up2 == 3) ? (p.form == GEMM_NT ? p.M == p.up_rows : p.form == GEMM_TN && p.K == p.up_rows) :
               p.up2 == 2 ? (p.form == GEMM_NN && p.M == 4 * p.up_plane && p.splitk <= 1)
                          : p.form == GEMM_NT ? p.M == 4 * p.up_plane : p.form == GEMM_NN ? p.M == p.up_rows : p.K == p.up_rows),
              "gemm: up2 needs the fast same-size gather (reduction %% 64 == 0), taps 4 (NT, M = 4 planes of a multiple of 128 rows) / 16 (NN, TN)");
  } else
  ARG_CHECK(p.taps == 1 || p.taps == 9, "gemm: taps=%d", p.taps);
  ARG_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 && ((uintptr_t)p.C & 15) == 0,
            "gemm: operands must be 16-byte aligned");
  if (p.form == GEMM_TN) {
    ARG_CHECK(p.M % 8 == 0, "gemm TN: M=%d must be a multiple of 8", p.M);
    ARG_CHECK(p.out_f32, "gemm TN: only fp32 output is implemented");
    ARG_CHECK(p.ldc % 4 == 0, "gemm TN: ldc=%ld must be a multiple of 4", p.ldc);
  } else {
    ARG_CHECK(p.K % 8 == 0, "gemm: K=%d must be a multiple of 8", p.K);
    ARG_CHECK(!p.out_f32, "gemm NT/NN: bf16 output");
    // split-K of the bf16-output forms: plain linear problems whose K is a whole number of 64-element steps
    const bool conv_fast = (p.taps == 9 || p.up2) && p.sm == 1 && p.sd == 1 && p.Hm == p.Hs && p.Wm == p.Ws;
    if (p.splitk > 1 && (!(p.taps == 1 || conv_fast) || p.geglu || p.K % 64 != 0 || p.N % 8 != 0)) p.splitk = 1;
    ARG_CHECK(p.ldc % 8 == 0, "gemm: ldc=%ld must be a multiple of 8", p.ldc);
    if (p.accumulate) { p.resid = (const bf16*)p.C; p.ldr = p.ldc; }
    if (p.geglu) {
      if (p.geglu_group == 0) p.geglu_group = 64;
      ARG_CHECK(p.geglu == 2 ? p.geglu_group % 8 == 0 : (p.geglu_group == 64 || p.geglu_group == 80), "gemm: geglu group %d (forward: 64 or 80)", p.geglu_group);
      const int G = p.geglu_group;
      ARG_CHECK((p.geglu == 1 && p.form == GEMM_NT && p.N % (2 * G) == 0) || (p.geglu == 2 && p.form == GEMM_NN && p.N % G == 0),
                "gemm: geglu mode %d does not fit form %d / N=%d / group %d", p.geglu, p.form, p.N, G);
      ARG_CHECK(p.aux && p.ldaux % 8 == 0 && ((uintptr_t)p.aux & 15) == 0 && p.taps == 1, "gemm: geglu needs an aligned aux matrix");
      ARG_CHECK(!p.resid && !p.rowvec && (p.geglu == 1 || !p.bias), "gemm: geglu epilogue takes no residual / row vector");
    }
  }
  if (p.form == GEMM_TN) {
    ARG_CHECK(!p.geglu, "gemm TN: no geglu epilogue");
    ARG_CHECK(!p.Cb || (((uintptr_t)p.Cb & 7) == 0), "gemm TN: bf16 destination must be 8-byte aligned");
  } else {
    ARG_CHECK(!p.Cb, "gemm: the bf16 gradient destination is a TN (wgrad) option");
  }
  if (p.splitk < 1) p.splitk = 1;
  if (p.delta_out)      // a launch that is to write Delta must reach the 128 x 128 4-wave NN kernel's register epilogue: no slab path, no GEGLU
    ARG_CHECK(p.form == GEMM_NN && p.splitk == 1 && p.taps == 1 && !p.geglu && p.delta_o && p.delta_nq > 0 && p.delta_heads > 0 && p.delta_ldo % 8 == 0,
              "gemm: the Delta epilogue takes a plain, unsplit NN problem (form %d, splitk %d)", p.form, p.splitk);
  if (p.group > 1) {
    ARG_CHECK(p.form == GEMM_TN && p.taps == 1 && p.splitk == 1 && p.group <= GEMM_MAX_GROUP,
              "gemm: grouped launches are TN, one tap, no split-K, at most %d problems", GEMM_MAX_GROUP);
    for (int i = 0; i < p.group; ++i)
      ARG_CHECK(p.gA[i] && p.gB[i] && p.gC[i] && (((uintptr_t)p.gA[i] | (uintptr_t)p.gB[i] | (uintptr_t)p.gC[i]) & 15) == 0,
                "gemm: grouped operands must be set and 16-byte aligned");
    p.A = p.gA[0]; p.B = p.gB[0]; p.C = p.gC[0]; p.bias_grad = p.gbias_grad[0]; p.Cb = p.gCb[0];
  } else {
    p.group = 0;
  }
  {
    p.xcd_px = 0;
    {   // px x (8/px) XCD grid over (n, m) tiles minimising per-XCD operand footprint ~ N/px + M/py
      const int gx = cdiv(p.N, 128), gy = cdiv(p.M, BM);
      double best = 1e30;
      for (int px = 1; px <= 8; px *= 2) {
        const int py = 8 / px;
        if (gx % px || gy % py) continue;
        double cost = (double)p.N / px + (double)p.M / py;
        if (cost < best) { best = cost; p.xcd_px = px; }
      }
    }
  }
  if (p.splitk > 1) {
    ARG_CHECK(p.slab != nullptr, "gemm: split-K needs a slab scratch buffer");
    if (p.form != GEMM_TN && p.splitk > p.K / 64 * p.taps) p.splitk = p.K / 64 * p.taps;
    p.slab_ld = p.form == GEMM_TN ? (long)p.N * p.taps : (long)p.N;
  }
  const bool conv = p.taps != 1;
  int rc;
  if (p.ln_x) {      // LayerNorm backward in the dgrad's epilogue (GemmP::ln_x)
    ARG_CHECK(SDXL_LN_EPILOGUE, "gemm: the LayerNorm-backward epilogue exists in the diagnostics build only");
    ARG_CHECK(p.form == GEMM_NN && !conv && p.splitk == 1 && !p.geglu && !p.resid && !p.bias && !p.rowvec && !p.delta_out && !p.up2,
              "gemm: the LayerNorm-backward epilogue takes a plain NN problem");
    const int lc = gemm_ln_cfg(p.M, p.N, p.K);
    ARG_CHECK(lc != 0 && (p.cfg == lc || p.cfg == 0), "gemm: LayerNorm-backward epilogue: M=%d N=%d K=%d cfg=%d does not fit (gemm_ln_cfg)", p.M, p.N, p.K, p.cfg);
    p.cfg = lc;
    ARG_CHECK(p.ln_stats && p.ln_gamma && p.ln_dx && p.ln_part && p.ln_epoch > 0 && p.ln_ldx % 8 == 0 && p.ln_ldo % 8 == 0 && ((uintptr_t)p.ln_part & 15) == 0 &&
              (((uintptr_t)p.ln_x | (uintptr_t)p.ln_dx | (uintptr_t)p.ln_addend | (uintptr_t)p.ln_gamma) & 15) == 0,
              "gemm: LayerNorm-backward epilogue: missing or misaligned buffers");
    return launch_one<GEMM_NN, false>(p, st);
  }
  {   // software-pipelined one-wave-per-SIMD kernel (gemm_pl.hip): forced configuration 7, or the policy of knob 30 (linear problems whose tiles fit one round)
    const int fc = p.cfg > 0 ? p.cfg : g_force_cfg;
    bool use = fc == 7 || fc == 8;                       // (8: WITH the L2 prefetch wave -- measured, not used: gemm_pl.hip)
    const bool pl_pf = fc == 8 || KNOB(31) == 1;            // diagnostics build only (the product has no such instantiation)
    // Policy (knob 30 = 0): the linear problems whose 128 x 160 tiles are ONE round of the chip (129 .. 256 workgroups: the 4096-token level's
    // 1280-column outputs) -- forward projections and dgrads.  In the step (profiles/r05e_*): NN 4096 x 1280 x 10240 162 -> 106 us, x 3840
    // 67 -> 46, NT x 5120 66 -> 59; step -0.6 ms (the weight-gradient stream loses its co-resident partner while such a dgrad runs).
    if (!use && fc == 0 && KNOB(30) != 64 && p.taps == 1 && p.form != GEMM_TN) {
      const int bnp = p.N % 160 == 0 ? 160 : 128;
      const long tiles = (long)cdiv(p.M, BM) * cdiv(p.N, bnp);
      const int k30 = KNOB(30) ? KNOB(30) : 3;
      const bool one_round = tiles <= 256 && tiles > 128;
      use = ((k30 & 1) && p.form == GEMM_NT && one_round) || ((k30 & 2) && p.form == GEMM_NN && one_round) || ((k30 & 512) && p.form == GEMM_NN && one_round && p.K >= 2560) ||
            ((k30 & 16) && tiles > 128 && tiles % 256 == 0) || ((k30 & 1024) && p.form == GEMM_NT && tiles > 256 && tiles % 256 == 0) || (k30 & 32);
    }
    if (use && pl_applicable(p)) return launch_pl(p, 0, st, pl_pf);
  }
  {   // co-resident 256-row kernel (gemm_cr256.hip): forced configurations 31 (160-column tiles) / 32 (128)
    const int fc = p.cfg > 0 ? p.cfg : g_force_cfg;
    if (fc >= 31 && fc <= 36 && !p.delta_out && cr256_applicable(p)) {      // (33 / 34: the exclusive 6-deep form, diagnostics build; 35 / 36: phased loop, 128 / 160 columns)
      rc = launch_cr256(p, (fc == 31 || fc == 33 || fc == 36) ? 160 : 128, st, fc == 33 || fc == 34, fc == 35 || fc == 36);
      if (rc == 0 && p.form == GEMM_TN && p.splitk > 1) {
        const long nv = (long)p.M * (p.N / 4);
        int g = (int)((nv + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p.slab, (float*)p.C, p.M, p.N, p.ldc, p.slab_ld,
                           p.splitk, p.accumulate, p.Cb, p.cb_scale, p.bias_grad);
        HIP_CHECK_RET(hipGetLastError());
      }
      return rc;
    }
  }
#ifdef SDXL_DIAG
  if (g_sk_mode && !p.delta_out && !(p.form != GEMM_TN && p.splitk > 1) && (g_sk_mode == 2 ? gemm_sk_applicable(p) : gemm_use_sk(p))) {
    GemmP q = p;
    q.splitk = 1;
    return launch_gemm_sk(&q, 1, st);      // persistent stream-K kernel (gemm_sk.hip)
  }
#endif
  {   // 256 x 256 kernel (gemm256.hip)
    if (g_mode256 && p.group <= 1 && !p.Cb && !p.delta_out && !(p.form != GEMM_TN && p.splitk > 1) && gemm256_applicable(p)) {      // (the Delta epilogue exists in the 128 x 128 4-wave kernel only)
      // the forward GEGLU projection packed in groups of 64: the 256 x 256 kernel's register epilogue (value and gate of a channel
      // in one lane) against the 128-row kernel's LDS-staged one -- 131 vs 156 us at 4096 x 10240 x 1280 although its 640 tiles
      // fill only 2.5 rounds (profiles/r03_notes)
      const bool geglu256 = p.geglu == 1 && p.geglu_group == 64 && (long)(p.M / 256) * (p.N / 256) >= 256;
      if (g_mode256 == 2 || gemm_use256(p.form, p.M, p.N, p.K, p.splitk) || geglu256) {
        rc = launch_gemm256(p, st);
        if (rc == 0 && p.splitk > 1) {
          const long nv = (long)p.M * (p.N / 4);
          int g = (int)((nv + 255) / 256);
          if (g > 2048) g = 2048;
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p.slab, (float*)p.C, p.M, p.N, p.ldc, p.slab_ld,
                             p.splitk, p.accumulate, p.Cb, p.cb_scale, p.bias_grad);
          HIP_CHECK_RET(hipGetLastError());
        }
        return rc;
      }
    }
  }
  if (p.form != GEMM_TN && p.splitk > 1) {     // partial tiles to the slabs, then the fixed-order sum + bf16 epilogue
    const bf16* resid = p.resid; const long ldr = p.ldr;
    if (conv) rc = p.form == GEMM_NT ? launch_one<GEMM_NT, true>(p, st) : launch_one<GEMM_NN, true>(p, st);
    else rc = p.form == GEMM_NT ? launch_one<GEMM_NT, false>(p, st) : launch_one<GEMM_NN, false>(p, st);
    if (rc) return rc;
    const long nv = (long)p.M * (p.N / 8);
    int g = (int)((nv + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(g), dim3(256), 0, st, p.slab, (bf16*)p.C, p.M, p.N, p.ldc, p.slab_ld, p.splitk,
                       p.bias, resid, ldr, p.rowvec, p.ldv, p.rows_per_batch > 0 ? p.rows_per_batch : 1);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  switch (p.form) {
    case GEMM_NT: return conv ? launch_one<GEMM_NT, true>(p, st) : launch_one<GEMM_NT, false>(p, st);
    case GEMM_NN: return conv ? launch_one<GEMM_NN, true>(p, st) : launch_one<GEMM_NN, false>(p, st);
    case GEMM_TN:
      if (conv && conv_wgrad3_applicable(p) && conv_wgrad3_policy(p.M, p.N, p.K, p.Wm, p.sm)) rc = launch_conv_wgrad3(p, st);
      else if (!conv && wgrad256_applicable(p) && wgrad256_policy(p.M, p.N, p.K)) rc = launch_wgrad256(p, st);
      else
      rc = conv ? launch_one<GEMM_TN, true>(p, st) : launch_one<GEMM_TN, false>(p, st);
      if (rc == 0 && p.splitk > 1) {
        const int cols = p.N * p.taps;
        long nv = (long)p.M * (cols / 4);
        int g = (int)((nv + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, st, p.slab, (float*)p.C, p.M, cols, p.ldc,
                           p.slab_ld, p.splitk, p.accumulate, p.Cb, p.cb_scale, p.up2 == 1 ? nullptr : p.bias_grad);      // (up2 == 1: gemm_kernel's own adds)
        HIP_CHECK_RET(hipGetLastError());
      }
      return rc;
  }
  ARG_CHECK(false, "gemm: unknown form %d", p.form);
}

// ---- 3x3 convolution of a nearest-2x upsampled image without the upsampled image (GemmP::up2) ----
// x [B][H][W][Cin] low resolution, w [Cout][9][Cin]; weff [Cout][16][Cin] and planar [4][upconv_plane_rows(B, H, W)][Cout] are caller-provided scratch
// (weff is what the dgrad multiplies by: keep it until then); y [B][2H][2W][Cout].  16 instead of 36 tap-pixels per output pixel quad.
int launch_upconv3x3_fwd(const bf16* x, const bf16* w, const bf16* bias, bf16* weff, bf16* planar, bf16* y, int B, int H, int W,
                         int Cin, int Cout, int splitk, float* slab, hipStream_t st) {
  if (int e = launch_upconv_fold_weights(w, weff, Cout, Cin, st)) return e;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NT;
  g.up2 = 1; g.taps = 4;
  g.A = x; g.B = weff; g.C = planar;
  g.up_rows = B * H * W; g.up_plane = (int)upconv_plane_rows(B, H, W);
  g.M = 4 * g.up_plane; g.N = Cout; g.K = Cin;
  g.lda = Cin; g.ldb = 16L * Cin; g.ldc = Cout;
  g.Hm = H; g.Wm = W; g.Hs = H; g.Ws = W;
  g.b_tap_stride = Cin;
  g.bias = bias;
  if (splitk > 1) { g.splitk = splitk; g.slab = slab; }
  if (int e = launch_gemm(g, st)) return e;
  return launch_pixel_shuffle2(planar, y, B, H, W, Cout, 1, st);
}
// dx [B][H][W][Cin] (= resid + ...) from dy [B][2H][2W][Cout]: the high-resolution gradient is de-interleaved into its four phases,
// one NN product over the 16 (phase, stencil) entries gathers them at the mirrored offsets
int launch_upconv3x3_dgrad(const bf16* dy, const bf16* weff, bf16* planar, bf16* dx, const bf16* addend, int B, int H, int W, int Cin,
                           int Cout, int splitk, float* slab, int prio, hipStream_t st) {
  if (dy) { if (int e = launch_pixel_shuffle2(dy, planar, B, H, W, Cout, 0, st)) return e; }      // (dy == nullptr: `planar` is ready)
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NN;
  g.up2 = 1; g.taps = 16;
  g.A = planar; g.B = weff; g.C = dx;
  g.up_rows = B * H * W; g.up_plane = (int)upconv_plane_rows(B, H, W);
  g.M = B * H * W; g.N = Cin; g.K = Cout;
  g.lda = Cout; g.ldb = 16L * Cin; g.ldc = Cin;
  g.Hm = H; g.Wm = W; g.Hs = H; g.Ws = W;
  g.b_tap_stride = Cin;
  if (addend) { g.resid = addend; g.ldr = Cin; }
  if (splitk > 1) { g.splitk = splitk; g.slab = slab; }
  g.prio = prio;
  return launch_gemm(g, st);
}


// dW [Cout][9][Cin] (fp32, = or +=; optionally also bf16 x scale into `emit`) and dbias from dy [B][2H][2W][Cout] ALREADY de-interleaved
// into `planar` (launch_upconv3x3_dgrad does that) and the low-resolution x: 16 (phase, stencil) products into dweff [Cout][16][Cin]
// fp32 scratch, folded back onto the nine taps
int launch_upconv3x3_wgrad(const bf16* planar, const bf16* x, float* dweff, float* dw, float* dbias, bf16* emit, float emit_scale,
                           int accumulate, int B, int H, int W, int Cin, int Cout, int splitk, float* slab, hipStream_t st) {
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_TN;
  g.up2 = 1; g.taps = 16;
  g.up_rows = B * H * W; g.up_plane = (int)upconv_plane_rows(B, H, W);
  g.A = planar; g.B = x; g.C = dweff;
  g.M = Cout; g.N = Cin; g.K = B * H * W;
  g.lda = Cout; g.ldb = Cin; g.ldc = 16L * Cin;
  g.Hm = H; g.Wm = W; g.Hs = H; g.Ws = W;
  g.c_tap_stride = Cin;
  g.out_f32 = 1;
  g.splitk = splitk < 1 ? 1 : splitk;
  g.slab = slab;
  g.bias_grad = dbias;
  if (int e = launch_gemm(g, st)) return e;
  return launch_upconv_unfold_grads(dweff, dw, emit, emit_scale, accumulate, Cout, Cin, st);
}

// dx [B][H][W][Cin] (= addend + ...) of a stride-2 3x3 convolution (pad 1, H and W even) from dy [B][H/2][W/2][Cout]: input pixel
// (2r + a, 2c + b) only receives the taps whose parity matches -- 1, 2, 2, 4 of the nine for the four phases (9 / 4 per pixel instead
// of 9 masked ones through the general gather): one NN launch over the four phase planes into `planar` [4][plane][Cin], then the
// pixel shuffle (+ addend) into dx.  w [Cout][9][Cin] as the forward uses it.
int launch_conv3x3_s2_dgrad(const bf16* dy, const bf16* w, bf16* planar, bf16* dx, const bf16* addend, int B, int H, int W, int Cin,
                            int Cout, int prio, hipStream_t st) {
  ARG_CHECK(H % 2 == 0 && W % 2 == 0, "stride-2 dgrad by phases: H=%d W=%d must be even", H, W);
  const int Hl = H / 2, Wl = W / 2;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NN;
  g.up2 = 2; g.taps = 4;
  g.up_rows = B * Hl * Wl; g.up_plane = (int)upconv_plane_rows(B, Hl, Wl);
  g.A = dy; g.B = w; g.C = planar;
  g.M = 4 * g.up_plane; g.N = Cin; g.K = Cout;
  g.lda = Cout; g.ldb = 9L * Cin; g.ldc = Cin;
  g.Hm = Hl; g.Wm = Wl; g.Hs = Hl; g.Ws = Wl;
  g.b_tap_stride = Cin;
  g.prio = prio;
  if (int e = launch_gemm(g, st)) return e;
  return launch_pixel_shuffle2(planar, dx, B, Hl, Wl, Cin, 1, st, addend);
}

#ifdef SDXL_DIAG
// Stride-2 3x3 convolution (pad 1, H and W even) through the fast same-size gather: x [B][H][W][Cin] is de-interleaved into its four
// phase planes once (`xplanar` [4][plane][Cin], kept for the weight gradient), tap (ky, kx) then reads ONE plane at row offset -1 / 0
int launch_conv3x3_s2_fwd(const bf16* x, const bf16* w, const bf16* bias, bf16* xplanar, bf16* y, int B, int H, int W, int Cin, int Cout,
                          hipStream_t st) {
  ARG_CHECK(H % 2 == 0 && W % 2 == 0, "stride-2 conv by phases: H=%d W=%d must be even", H, W);
  const int Hl = H / 2, Wl = W / 2;
  if (int e = launch_pixel_shuffle2(x, xplanar, B, Hl, Wl, Cin, 0, st)) return e;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NT;
  g.up2 = 3; g.taps = 9;
  g.up_rows = B * Hl * Wl; g.up_plane = (int)upconv_plane_rows(B, Hl, Wl);
  g.A = xplanar; g.B = w; g.C = y;
  g.M = B * Hl * Wl; g.N = Cout; g.K = Cin;
  g.lda = Cin; g.ldb = 9L * Cin; g.ldc = Cout;
  g.Hm = Hl; g.Wm = Wl; g.Hs = Hl; g.Ws = Wl;
  g.b_tap_stride = Cin;
  g.bias = bias;
  return launch_gemm(g, st);
}
// ... and its weight / bias gradient from the same planes: dw [Cout][9][Cin] fp32
int launch_conv3x3_s2_wgrad(const bf16* dy, const bf16* xplanar, float* dw, float* dbias, bf16* emit, float emit_scale, int accumulate, int B,
                            int H, int W, int Cin, int Cout, int splitk, float* slab, hipStream_t st) {
  const int Hl = H / 2, Wl = W / 2;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_TN;
  g.up2 = 3; g.taps = 9;
  g.up_rows = B * Hl * Wl; g.up_plane = (int)upconv_plane_rows(B, Hl, Wl);
  g.A = dy; g.B = xplanar; g.C = dw;
  g.M = Cout; g.N = Cin; g.K = B * Hl * Wl;
  g.lda = Cout; g.ldb = Cin; g.ldc = 9L * Cin;
  g.Hm = Hl; g.Wm = Wl; g.Hs = Hl; g.Ws = Wl;
  g.c_tap_stride = Cin;
  g.out_f32 = 1;
  g.splitk = splitk < 1 ? 1 : splitk;
  g.slab = slab;
  g.accumulate = accumulate;
  g.bias_grad = dbias;
  if (emit) { g.Cb = emit; g.cb_scale = emit_scale; }
  return launch_gemm(g, st);
}
#endif
