// Launcher declarations for every hand-written gfx950 kernel family of the SDXL training step.
// All tensors are bf16, token-major ("NHWC"): an activation is a row-major [rows = B*H*W, C] matrix.
// All launchers are stream-ordered and return 0 on success (non-zero + sdxl_last_error() otherwise).
#pragma once
#include "common.h"

// experiment knobs (sdxl_set_knob / bench.py --knob id=value; 0 = the shipped policy everywhere):
//   0 wave priority of the main-stream dgrad GEMMs in the backward, 1 of the attention backward kernels
//   2 = 1: no split of the one-round 3x3 convolutions' reduction; = 16: every side-stream leaf behind its own fork event;
//       = 4 / 8 / 12: the same split for the long linear dgrads / forward projections / both
//       = 64: upsampler convolutions on the upsampled image; = 128: only their weight gradient; = 256: stride-2 dgrad through the general
//       gather; = 512: stride-2 forward / weight gradient on phase planes (OFF by default: neutral in the step)
//   3, 4 split-K factor of the long linear dgrads and its N threshold; 5 = 80: GEGLU packed in groups of 80
//   6, 7 forced configuration of the linear / conv dgrads, 8 N threshold of knob 6
//   9 = 1: no wgrad256 kernel; 10 = 1: LayerNorm backward as lean dx kernel + parameter-gradient leaf pass (0 / 2: dx and parameter partials in one pass, shipped), 3: lean + the atomic column-sum pass; 11 = 1: the round-2 LayerNorm dx kernel
//   12 = 1: no three-tap conv weight gradient; 13 split-K workgroup target of conv_wgrad3 / wgrad256 (144); 14 = 2: its W = 32 form
//   15 = 1: no half-height tail workgroups in the 256 x 256 kernel
//   16 co-resident 256-row kernel for the level-2 linear weight gradients: 0 policy, 1 off, 2 / 3 = 128 / 160-column tiles wherever allowed;
//   17 its split-K workgroup target for long reductions (0: 256); 19 = 1: also the 16 384-row level's linear weight gradients on it; 20 = 1: self-attention backward as two launches (dQ, then dK / dV); 23: loop of the co-resident 256-row kernel's 128-column tiles: 0 / 1 lockstep (shipped), 5 phased everywhere (rounds 4-5), 2 phased only for >= 256 tiles, 3 / 4 phased + 160-column tiles for >= 8192 rows / wherever they divide; 22 configuration of the GEGLU (FF2) dgrad; 21 = 1: Delta always from its own pass; 18 = 1: no padding columns on the feed-forward hidden tensors
//   24 = 1: GroupNorm backward partial sums by the LDS-free row-lane kernel (neutral in the step: r04i_ab_gn_nolds.txt)
//   37 problems per grouped weight-gradient launch (0: 256 / tiles = 3 for the 1280 x 1280 ones; 1 = alone with their split-K, 2, 4)
//   36 workgroup target of the cross-attention dK / dV kernel's query split (0: the shipped 160; 640 = rounds 2 - 4)
//   25 LayerNorm backward knock-outs (timing only, wrong gradients): 1 no dgamma / dbeta pass, 2 no dx pass either, 3 no dx pass
//   26 LayerNorm backward in the consumer's dgrad epilogue (GemmP::ln_x): 0 separate passes (shipped), 1 fused (dx + dgamma | dbeta partials), 2 fused dx, dy stored and the parameter pass kept
//   27 timing knock-outs by op type (bit mask, wrong results): 1 GroupNorm backward, 2 GroupNorm forward, 4 LayerNorm forward, 8 cross-attention
//      backward, 16 self-attention backward, 32 self-attention forward, 64 cross-attention forward; 128 / 256 GroupNorm / LayerNorm forward replaced by ONE elementwise launch
//      (live data downstream: the skipped ops leave stale zeros behind, and a chip at its power cap runs zeros ~5 % faster -- profiles/r05t_knockouts_live_data.txt)
//   28 wave priority (s_setprio 0..3) of the LayerNorm backward dx kernel
//   30 pipelined one-wave-per-SIMD kernels, bit mask (0 = the shipped policy = 3; 64 = none): 1 = the forward's one-round linear problems on gemm_pl.hip, 2 = the one-round linear dgrads,
//      16 = every plain linear NT / NN problem whose tiles are whole rounds of 256, 1024 = the FORWARD ones of those with more than one round, 32 = every plain linear NT / NN problem;
//      4 = the one-round 3x3 convolutions forward on gemm.hip's configuration 5 / 6 (unsplit instead of two co-resident halves), 8 = their dgrads
//   31 = 1: gemm_pl.hip WITH its L2 prefetch wave (measured: no gain)
//   32 = 1: attention workgroups in plain (block, pair) order instead of the XCD-aware one
//   33 attention forward kernel (Nk >= 256): 0 = the tiled kernel of attention.hip (shipped), 2 = the software-pipelined kernel of attention_pl.hip
//      (4 waves, one per SIMD, 512 registers): parity-green, at parity in time (profiles/r06f_attn_ab.txt)
//   35 self-attention backward: 0 policy (attention_bwd_pl.hip for Nq, Nk >= 2048, attn_bwd_fused_kernel below), 1 = attn_bwd_fused_kernel everywhere, 2 = pipelined wherever it applies
//   34 = 1: no generic XCD order (xcd_seq_map) in the weight-gradient kernels: identity where no XCD rectangle fits, = 2: the generic order also where a rectangle fits (cr256)
//   38 LDS KiB requested per workgroup of the co-resident 256-row weight-gradient kernel (0: the shipped 84 = one per CU + room for the other stream; 1: its ring's own size = two per CU; N: N KiB)
//   39 = a + 1000 b: a KiB of LDS per workgroup of the 128-row kernel's NN (dgrad) launches, b KiB for its TN launches (one per CU; measured: +12 / +1.1 ms)
//   29 = 1: weight-gradient GEMMs that use no split-K slab and whose operands come from the caller's stream are launched any-order on the side stream
// The product library has NO knobs: KNOB(i) is the constant 0 (= the shipped policy) and every experiment branch below it folds away.
// The diagnostics build (`build.py --diag`: -DSDXL_DIAG -> libsdxlstep_diag.so, include/sdxlstep_diag.h) keeps the table, sdxl_set_knob,
// the stream-K kernel (gemm_sk.hip), the stride-2 forward / weight gradient on phase planes (GemmP::up2 == 3) and the W = 32 three-tap form:
// measured, parity-tested experiments the step does not run (DESIGN.md sections 10, 11).
#define SDXL_NKNOBS 40
#ifdef SDXL_DIAG
extern int g_knobs[SDXL_NKNOBS];
#define KNOB(i) (g_knobs[(i)])
constexpr bool SDXL_UP2_3 = true;
constexpr bool SDXL_LN_EPILOGUE = true;
#else
#define KNOB(i) 0
constexpr bool SDXL_UP2_3 = false;
constexpr bool SDXL_LN_EPILOGUE = false;      // GemmP::ln_x (LayerNorm backward in the dgrad epilogue): measured, not shipped
#endif

// ------------------------------------------------------------------------------------------------
// bf16 MFMA GEMM family (gemm.hip).  C[M,N] = sum_k A(m,k) * B(k,n), fp32 accumulate.
//   NT: A [M][K] K-contiguous (optionally rows gathered from an image = implicit-GEMM conv),
//       B [N][K] K-contiguous                     -> Linear forward, conv3x3 forward
//   NN: A as NT, B [K][N] N-contiguous            -> Linear dgrad, conv3x3 dgrad
//   TN: A [K][M] M-contiguous, B [K][N] N-contiguous (optionally rows gathered) -> wgrad
// ------------------------------------------------------------------------------------------------
enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
// kernel launch of a GemmP problem: in stream order, or (GemmP::anyorder) without the barrier against the stream's earlier launches
#define GEMM_LAUNCH(kernel, grid, block, smem, st, p)                                                          \
  do {                                                                                                         \
    if ((p).anyorder) hipExtLaunchKernelGGL(kernel, grid, block, smem, st, nullptr, nullptr, hipExtAnyOrderLaunch, p); \
    else hipLaunchKernelGGL(kernel, grid, block, smem, st, p);                                                 \
  } while (0)

#define GEMM_MAX_GROUP 4
struct GemmP {
  int form;
  const bf16* A;
  const bf16* B;
  void* C;
  int M, N, K;      // K = reduction length per tap (channels for conv forms, pixels for TN)
  long lda, ldb, ldc;
  // implicit-GEMM 3x3 gather (taps == 9).  The gathered operand's rows are pixels (b,y,x) of an
  // Hm x Wm image; tap (dy,dx) reads source pixel ((y*sm+dy-1)/sd, (x*sm+dx-1)/sd) of an Hs x Ws
  // image when divisible and in range, else zero.
  int taps;
  int Hm, Wm, Hs, Ws, sm, sd;
  int flip;            // NT/NN: weight tap index = flip ? 8-tap : tap
  // up2 (NT / NN, same-size stride-1 gather): a nearest-2x upsampling folded into the 3x3 convolution that follows it.  Output pixel
  // (2i + a, 2j + b) = phase 2a + b sees a 2 x 2 stencil (u, v) of the LOW-resolution input at (i + u - (1 - a), j + v - (1 - b)) whose
  // weights are sums of the nine taps (elementwise.hip: launch_upconv_fold_weights): weight entry e = 4 phase + 2u + v of a
  // [Cout][16][Cin] matrix (ldb = 16 Cin, b_tap_stride = Cin).  NT (forward): taps = 4, M = 4 x up_plane rows in PLANAR phase-major order
  // (row m: phase m / up_plane, low-resolution pixel m % up_plane; rows >= up_rows of a plane are padding), A = the low-resolution image.  NN (dgrad):
  // taps = 16, M = up_rows low-resolution pixels, A = the planar phase-major output gradient [4][up_plane][K], gathered at the mirrored offsets.
  int up2;             // 1: as above.  3 (NT, TN): a STRIDE-2 3x3 convolution on the four phase planes of its input (A resp. B = [4][up_plane][Cin]): tap
                       // (ky, kx) reads plane (ky != 1, kx != 1) at row / column offset -1 (ky, kx = 0) or 0; M resp. K = up_rows output pixels.
                       //  2 (NN only): the input gradient of a STRIDE-2 3x3 convolution by output phase -- M = 4 x up_plane planar
                       // rows, A = the low-resolution output gradient, plain [Cout][9][Cin] weights, taps = 4 (1 / 2 / 2 / 4 used by phase)
  int up_plane, up_rows;      // up2: rows per phase plane of the planar matrix (a multiple of 128, >= up_rows) and low-resolution pixels B Hm Wm
  long b_tap_stride;   // NT/NN: elements added to B per weight tap
  long c_tap_stride;   // TN: elements added to C per tap
  // epilogue, bf16 output:  C = acc + bias[n] + rowvec[m / rows_per_batch][n] + resid[m][n]
  const bf16* bias;
  const bf16* resid;
  long ldr;
  const bf16* rowvec;
  long ldv;
  int rows_per_batch;
  // GEGLU fused into the feed-forward GEMMs (bf16 output forms, tile width = 2 * group).  The 2*C4 columns of the first
  // projection are stored interleaved in groups of G = 64 or 80: column (c/G)*2G + c%G is the value half of channel c,
  // +G its gate half (the weight rows are packed in the same order), so one tile holds both halves of 64 channels.
  //   geglu == 1 (NT, N = 2*C4):  C = u (pre-activation, kept for the backward), aux[m][c] = value * gelu(gate)
  //   geglu == 2 (NN, N = C4):    acc = dG;  C[m][2*C4] = dU from aux = u:  d value = dG * gelu(gate),
  //                                                                        d gate  = dG * value * gelu'(gate)
  int geglu;
  int geglu_group;   // 64 (128-column tiles) or 80 (160-column tiles); 0 = 64
  bf16* aux;
  long ldaux;
  // NT / NN (bf16 output), taps == 1, no GEGLU, K % 64 == 0: splitk > 1 -> fp32 partial tiles to slab[split], then
  // splitk_epilogue_kernel sums them in a fixed order and applies bias / row vector / residual (small-M problems).
  // fp32 output (wgrad): C_f32 (+)= acc.  splitk > 1: each split writes its partial [M][N*taps] tile set to
  // slab[split] (plain stores) and a reduce kernel sums the slabs in a fixed order (deterministic, no atomics);
  // the conv form requires C to be the dense [M][taps*N] weight-gradient matrix (ldc == N*taps).
  int out_f32;
  int accumulate;
  int splitk;
  float* slab;     // gemm_slab_floats(M, N, taps, splitk): splitk * M * N * taps partial tiles + a [splitk][M] tail (bias-gradient partials)
  long slab_ld;    // set by the launcher
  // TN only, optional: the FINAL value (acc [+ C when accumulate]) * cb_scale goes out as bf16 to Cb (same [M][ldc] geometry as C)
  // and C itself is not written -- the gradient-exchange micro-step under data parallelism wants the bf16 comm arena, and
  // a separate cast pass over the 10 GB fp32 arena costs 3.4 ms per step (grouped launches: gCb[i])
  bf16* Cb;
  float cb_scale;
  float* bias_grad;  // TN only, optional: bias_grad[m] += sum_k A(k, m)  (column sums of dY, computed on the matrix pipe
                     // by the n-tile-0 / tap-0 workgroups with an all-ones B fragment).  Bitwise reproducible: an unsplit launch adds ONE
                     // value per element; a split launch stores the splits' sums as rows of the slab's tail and splitk_reduce_kernel adds
                     // them in a fixed order (gemm_bias_out below)
  // NN (bf16 output), 128-column tiles of the 4-wave kernel only (a wave then owns whole 64-column heads): the output is an attention
  // layer's dO; also write Delta[b * delta_heads + head][q] = sum_d dO[m][64 head + d] * O[m][64 head + d] (m = b * delta_nq + q, O = delta_o,
  // row stride delta_ldo) from the bf16-rounded output -- what attn_delta_kernel would compute in a pass of its own
  const bf16* delta_o;
  long delta_ldo;
  float* delta_out;
  int delta_nq, delta_heads;
  // (Diagnostics build only -- SDXL_LN_EPILOGUE; the product kernel has no such path.)  NN, plain bf16 epilogue, configuration 1 (128 x 128 tiles): this dgrad's output IS the output gradient dy of a LayerNorm (y = LN(x) feeds only
  // this linear layer).  With ln_x set the epilogue also runs that LayerNorm's backward: per-row partial sums (sum dy g, sum dy g xhat) of
  // the workgroup's columns -> ln_part[column tile][row] as 8-byte (value, tag) granules, write-through -> every workgroup of the row block
  // gathers the granules of all its column tiles, polling until they carry this launch's tag (ln_epoch: unique among the launches that
  // share ln_part since it was last zeroed) -> dx = rstd (dy g - S1 / N - xhat S2 / N) + addend, bf16, to ln_dx; and -- ln_pcol set -- the
  // block's dgamma | dbeta column partial sums to ln_pcol[row block][2][N] (folded later by ln_param_reduce_kernel).  C == nullptr: dy itself
  // is not stored.  N = the LayerNorm width <= 1280; launches of <= 512 workgroups only (all resident at once: the meeting cannot starve).
  const bf16* ln_x;        // the LayerNorm's input [M][N], row stride ln_ldx
  long ln_ldx;
  const float* ln_stats;   // [M][2] mean, rstd (forward)
  const bf16* ln_gamma;    // [N]
  bf16* ln_dx;             // [M][N], row stride ln_ldo
  const bf16* ln_addend;   // or nullptr
  long ln_ldo;
  float* ln_part;          // gemm_ln_part_floats(M, N) floats of scratch, 16-byte aligned, zeroed before epoch 1
  int ln_epoch;            // > 0
  float* ln_pcol;          // or nullptr: [gridDim.y][2][N]
  int xcd_px;      // set by the launcher: XCD grid width over n-tiles (0 = identity order)
  int xcd_bh;      // set by the launcher: band height of the generic XCD order (gemm_tiles.h: xcd_seq_map) where no px x py rectangle fits; 0 = off
  int tail_n0;     // 256 x 256 kernel, set by its launcher: > 0 = the tile columns from tail_n0 on are computed by HALF-HEIGHT workgroups
                   // (128 x 256: waves 4-7 only stage data) so that a launch of 2.5 rounds of tiles takes ~2.6 rounds, not 3
  int anyorder;    // host side, experiment (knob 29): launch without the in-stream barrier (hipExtAnyOrderLaunch): the kernel depends on nothing the stream ran before it
  int cfg;         // > 0: this launch's configuration of the 128-row kernel (1, 2, 3, 13, 23), overriding the selection policy
  int prio;        // wave priority (s_setprio 0..3) of the whole kernel: the backward's critical-path launches (dgrad chain, caller's
                   // stream) outrank the co-resident weight-gradient workgroups of the side stream on every SIMD they share
  // grouped launch (TN, taps == 1, splitk == 1): `group` > 1 problems of one shape in one grid (blockIdx.z = problem i, which
  // uses gA[i], gB[i], gC[i], gbias_grad[i] in place of A, B, C, bias_grad).  Small weight gradients (1280 x 1280: 80 tiles)
  // fill the chip three at a time instead of each being cut into split-K slabs and reduced.
  int group;
  const bf16* gA[GEMM_MAX_GROUP];
  const bf16* gB[GEMM_MAX_GROUP];
  float* gC[GEMM_MAX_GROUP];
  float* gbias_grad[GEMM_MAX_GROUP];
  bf16* gCb[GEMM_MAX_GROUP];
};

// Bias-gradient value of row m from reduction split `split` (the kernels' epilogues call this from the one workgroup per row block and split
// that owns it).  Unsplit: the only writer of the element in this launch.  Split: row `split` of the slab's tail, folded by splitk_reduce_kernel.
__device__ __forceinline__ void gemm_bias_out(float* bias_grad, float* slab, long slab_ld, int splitk, int split, int M, int m, float v) {
  if (splitk > 1) slab[(long)splitk * M * slab_ld + (long)split * M + m] = v;
  else atomicAdd(bias_grad + m, v);
}
int gemm_pick_group(int M, int N, int taps, long red, int splitk);   // problems per grouped wgrad launch (1 = launch alone)
size_t gemm_slab_floats(int M, int N, int taps, int splitk);
void gemm_defaults(GemmP* p);
int launch_gemm(const GemmP& p, hipStream_t st);
// 256 x 256 tile, 8-phase kernel (gemm256.hip): M, N multiples of 256, K of 64, no 3x3 gather
bool gemm256_applicable(const GemmP& p);
void gemm_set_mode(int mode);   // bits 0-1: 0 never / 1 policy / 2 wherever applicable; bits 2..: force a 128-row configuration
bool gemm_use256(int form, int M, int N, int K, int splitk);   // the policy of mode 1
int gemm_pick_splitk(int M, int N, int taps, long red);        // split-K factor the wgrad launchers should request
int gemm_pick_splitk_small(int M, int N, int K, int kind = -1);               // split-K factor for NT / NN (bf16 output) launches of small problems
int launch_gemm256(const GemmP& p, hipStream_t st);
void gemm256_set_tail(bool on);     // half-height workgroups for the last partial round (default on; A/B runs)
#ifdef SDXL_DIAG
// persistent stream-K kernel (gemm_sk.hip): up to 4 problems (M, N multiples of 256, K of 64, no gather) in ONE launch, their
// K-steps cut evenly over the CUs; partial tiles are handed to the tile's owner inside the launch (fixed order: reproducible)
bool gemm_sk_applicable(const GemmP& p);
int launch_gemm_sk(const GemmP* problems, int n, hipStream_t st);
int launch_gemm_multi(const GemmP* problems, int n, hipStream_t st);   // = launch_gemm_sk + the per-launch profiling of launch_gemm
void gemm_set_sk_mode(int mode);                      // 0 never / 1 policy (gemm_use_sk) / 2 wherever applicable
int gemm_sk_mode();
bool gemm_use_sk(const GemmP& p);
void gemm_sk_set_workers(int n);                      // > 0: force the worker count (microbenchmarks), 0: policy
int gemm_ln_error(unsigned* out);                         // != 0: a LayerNorm-backward epilogue (GemmP::ln_x) gave up waiting for its row block's other tiles; clears it
int gemm_sk_error(hipStream_t st, unsigned* out);     // != 0: an owner gave up waiting for a partial tile (results invalid)
#endif
// 3x3 weight gradient, three taps per workgroup (conv_wgrad3.hip): same-size stride-1 convolutions whose image width is a multiple of 64
bool conv_wgrad3_applicable(const GemmP& p);
bool conv_wgrad3_policy(int M, int N, long red, int Wm, int stride);   // the plan's choice (long reductions: the 128^2 / 64^2 levels)
int conv_wgrad3_pick_splitk(int M, int N, long red);
int launch_conv_wgrad3(const GemmP& p, hipStream_t st);
void conv_wgrad3_set_enabled(bool on);
// long-reduction linear weight gradient, 256 x 160 tiles, 8 waves (wgrad256.hip)
bool wgrad256_applicable(const GemmP& p);
bool wgrad256_policy(int M, int N, long red);
int wgrad256_pick_splitk(int M, int N, long red);
int launch_wgrad256(const GemmP& p, hipStream_t st);
void wgrad256_set_enabled(bool on);
// co-resident 256-row tile, 8 waves, <= 78 KiB LDS, <= 128 registers (gemm_cr256.hip): linear NT / NN / TN problems, K % 32 == 0
bool cr256_applicable(const GemmP& p);
int launch_cr256(const GemmP& p, int bn, hipStream_t st, bool deep = false, bool phased = false);      // bn = 160 / 128 / 0 (pick); deep: exclusive 6-deep ring (diagnostics build)
int cr256_wgrad_cfg(int M, int N, long red, bool bias);
// software-pipelined one-wave-per-SIMD kernel (gemm_pl.hip): 128 x 160 / 128 x 128 x 64 tiles, 4 waves, 4-deep ring; linear NT / NN, plain bf16 epilogue
bool pl_applicable(const GemmP& p);
int launch_pl(const GemmP& p, int bn, hipStream_t st, bool prefetch = false);
int launch_pl_prefetch_b(const GemmP& p, int parts, hipStream_t st);      // diagnostics build: L2 prefetch of B for a coming launch_pl (another stream)      // bn = 160 / 128 / 0 (160 where N % 160 == 0)
int gemm_ln_cfg(int M, int N, int K);                  // GemmP::ln_x: the configuration such a launch takes (GemmP::cfg), 0 = not possible for this shape
size_t gemm_ln_part_floats(int M, int N);
size_t gemm_ln_pcol_floats(int M, int N);
int gemm_ln_rowblocks(int M);
int cr256_pick_splitk(int M, int N, long red, int cfg);        // the plan's choice for a linear weight gradient [M][N] over `red` rows: 0 / 31 / 32 (GemmP::cfg)
// per-launch HIP-event timing of every GEMM launch between begin and end (end synchronises the device)
int gemm_profile_begin();
FILE* launch_log();      // SDXL_LAUNCH_LOG (gemm.hip)
bool gemm_profiling();   // true between begin and end: the engine then runs everything on one stream (clean durations)
int gemm_profile_end(double* flops, double* ms, int* launches);

// ------------------------------------------------------------------------------------------------
// flash attention, head_dim 64 (attention.hip).  Q rows [B*Nq], K/V rows [B*Nk]; head h lives in
// columns [h*64, h*64+64) of each; row strides ldq/ldk/ldv/ldo in elements.  softmax scale 1/8.
// ------------------------------------------------------------------------------------------------
struct AttnP {
  const bf16 *Q, *K, *V;
  bf16* O;
  float* LSE;          // [B*H][Nq]  log-sum-exp of the scaled scores (natural log)
  int B, H, Nq, Nk;
  long ldq, ldk, ldv, ldo;
  // backward
  const bf16* dO;
  long lddo;
  bf16 *dQ, *dK, *dV;
  long lddq, lddk, lddv;
  float* Delta;        // [B*H][Nq]  rowsum(dO * O)
  int accumulate;      // dQ/dK/dV += (else =)
  // dK/dV kernel, short key sequences (cross attention, Nk = 77): the query loop is split over grid.z so the
  // kernel fills the chip; each split writes fp32 partials to `part` and a reduce kernel sums them (fixed order)
  int qsplit;
  float* part;         // qsplit * B*H * kvtiles*64 * 64 * 2 floats
  int prio;            // wave priority of the backward kernels (see GemmP::prio)
  int delta_ready;     // fused backward: Delta is already in place (written by the epilogue of the GEMM that produced dO, GemmP::delta_out)
  int xcd;             // set by the launchers: XCD-aware (block, pair) order of the workgroups (attention.hip: attn_xcd_map)
};
size_t attn_part_floats(int B, int H, int Nk, int qsplit);
int attn_pick_qsplit(int B, int H, int Nq, int Nk);
int launch_attn_fwd(const AttnP& p, hipStream_t st);
int launch_attn_fwd_tiled(const AttnP& p, hipStream_t st); // attention.hip: 128-query workgroups, three per CU (what launch_attn_fwd runs)
bool attn_fwd_pl_applicable(const AttnP& p);               // diagnostics build (attention_pl.hip): software-pipelined forward, Nk >= 256
int launch_attn_fwd_pl(const AttnP& p, hipStream_t st);
bool attn_bwd_pl_applicable(const AttnP& p);               // attention_bwd_pl.hip: software-pipelined dQ | dK | dV in one grid (Nq, Nk >= 256; Delta in place)
int launch_attn_bwd_pl(const AttnP& p, hipStream_t st);
int launch_attn_bwd(const AttnP& p, hipStream_t st);       // = dq, then dkv
int launch_attn_bwd_dq(const AttnP& p, hipStream_t st);
int launch_attn_bwd_dkv(const AttnP& p, hipStream_t st);
int launch_attn_bwd_fused(const AttnP& p, hipStream_t st);   // self-attention: Delta pass + dK / dV and dQ workgroups in one grid

// ------------------------------------------------------------------------------------------------
// normalisation (norm.hip)
// ------------------------------------------------------------------------------------------------
// GroupNorm over [B][HW][C] with G groups (+ optional SiLU).  stats: [B][G][2] = mean, rstd.
// ws: groupnorm_ws_floats(B, C, G) floats of scratch (may be shared by stream-ordered calls)
size_t groupnorm_ws_floats(int B, int C, int G);
int launch_groupnorm_fwd(const bf16* x, bf16* y, const bf16* gamma, const bf16* beta, float* stats,
                         float* ws, int B, int HW, int C, int G, float eps, int silu,
                         hipStream_t st);
// dx (+)= ; dgamma/dbeta fp32 += .  ws: same scratch
// dx = (addend ? addend : 0) + grad   (addend may alias dx; a distinct addend keeps gradient buffers write-once)
int launch_groupnorm_bwd(const bf16* x, const bf16* dy, const bf16* gamma, const bf16* beta,
                         const float* stats, bf16* dx, const bf16* addend, float* dgamma, float* dbeta, float* ws,
                         int B, int HW, int C, int G, int silu, hipStream_t st, float* prow = nullptr);
// LayerNorm over rows of [M][C]; stats [M][2] = mean, rstd
int launch_layernorm_fwd(const bf16* x, bf16* y, const bf16* gamma, const bf16* beta, float* stats, int M,
                         int C, float eps, hipStream_t st);
// dx = (addend ? addend : 0) + grad; with `part` (layernorm_bwd_part_floats(M, C) floats) also the per-block partial sums of
// dgamma | dbeta, which launch_ln_param_reduce folds into the fp32 gradients (+=) for a whole batch of LayerNorms at once
size_t layernorm_bwd_part_floats(int M, int C);
int launch_layernorm_bwd(const bf16* x, const bf16* dy, const bf16* gamma, const float* stats, bf16* dx,
                         const bf16* addend, float* part, int* nblk_out, int M, int C, hipStream_t st);
int launch_layernorm_param_grads(const bf16* x, const bf16* dy, const float* stats, float* dgamma, float* dbeta, int M, int C, hipStream_t st);
// the same sums as layernorm_param_partial_rows(M, C) partial rows part[rows][2][C] (plain stores, inside layernorm_bwd_part_floats(M, C));
// an LnRedEntry with nblk = that row count folds them into the gradients (launch_ln_param_reduce)
int layernorm_param_partial_rows(int M, int C);
int launch_layernorm_param_partials(const bf16* x, const bf16* dy, const float* stats, float* part, int M, int C, hipStream_t st);
#define LN_RED_MAX 64
struct LnRedEntry { const float* part; float* dgamma; float* dbeta; int nblk, C; };
struct LnRedBatch { LnRedEntry e[LN_RED_MAX]; int n; };   // passed by value as the kernel argument (2 KiB)
int launch_ln_param_reduce(const LnRedBatch& b, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// elementwise / small (elementwise.hip)
// ------------------------------------------------------------------------------------------------
int launch_silu_fwd(const bf16* x, bf16* y, long n, hipStream_t st);
int launch_silu_bwd(const bf16* x, const bf16* dy, bf16* dx, const bf16* addend, long n, hipStream_t st);
int launch_add(const bf16* a, const bf16* b, bf16* o, long n, hipStream_t st);           // o = a + b
size_t colsum_part_floats(int batches, int rows, int N);
int launch_colsum_f32_batched(const bf16* x, float* out, int batches, int rows, int N, long ldx, long ld_out, float* part, hipStream_t st);  // out[b][n] += sum_m, fixed order
int launch_colsum_partials(const bf16* x, float* out, int batches, int rows, int N, long ldx, long ld_out, float* part, LnRedEntry* entries, hipStream_t st);
int launch_concat(const bf16* a, int Ca, const bf16* b, int Cb, bf16* o, long rows, hipStream_t st);
int launch_split_add(const bf16* g, bf16* ga, int Ca, const bf16* add_a, bf16* gb, int Cb, const bf16* add_b,
                     long rows, hipStream_t st);                                                  // concat backward
int launch_upsample2x(const bf16* x, bf16* y, int B, int H, int W, int C, hipStream_t st);
// GemmP::up2 companions: the [Cout][16][Cin] stencil weights of a [Cout][9][Cin] kernel; planar phase-major <-> high-resolution layout
int launch_upconv_fold_weights(const bf16* w, bf16* weff, int Cout, int Cin, hipStream_t st);
int launch_pixel_shuffle2(const bf16* src, bf16* dst, int B, int H, int W, int C, int to_hi, hipStream_t st, const bf16* addend = nullptr);   // plane stride = upconv_plane_rows
static inline long upconv_plane_rows(int B, int H, int W) { return ((long)B * H * W + 127) / 128 * 128; }
int launch_upconv3x3_fwd(const bf16* x, const bf16* w, const bf16* bias, bf16* weff, bf16* planar, bf16* y, int B, int H, int W, int Cin,
                         int Cout, int splitk, float* slab, hipStream_t st);          // gemm.hip
#ifdef SDXL_DIAG
int launch_conv3x3_s2_fwd(const bf16* x, const bf16* w, const bf16* bias, bf16* xplanar, bf16* y, int B, int H, int W, int Cin, int Cout,
                          hipStream_t st);      // gemm.hip (GemmP::up2 == 3)
int launch_conv3x3_s2_wgrad(const bf16* dy, const bf16* xplanar, float* dw, float* dbias, bf16* emit, float emit_scale, int accumulate, int B,
                            int H, int W, int Cin, int Cout, int splitk, float* slab, hipStream_t st);
#endif
int launch_conv3x3_s2_dgrad(const bf16* dy, const bf16* w, bf16* planar, bf16* dx, const bf16* addend, int B, int H, int W, int Cin,
                            int Cout, int prio, hipStream_t st);      // gemm.hip (GemmP::up2 == 2)
int launch_upconv_unfold_grads(const float* dweff, float* dw, bf16* emit, float emit_scale, int accumulate, int Cout, int Cin, hipStream_t st);
int launch_upconv3x3_wgrad(const bf16* planar, const bf16* x, float* dweff, float* dw, float* dbias, bf16* emit, float emit_scale,
                           int accumulate, int B, int H, int W, int Cin, int Cout, int splitk, float* slab, hipStream_t st);
int launch_upconv3x3_dgrad(const bf16* dy, const bf16* weff, bf16* planar, bf16* dx, const bf16* addend, int B, int H, int W, int Cin,
                           int Cout, int splitk, float* slab, int prio, hipStream_t st);
int launch_upsample2x_bwd(const bf16* dy, bf16* dx, const bf16* addend, int B, int H, int W, int C, hipStream_t st);
// sinusoidal embedding, cos first: out[r][0:half]=cos(t*f_i), out[r][half:]=sin ; out row stride ldo
int launch_sincos(const float* t, bf16* out, int rows, int dim, long ldo, hipStream_t st);
int launch_copy_cols(const bf16* src, long lds, bf16* dst, long ldd, int rows, int cols, hipStream_t st);
int launch_f32_to_bf16(const float* x, bf16* y, long n, float scale, hipStream_t st);
int launch_bf16_to_f32(const bf16* x, float* y, long n, hipStream_t st);
int launch_sumsq_f32(const float* x, long n, float* out /* += */, hipStream_t st);
int launch_sumsq_bf16(const bf16* x, long n, float* out /* += */, hipStream_t st);
int launch_clip_coef(const float* sumsq, float max_norm, float* coef, hipStream_t st);
int launch_scale_f32(float* x, long n, const float* scale_dev, hipStream_t st);
int launch_exchange_shadow(void* buf, size_t bytes, int workgroups, int lds_bytes, float busy_us, hipStream_t st);   // measurement hook

// ------------------------------------------------------------------------------------------------
// loss side (loss.hip) -- restates reference compute_loss arithmetic on device
// ------------------------------------------------------------------------------------------------
struct LossP {
  int method;             // 0 = ddpm, 1 = flow matching
  int prediction_type;    // ddpm: 0 = epsilon, 1 = v_prediction
  int use_min_snr;        // ddpm
  float min_snr_gamma;
  int use_ztsnr;          // clamp noisy latents to +-20000
  int B, HW, C;           // latents [B][C][HW] (NCHW, fp32), C = 4
  const float* latents;   // x (ddpm) / x1 (flow)
  const float* noise;     // noise (ddpm) / x0 (flow)
  const float* sigma;     // ddpm: [B] sigma ; flow: [B] t
  const float* tag_w;     // optional [B]
  bf16* unet_in;          // [B*HW][8] (channels 4..7 zero)
  const bf16* pred;       // [B*HW][8]
  bf16* dpred;            // [B*HW][8]
  float grad_scale;       // multiplies d(loss)/d(pred) (1/grad_accum, 1/world)
  float* out;             // device: [0]=loss [1]=raw loss [2]=sum|pred| [3]=sum pred^2 [4]=sum|noise|
                          //         [5]=sum x0^2 [6]=sum x1^2 [7]=gate
  float* part;            // device scratch, loss_part_floats(B, HW): the blocks' partial sums (no atomics: the loss is bitwise reproducible)
};
static inline size_t loss_part_floats(int B, int HW) { return 6 * (((size_t)B * HW + 255) / 256); }
int launch_loss_prepare(const LossP& p, hipStream_t st);
int launch_loss_fwd(const LossP& p, hipStream_t st);
int launch_loss_bwd(const LossP& p, hipStream_t st);

// ---- fused AdamW_BF16 step (optimizer.hip; reference adamw_bfloat16/__init__.py:146-197) ----
struct AdamWP {
  bf16* p;
  const float* grad_f32;       // native fp32 gradient arena, or
  const bf16* grad_bf16;       //   bf16 gradients (exactly one of the two)
  bf16 *m, *v, *shift;
  size_t n;                    // elements, multiple of 8
  float beta1, beta2, one_minus_beta1, one_minus_beta2;
  float eps_bf16;              // eps rounded to bf16 (torch casts the scalar it adds to a bf16 tensor)
  float value;                 // -lr * sqrt(1 - beta2^step)
  float decay_alpha_bf16;      // -decay_this_iteration rounded to bf16, 0 = no decay in this launch
  int reference_ema;           // 1: the reference's actual first-moment update  m <- SR(g + (1-b1) * b1*m)   (quirk D17)
                               // 0: the documented EMA                          m <- SR(b1*m + (1-b1) * g)
  int grad_round_bf16;         // round the (scaled) gradient to bf16 first, as the reference's bf16 autograd does
  const float* grad_scale;     // device scalar multiplied into the gradient (unscale / clip), or nullptr
  const unsigned short* rand;  // [4][n] injected random 16-bit integers (parity tests), or nullptr = Philox
  unsigned seed_lo, seed_hi, step_counter;
  size_t elem_offset;          // arena index of element 0 of this launch (multiple of 8): the Philox counters are those of the
                               // full-arena launch, so a sharded (ZeRO-1) update is bit-identical to the unsharded one
};
int launch_adamw_bf16(const AdamWP& q, hipStream_t st);
int launch_adamw_decay(bf16* shift, const bf16* p, size_t n, float alpha_bf16, hipStream_t st);

