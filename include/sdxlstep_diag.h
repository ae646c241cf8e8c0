/* libsdxlstep -- test hooks and experiment ABI.  NOT part of the drop-in boundary (include/sdxlstep.h is): nothing a caller of the
 * training step needs is declared here.
 *
 * Part 1, exported by the product library too: hooks the parity tests and the measurement tools use to force one kernel
 * configuration or to look inside a plan.
 * Part 2, exported ONLY by the diagnostics build (`python sdxl-training-improvements_amd/build.py --diag` -> libsdxlstep_diag.so,
 * compiled with -DSDXL_DIAG): measured, parity-tested experiments that the shipped plan never runs -- the knob table, the
 * persistent stream-K GEMM (csrc/gemm_sk.hip), the stride-2 convolution forward / weight gradient on phase planes
 * (GemmP::up2 == 3), the W = 32 form of the three-tap convolution weight gradient.  DESIGN.md sections 10-12 hold their numbers;
 * tests of them carry the `diag` marker and run with SDXL_DIAG=1. */
#ifndef SDXLSTEP_DIAG_H
#define SDXLSTEP_DIAG_H
#include "sdxlstep.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- part 1: test hooks (product library) ---- */
/* run ds_read_b64_tr_b16 / MFMA layout probes (tests/test_gpu_ops.py::test_hw_layout_probe) */
SDXL_API int sdxl_probe_layout(void* out_dev, void* stream);
/* tile-kernel selection of the GEMM family, for A/B measurements and parity tests: 0 = 128-row kernel only,
 * 1 = the plan's policy (default), 2 = 256 x 256 kernel wherever it is applicable;
 * + 4 * c forces configuration c of the 128-row kernel (1, 2, 3, 13, 23) or the co-resident 256-row kernel (31: 256 x 160 tiles,
 * 32: 256 x 128) wherever applicable.  Process-global: restore 1 after use. */
SDXL_API int sdxl_set_gemm_mode(int mode);
/* checksum of every activation (grads != 0: of every activation gradient) of the current plan, in creation
 * order; synchronises the device.  n_out receives the number of activations. */
SDXL_API int sdxl_debug_act_checksums(sdxl_handle* h, unsigned long long* out_host, int cap, int* n_out, int grads);

/* stand-in for a collective's device kernel on one GPU (bench.py --exchange-shadow): `workgroups` x 256 threads holding `lds_bytes` of
 * LDS each stream `buf` for `busy_us` microseconds; run beside the backward it prices the co-residency of the gradient exchange. */
SDXL_API int sdxl_op_exchange_shadow(void* buf, size_t bytes, int workgroups, int lds_bytes, float busy_us, void* stream);

/* sdxl_op_gemm's NT (form 0) / NN (form 1) with explicit leading dimensions (elements) and a configuration forced for this launch only
 * (cfg as in sdxl_set_gemm_mode's upper bits; 0 = the policy): C [M][N] = A [M][K] . B (+ bias [N]) (+ resid [M][ldr]) */
SDXL_API int sdxl_op_gemm_ld(int form, const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, const void* bias,
                    const void* resid, long ldr, int cfg, void* stream);
/* the out-projection dgrad whose epilogue also writes the self-attention backward's Delta (csrc/kernels.h, GemmP::delta_out; the plan
 * uses it at the 1280-channel level, csrc/engine.hip LinearOp::plan_bwd): dO [M][N] = dY [M][K] . W [K][N] (+ addend, or null), bf16, and
 * Delta[(b * heads + h) * Nq + q] = sum_d bf16(dO[m][64 h + d]) * O[m][64 h + d] with m = b * Nq + q, heads = N / 64.  M = B * Nq,
 * N % 128 == 0, K % 64 == 0. */
SDXL_API int sdxl_op_linear_dgrad_delta(const void* dy, const void* w, const void* o, const void* addend, void* d_o, float* delta, int B, int Nq,
                               int N, int K, void* stream);

/* ---- part 2: experiment ABI (diagnostics build only) ---- */
/* the linear dgrad whose epilogue runs the backward of the LayerNorm that produced its input (csrc/kernels.h, GemmP::ln_x): dY [M][K] bf16,
 * W [K][N] bf16 (N = the LayerNorm width), x [M][N] the LayerNorm's input, stats [M][2] its (mean, rstd), gamma [N]; dx [M][N] = the
 * LayerNorm's input gradient (+ addend, or null); dy_out (or null) [M][N] = dY W; pcol (or null) [cdiv(M, 128)][2][N] fp32 partial sums of
 * dgamma | dbeta per 128-row block.  N <= 1280, cdiv(M, 128) * cdiv(N, 128) <= 512.  Runs the launch three times (epochs 1, 2, 3) on one scratch buffer. */
SDXL_API int sdxl_op_linear_dgrad_ln_bwd(const void* dy, const void* w, const void* x, const float* stats, const void* gamma, const void* addend,
                                void* dx, void* dy_out, float* pcol, int M, int N, int K, void* stream);

/* experiment knobs of the plan (A/B runs; 0 = the shipped policy): see csrc/kernels.h.  Process-global, read at plan-build, forward
 * and backward time: set them before sdxl_plan / the first step and do not change them while a handle is in use. */
SDXL_API int sdxl_set_knob(int id, int value);
/* Persistent stream-K GEMM (csrc/gemm_sk.hip; 256 x 256 tiles, one workgroup per CU, the K-steps of ALL problems of a launch
 * cut evenly over the CUs).  sdxl_set_sk_mode: mode 0 = never (default), 1 = the policy gemm_use_sk, 2 = wherever a problem is
 * applicable (M, N multiples of 256, K of 64); workers > 0 forces the worker count (microbenchmarks), 0 = policy.
 * sdxl_sk_error: *out != 0 iff an owner workgroup ever gave up waiting for a partial tile on that stream (results invalid:
 * callers of modes 1 / 2 must check it).
 * sdxl_op_gemm_sk: n (<= 4) problems in ONE launch, arguments per problem as the single-problem GEMM op takes them (form 2: bias[i] = fp32 bias
 * gradient accumulator or NULL, C fp32). */
SDXL_API int sdxl_set_sk_mode(int mode, int workers);
SDXL_API int sdxl_sk_error(void* stream, unsigned* out);
/* *out != 0 iff a LayerNorm-backward epilogue (sdxl_op_linear_dgrad_ln_bwd, knob 26) gave up its in-launch meeting since the last call (bit 0: the ready
 * flags of its row block's other column tiles, bit 1: a granule): the results of that launch are invalid.  Synchronises the device; clears the word. */
SDXL_API int sdxl_ln_error(unsigned* out);
/* L2 prefetch of the weight (B operand) of a coming one-round launch of the pipelined kernel (form 0 NT: B [N][ldb]; 1 NN: B [K][ldb]), on `stream`:
 * workgroup i (XCD i % 8) reads the columns that XCD's tiles will stage, `parts` workgroups per XCD (profiles/tools/l2_prefetch_bench.py). */
SDXL_API int sdxl_op_pl_prefetch_b(int form, const void* B, int M, int N, int K, long ldb, int parts, void* stream);
SDXL_API int sdxl_op_gemm_sk(int n, const int* form, const void* const* A, const void* const* B, void* const* C, const int* M,
                    const int* N, const int* K, const void* const* bias, const void* const* resid, const int* accumulate,
                    void* stream);
/* The stride-2 3x3 convolution (pad 1, H and W even) on the four phase planes of its input, and its weight / bias gradient from the
   same planes: xplanar [4 * roundup(B*(H/2)*(W/2), 128)][Cin] bf16 is written by _fwd and read by _wgrad; y / dy [B][H/2][W/2][Cout];
   dw [Cout][9][Cin] fp32 (accumulate 0: =, 1: +=), dbias += (may be NULL). */
SDXL_API int sdxl_op_conv3x3_s2_fwd(const void* x, const void* w, const void* bias, void* xplanar, void* y, int B, int H, int W, int Cin,
                           int Cout, void* stream);
SDXL_API int sdxl_op_conv3x3_s2_wgrad(const void* dy, const void* xplanar, float* dw, float* dbias, int accumulate, int B, int H, int W, int Cin,
                             int Cout, int splitk, void* stream);

#ifdef __cplusplus
}
#endif
#endif
