/* libsdxlstep -- C ABI of the MI355X-native SDXL training step.
 *
 * Drop-in boundary for the hot path of DataCTE/SDXL-Training-Improvements (reference is 100 % Python and
 * has no native interface of its own; each entry point cites the reference call it replaces, paths under
 * /root/reference/src).  Plain pointers and sizes only -- no torch types.  All device pointers are raw HIP
 * device addresses (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*.
 *
 * Conventions: every function returns 0 on success, non-zero on failure (1 = bad argument, 2 = HIP error,
 * 3 = wrong state); sdxl_last_error() returns a thread-local message.  One handle per rank/GPU; a handle is not
 * thread-safe.  Calls are stream-ordered and do not synchronise unless stated.
 */
#ifndef SDXLSTEP_H
#define SDXLSTEP_H
#include <stddef.h>
#include <stdint.h>
/* Every entry point is marked SDXL_API: the library is built with -fvisibility=hidden, so these declarations (and the test hooks of
 * sdxlstep_diag.h) are the ONLY dynamic symbols libsdxlstep.so exports (tests/test_host_boundary.py checks `nm -D`). */
#ifndef SDXL_API
#define SDXL_API __attribute__((visibility("default")))
#endif
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdxl_handle sdxl_handle;

/* UNet2DConditionModel config (the fields of diffusers' unet/config.json that determine the arithmetic;
 * loaded by the reference at models/sdxl.py:25-40). */
typedef struct {
  int in_channels, out_channels;       /* 4, 4 */
  int block_out_channels[3];           /* 320, 640, 1280 */
  int layers_per_block;                /* 2 */
  int transformer_layers[3];           /* 0, 2, 10 */
  int head_dim;                        /* 64 (only value supported) */
  int cross_attention_dim;             /* 2048 */
  int norm_num_groups;                 /* 32 */
  int addition_time_embed_dim;         /* 256 */
  int pooled_dim;                      /* 1280 */
  float resnet_eps, tf_gn_eps, ln_eps; /* 1e-5, 1e-6, 1e-5 */
} sdxl_unet_config;

/* Loss configuration: config.yaml keys read by the path (training.method, training.prediction_type,
 * model.min_snr_gamma, model.use_ztsnr; ddpm_trainer.py:328,336, novelai_v3.py:106,117). */
typedef struct {
  int method;            /* 0 = ddpm (ddpm_trainer.py:280-405), 1 = flow_matching (flow_matching_trainer.py:267-356) */
  int prediction_type;   /* ddpm: 0 = epsilon, 1 = v_prediction */
  int use_min_snr;       /* model.min_snr_gamma is not None */
  float min_snr_gamma;
  int use_ztsnr;         /* clamp noisy latents to +-20000 */
} sdxl_loss_config;

/* One micro-batch, all device pointers.  RNG is the caller's: `noise` and `sigma_or_t` are inputs so that
 * fixtures are exact (reference draws them at ddpm_trainer.py:303-304 / flow_matching_trainer.py:298-306). */
typedef struct {
  int B, H, W;                 /* latent batch / height / width (pixels / 8) */
  int ctx_len;                 /* 77 */
  const float* latents;        /* [B,4,H,W] fp32 NCHW : batch["vae_latents"] (x for ddpm, x1 for flow matching) */
  const float* noise;          /* [B,4,H,W] fp32      : ddpm noise / flow-matching x0 */
  const float* sigma_or_t;     /* [B] fp32            : ddpm sigma = karras[timestep] ; flow matching t in (0,1) */
  const float* timestep;       /* [B] fp32            : value fed to the UNet time embedding (ddpm: index; fm: t) */
  const void* prompt_embeds;   /* [B,ctx_len,cross_attention_dim] bf16 : batch["prompt_embeds"] */
  const void* pooled;          /* [B,pooled_dim] bf16                 : batch["pooled_prompt_embeds"] */
  const float* time_ids;       /* [B,6] fp32                          : batch["time_ids"] */
  const float* tag_weights;    /* optional [B] fp32 (NULL = none)     : batch["tag_weights"] */
} sdxl_batch;

SDXL_API const char* sdxl_last_error(void);
SDXL_API int sdxl_version(void);

/* ---- lifetime ---------------------------------------------------------------------------------------------- */
SDXL_API int sdxl_default_config(sdxl_unet_config* cfg);                 /* SDXL-base-1.0 */
SDXL_API int sdxl_create(const sdxl_unet_config* cfg, int device, sdxl_handle** out);
SDXL_API int sdxl_destroy(sdxl_handle* h);

/* ---- parameters (replaces model.unet.parameters()/state_dict(), models/sdxl.py:237-240) ---------------------- */
/* bytes of the packed bf16 weight arena and the fp32 gradient arena */
SDXL_API int sdxl_param_bytes(sdxl_handle* h, size_t* weight_bytes, size_t* grad_bytes);
/* bind caller-allocated arenas (NULL = library allocates with hipMalloc) */
SDXL_API int sdxl_bind_params(sdxl_handle* h, void* weights_dev, void* grads_dev);
SDXL_API int sdxl_num_params(sdxl_handle* h);                            /* diffusers state-dict tensors */
SDXL_API int sdxl_param_info(sdxl_handle* h, int i, char* name, int name_cap, int* ndim, long shape[4]);
/* the contiguous element range tensor i occupies in the packed weight / gradient arenas (per-tensor optimizer
 * bookkeeping such as AdamWBF16's lazy decay) */
SDXL_API int sdxl_param_range(sdxl_handle* h, int i, size_t* elem_off, size_t* elems);
/* copy one tensor in PyTorch layout ([out,in] / [cout,cin,kh,kw]) from device memory into the packed arena
 * (dtype: 0 = fp32, 1 = bf16).  Caller keeps ownership of src. */
SDXL_API int sdxl_load_weight(sdxl_handle* h, const char* name, const void* src_dev, int dtype, void* stream);
SDXL_API int sdxl_export_weight(sdxl_handle* h, const char* name, void* dst_dev, int dtype, void* stream);
SDXL_API int sdxl_export_grad(sdxl_handle* h, const char* name, void* dst_dev, int dtype, void* stream);

/* ---- plan: static execution plan for one bucket shape (B, H, W) ------------------------------------------------ */
SDXL_API int sdxl_plan(sdxl_handle* h, int B, int H, int W, int ctx_len, size_t* workspace_bytes);
SDXL_API int sdxl_bind_workspace(sdxl_handle* h, void* ws_dev, size_t bytes);   /* NULL = library allocates */

/* ---- the step (replaces compute_loss()/training_step() + loss.backward()) -------------------------------------- */
/* start of an accumulation cycle: zeroes the bias / norm gradient vectors (accumulated with atomics); the weight-matrix
 * gradients are NOT touched -- the first micro-step (first_micro != 0) overwrites them.  A cycle must therefore start
 * with first_micro = 1. */
SDXL_API int sdxl_zero_grads(sdxl_handle* h, void* stream);
/* loss preparation + UNet forward + loss.  Leaves loss/metrics on device. */
SDXL_API int sdxl_forward_loss(sdxl_handle* h, const sdxl_loss_config* lc, const sdxl_batch* b, void* stream);
/* backward, split into segments (reverse execution order) so the caller can overlap gradient all-reduce of a
 * finished segment with the remaining backward (replaces DDP hooks, core/distributed.py:153-157).
 * first_micro != 0: gradients are overwritten (first micro-step after sdxl_zero_grads); else accumulated. */
SDXL_API int sdxl_num_segments(sdxl_handle* h);
SDXL_API int sdxl_segment_range(sdxl_handle* h, int seg, size_t* grad_elem_offset, size_t* grad_elems);
SDXL_API int sdxl_backward_segment(sdxl_handle* h, int seg, float grad_scale, int first_micro, void* stream);
/* By default the stream waits, at the end of every segment, for that segment's weight gradients (they run on an internal
 * side stream), so a caller can hand the segment to the gradient exchange.  A caller that consumes gradients only after
 * the whole backward (single GPU, or accumulation micro-steps without exchange) may set last_only = 1: the wait then
 * happens once, in the last segment (sdxl_loss_fwd_bwd always ends with it).
 * mode 2: as 1, and at every segment end the SIDE stream (sdxl_side_stream) waits for the caller's stream instead: a
 * caller that enqueues the segment's cast (sdxl_grads_to_bf16) and collective on the side stream gets the exchange
 * started without the caller's stream -- the critical path of the backward -- ever waiting or running the casts. */
SDXL_API int sdxl_set_join_mode(sdxl_handle* h, int mode);
/* the engine's side stream (hipStream_t; NULL when the serialized measurement mode is on) */
SDXL_API int sdxl_side_stream(sdxl_handle* h, void** stream);
/* convenience: forward + all backward segments */
SDXL_API int sdxl_loss_fwd_bwd(sdxl_handle* h, const sdxl_loss_config* lc, const sdxl_batch* b, float grad_scale,
                      int first_micro, void* stream);
/* every backward segment in one call (what a caller without a per-segment gradient exchange uses: one captured graph) */
SDXL_API int sdxl_backward_all(sdxl_handle* h, float grad_scale, int first_micro, void* stream);
/* hipGraph replay of forward / backward (default OFF: measured slower than eager two-stream launches on ROCm 7.2, see
 * DESIGN.md): the second call with a given (plan, loss configuration, first_micro,
 * grad_scale) captures the launch sequence of both streams, later calls replay it with one hipGraphLaunch.  0 = launch
 * kernel by kernel.  Inputs are staged at fixed addresses inside the plan, so the caller's tensors may move between steps. */
SDXL_API int sdxl_set_graph_mode(sdxl_handle* h, int on);
/* synchronises `stream`; out[0]=loss out[1]=sum w*(pred-target)^2 out[2]=sum|pred| out[3]=sum pred^2
 * out[4]=sum|noise| out[5]=sum noise^2 (x0) out[6]=sum latents^2 (x1) out[7]=gradient gate */
SDXL_API int sdxl_read_loss(sdxl_handle* h, float out[8], void* stream);

/* UNet only: sample_nhwc8 [B*H*W][8] bf16 in (channels 4..7 ignored) -> pred [B*H*W][8] bf16 out.
 * (replaces unet(sample, t, ehs, added_cond_kwargs).sample, ddpm_trainer.py:320-325) */
SDXL_API int sdxl_unet_forward(sdxl_handle* h, const void* sample_nhwc8, const sdxl_batch* cond, void* pred_nhwc8, void* stream);
/* d(pred) in -> runs every backward segment; d(sample) is not produced (inputs carry no gradient) */
SDXL_API int sdxl_unet_backward(sdxl_handle* h, const void* dpred_nhwc8, int first_micro, void* stream);

/* fp32 grads -> bf16 (scaled) for the gradient exchange; global L2 norm of the fp32 grads */
SDXL_API int sdxl_grads_to_bf16(sdxl_handle* h, size_t elem_offset, size_t elems, void* dst_bf16, float scale, void* stream);
/* Exchange micro-step without the cast pass: with a bf16 arena set (element offsets = the gradient arena's; NULL turns it
 * off), every weight-gradient GEMM of the following backward calls writes its FINAL value (including what earlier
 * micro-steps accumulated in fp32) x scale as bf16 there and leaves the fp32 arena alone; sdxl_small_grads_to_bf16 then
 * casts what the GEMMs do not produce (biases, norm parameters) for a segment range.  The fp32 arena is NOT the step's
 * gradient afterwards: use this only when the bf16 arena is what the exchange / optimizer consume. */
SDXL_API int sdxl_set_grad_emit(sdxl_handle* h, void* bf16_arena, float scale);
SDXL_API int sdxl_small_grads_to_bf16(sdxl_handle* h, size_t elem_offset, size_t elems, void* dst_bf16, float scale, void* stream);
SDXL_API int sdxl_grad_sumsq(sdxl_handle* h, float* out_dev, void* stream);
/* row f3 pieces: squared L2 norm of any fp32 (dtype 0) / bf16 (1) device array -- e.g. the all-reduced bf16 gradient
 * arena -- and torch.nn.utils.clip_grad_norm_'s coefficient min(1, max_norm / (norm + 1e-6)) computed on the device;
 * the coefficient is consumed by sdxl_adamw_bf16_step's grad_scale_dev, so clipping costs no pass over the gradients
 * (reference: clip_grad_norm_ then optimizer.step, flow_matching_trainer.py:181-189). */
SDXL_API int sdxl_sumsq(const void* x_dev, int dtype, size_t n, float* out_dev, void* stream);
SDXL_API int sdxl_clip_coef(const float* sumsq_dev, float max_norm, float* coef_dev, void* stream);

/* ---- single-kernel entry points (parity tests call these; same kernels the plan launches) --------------------- */
/* C[M,N] = A.B ; form 0: A[M,K],B[N,K] ; 1: A[M,K],B[K,N] ; 2: A[K,M],B[K,N] -> fp32 C (+= if accumulate).
 * form 2 (wgrad): `bias`, when given, is the fp32 bias-GRADIENT accumulator float[M]: += column sums of A.
 * splitk > 1: deterministic split-K (fp32 partial slabs, fixed-order sum): the wgrad form always; forms 0 / 1 when K is a
 * multiple of 64 (what the plan does for problems with fewer than 128 output tiles), otherwise ignored. */
SDXL_API int sdxl_op_gemm(int form, const void* A, const void* B, void* C, int M, int N, int K, const void* bias,
                 const void* resid, int accumulate, int splitk, void* stream);
/* n (<= 4) weight gradients of one shape in one launch, as the plan groups them: dw[i][Mo][No] (+)= dy[i]^T . x[i] with
 * dy[i] [rows][Mo], x[i] [rows][No] bf16; dbias (may be NULL, entries may be NULL): dbias[i][Mo] += column sums of dy[i]. */
SDXL_API int sdxl_op_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* dw, float* const* dbias, int Mo,
                        int No, int rows, int accumulate, void* stream);
/* 3x3 conv, pad 1, token-major: x [B,H,W,Cin], w [Cout][9][Cin] ; y [B,Ho,Wo,Cout] */
SDXL_API int sdxl_op_conv3x3_fwd(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin,
                        int Cout, int stride, void* stream);
/* conv3x3(nearest-2x upsample(x)) and its input gradient WITHOUT the upsampled image (the pair `Upsample2D` of diffusers =
   F.interpolate(scale 2, nearest) + Conv2d 3x3, which the reference's UNet runs at the two up-level transitions): per output phase a
   2 x 2 stencil on the low-resolution image with summed taps.  x [B][H][W][Cin], w [Cout][9][Cin], y / dy [B][2H][2W][Cout];
   weff [Cout][16][Cin] and planar [4 * roundup(B*H*W, 128)][Cout] are bf16 scratch of the caller (weff: written by _fwd, read by _dgrad);
   dx = addend (may be NULL) + gradient. */
SDXL_API int sdxl_op_upconv3x3_fwd(const void* x, const void* w, const void* bias, void* weff, void* planar, void* y, int B, int H, int W,
                          int Cin, int Cout, void* stream);
SDXL_API int sdxl_op_upconv3x3_dgrad(const void* dy, const void* weff, void* planar, void* dx, const void* addend, int B, int H, int W, int Cin,
                            int Cout, void* stream);
/* Input gradient of the stride-2 3x3 convolution (the two `Downsample2D` convs; pad 1, H and W even) by output phase: input pixel
   (2r + a, 2c + b) receives 1 / 2 / 2 / 4 of the nine taps.  dy [B][H/2][W/2][Cout], w [Cout][9][Cin], dx [B][H][W][Cin] = addend
   (may be NULL) + gradient; planar [4 * roundup(B*(H/2)*(W/2), 128)][Cin] bf16 scratch. */
SDXL_API int sdxl_op_conv3x3_s2_dgrad(const void* dy, const void* w, void* planar, void* dx, const void* addend, int B, int H, int W, int Cin,
                             int Cout, void* stream);
/* ... and its weight / bias gradient: `planar` as _dgrad left it (dy de-interleaved into its four phases), x the low-resolution input;
   dweff [Cout][16][Cin] fp32 scratch; dw [Cout][9][Cin] fp32 (accumulate 0: =, 1: +=), dbias[Cout] += (may be NULL); splitk >= 1. */
SDXL_API int sdxl_op_upconv3x3_wgrad(const void* planar, const void* x, float* dweff, float* dw, float* dbias, int accumulate, int B, int H,
                            int W, int Cin, int Cout, int splitk, void* stream);
SDXL_API int sdxl_op_conv3x3_dgrad(const void* dy, const void* w, void* dx, int B, int H, int W, int Cin, int Cout,
                          int stride, void* stream);
SDXL_API int sdxl_op_conv3x3_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                          int stride, int splitk, void* stream);
/* the same with the plan's options: dbias[Cout] += column sums of dy (may be NULL), accumulate 0 / 1, splitk <= 0 = the plan's choice.
 * Same-size stride-1 convolutions with W % 64 == 0 and >= 16 384 pixels run on the three-taps-per-workgroup kernel (conv_wgrad3.hip). */
SDXL_API int sdxl_op_conv3x3_wgrad2(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                           int stride, int splitk, int accumulate, void* stream);
SDXL_API int sdxl_op_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int heads,
                          int Nq, int Nk, long ldq, long ldk, long ldv, long ldo, void* stream);
SDXL_API int sdxl_op_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                          const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int heads, int Nq,
                          int Nk, long ldq, long ldk, long ldv, long ldo, void* stream);
SDXL_API int sdxl_op_groupnorm_fwd(const void* x, void* y, const void* gamma, const void* beta, float* stats, float* ws,
                          int B, int HW, int C, int G, float eps, int silu, void* stream);
SDXL_API int sdxl_op_groupnorm_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* stats,
                          void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G, int silu,
                          int accumulate, void* stream);
SDXL_API int sdxl_op_layernorm_fwd(const void* x, void* y, const void* gamma, const void* beta, float* stats, int M, int C,
                          float eps, void* stream);
SDXL_API int sdxl_op_layernorm_bwd(const void* x, const void* dy, const void* gamma, const float* stats, void* dx,
                          float* dgamma, float* dbeta, int M, int C, int accumulate, void* stream);
/* GEGLU feed-forward pair, fused into the two projections' GEMM epilogues.  w1 [2*C4][K] / b1 [2*C4] and the
 * pre-activation u [M][2*C4] use the library's packed order with group G = 64 or 80 (C4 % G == 0): channel c's value
 * half at row/column (c/G)*2G + c%G, its gate half G further.  sdxl_load_weight applies this to "ff.net.0.proj" with
 * G = 80 where 4*C divides (SDXL-base: 5120, 2560), else 64; callers only ever see the diffusers layout.
 * fwd: u = x @ w1^T + b1,  g[M][C4] = value * gelu(gate).
 * bwd: du[M][2*C4] from dy[M][C] @ w2[C][C4] (the second projection's input gradient) and u. */
SDXL_API int sdxl_op_ff_geglu_fwd(const void* x, const void* w1, const void* b1, void* u, void* g, int M, int K, int C4,
                         int group, void* stream);
SDXL_API int sdxl_op_ff_geglu_bwd(const void* dy, const void* w2, const void* u, void* du, int M, int C, int C4, int group,
                         void* stream);
SDXL_API int sdxl_op_loss(const sdxl_loss_config* lc, const sdxl_batch* b, void* unet_in, const void* pred, void* dpred,
                 float grad_scale, float* out8_dev, int phase /*0 prepare,1 loss,2 dpred*/, void* stream);
/* ---- row f1: fused AdamW_BF16 step (replaces AdamWBF16.step / _make_step,
 * reference src/training/optimizers/adamw_bfloat16/__init__.py:87-197 and stochastic/__init__.py:46-124).
 * One launch over flat arrays (the packed weight / gradient arenas, or any 16-byte aligned slice of them):
 * p, m (exp_avg), v (exp_avg_sq), shift are bf16 and updated in place; grad is fp32 (grad_dtype 0, the native arena) or
 * bf16 (1).  n must be a multiple of 8.  grad_scale_dev: optional device float multiplied into the gradient
 * (1/accumulation, clip coefficient).  rand_inject: optional uint16 [4][n] table of the random integers of the four
 * stochastic roundings in the reference's draw order (exp_avg, shift, p, shift) -- parity tests; NULL = Philox keyed
 * by (seed, step, element).  Weight decay is lazy in the reference (per tensor, paid when wd*lr accumulates past
 * 5e-3): decay_this_iteration applies to the whole launch; sdxl_adamw_decay pays it for one tensor's range. */
typedef struct {
  double lr, beta1, beta2, eps; /* doubles: the reference derives 1-beta, -lr*sqrt(1-beta2^t) in python floats */
  double step;                 /* t >= 1 of this update (denominator correction sqrt(1 - beta2^t)) */
  double decay_this_iteration; /* 0 = none */
  int reference_ema;           /* 1: the reference's arithmetic exactly, m <- SR(g + (1-beta1)*(beta1*m))  [its
                                  add_stochastic_ applies alpha to the wrong operand, stochastic/__init__.py:96];
                                  0: the documented EMA m <- SR(beta1*m + (1-beta1)*g) */
  int grad_round_bf16;         /* 1: round the scaled gradient to bf16 first (the reference's gradients are bf16) */
  unsigned long long seed;
  unsigned long long elem_offset; /* arena index of element 0 of this call (multiple of 8).  The stochastic-rounding counters are
                                   * (seed, step, arena index), so updating a sub-range [elem_offset, elem_offset + n) of the arena
                                   * (one rank's ZeRO-1 shard) gives exactly the bits of the full-arena update. */
} sdxl_adamw_config;
SDXL_API int sdxl_adamw_default_config(sdxl_adamw_config* c);   /* lr 1e-4, betas (0.9, 0.999), eps 1e-8, reference_ema 1 */
SDXL_API int sdxl_adamw_bf16_step(void* p, const void* grad, int grad_dtype, void* m, void* v, void* shift, size_t n,
                         const sdxl_adamw_config* c, const float* grad_scale_dev, const unsigned short* rand_inject,
                         void* stream);
SDXL_API int sdxl_adamw_decay(void* shift, const void* p, size_t n, float decay, void* stream);

/* measurement: between begin and end every launch of the bf16 MFMA GEMM family (Linear / conv fwd, dgrad, wgrad) is
 * bracketed by HIP events on its launch stream; end synchronises and returns the summed algorithmic FLOPs
 * (2*M*N*K*taps), the summed event time and the number of launches (bench.py's roofline block). */
SDXL_API int sdxl_profile_gemm_begin(void);
SDXL_API int sdxl_profile_gemm_end(double* flops, double* ms, int* launches);

/* Test hooks (layout probe, forced kernel configurations, activation checksums) and the experiment ABI of the diagnostics build
 * (knobs, stream-K, phase-plane stride-2 convolution) are NOT part of this boundary: include/sdxlstep_diag.h. */

#ifdef __cplusplus
}
#endif
#endif
