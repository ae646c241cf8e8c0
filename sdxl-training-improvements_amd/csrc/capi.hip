// extern "C" surface of libsdxlstep (see include/sdxlstep.h for the contract of every entry point).
#include <math.h>
#include "engine.h"
#include "../../include/sdxlstep_diag.h"
#include <functional>

#include <stdlib.h>

const char* sdxl_get_error();

struct StepState {  // what backward needs from the preceding forward
  sdxl_loss_config lc;
  sdxl_batch b;
  bool valid = false;
};
struct sdxl_handle {
  Engine e;
  StepState step;
};

#define H_CHECK(h) ARG_CHECK((h) != nullptr, "null handle")
#define CHK(x)         \
  do {                 \
    int _r = (x);      \
    if (_r) return _r; \
  } while (0)

extern "C" {

const char* sdxl_last_error(void) { return sdxl_get_error(); }
int sdxl_version(void) { return 1; }

int sdxl_default_config(sdxl_unet_config* c) {
  ARG_CHECK(c, "null config");
  c->in_channels = 4; c->out_channels = 4;
  c->block_out_channels[0] = 320; c->block_out_channels[1] = 640; c->block_out_channels[2] = 1280;
  c->layers_per_block = 2;
  c->transformer_layers[0] = 0; c->transformer_layers[1] = 2; c->transformer_layers[2] = 10;
  c->head_dim = 64; c->cross_attention_dim = 2048; c->norm_num_groups = 32;
  c->addition_time_embed_dim = 256; c->pooled_dim = 1280;
  c->resnet_eps = 1e-5f; c->tf_gn_eps = 1e-6f; c->ln_eps = 1e-5f;
  return 0;
}

int sdxl_create(const sdxl_unet_config* cfg, int device, sdxl_handle** out) {
  ARG_CHECK(cfg && out, "null argument");
  ARG_CHECK(cfg->head_dim == 64, "head_dim must be 64 (got %d)", cfg->head_dim);
  ARG_CHECK(cfg->in_channels == 4 && cfg->out_channels == 4, "in/out channels must be 4");
  ARG_CHECK(cfg->transformer_layers[0] == 0, "level 0 must not have attention");
  for (int i = 0; i < 3; ++i) {
    int c = cfg->block_out_channels[i];
    ARG_CHECK(c % cfg->norm_num_groups == 0 && c % 8 == 0, "block_out_channels[%d]=%d unsupported", i, c);
    if (cfg->transformer_layers[i] > 0) ARG_CHECK(c % 128 == 0, "attention level width %d must be a multiple of 128", c);
  }
  ARG_CHECK(cfg->cross_attention_dim % 8 == 0 && cfg->pooled_dim % 8 == 0 && cfg->addition_time_embed_dim % 8 == 0,
            "conditioning widths must be multiples of 8");
  int ndev = 0;
  HIP_CHECK_RET(hipGetDeviceCount(&ndev));
  ARG_CHECK(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
  HIP_CHECK_RET(hipSetDevice(device));
  sdxl_handle* h = new sdxl_handle();
  h->e.cfg = *cfg;
  h->e.device = device;
  h->e.build(nullptr);
  const char* ns = getenv("SDXL_NO_SIDE_STREAM");     // measurement mode: everything on the caller's stream (clean per-kernel
  h->e.use_side = !(ns && ns[0] == '1');              // durations for the serialized rocprof summaries under profiles/)
  if (h->e.use_side) {
    if (const char* sp = getenv("SDXL_SIDE_PRIO")) {   // measurement only: queue priority of the side stream (1 low, 0 normal, -1 high)
      HIP_CHECK_RET(hipStreamCreateWithPriority(&h->e.side, hipStreamNonBlocking, atoi(sp)));
    } else
    HIP_CHECK_RET(hipStreamCreateWithFlags(&h->e.side, hipStreamNonBlocking));   // (stream priorities: measured, neutral)
    HIP_CHECK_RET(hipStreamCreateWithFlags(&h->e.gstream, hipStreamNonBlocking));
    HIP_CHECK_RET(hipEventCreateWithFlags(&h->e.ev_gin, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&h->e.ev_gout, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&h->e.ev_join, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_CHECK_RET(hipEventCreateWithFlags(&h->e.ev_seg, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_CHECK_RET(hipEventCreateWithFlags(&h->e.ev_hoist, hipEventDisableTiming | hipEventDisableSystemFence));
  }
  *out = h;
  return 0;
}

int sdxl_destroy(sdxl_handle* h) {
  if (!h) return 0;
  if (h->e.own_weights && h->e.weights) (void)hipFree(h->e.weights);
  if (h->e.own_grads && h->e.grads) (void)hipFree(h->e.grads);
  if (h->e.own_ws && h->e.ws) (void)hipFree(h->e.ws);
  h->e.clear_graphs();
  if (h->e.gstream) (void)hipStreamDestroy(h->e.gstream);
  if (h->e.ev_gin) (void)hipEventDestroy(h->e.ev_gin);
  if (h->e.ev_gout) (void)hipEventDestroy(h->e.ev_gout);
  for (hipEvent_t ev : h->e.ev_pool) (void)hipEventDestroy(ev);
  if (h->e.ev_join) (void)hipEventDestroy(h->e.ev_join);
  if (h->e.ev_seg) (void)hipEventDestroy(h->e.ev_seg);
  if (h->e.ev_hoist) (void)hipEventDestroy(h->e.ev_hoist);
  if (h->e.side) (void)hipStreamDestroy(h->e.side);
  if (h->e.small_ranges_dev) (void)hipFree(h->e.small_ranges_dev);
  delete h;
  return 0;
}

int sdxl_param_bytes(sdxl_handle* h, size_t* wb, size_t* gb) {
  H_CHECK(h);
  if (wb) *wb = h->e.param_elems * sizeof(bf16);
  if (gb) *gb = h->e.param_elems * sizeof(float);
  return 0;
}

int sdxl_bind_params(sdxl_handle* h, void* w, void* g) {
  H_CHECK(h);
  Engine& e = h->e;
  e.clear_graphs();                                   // captured kernels hold the old arena addresses
  if (w) { e.weights = (bf16*)w; e.own_weights = false; }
  else {
    HIP_CHECK_RET(hipMalloc((void**)&e.weights, e.param_elems * sizeof(bf16)));
    HIP_CHECK_RET(hipMemset(e.weights, 0, e.param_elems * sizeof(bf16)));
    e.own_weights = true;
  }
  if (g) { e.grads = (float*)g; e.own_grads = false; }
  else {
    HIP_CHECK_RET(hipMalloc((void**)&e.grads, e.param_elems * sizeof(float)));
    HIP_CHECK_RET(hipMemset(e.grads, 0, e.param_elems * sizeof(float)));
    e.own_grads = true;
  }
  ARG_CHECK(((uintptr_t)e.weights & 255) == 0 && ((uintptr_t)e.grads & 255) == 0, "arenas must be 256-byte aligned");
  return 0;
}

int sdxl_num_params(sdxl_handle* h) { return h ? (int)h->e.src.size() : -1; }

int sdxl_param_info(sdxl_handle* h, int i, char* name, int cap, int* ndim, long shape[4]) {
  H_CHECK(h);
  ARG_CHECK(i >= 0 && i < (int)h->e.src.size(), "parameter index %d out of range", i);
  const SrcParam& s = h->e.src[i];
  if (name && cap > 0) snprintf(name, cap, "%s", s.name.c_str());
  if (ndim) *ndim = s.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
  return 0;
}

int sdxl_param_range(sdxl_handle* h, int i, size_t* elem_off, size_t* elems) {
  H_CHECK(h);
  ARG_CHECK(i >= 0 && i < (int)h->e.src.size(), "parameter index %d out of range", i);
  const SrcParam& s = h->e.src[i];
  size_t n = 1;
  for (int k = 0; k < s.ndim; ++k) n *= (size_t)s.shape[k];
  if (s.kind == 1) n = (size_t)s.shape[0] * 9 * (size_t)s.ci_pad;   // conv: [cout][tap][cin padded]
  if (elem_off) *elem_off = s.native.off + s.elem_off;
  if (elems) *elems = n;
  return 0;
}

int sdxl_load_weight(sdxl_handle* h, const char* name, const void* src, int dtype, void* st) {
  H_CHECK(h);
  ARG_CHECK(name && src, "null argument");
  return engine_load_weight(h->e, name, src, dtype, (hipStream_t)st);
}
int sdxl_export_weight(sdxl_handle* h, const char* name, void* dst, int dtype, void* st) {
  H_CHECK(h);
  ARG_CHECK(name && dst, "null argument");
  return engine_export(h->e, name, dst, dtype, false, (hipStream_t)st);
}
int sdxl_export_grad(sdxl_handle* h, const char* name, void* dst, int dtype, void* st) {
  H_CHECK(h);
  ARG_CHECK(name && dst, "null argument");
  return engine_export(h->e, name, dst, dtype, true, (hipStream_t)st);
}

int sdxl_plan(sdxl_handle* h, int B, int H, int W, int ctx, size_t* ws_bytes) {
  H_CHECK(h);
  ARG_CHECK(B > 0 && H > 0 && W > 0 && ctx > 0, "bad plan shape B=%d H=%d W=%d ctx=%d", B, H, W, ctx);
  ARG_CHECK(H % 4 == 0 && W % 4 == 0, "latent H=%d W=%d must be multiples of 4 (two stride-2 levels)", H, W);
  Engine& e = h->e;
  auto key = std::make_tuple(B, H, W, ctx);
  auto it = e.plans.find(key);
  if (it == e.plans.end()) {
    std::unique_ptr<Plan> p(new Plan());
    p->eng = &e;
    p->B = B; p->H = H; p->W = W; p->ctx = ctx;
    e.build(p.get());
    it = e.plans.emplace(key, std::move(p)).first;
  }
  if (e.cur != it->second.get()) e.drop_pending();
  e.cur = it->second.get();
  if (ws_bytes) *ws_bytes = e.cur->ws_bytes;
  return 0;
}

int sdxl_bind_workspace(sdxl_handle* h, void* ws, size_t bytes) {
  H_CHECK(h);
  Engine& e = h->e;
  if (ws) {
    if (ws == (void*)e.ws && bytes == e.ws_cap) return 0;     // unchanged: captured graphs stay valid
    e.clear_graphs();
    if (e.own_ws && e.ws) (void)hipFree(e.ws);
    e.ws = (char*)ws; e.ws_cap = bytes; e.own_ws = false;
  } else {
    size_t need = bytes;
    for (auto& kv : e.plans) if (kv.second->ws_bytes > need) need = kv.second->ws_bytes;
    if (e.own_ws && e.ws && e.ws_cap >= need) return 0;
    e.clear_graphs();
    if (e.own_ws && e.ws) (void)hipFree(e.ws);
    HIP_CHECK_RET(hipMalloc((void**)&e.ws, need));
    e.ws_cap = need; e.own_ws = true;
  }
  ARG_CHECK(((uintptr_t)e.ws & 255) == 0, "workspace must be 256-byte aligned");
  return 0;
}

static int ready(Engine& e) {
  if (!e.cur) { sdxl_set_error("no plan: call sdxl_plan first"); return 3; }
  if (!e.weights || !e.grads) { sdxl_set_error("parameters are not bound: call sdxl_bind_params"); return 3; }
  if (!e.ws || e.ws_cap < e.cur->ws_bytes) {
    sdxl_set_error("workspace too small: have %zu bytes, plan needs %zu", e.ws_cap, e.cur->ws_bytes);
    return 3;
  }
  return 0;
}

// zero the gradient ranges that are accumulated with += by every backward (bias / norm vectors: ~2.6 M of the 2.57 G elements; one writer per element and launch);
// the weight matrices are overwritten by the first micro-step's wgrad GEMMs (first_micro) and need no zeroing
static int small_ranges_on_device(Engine& e);
__global__ void zero_ranges_kernel(float* __restrict__ g, const unsigned long long* __restrict__ ranges) {
  const unsigned long long off = ranges[2 * blockIdx.x], n = ranges[2 * blockIdx.x + 1];
  for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) g[off + i] = 0.f;
}

int sdxl_zero_grads(sdxl_handle* h, void* st) {
  H_CHECK(h);
  Engine& e = h->e;
  ARG_CHECK(e.grads, "grads are not bound");
  CHK(small_ranges_on_device(e));
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)e.small_ranges.size()), dim3(256), 0, (hipStream_t)st, e.grads,
                     e.small_ranges_dev);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int check_batch(Engine& e, const sdxl_batch* b) {
  Plan& p = *e.cur;
  ARG_CHECK(b, "null batch");
  ARG_CHECK(b->B == p.B && b->H == p.H && b->W == p.W && b->ctx_len == p.ctx,
            "batch shape (B=%d,H=%d,W=%d,ctx=%d) does not match the current plan (B=%d,H=%d,W=%d,ctx=%d)", b->B, b->H,
            b->W, b->ctx_len, p.B, p.H, p.W, p.ctx);
  ARG_CHECK(b->timestep && b->prompt_embeds && b->pooled && b->time_ids, "batch is missing conditioning pointers");
  return 0;
}

static int upload_cond(Engine& e, const sdxl_batch* b, hipStream_t st) {
  Plan& p = *e.cur;
  const sdxl_unet_config& c = e.cfg;
  HIP_CHECK_RET(hipMemcpyAsync(p.P(p.ehs), b->prompt_embeds, (size_t)p.B * p.ctx * c.cross_attention_dim * sizeof(bf16),
                               hipMemcpyDeviceToDevice, st));
  HIP_CHECK_RET(hipMemcpyAsync(p.F(p.t_off), b->timestep, sizeof(float) * p.B, hipMemcpyDeviceToDevice, st));
  HIP_CHECK_RET(hipMemcpyAsync(p.F(p.tid_off), b->time_ids, sizeof(float) * p.B * 6, hipMemcpyDeviceToDevice, st));
  CHK(launch_copy_cols((const bf16*)b->pooled, c.pooled_dim, p.P(p.aug_in), c.pooled_dim + 6L * c.addition_time_embed_dim,
                       p.B, c.pooled_dim, st));
  return 0;
}

static void fill_loss(Engine& e, const sdxl_loss_config* lc, const sdxl_batch* b, float grad_scale, LossP& L) {
  Plan& p = *e.cur;
  memset(&L, 0, sizeof(L));
  L.method = lc->method; L.prediction_type = lc->prediction_type; L.use_min_snr = lc->use_min_snr;
  L.min_snr_gamma = lc->min_snr_gamma; L.use_ztsnr = lc->use_ztsnr;
  L.B = p.B; L.HW = p.H * p.W; L.C = 4;
  L.latents = b->latents; L.noise = b->noise; L.sigma = b->sigma_or_t; L.tag_w = b->tag_weights;
  L.unet_in = p.P(p.x_in); L.pred = p.P(p.pred); L.dpred = p.G(p.pred);
  L.grad_scale = grad_scale;
  L.out = p.F(p.loss_off);
  L.part = p.F(p.loss_part_off);
}


// run `body(stream)` once eagerly (first call for a key: one-time function attributes), capture it on the second call, replay
// the instantiated graph from then on; everything on the engine's graph stream, fenced against the caller's stream
static int run_graphed(Engine& e, Engine::GraphKey key, hipStream_t user, const std::function<int(hipStream_t)>& body) {
  if (!e.use_graphs || !e.gstream || gemm_profiling()) return body(user);
  Engine::GraphEntry& g = e.graphs[key];
  hipStream_t gs = e.gstream;
  HIP_CHECK_RET(hipEventRecord(e.ev_gin, user));
  HIP_CHECK_RET(hipStreamWaitEvent(gs, e.ev_gin, 0));
  if (g.exec) {
    HIP_CHECK_RET(hipGraphLaunch(g.exec, gs));
  } else if (g.seen++ == 0) {
    CHK(body(gs));
  } else {
    HIP_CHECK_RET(hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal));
    const int rc = body(gs);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(gs, &graph);
    if (rc || ce != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      if (rc) return rc;
      sdxl_set_error("hipStreamEndCapture: %s", hipGetErrorString(ce));
      return 2;
    }
    HIP_CHECK_RET(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    HIP_CHECK_RET(hipGraphLaunch(g.exec, gs));
  }
  HIP_CHECK_RET(hipEventRecord(e.ev_gout, gs));
  HIP_CHECK_RET(hipStreamWaitEvent(user, e.ev_gout, 0));
  return 0;
}
static unsigned loss_cfg_bits(const sdxl_loss_config& lc, bool tag) {
  unsigned g;
  memcpy(&g, &lc.min_snr_gamma, 4);
  return (unsigned)lc.method | ((unsigned)lc.prediction_type << 2) | ((unsigned)lc.use_min_snr << 4) | ((unsigned)lc.use_ztsnr << 5) |
         ((unsigned)tag << 6) | (g << 7);
}

static int run_forward_ops(Engine& e, hipStream_t st) {
  Plan& p = *e.cur;
  e.drop_pending();      // (leftovers of a backward that failed half way)
  const bool side = e.use_side && e.side && !gemm_profiling();
  if (side) {   // hoisted ops (inputs-only dependencies) run on the side stream, concurrently with the first layers
    HIP_CHECK_RET(hipEventRecord(e.ev_hoist, st));
    HIP_CHECK_RET(hipStreamWaitEvent(e.side, e.ev_hoist, 0));
    for (auto& op : p.ops) if (op->hoist_fwd) CHK(op->fwd(p, e.side));
    HIP_CHECK_RET(hipEventRecord(e.ev_hoist, e.side));
  }
  bool waited = false;
  for (auto& op : p.ops) {
    if (side && op->hoist_fwd) continue;
    if (side && op->needs_hoisted && !waited) {
      HIP_CHECK_RET(hipStreamWaitEvent(st, e.ev_hoist, 0));
      waited = true;
    }
    CHK(op->fwd(p, st));
  }
  if (side && e.side_dirty) {      // forward work put on the side stream (an upsampled image only a weight gradient reads): the caller's
    HIP_CHECK_RET(hipEventRecord(e.ev_join, e.side));      // stream owns everything the forward produced once this returns
    HIP_CHECK_RET(hipStreamWaitEvent(st, e.ev_join, 0));
    e.side_dirty = false;
  }
  return 0;
}

int sdxl_forward_loss(sdxl_handle* h, const sdxl_loss_config* lc, const sdxl_batch* b, void* stp) {
  H_CHECK(h);
  Engine& e = h->e;
  hipStream_t st = (hipStream_t)stp;
  CHK(ready(e));
  ARG_CHECK(lc, "null loss config");
  ARG_CHECK(lc->method == 0 || lc->method == 1, "unknown method %d", lc->method);
  CHK(check_batch(e, b));
  ARG_CHECK(b->latents && b->noise && b->sigma_or_t, "batch is missing latents/noise/sigma");
  CHK(upload_cond(e, b, st));
  // the step's inputs are staged at fixed addresses inside the plan (the caller's tensors move from step to step; the captured
  // kernels must not)
  Plan& p = *e.cur;
  sdxl_batch sb = *b;
  if (e.use_graphs) {
    const size_t nlat = sizeof(float) * (size_t)p.B * 4 * p.H * p.W;
    HIP_CHECK_RET(hipMemcpyAsync(p.F(p.in_lat_off), b->latents, nlat, hipMemcpyDeviceToDevice, st));
    HIP_CHECK_RET(hipMemcpyAsync(p.F(p.in_noise_off), b->noise, nlat, hipMemcpyDeviceToDevice, st));
    HIP_CHECK_RET(hipMemcpyAsync(p.F(p.in_sig_off), b->sigma_or_t, sizeof(float) * p.B, hipMemcpyDeviceToDevice, st));
    if (b->tag_weights) HIP_CHECK_RET(hipMemcpyAsync(p.F(p.in_tag_off), b->tag_weights, sizeof(float) * p.B, hipMemcpyDeviceToDevice, st));
    sb.latents = p.F(p.in_lat_off); sb.noise = p.F(p.in_noise_off); sb.sigma_or_t = p.F(p.in_sig_off);
    sb.tag_weights = b->tag_weights ? p.F(p.in_tag_off) : nullptr;
  }
  LossP L;
  fill_loss(e, lc, &sb, 1.f, L);
  Engine::GraphKey key{&p, 0, 0, 0, 0, 0u, loss_cfg_bits(*lc, b->tag_weights != nullptr)};
  CHK(run_graphed(e, key, st, [&](hipStream_t s) -> int {
    CHK(launch_loss_prepare(L, s));
    CHK(run_forward_ops(e, s));
    CHK(launch_loss_fwd(L, s));
    return 0;
  }));
  h->step.lc = *lc; h->step.b = sb; h->step.valid = true;
  return 0;
}

int sdxl_num_segments(sdxl_handle* h) { return h ? h->e.nseg : -1; }
int sdxl_set_join_mode(sdxl_handle* h, int mode) {
  H_CHECK(h);
  ARG_CHECK(mode >= 0 && mode <= 2, "join mode %d (0, 1, 2)", mode);
  h->e.join_last_only = mode != 0;
  h->e.seg_on_side = mode == 2 && h->e.use_side && h->e.side;
  return 0;
}
int sdxl_side_stream(sdxl_handle* h, void** stream) {
  H_CHECK(h);
  ARG_CHECK(stream, "null output");
  *stream = h->e.use_side ? (void*)h->e.side : nullptr;
  return 0;
}

int sdxl_segment_range(sdxl_handle* h, int k, size_t* off, size_t* n) {
  H_CHECK(h);
  ARG_CHECK(k >= 0 && k < h->e.nseg, "segment %d out of range", k);
  int s = h->e.nseg - 1 - k;
  if (off) *off = h->e.seg_begin[s];
  if (n) *n = h->e.seg_end[s] - h->e.seg_begin[s];
  return 0;
}

static int run_backward_segment_body(Engine& e, int k, bool first, hipStream_t st);
static int run_backward_segment(Engine& e, int k, bool first, hipStream_t st) {
  if (k == 0) e.drop_pending();
  const int rc = run_backward_segment_body(e, k, first, st);
  if (rc) e.drop_pending();      // an op failed: its queued leaves / grouped weight gradients must not ride on a later step's fork event
  return rc;
}
static int run_backward_segment_body(Engine& e, int k, bool first, hipStream_t st) {
  Plan& p = *e.cur;
  int s = e.nseg - 1 - k;
  e.ev_used = 0;   // per-op events are consumed in order; a segment's waits are all enqueued before the pool is reused
  if (k == 0 && p.tp32_off != NONE) HIP_CHECK_RET(hipMemsetAsync(p.F(p.tp32_off), 0, p.tp32_bytes, st));
  if (k == 0 && p.ln_part_floats) {      // the fused LayerNorm backward's tagged partial sums: epochs count from 1 in every backward
    HIP_CHECK_RET(hipMemsetAsync(p.F(p.ln_part_off), 0, p.ln_part_floats * sizeof(float), st));
    p.ln_epoch = 0;
  }
  for (int i = p.seg_last_op[s]; i >= p.seg_first_op[s] && i >= 0; --i) CHK(p.ops[i]->bwd(p, st, first));
  // the segment's weight gradients are complete once `st` passes this point -- unless the caller declared (sdxl_set_join_mode)
  // that it only needs that of the whole backward (no per-segment gradient exchange): then the side stream runs free
  // until the last segment (nothing on the main stream reads a weight gradient, and no gradient buffer is reused)
  CHK(e.flush_wgrads(p, st));
  CHK(e.flush_ln_params(p, st));
  if (e.seg_on_side && !gemm_profiling()) {   // the side stream sees the segment's main-stream gradients (norm parameters, ...)
    HIP_CHECK_RET(hipEventRecord(e.ev_seg, st));
    HIP_CHECK_RET(hipStreamWaitEvent(e.side, e.ev_seg, 0));
    e.side_dirty = true;                       // (the caller's cast + collective follow on the side stream)
  }
  if (e.side_dirty && !(e.join_last_only && k != e.nseg - 1)) {
    HIP_CHECK_RET(hipEventRecord(e.ev_join, e.side));
    HIP_CHECK_RET(hipStreamWaitEvent(st, e.ev_join, 0));
    e.side_dirty = false;
  }
  return 0;
}

int sdxl_backward_segment(sdxl_handle* h, int k, float grad_scale, int first_micro, void* stp) {
  H_CHECK(h);
  Engine& e = h->e;
  hipStream_t st = (hipStream_t)stp;
  CHK(ready(e));
  ARG_CHECK(k >= 0 && k < e.nseg, "segment %d out of range", k);
  if (k == 0) ARG_CHECK(h->step.valid, "sdxl_backward_segment(0) needs a preceding sdxl_forward_loss");
  auto body = [&](hipStream_t s) -> int {
    if (k == 0) {
      LossP L;
      fill_loss(e, &h->step.lc, &h->step.b, grad_scale, L);
      CHK(launch_loss_bwd(L, s));
    }
    return run_backward_segment(e, k, first_micro != 0, s);
  };
  // a segment can only be captured on its own when it ends with the side stream joined (per-segment join mode, or the last one)
  if (e.join_last_only) return body(st);
  unsigned sbits;
  memcpy(&sbits, &grad_scale, 4);
  Engine::GraphKey key{e.cur, 1, k, first_micro != 0, 0, sbits, loss_cfg_bits(h->step.lc, h->step.b.tag_weights != nullptr)};
  return run_graphed(e, key, st, body);
}

int sdxl_backward_all(sdxl_handle* h, float grad_scale, int first_micro, void* stp) {
  H_CHECK(h);
  Engine& e = h->e;
  hipStream_t st = (hipStream_t)stp;
  CHK(ready(e));
  ARG_CHECK(h->step.valid, "sdxl_backward_all needs a preceding sdxl_forward_loss");
  unsigned sbits;
  memcpy(&sbits, &grad_scale, 4);
  Engine::GraphKey key{e.cur, 2, 0, first_micro != 0, e.join_last_only, sbits, loss_cfg_bits(h->step.lc, h->step.b.tag_weights != nullptr)};
  return run_graphed(e, key, st, [&](hipStream_t s) -> int {
    LossP L;
    fill_loss(e, &h->step.lc, &h->step.b, grad_scale, L);
    CHK(launch_loss_bwd(L, s));
    for (int k = 0; k < e.nseg; ++k) CHK(run_backward_segment(e, k, first_micro != 0, s));
    return 0;
  });
}
int sdxl_set_graph_mode(sdxl_handle* h, int on) {
  H_CHECK(h);
  h->e.use_graphs = on != 0 && h->e.use_side;
  if (!h->e.use_graphs) h->e.clear_graphs();
  return 0;
}

int sdxl_loss_fwd_bwd(sdxl_handle* h, const sdxl_loss_config* lc, const sdxl_batch* b, float grad_scale,
                      int first_micro, void* st) {
  CHK(sdxl_forward_loss(h, lc, b, st));
  return sdxl_backward_all(h, grad_scale, first_micro, st);
}

int sdxl_read_loss(sdxl_handle* h, float out[8], void* stp) {
  H_CHECK(h);
  CHK(ready(h->e));
  hipStream_t st = (hipStream_t)stp;
  HIP_CHECK_RET(hipMemcpyAsync(out, h->e.cur->F(h->e.cur->loss_off), 8 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK_RET(hipStreamSynchronize(st));
  return 0;
}

int sdxl_unet_forward(sdxl_handle* h, const void* sample, const sdxl_batch* cond, void* pred, void* stp) {
  H_CHECK(h);
  Engine& e = h->e;
  hipStream_t st = (hipStream_t)stp;
  CHK(ready(e));
  CHK(check_batch(e, cond));
  Plan& p = *e.cur;
  CHK(upload_cond(e, cond, st));
  size_t bytes = (size_t)p.B * p.H * p.W * 8 * sizeof(bf16);
  HIP_CHECK_RET(hipMemcpyAsync(p.P(p.x_in), sample, bytes, hipMemcpyDeviceToDevice, st));
  CHK(run_forward_ops(e, st));
  if (pred) HIP_CHECK_RET(hipMemcpyAsync(pred, p.P(p.pred), bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

int sdxl_unet_backward(sdxl_handle* h, const void* dpred, int first_micro, void* stp) {
  H_CHECK(h);
  Engine& e = h->e;
  hipStream_t st = (hipStream_t)stp;
  CHK(ready(e));
  Plan& p = *e.cur;
  size_t bytes = (size_t)p.B * p.H * p.W * 8 * sizeof(bf16);
  HIP_CHECK_RET(hipMemcpyAsync(p.G(p.pred), dpred, bytes, hipMemcpyDeviceToDevice, st));
  for (int k = 0; k < e.nseg; ++k) CHK(run_backward_segment(e, k, first_micro != 0, st));
  return 0;
}

int sdxl_grads_to_bf16(sdxl_handle* h, size_t off, size_t n, void* dst, float scale, void* st) {
  H_CHECK(h);
  ARG_CHECK(h->e.grads && off + n <= h->e.param_elems, "range out of bounds");
  return launch_f32_to_bf16(h->e.grads + off, (bf16*)dst, (long)n, scale, (hipStream_t)st);
}

// bf16 cast of the SMALL parameter ranges (biases, norm weights: the fp32 += accumulators) inside [off, off + n): what is
// left to cast when the weight-gradient GEMMs emit bf16 themselves (sdxl_set_grad_emit)
__global__ void cast_small_ranges_kernel(const float* __restrict__ g, bf16* __restrict__ dst, const unsigned long long* __restrict__ ranges,
                                         unsigned long long lo, unsigned long long hi, float scale) {
  const unsigned long long off = ranges[2 * blockIdx.x], n = ranges[2 * blockIdx.x + 1];
  if (off < lo || off + n > hi) return;
  for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) dst[off - lo + i] = (bf16)(g[off + i] * scale);
}
static int small_ranges_on_device(Engine& e) {
  if (!e.small_ranges_dev) {
    std::vector<unsigned long long> flat;
    for (auto& r : e.small_ranges) { flat.push_back(r.first); flat.push_back(r.second); }
    HIP_CHECK_RET(hipMalloc((void**)&e.small_ranges_dev, flat.size() * sizeof(unsigned long long)));
    HIP_CHECK_RET(hipMemcpy(e.small_ranges_dev, flat.data(), flat.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  return 0;
}
int sdxl_small_grads_to_bf16(sdxl_handle* h, size_t off, size_t n, void* dst, float scale, void* st) {
  H_CHECK(h);
  Engine& e = h->e;
  ARG_CHECK(e.grads && off + n <= e.param_elems && dst, "range out of bounds");
  CHK(small_ranges_on_device(e));
  if (e.small_ranges.empty()) return 0;
  hipLaunchKernelGGL(cast_small_ranges_kernel, dim3((unsigned)e.small_ranges.size()), dim3(256), 0, (hipStream_t)st, e.grads, (bf16*)dst,
                     e.small_ranges_dev, (unsigned long long)off, (unsigned long long)(off + n), scale);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int sdxl_set_grad_emit(sdxl_handle* h, void* bf16_arena, float scale) {
  H_CHECK(h);
  ARG_CHECK(((uintptr_t)bf16_arena & 15) == 0, "bf16 gradient arena must be 16-byte aligned");
  if (h->e.emit_base != (bf16*)bf16_arena || h->e.emit_scale != scale) h->e.clear_graphs();   // captured wgrad launches hold the old target
  h->e.emit_base = (bf16*)bf16_arena;
  h->e.emit_scale = scale;
  return 0;
}

int sdxl_grad_sumsq(sdxl_handle* h, float* out, void* st) {
  H_CHECK(h);
  ARG_CHECK(h->e.grads, "grads are not bound");
  HIP_CHECK_RET(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)st));
  return launch_sumsq_f32(h->e.grads, (long)h->e.param_elems, out, (hipStream_t)st);
}

int sdxl_sumsq(const void* x, int dtype, size_t n, float* out, void* st) {
  ARG_CHECK(x && out && (dtype == 0 || dtype == 1), "sumsq: bad arguments");
  ARG_CHECK(((uintptr_t)x & 15) == 0, "sumsq: x must be 16-byte aligned");
  HIP_CHECK_RET(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)st));
  return dtype == 0 ? launch_sumsq_f32((const float*)x, (long)n, out, (hipStream_t)st)
                    : launch_sumsq_bf16((const bf16*)x, (long)n, out, (hipStream_t)st);
}
int sdxl_clip_coef(const float* sumsq_dev, float max_norm, float* coef_dev, void* st) {
  ARG_CHECK(sumsq_dev && coef_dev && max_norm > 0.f, "clip_coef: bad arguments");
  return launch_clip_coef(sumsq_dev, max_norm, coef_dev, (hipStream_t)st);
}

// ------------------------------------------------------------------------------------------------------------
// single-kernel entry points
// ------------------------------------------------------------------------------------------------------------
static float* g_test_slab = nullptr;
static size_t g_test_slab_floats = 0;
static int test_slab(size_t floats, float** out) {   // scratch for the split-K slabs of the test entry points
  if (floats > g_test_slab_floats) {
    if (g_test_slab) (void)hipFree(g_test_slab);
    HIP_CHECK_RET(hipMalloc((void**)&g_test_slab, floats * sizeof(float)));
    g_test_slab_floats = floats;
  }
  *out = g_test_slab;
  return 0;
}

int sdxl_op_gemm(int form, const void* A, const void* B, void* C, int M, int N, int K, const void* bias,
                 const void* resid, int accumulate, int splitk, void* st) {
  GemmP g;
  gemm_defaults(&g);
  g.form = form;
  g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  if (form == GEMM_NT) { g.lda = K; g.ldb = K; }
  else if (form == GEMM_NN) { g.lda = K; g.ldb = N; }
  if (form != GEMM_TN && splitk > 1) {
    g.splitk = splitk;
    CHK(test_slab(gemm_slab_floats(M, N, 1, splitk), &g.slab));
  }
  if (form == GEMM_TN) {
    if (splitk <= 0) splitk = wgrad256_policy(M, N, K) ? wgrad256_pick_splitk(M, N, K) : gemm_pick_splitk(M, N, 1, K);     // the plan's choice
    g.lda = M; g.ldb = N; g.out_f32 = 1; g.splitk = splitk;
    if (splitk > 1) CHK(test_slab(gemm_slab_floats(M, N, 1, splitk), &g.slab));
  }
  g.ldc = N;
  if (form == GEMM_TN) g.bias_grad = (float*)bias;
  else g.bias = (const bf16*)bias;
  if (resid) { g.resid = (const bf16*)resid; g.ldr = N; }
  g.accumulate = accumulate;
  return launch_gemm(g, (hipStream_t)st);
}

int sdxl_op_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* dw, float* const* dbias, int Mo, int No,
                        int rows, int accumulate, void* st) {
  ARG_CHECK(n >= 1 && n <= GEMM_MAX_GROUP, "wgrad group of %d (1..%d)", n, GEMM_MAX_GROUP);
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_TN;
  g.M = Mo; g.N = No; g.K = rows;
  g.lda = Mo; g.ldb = No; g.ldc = No;
  g.out_f32 = 1;
  g.accumulate = accumulate;
  g.A = (const bf16*)dy[0]; g.B = (const bf16*)x[0]; g.C = dw[0]; g.bias_grad = dbias ? dbias[0] : nullptr;
  g.group = n;
  for (int i = 0; i < n; ++i) {
    g.gA[i] = (const bf16*)dy[i]; g.gB[i] = (const bf16*)x[i]; g.gC[i] = dw[i]; g.gbias_grad[i] = dbias ? dbias[i] : nullptr;
  }
  return launch_gemm(g, (hipStream_t)st);
}

int sdxl_op_conv3x3_fwd(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                        int stride, void* st) {
  int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NT;
  g.A = (const bf16*)x; g.B = (const bf16*)w; g.C = y;
  g.M = B * Ho * Wo; g.N = Cout; g.K = Cin;
  g.lda = Cin; g.ldb = 9L * Cin; g.ldc = Cout;
  g.taps = 9; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.sm = stride; g.sd = 1;
  g.b_tap_stride = Cin;
  g.bias = (const bf16*)bias;
  if (stride == 1 && Cin % 64 == 0) {     // as the plan does: small images split the (tap, channel) reduction
    g.splitk = gemm_pick_splitk_small(g.M, Cout, 9 * Cin, 0);
    if (g.splitk > 1) CHK(test_slab(gemm_slab_floats(g.M, Cout, 1, g.splitk), &g.slab));
  }
  return launch_gemm(g, (hipStream_t)st);
}
// y = conv3x3(upsample2x(x)) and its input gradient without the upsampled image (GemmP::up2); weff [Cout][16][Cin] and planar
// [4 B H W][Cout] bf16 are scratch of the caller (weff written by the forward, read by the dgrad)
int sdxl_op_upconv3x3_fwd(const void* x, const void* w, const void* bias, void* weff, void* planar, void* y, int B, int H, int W,
                          int Cin, int Cout, void* st) {
  const int Mp = 4 * (int)upconv_plane_rows(B, H, W);
  int splitk = gemm_pick_splitk_small(Mp, Cout, 4 * Cin, 0);
  float* slab = nullptr;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(Mp, Cout, 1, splitk), &slab));
  return launch_upconv3x3_fwd((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)weff, (bf16*)planar, (bf16*)y, B, H, W, Cin, Cout,
                              splitk, slab, (hipStream_t)st);
}
int sdxl_op_upconv3x3_dgrad(const void* dy, const void* weff, void* planar, void* dx, const void* addend, int B, int H, int W, int Cin,
                            int Cout, void* st) {
  int splitk = gemm_pick_splitk_small(B * H * W, Cin, 16 * Cout, 1);
  float* slab = nullptr;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(B * H * W, Cin, 1, splitk), &slab));
  return launch_upconv3x3_dgrad((const bf16*)dy, (const bf16*)weff, (bf16*)planar, (bf16*)dx, (const bf16*)addend, B, H, W, Cin, Cout, splitk,
                                slab, 0, (hipStream_t)st);
}
#ifdef SDXL_DIAG
// stride-2 3x3 convolution through the fast gather on the four phase planes of x (GemmP::up2 == 3); xplanar [4 * roundup(B*(H/2)*(W/2), 128)][Cin]
// bf16: written by _fwd, read by _wgrad
int sdxl_op_conv3x3_s2_fwd(const void* x, const void* w, const void* bias, void* xplanar, void* y, int B, int H, int W, int Cin, int Cout,
                           void* st) {
  return launch_conv3x3_s2_fwd((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)xplanar, (bf16*)y, B, H, W, Cin, Cout, (hipStream_t)st);
}
int sdxl_op_conv3x3_s2_wgrad(const void* dy, const void* xplanar, float* dw, float* dbias, int accumulate, int B, int H, int W, int Cin,
                             int Cout, int splitk, void* st) {
  float* slab = nullptr;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(Cout, Cin, 9, splitk), &slab));
  return launch_conv3x3_s2_wgrad((const bf16*)dy, (const bf16*)xplanar, dw, dbias, nullptr, 1.f, accumulate, B, H, W, Cin, Cout, splitk, slab,
                                 (hipStream_t)st);
}
#endif
// input gradient of the stride-2 3x3 convolution by output phase (GemmP::up2 == 2); planar [4 * roundup(B*(H/2)*(W/2), 128)][Cin] scratch
int sdxl_op_conv3x3_s2_dgrad(const void* dy, const void* w, void* planar, void* dx, const void* addend, int B, int H, int W, int Cin,
                             int Cout, void* st) {
  return launch_conv3x3_s2_dgrad((const bf16*)dy, (const bf16*)w, (bf16*)planar, (bf16*)dx, (const bf16*)addend, B, H, W, Cin, Cout, 0,
                                 (hipStream_t)st);
}
// the weight / bias gradient of the same pair from `planar` as sdxl_op_upconv3x3_dgrad left it (the de-interleaved dy) and the
// low-resolution x; dweff [Cout][16][Cin] fp32 scratch; dw [Cout][9][Cin] (= or +=), dbias += (may be NULL)
int sdxl_op_upconv3x3_wgrad(const void* planar, const void* x, float* dweff, float* dw, float* dbias, int accumulate, int B, int H, int W,
                            int Cin, int Cout, int splitk, void* st) {
  float* slab = nullptr;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(Cout, Cin, 16, splitk), &slab));
  return launch_upconv3x3_wgrad((const bf16*)planar, (const bf16*)x, dweff, dw, dbias, nullptr, 1.f, accumulate, B, H, W, Cin, Cout, splitk,
                                slab, (hipStream_t)st);
}
int sdxl_op_conv3x3_dgrad(const void* dy, const void* w, void* dx, int B, int H, int W, int Cin, int Cout, int stride,
                          void* st) {
  int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NN;
  g.A = (const bf16*)dy; g.B = (const bf16*)w; g.C = dx;
  g.M = B * H * W; g.N = Cin; g.K = Cout;
  g.lda = Cout; g.ldb = 9L * Cin; g.ldc = Cin;
  g.taps = 9; g.Hm = H; g.Wm = W; g.Hs = Ho; g.Ws = Wo; g.sm = 1; g.sd = stride;
  g.flip = 1; g.b_tap_stride = Cin;
  if (stride == 1 && Cout % 64 == 0) {
    g.splitk = gemm_pick_splitk_small(g.M, Cin, 9 * Cout, 1);
    if (g.splitk > 1) CHK(test_slab(gemm_slab_floats(g.M, Cin, 1, g.splitk), &g.slab));
  }
  return launch_gemm(g, (hipStream_t)st);
}
int sdxl_op_conv3x3_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int stride,
                          int splitk, void* st) {
  int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_TN;
  g.A = (const bf16*)dy; g.B = (const bf16*)x; g.C = dw;
  g.M = Cout; g.N = Cin; g.K = B * Ho * Wo;
  g.lda = Cout; g.ldb = Cin; g.ldc = 9L * Cin;
  g.taps = 9; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.sm = stride; g.sd = 1;
  g.c_tap_stride = Cin;
  g.out_f32 = 1; g.splitk = splitk; g.accumulate = 1;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(Cout, Cin, 9, splitk), &g.slab));
  return launch_gemm(g, (hipStream_t)st);
}

int sdxl_op_conv3x3_wgrad2(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout, int stride,
                           int splitk, int accumulate, void* st) {
  int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_TN;
  g.A = (const bf16*)dy; g.B = (const bf16*)x; g.C = dw;
  g.M = Cout; g.N = Cin; g.K = B * Ho * Wo;
  g.lda = Cout; g.ldb = Cin; g.ldc = 9L * Cin;
  g.taps = 9; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.sm = stride; g.sd = 1;
  g.c_tap_stride = Cin;
  g.out_f32 = 1; g.accumulate = accumulate; g.bias_grad = dbias;
  // splitk <= 0: the plan's choice (the three-taps-per-workgroup kernel has its own)
  if (splitk <= 0) splitk = conv_wgrad3_policy(Cout, Cin, g.K, Wo, stride) ? conv_wgrad3_pick_splitk(Cout, Cin, g.K) : gemm_pick_splitk(Cout, Cin, 9, g.K);
  g.splitk = splitk;
  if (splitk > 1) CHK(test_slab(gemm_slab_floats(Cout, Cin, 9, splitk), &g.slab));
  return launch_gemm(g, (hipStream_t)st);
}

int sdxl_op_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int heads, int Nq,
                          int Nk, long ldq, long ldk, long ldv, long ldo, void* st) {
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (bf16*)o; a.LSE = lse;
  a.B = B; a.H = heads; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  return launch_attn_fwd(a, (hipStream_t)st);
}
int sdxl_op_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                          float* delta, void* dq, void* dk, void* dv, int B, int heads, int Nq, int Nk, long ldq,
                          long ldk, long ldv, long ldo, void* st) {
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (bf16*)o; a.LSE = (float*)lse;
  a.B = B; a.H = heads; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.dO = (const bf16*)d_o; a.lddo = ldo; a.Delta = delta;
  a.dQ = (bf16*)dq; a.dK = (bf16*)dk; a.dV = (bf16*)dv; a.lddq = ldq; a.lddk = ldk; a.lddv = ldv;
  a.qsplit = attn_pick_qsplit(B, heads, Nq, Nk);
  if (a.qsplit > 1) CHK(test_slab(attn_part_floats(B, heads, Nk, a.qsplit), &a.part));
  return launch_attn_bwd(a, (hipStream_t)st);
}

int sdxl_op_groupnorm_fwd(const void* x, void* y, const void* gamma, const void* beta, float* stats, float* ws, int B,
                          int HW, int C, int G, float eps, int silu, void* st) {
  return launch_groupnorm_fwd((const bf16*)x, (bf16*)y, (const bf16*)gamma, (const bf16*)beta, stats, ws, B, HW, C, G, eps,
                              silu, (hipStream_t)st);
}
int sdxl_op_groupnorm_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* stats, void* dx,
                          float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G, int silu, int accumulate,
                          void* st) {
  return launch_groupnorm_bwd((const bf16*)x, (const bf16*)dy, (const bf16*)gamma, (const bf16*)beta, stats, (bf16*)dx,
                              accumulate ? (const bf16*)dx : nullptr, dgamma, dbeta, ws, B, HW, C, G, silu, (hipStream_t)st);
}
int sdxl_op_layernorm_fwd(const void* x, void* y, const void* gamma, const void* beta, float* stats, int M, int C,
                          float eps, void* st) {
  return launch_layernorm_fwd((const bf16*)x, (bf16*)y, (const bf16*)gamma, (const bf16*)beta, stats, M, C, eps,
                              (hipStream_t)st);
}
int sdxl_op_layernorm_bwd(const void* x, const void* dy, const void* gamma, const float* stats, void* dx, float* dgamma,
                          float* dbeta, int M, int C, int accumulate, void* st) {
  if (!dgamma)
    return launch_layernorm_bwd((const bf16*)x, (const bf16*)dy, (const bf16*)gamma, stats, (bf16*)dx,
                                accumulate ? (const bf16*)dx : nullptr, nullptr, nullptr, M, C, (hipStream_t)st);
  ARG_CHECK(dbeta, "layernorm bwd: dgamma and dbeta go together");
  if (KNOB(10) == 1 || KNOB(10) == 3) {     // (A/B runs) lean dx kernel + parameter gradients as a pass of their own; the plan's default is the one-pass form below
    CHK(launch_layernorm_bwd((const bf16*)x, (const bf16*)dy, (const bf16*)gamma, stats, (bf16*)dx,
                             accumulate ? (const bf16*)dx : nullptr, nullptr, nullptr, M, C, (hipStream_t)st));
    if (KNOB(10) == 3) return launch_layernorm_param_grads((const bf16*)x, (const bf16*)dy, stats, dgamma, dbeta, M, C, (hipStream_t)st);
    LnRedBatch b;
    b.n = 1;
    float* part;
    CHK(test_slab(layernorm_bwd_part_floats(M, C), &part));
    b.e[0].part = part; b.e[0].dgamma = dgamma; b.e[0].dbeta = dbeta; b.e[0].C = C; b.e[0].nblk = layernorm_param_partial_rows(M, C);
    CHK(launch_layernorm_param_partials((const bf16*)x, (const bf16*)dy, stats, part, M, C, (hipStream_t)st));
    return launch_ln_param_reduce(b, (hipStream_t)st);
  }
  LnRedBatch b;
  b.n = 1;
  float* part;
  CHK(test_slab(layernorm_bwd_part_floats(M, C), &part));
  b.e[0].part = part; b.e[0].dgamma = dgamma; b.e[0].dbeta = dbeta; b.e[0].C = C;
  CHK(launch_layernorm_bwd((const bf16*)x, (const bf16*)dy, (const bf16*)gamma, stats, (bf16*)dx,
                           accumulate ? (const bf16*)dx : nullptr, part, &b.e[0].nblk, M, C, (hipStream_t)st));
  return launch_ln_param_reduce(b, (hipStream_t)st);
}
int sdxl_op_ff_geglu_fwd(const void* x, const void* w1, const void* b1, void* u, void* g, int M, int K, int C4, int group,
                         void* st) {
  GemmP p;
  gemm_defaults(&p);
  p.form = GEMM_NT;
  p.A = (const bf16*)x; p.B = (const bf16*)w1; p.C = u;
  p.M = M; p.N = 2 * C4; p.K = K;
  p.lda = K; p.ldb = K; p.ldc = 2L * C4;
  p.bias = (const bf16*)b1;
  p.geglu = 1; p.geglu_group = group; p.aux = (bf16*)g; p.ldaux = C4;
  return launch_gemm(p, (hipStream_t)st);
}
int sdxl_op_ff_geglu_bwd(const void* dy, const void* w2, const void* u, void* du, int M, int C, int C4, int group,
                         void* st) {
  GemmP p;
  gemm_defaults(&p);
  p.form = GEMM_NN;
  p.A = (const bf16*)dy; p.B = (const bf16*)w2; p.C = du;
  p.M = M; p.N = C4; p.K = C;
  p.lda = C; p.ldb = C4; p.ldc = 2L * C4;
  p.geglu = 2; p.geglu_group = group; p.aux = (bf16*)u; p.ldaux = 2L * C4;
  return launch_gemm(p, (hipStream_t)st);
}

int sdxl_op_loss(const sdxl_loss_config* lc, const sdxl_batch* b, void* unet_in, const void* pred, void* dpred,
                 float grad_scale, float* out8, int phase, void* st) {
  ARG_CHECK(lc && b, "null argument");
  LossP L;
  memset(&L, 0, sizeof(L));
  L.method = lc->method; L.prediction_type = lc->prediction_type; L.use_min_snr = lc->use_min_snr;
  L.min_snr_gamma = lc->min_snr_gamma; L.use_ztsnr = lc->use_ztsnr;
  L.B = b->B; L.HW = b->H * b->W; L.C = 4;
  L.latents = b->latents; L.noise = b->noise; L.sigma = b->sigma_or_t; L.tag_w = b->tag_weights;
  L.unet_in = (bf16*)unet_in; L.pred = (const bf16*)pred; L.dpred = (bf16*)dpred;
  L.grad_scale = grad_scale; L.out = out8;
  if (phase == 1) CHK(test_slab(loss_part_floats(L.B, L.HW), &L.part));
  if (phase == 0) return launch_loss_prepare(L, (hipStream_t)st);
  if (phase == 1) return launch_loss_fwd(L, (hipStream_t)st);
  if (phase == 2) return launch_loss_bwd(L, (hipStream_t)st);
  ARG_CHECK(false, "phase %d", phase);
}

// ---- row f1: fused AdamW_BF16 ----
static float bf16_round_host(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  u &= 0xFFFF0000u;
  memcpy(&x, &u, 4);
  return x;
}
int sdxl_adamw_default_config(sdxl_adamw_config* c) {
  ARG_CHECK(c, "null config");
  memset(c, 0, sizeof(*c));
  c->lr = 1e-4; c->beta1 = 0.9; c->beta2 = 0.999; c->eps = 1e-8;   // AdamWBF16.__init__ defaults (:31-36)
  c->step = 1.0;
  c->reference_ema = 1;
  return 0;
}
int sdxl_adamw_bf16_step(void* p, const void* grad, int grad_dtype, void* m, void* v, void* shift, size_t n,
                         const sdxl_adamw_config* c, const float* grad_scale_dev, const unsigned short* rand_inject,
                         void* st) {
  ARG_CHECK(c, "null config");
  ARG_CHECK(grad_dtype == 0 || grad_dtype == 1, "adamw: grad_dtype %d (0 = fp32, 1 = bf16)", grad_dtype);
  ARG_CHECK(c->step >= 1.0 && c->beta1 >= 0.0 && c->beta1 < 1.0 && c->beta2 >= 0.0 && c->beta2 < 1.0 && c->eps >= 0.0,
            "adamw: invalid hyper-parameters");
  AdamWP q;
  memset(&q, 0, sizeof(q));
  q.p = (bf16*)p; q.m = (bf16*)m; q.v = (bf16*)v; q.shift = (bf16*)shift; q.n = n;
  if (grad_dtype == 0) q.grad_f32 = (const float*)grad; else q.grad_bf16 = (const bf16*)grad;
  // scalars exactly as the reference's python floats reach the torch kernels: double arithmetic, then float32
  const double b1 = c->beta1, b2 = c->beta2;
  q.beta1 = (float)b1; q.beta2 = (float)b2;
  q.one_minus_beta1 = (float)(1.0 - b1);
  q.one_minus_beta2 = (float)(1.0 - b2);
  q.eps_bf16 = bf16_round_host((float)c->eps);
  q.value = (float)(-c->lr * sqrt(1.0 - pow(b2, c->step)));
  q.decay_alpha_bf16 = c->decay_this_iteration > 0.0 ? bf16_round_host((float)-c->decay_this_iteration) : 0.f;
  q.reference_ema = c->reference_ema;
  q.grad_round_bf16 = c->grad_round_bf16;
  q.grad_scale = grad_scale_dev;
  q.rand = rand_inject;
  q.seed_lo = (unsigned)c->seed; q.seed_hi = (unsigned)(c->seed >> 32);
  q.step_counter = (unsigned)c->step;
  q.elem_offset = (size_t)c->elem_offset;
  return launch_adamw_bf16(q, (hipStream_t)st);
}
int sdxl_adamw_decay(void* shift, const void* p, size_t n, float decay, void* st) {
  ARG_CHECK(shift && p, "adamw decay: missing buffers");
  if (decay <= 0.f) return 0;
  return launch_adamw_decay((bf16*)shift, (const bf16*)p, n, bf16_round_host(-decay), (hipStream_t)st);
}

int sdxl_probe_layout(void* out, void* st) { return probe_layout(out, (hipStream_t)st); }
int sdxl_op_exchange_shadow(void* buf, size_t bytes, int workgroups, int lds_bytes, float busy_us, void* st) {
  return launch_exchange_shadow(buf, bytes, workgroups, lds_bytes, busy_us, (hipStream_t)st);
}
// test hook (include/sdxlstep_diag.h part 1): sdxl_op_gemm's NT / NN forms with explicit leading dimensions (padded activations, as the plan's
// feed-forward hidden tensors are) and a forced configuration for this launch only (0: the policy)
int sdxl_op_gemm_ld(int form, const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, const void* bias,
                    const void* resid, long ldr, int cfg, void* st) {
  ARG_CHECK(form == GEMM_NT || form == GEMM_NN, "gemm_ld: NT / NN only (form %d)", form);
  GemmP g;
  gemm_defaults(&g);
  g.form = form;
  g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.bias = (const bf16*)bias;
  if (resid) { g.resid = (const bf16*)resid; g.ldr = ldr; }
  g.cfg = cfg;
  return launch_gemm(g, (hipStream_t)st);
}
// test hook (include/sdxlstep_diag.h part 1): the NN dgrad with the Delta epilogue, as LinearOp::bwd launches it for a self-attention
// layer's out-projection
int sdxl_op_linear_dgrad_delta(const void* dy, const void* w, const void* o, const void* addend, void* d_o, float* delta, int B, int Nq,
                               int N, int K, void* st) {
  ARG_CHECK(dy && w && o && d_o && delta && B > 0 && Nq > 0 && N % 128 == 0 && K % 64 == 0, "linear_dgrad_delta: B=%d Nq=%d N=%d K=%d", B, Nq, N, K);
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NN;
  g.A = (const bf16*)dy; g.B = (const bf16*)w; g.C = d_o;
  g.M = B * Nq; g.N = N; g.K = K;
  g.lda = K; g.ldb = N; g.ldc = N;
  if (addend) { g.resid = (const bf16*)addend; g.ldr = N; }
  g.delta_o = (const bf16*)o; g.delta_ldo = N;
  g.delta_out = delta;
  g.delta_nq = Nq; g.delta_heads = N / 64;
  g.cfg = 1;
  return launch_gemm(g, (hipStream_t)st);
}
int sdxl_profile_gemm_begin(void) { return gemm_profile_begin(); }
int sdxl_set_gemm_mode(int mode) {
  const int cfg = mode >> 2;
  ARG_CHECK(mode >= 0 && (mode & 3) <= 2 && (cfg == 0 || cfg == 1 || cfg == 2 || cfg == 3 || cfg == 5 || cfg == 6 || cfg == 7 || cfg == 8 || cfg == 13 || cfg == 23 || cfg == 43 || cfg == 31 || cfg == 32 || cfg == 33 || cfg == 34 || cfg == 35 || cfg == 36), "gemm mode %d", mode);
  gemm_set_mode(mode);
  return 0;
}
int sdxl_profile_gemm_end(double* flops, double* ms, int* launches) { return gemm_profile_end(flops, ms, launches); }
#ifdef SDXL_DIAG     // ---- experiment ABI of the diagnostics build (include/sdxlstep_diag.h): not in the product library ----
// dY [M][Kr] x W [Kr][N] (the NN dgrad form) with the LayerNorm backward of GemmP::ln_x in the epilogue: dx = LN_bwd(dY W | x, stats, gamma)
// (+ addend), dy_out (or null) = dY W, pcol (or null) = [cdiv(M, 128)][2][N] dgamma | dbeta partial sums.  Scratch is allocated, zeroed and
// freed per call (test hook; the plan shares its own); N <= 1280.
int sdxl_op_linear_dgrad_ln_bwd(const void* dy, const void* w, const void* x, const float* stats, const void* gamma, const void* addend,
                                void* dx, void* dy_out, float* pcol, int M, int N, int Kr, void* st) {
  ARG_CHECK(gemm_ln_cfg(M, N, Kr) != 0, "linear_dgrad_ln_bwd: M=%d N=%d K=%d does not fit the fused epilogue", M, N, Kr);
  float* part = nullptr;
  const size_t pbytes = gemm_ln_part_floats(M, N) * sizeof(float);
  HIP_CHECK_RET(hipMalloc((void**)&part, pbytes));
  if (hipError_t e = hipMemsetAsync(part, 0, pbytes, (hipStream_t)st)) {      // (every exit path frees the scratch)
    (void)hipFree(part);
    sdxl_set_error("linear_dgrad_ln_bwd: hipMemsetAsync -> %s", hipGetErrorString(e));
    return 2;
  }
  GemmP g;
  gemm_defaults(&g);
  g.form = GEMM_NN;
  g.A = (const bf16*)dy; g.B = (const bf16*)w; g.C = dy_out;
  g.M = M; g.N = N; g.K = Kr;
  g.lda = Kr; g.ldb = N; g.ldc = N;
  g.ln_x = (const bf16*)x; g.ln_ldx = N; g.ln_stats = stats; g.ln_gamma = (const bf16*)gamma;
  g.ln_dx = (bf16*)dx; g.ln_addend = (const bf16*)addend; g.ln_ldo = N;
  g.ln_part = part; g.ln_pcol = pcol;
  int rc = 0;
  for (int epoch = 1; epoch <= 3 && rc == 0; ++epoch) {      // three launches on the same scratch: later ones find the earlier epochs' granules
    g.ln_epoch = epoch;
    rc = launch_gemm(g, (hipStream_t)st);
  }
  const hipError_t se = hipStreamSynchronize((hipStream_t)st);
  (void)hipFree(part);
  if (se != hipSuccess) { sdxl_set_error("linear_dgrad_ln_bwd: hipStreamSynchronize -> %s", hipGetErrorString(se)); return 2; }
  return rc;
}
// Knobs are process-global and read at plan-build, forward and backward time: set them BEFORE sdxl_plan / the first step of a handle and
// leave them alone afterwards (A/B runs restart the process per setting, profiles/tools/ab.sh).
int sdxl_set_knob(int id, int value) {
  ARG_CHECK(id >= 0 && id < SDXL_NKNOBS, "knob %d out of range", id);
  g_knobs[id] = value;
  if (id == 9) wgrad256_set_enabled(value == 0);         // knob 9 = 1: long-reduction linear weight gradients on the 128 x 160 kernel
  if (id == 12) conv_wgrad3_set_enabled(value == 0);    // knob 12 = 1: 3x3 weight gradients on the one-tap-per-workgroup kernel only
  if (id == 15) gemm256_set_tail(value == 0);     // knob 15 = 1: no half-height tail workgroups in the 256 x 256 kernel
  return 0;
}
int sdxl_set_sk_mode(int mode, int workers) {
  ARG_CHECK(mode >= 0 && mode <= 2 && workers >= 0 && workers <= 256, "stream-K mode %d / workers %d", mode, workers);
  gemm_set_sk_mode(mode);
  gemm_sk_set_workers(workers);
  return 0;
}
int sdxl_op_pl_prefetch_b(int form, const void* B, int M, int N, int K, long ldb, int parts, void* st) {
  GemmP p;
  gemm_defaults(&p);
  p.form = form; p.B = (const bf16*)B; p.M = M; p.N = N; p.K = K; p.ldb = ldb;
  return launch_pl_prefetch_b(p, parts, (hipStream_t)st);
}
int sdxl_ln_error(unsigned* out) {
  ARG_CHECK(out != nullptr, "ln_error: null output");
  return gemm_ln_error(out);
}
int sdxl_sk_error(void* st, unsigned* out) {
  ARG_CHECK(out, "null output");
  return gemm_sk_error((hipStream_t)st, out);
}
int sdxl_op_gemm_sk(int n, const int* form, const void* const* A, const void* const* B, void* const* C, const int* M, const int* N,
                    const int* K, const void* const* bias, const void* const* resid, const int* accumulate, void* st) {
  ARG_CHECK(n >= 1 && n <= 4 && form && A && B && C && M && N && K, "gemm_sk: bad arguments");
  GemmP g[4];
  for (int i = 0; i < n; ++i) {
    gemm_defaults(&g[i]);
    g[i].form = form[i];
    g[i].A = (const bf16*)A[i]; g[i].B = (const bf16*)B[i]; g[i].C = C[i];
    g[i].M = M[i]; g[i].N = N[i]; g[i].K = K[i];
    if (form[i] == GEMM_NT) { g[i].lda = K[i]; g[i].ldb = K[i]; }
    else if (form[i] == GEMM_NN) { g[i].lda = K[i]; g[i].ldb = N[i]; }
    else { g[i].lda = M[i]; g[i].ldb = N[i]; g[i].out_f32 = 1; }
    g[i].ldc = N[i];
    if (form[i] == GEMM_TN) g[i].bias_grad = bias ? (float*)bias[i] : nullptr;
    else g[i].bias = bias ? (const bf16*)bias[i] : nullptr;
    if (resid && resid[i]) { g[i].resid = (const bf16*)resid[i]; g[i].ldr = N[i]; }
    g[i].accumulate = accumulate ? accumulate[i] : 0;
  }
  return launch_gemm_multi(g, n, (hipStream_t)st);
}
#endif   // SDXL_DIAG

// debug: order-independent checksum (sum of raw 16-bit patterns) of every activation of the current plan, in
// creation order.  Synchronises.  Used to localise run-to-run differences.
__global__ void checksum_kernel(const unsigned short* __restrict__ x, long n, unsigned long long* out) {
  unsigned long long s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += (unsigned long long)x[i] * (unsigned long long)((i % 251) + 1);
  atomicAdd(out, s);
}
int sdxl_debug_act_checksums(sdxl_handle* h, unsigned long long* out_host, int cap, int* n_out, int grads) {
  H_CHECK(h);
  CHK(ready(h->e));
  Plan& p = *h->e.cur;
  int n = (int)p.acts.size();
  if (n_out) *n_out = n;
  if (n > cap) n = cap;
  unsigned long long* d = nullptr;
  HIP_CHECK_RET(hipMalloc((void**)&d, sizeof(unsigned long long) * (n > 0 ? n : 1)));
  HIP_CHECK_RET(hipMemset(d, 0, sizeof(unsigned long long) * (n > 0 ? n : 1)));
  for (int i = 0; i < n; ++i) {
    Act* a = p.acts[i].get();
    const bf16* src = grads ? p.G(a) : p.P(a);
    if (!src || a->parent) continue;     // (column-slice views are covered by their parent)
    long cnt = a->rows * a->cols;
    hipLaunchKernelGGL(checksum_kernel, dim3(64), dim3(256), 0, 0, (const unsigned short*)src, cnt, d + i);
  }
  HIP_CHECK_RET(hipDeviceSynchronize());
  HIP_CHECK_RET(hipMemcpy(out_host, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return 0;
}

}  // extern "C"
