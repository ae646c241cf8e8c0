// UNet plan builder, op implementations (forward + backward), parameter import/export.
// Architecture = public SDXL-base UNet2DConditionModel semantics; the reference
// reaches it through diffusers at ddpm_trainer.py:320-325 and flow_matching_trainer.py:400-405.
#include "engine.h"

#include <stdarg.h>
#include <stdlib.h>

#ifdef SDXL_DIAG
int g_knobs[SDXL_NKNOBS] = {0};
#endif
static thread_local char g_err[1024] = "";
void sdxl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* sdxl_get_error() { return g_err; }

#define CHK(x)            \
  do {                    \
    int _r = (x);         \
    if (_r) return _r;    \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ================================================================================================
// Plan memory
// ================================================================================================
size_t Plan::alloc(size_t bytes) {
  size_t off = cursor;
  cursor = align_up(cursor + bytes, 256);
  ws_bytes = cursor;
  return off;
}
Act* Plan::new_act(long rows, int cols, bool need_grad, int pad_rows) {
  Act* a = new Act();
  a->rows = rows;
  a->cols = cols;
  a->pad_rows = pad_rows;
  a->need_grad = need_grad;
  a->off = alloc((size_t)(rows + pad_rows) * cols * sizeof(bf16));
  acts.emplace_back(a);
  return a;
}
Act* Plan::view(Act* parent, int col0, int cols) {
  Act* a = new Act();
  a->rows = parent->rows;
  a->cols = cols;
  a->need_grad = parent->need_grad;
  a->parent = parent;
  a->col0 = col0;
  a->off = parent->off + (size_t)col0 * sizeof(bf16);
  acts.emplace_back(a);
  return a;
}
Plan::GradDst Plan::grad_dst(Act* a) {
  GradDst d;
  if (a->parent) {   // slice of the parent's gradient tensor (allocated by the first slice that asks): written once, by one op
    if (a->parent->goff == NONE) a->parent->goff = alloc((size_t)(a->parent->rows + a->parent->pad_rows) * a->parent->cols * sizeof(bf16));
    d.addend = a->goff;
    a->goff = a->parent->goff + (size_t)a->col0 * sizeof(bf16);
    d.out = a->goff;
    return d;
  }
  d.addend = a->goff;
  a->goff = alloc((size_t)a->rows * a->cols * sizeof(bf16));
  d.out = a->goff;
  return d;
}
hipEvent_t Engine::next_event() {
  if (ev_used == ev_pool.size()) {
    hipEvent_t e;
    // device-to-device ordering only (the host never waits on these): no system-scope fence at the marker (-0.7 ms per step)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
    ev_pool.push_back(e);
  }
  return ev_pool[ev_used++];
}
// run `fn(stream)` for the weight-gradient work of one op: on the side stream once `main` has reached this point.
// (Batching several ops behind one event record was measured: every step of delay for the side stream lengthens the
// segment-end join -- 152 / 155 / 158 / 163 / 171 ms per step for batches of 1 / 2 / 4 / 8 / 16 -- so each op gets its own.)
template <class F>
static int on_side(Plan& p, hipStream_t main, F&& fn) {
  Engine& e = *p.eng;
  if (!e.use_side || !e.side || gemm_profiling()) return fn(main);
  // (Timing experiment, r03: WITHOUT the fork event -- wrong results, the side stream free to run ahead -- the step takes 124.8 ms
  //  instead of 116.8: the event is also what pairs each weight-gradient GEMM with the dgrad of its own layer on the CUs.)
  hipEvent_t ev = e.next_event();
  if (!ev) { sdxl_set_error("hipEventCreate failed"); return 2; }
  HIP_CHECK_RET(hipEventRecord(ev, main));
  HIP_CHECK_RET(hipStreamWaitEvent(e.side, ev, 0));
  e.side_dirty = true;
  if (!e.side_leaves.empty()) {              // queued leaves: everything they read was enqueued on `main` before this event
    std::vector<std::function<int(hipStream_t)>> v;
    v.swap(e.side_leaves);
    for (auto& leaf : v) CHK(leaf(e.side));
  }
  return fn(e.side);
}
// queue a leaf for the side stream (runs behind the next fork event; immediately, on `main`, in the single-stream modes)
template <class F>
static int side_leaf(Plan& p, hipStream_t main, F&& fn) {
  Engine& e = *p.eng;
  if (!e.use_side || !e.side || gemm_profiling() || KNOB(2) == 16) return on_side(p, main, fn);     // (knob 2 = 16: own event each, A/B runs)
  e.side_leaves.emplace_back(std::function<int(hipStream_t)>(fn));
  return 0;
}
// weight gradients of one shape wait here until `grp` of them fill a launch (GemmP::group); flush_wgrads() at the end of a
// backward segment launches the stragglers (their dY / X operands are write-once buffers of the plan, nothing reuses them)
static int launch_wgrad_bucket(Plan& p, hipStream_t main, std::vector<GemmP>& v) {
  if (v.empty()) return 0;
  GemmP g = v[0];
  if (v.size() > 1) {
    g.splitk = 1;
    g.group = (int)v.size();
    for (size_t i = 0; i < v.size(); ++i) {
      g.gA[i] = v[i].A; g.gB[i] = v[i].B; g.gC[i] = (float*)v[i].C; g.gbias_grad[i] = v[i].bias_grad; g.gCb[i] = v[i].Cb;
    }
  }
  v.clear();
  if (g.splitk > 1) g.anyorder = 0;      // (a straggler that keeps its split: the shared slab orders it)
  return on_side(p, main, [&g](hipStream_t s2) -> int { return launch_gemm(g, s2); });
}
int Engine::defer_wgrad(Plan& p, hipStream_t main, const GemmP& g, int grp) {
  for (auto& b : wg_pending) {
    if (b.empty() || (b[0].M == g.M && b[0].N == g.N && b[0].K == g.K && b[0].accumulate == g.accumulate)) {
      b.push_back(g);
      if ((int)b.size() >= grp) return launch_wgrad_bucket(p, main, b);
      return 0;
    }
  }
  wg_pending.emplace_back(1, g);
  if (grp <= 1) return launch_wgrad_bucket(p, main, wg_pending.back());
  return 0;
}
int Engine::flush_wgrads(Plan& p, hipStream_t main) {
  if (!side_leaves.empty()) CHK(on_side(p, main, [](hipStream_t) -> int { return 0; }));     // the stragglers, behind one event
  for (auto& b : wg_pending) CHK(launch_wgrad_bucket(p, main, b));
  return 0;
}
// dgamma / dbeta of the LayerNorms since the last call: one reduce launch per LN_RED_MAX of them, on the side stream
int Engine::flush_ln_params(Plan& p, hipStream_t main) {
  size_t i = 0;
  while (i < ln_pending.size()) {
    LnRedBatch b;
    b.n = 0;
    while (i < ln_pending.size() && b.n < LN_RED_MAX) b.e[b.n++] = ln_pending[i++];
    CHK(on_side(p, main, [&b](hipStream_t s2) -> int { return launch_ln_param_reduce(b, s2); }));
  }
  ln_pending.clear();
  return 0;
}
bool Plan::grad_alias(Act* x, Act* y) {
  if (x->goff == NONE) {
    x->goff = y->goff;
    return true;
  }
  return false;
}

// ================================================================================================
// Ops
// ================================================================================================
static int pick_splitk(long out_rows, long out_cols, int taps, long red) { return gemm_pick_splitk((int)out_rows, (int)out_cols, taps, red); }
static void want_slab_main(Plan& p, int M, int N, int splitk) {     // slabs of the caller's-stream launches (forward, dgrad)
  size_t need = gemm_slab_floats(M, N, 1, splitk);
  if (need > p.slab_main_floats) p.slab_main_floats = need;
}
static void want_slab(Plan& p, int M, int N, int taps, int splitk) {
  size_t need = gemm_slab_floats(M, N, taps, splitk);
  if (need > p.slab_floats) p.slab_floats = need;
}

static int kv_pad_rows(int B, int ctx) { return (int)((64 - ((long)B * ctx) % 64) % 64); }   // B * 77 prompt tokens -> multiple of 64

struct AttnOp;
static void attn_delta_target(AttnOp* a, Plan& p, GemmP& g);      // (defined behind AttnOp)
static bool attn_delta_wanted(const AttnOp* a);
static void attn_set_delta_fused(AttnOp* a, bool on);
struct LayerNormOp;
static bool ln_fuse_plan(LayerNormOp* ln, Plan& p, int mode);      // (defined behind LayerNormOp)
static void ln_fuse_target(LayerNormOp* ln, Plan& p, GemmP& g, int mode);
struct LinearOp : Op {
  Act *x, *y, *resid;
  PRef w, b;
  int K, N;
  int resid_alias = 0, splitk = 1, wgroup = 1;
  struct AttnOp* delta_attn = nullptr;      // x is the output of this self-attention layer: the dgrad's epilogue also writes its Delta (GemmP::delta_out)
  bool delta_on = false;                    // ... decided in plan_bwd (the same predicate tells the attention backward that Delta is ready)
  // x is the output of this LayerNorm and feeds nothing else: the dgrad's epilogue runs the LayerNorm's backward (GemmP::ln_x) -- dx of the
  // LayerNorm straight from the accumulators (the knock-out of the separate dx pass: -4.6 ms per step), its dgamma | dbeta partial sums too.
  // ln_mode (plan_bwd): 0 not fused (the shipped plan; shape / split-K), 1 fused (knob 26 = 1, diagnostics build), 2 fused but dy is still
  // stored and the parameter gradients take their own pass over it (knob 26 = 2).  Measured: -0.35 ms per step, not shipped.
  struct LayerNormOp* ln_src = nullptr;
  int ln_mode = 0;
  int crcfg = 0, crsplit = 1;   // weight gradient on the co-resident 256-row kernel (gemm_cr256.hip): GemmP::cfg 31 / 32, its split-K factor
  int fsplit = 1, dsplit = 1;   // split-K of the forward / dgrad launch (small-M problems, gemm_pick_splitk_small)
  size_t dy32_off = NONE;   // the output gradient arrives as fp32 sums (grouped time-embedding projection): cast first
  size_t dy_off = NONE;
  Plan::GradDst dx, dres;
  // GEGLU feed-forward pair (GemmP::geglu): the first projection (y = u, interleaved value | gate columns) also
  // writes gact = value * gelu(gate); the second projection (x = gact) turns its input gradient straight into dU
  int ggroup = 64;       // packing group of the GEGLU projection (64: 128-column tiles, 80: 160-column tiles)
  Act* gact = nullptr;   // first projection: fused activation output [rows][N/2]
  Act* gu = nullptr;     // second projection: the first projection's pre-activation u [rows][2K]
  LinearOp(Act* x_, Act* y_, PRef w_, PRef b_, int K_, int N_, Act* resid_) : x(x_), y(y_), resid(resid_), w(w_), b(b_), K(K_), N(N_) {}
  int fwd(Plan& p, hipStream_t st) override {
    GemmP g;
    gemm_defaults(&g);
    g.form = GEMM_NT;
    g.A = p.P(x); g.B = p.eng->Wp(w); g.C = p.P(y);
    g.M = (int)x->rows; g.N = N; g.K = K;
    g.lda = x->ld(); g.ldb = K; g.ldc = y->ld();
    g.bias = b.off == NONE ? nullptr : p.eng->Wp(b);
    if (resid) { g.resid = p.P(resid); g.ldr = resid->ld(); }
    if (gact) { g.geglu = 1; g.geglu_group = ggroup; g.aux = p.P(gact); g.ldaux = gact->ld(); }
    if (fsplit > 1 && !hoist_fwd) { g.splitk = fsplit; g.slab = p.F(p.slab_main_off); }     // (the hoisted projection runs on the side stream)
    return launch_gemm(g, st);
  }
  void plan_bwd(Plan& p) override {
    dy_off = y->goff;
    if (resid) {
      resid_alias = p.grad_alias(resid, y) ? 1 : 0;
      if (!resid_alias) dres = p.grad_dst(resid);
    }
    if (gu) dx = p.grad_dst(gu);           // the activation itself gets no gradient buffer: dgrad emits dU
    else if (x->need_grad) dx = p.grad_dst(x);
    splitk = pick_splitk(N, K, 1, x->rows);
    if (wgrad256_policy(N, K, x->rows)) splitk = wgrad256_pick_splitk(N, K, x->rows);
    wgroup = gemm_pick_group(N, K, 1, x->rows, splitk);
    want_slab(p, N, K, 1, splitk);
    crcfg = cr256_wgrad_cfg(N, K, x->rows, b.off != NONE);
    if (crcfg) { crsplit = cr256_pick_splitk(N, K, x->rows, crcfg); want_slab(p, N, K, 1, crsplit); }
    if (!gact) { fsplit = gemm_pick_splitk_small((int)x->rows, N, K, 3); want_slab_main(p, (int)x->rows, N, fsplit); }
    if (delta_attn && !(!gu && x->need_grad)) attn_set_delta_fused(delta_attn, false);      // no dgrad launch: nobody would write Delta
    if (!gu && x->need_grad) {
      dsplit = gemm_pick_splitk_small((int)x->rows, K, N, 2);
      // knob 3 (experiment): split the reduction of the long-K dgrads (N >= knob 4) s ways although their tiles fill half the chip
      if (KNOB(3) > 1 && N >= (KNOB(4) > 0 ? KNOB(4) : 8192) && N % (64 * KNOB(3)) == 0 && dsplit == 1) dsplit = KNOB(3);
      want_slab_main(p, (int)x->rows, K, dsplit);
      // the Delta epilogue is decided HERE, once, for both sides: this dgrad attaches it iff the attention backward (planned after this
      // op: the plan walks the ops in reverse) is told that Delta is ready.  A split reduction (small batches: dsplit > 1) goes through the
      // slab epilogue, which has no Delta path -- the attention backward then runs its own Delta pass.
      if (delta_attn) {
        delta_on = attn_delta_wanted(delta_attn) && dsplit <= 1 && K % 128 == 0;
        attn_set_delta_fused(delta_attn, delta_on);
      }
      if (SDXL_LN_EPILOGUE && ln_src && KNOB(26) != 0 && dsplit == 1 && dx.addend == NONE && !x->parent && gemm_ln_cfg((int)x->rows, K, N) &&
          ln_fuse_plan(ln_src, p, KNOB(26) == 2 ? 2 : 1))
        ln_mode = KNOB(26) == 2 ? 2 : 1;
    }
  }
  int bwd(Plan& p, hipStream_t st, bool first) override {
    const bf16* dy = p.GP(dy_off);
    const int M = (int)x->rows;
    if (dy32_off != NONE) {
      auto& cs = p.eng->cs_pending;      // the convolutions' per-sample column sums: partial rows -> the fp32 buffer, LN_RED_MAX sums per launch
      for (size_t i = 0; i < cs.size();) {
        LnRedBatch rb;
        rb.n = 0;
        while (i < cs.size() && rb.n < LN_RED_MAX) rb.e[rb.n++] = cs[i++];
        CHK(launch_ln_param_reduce(rb, st));
      }
      cs.clear();
      CHK(launch_f32_to_bf16(p.F(dy32_off), p.GP(dy_off), (long)M * N, 1.f, st));
    }
    if (gu && dx.addend != NONE) { sdxl_set_error("geglu: pre-activation gradient has another writer"); return 3; }
    if (resid && !resid_alias) CHK(launch_add(p.GP(dres.addend), dy, p.GP(dres.out), (long)M * N, st));
    {
      GemmP g;
      gemm_defaults(&g);
      g.form = GEMM_TN;
      g.A = dy; g.B = p.P(x); g.C = p.eng->Gp(w);
      g.M = N; g.N = K; g.K = M;
      g.lda = y->ld(); g.ldb = x->ld(); g.ldc = K;
      g.out_f32 = 1;
      g.splitk = splitk;
      g.slab = p.F(p.slab_off);
      g.accumulate = first ? 0 : 1;
      g.bias_grad = b.off != NONE ? p.eng->Gp(b) : nullptr;   // column sums of dY ride along on the matrix pipe
      if (p.eng->emit_base) { g.Cb = p.eng->emit_base + w.off; g.cb_scale = p.eng->emit_scale; }
      if (crcfg) { g.cfg = crcfg; if (wgroup <= 1) g.splitk = crsplit; }     // (a grouped launch's stragglers keep the 128-row policy's split)
      // knob 29 (experiment): no in-stream barrier for weight gradients whose operands were written on the caller's stream (everything but the
      // prompt-side K | V projection, whose dY the side stream's own dK / dV kernels produce) and that use no shared split-K slab
      if (KNOB(29) == 1 && M >= 1024 && dy32_off == NONE && (g.splitk <= 1 || wgroup > 1)) g.anyorder = 1;
      const int pad = x->pad_rows < y->pad_rows ? x->pad_rows : y->pad_rows;
      if (pad > 0 && (M + pad) % 64 == 0) {     // zero rows appended to both operands: every reduction step is a full one
        g.K = M + pad;
        g.anyorder = 0;      // (behind its memsets)
        bf16* dyp = p.GP(dy_off) + (size_t)M * N;
        bf16* xp = p.P(x) + (size_t)M * K;
        const size_t nb_dy = (size_t)pad * N * sizeof(bf16), nb_x = (size_t)pad * K * sizeof(bf16);
        CHK(on_side(p, st, [=](hipStream_t s2) -> int {
          HIP_CHECK_RET(hipMemsetAsync(dyp, 0, nb_dy, s2));
          HIP_CHECK_RET(hipMemsetAsync(xp, 0, nb_x, s2));
          return launch_gemm(g, s2);
        }));
      } else
      if (wgroup > 1) CHK(p.eng->defer_wgrad(p, st, g, wgroup));     // (also in the serialized measurement modes: the launch structure that ships)
      else CHK(on_side(p, st, [&](hipStream_t s2) -> int { return launch_gemm(g, s2); }));
    }
    if (x->need_grad) {
      GemmP g;
      gemm_defaults(&g);
      g.form = GEMM_NN;
      g.A = dy; g.B = p.eng->Wp(w); g.C = p.GP(dx.out);
      g.M = M; g.N = K; g.K = N;
      g.lda = y->ld(); g.ldb = K; g.ldc = x->ld();
      if (gu) { g.geglu = 2; g.geglu_group = ggroup; g.aux = p.P(gu); g.ldaux = gu->ld(); g.ldc = gu->ld(); }
      else if (dx.addend != NONE) { g.resid = p.GP(dx.addend); g.ldr = x->ld(); }
      if (dsplit > 1) { g.splitk = dsplit; g.slab = p.F(p.slab_main_off); }
      g.prio = KNOB(0);
      if (KNOB(6) > 0 && !gu && (KNOB(8) <= 0 || N >= KNOB(8))) g.cfg = KNOB(6);     // experiment: configuration of the linear dgrads
      if (KNOB(22) > 0 && gu) g.cfg = KNOB(22);     // experiment: configuration of the GEGLU (FF2) dgrad
      // (last: the epilogue that writes Delta exists on the 128 x 128 tiles of the 4-wave kernel only)
      if (delta_on) { attn_delta_target(delta_attn, p, g); g.cfg = 1; }
      if (ln_mode) { ln_fuse_target(ln_src, p, g, ln_mode); g.cfg = 0; g.resid = nullptr; }
      CHK(launch_gemm(g, st));
    }
    return 0;
  }
};

struct ConvOp : Op {
  Act *x, *y, *resid, *rowvec;
  PRef w, b;
  int Bn, H, W, Cin, Cout, stride, Ho, Wo;
  int resid_alias = 0, splitk = 1;
  int fsplit = 1, dsplit = 1;   // split-K of the forward / dgrad launch (small images, gemm_pick_splitk_small)
  size_t rv32_off = NONE;  // rowvec gradient: this conv's column slice of the plan's fp32 [B][sum Cout] buffer
  long rv32_ld = 0;
  size_t cs_part_off = NONE;   // partial rows of the fixed-order column sums (time-embedding row vector; the up-sampler's bias gradient)
  size_t dy_off = NONE;
  Plan::GradDst dx, dres, drv;
  // x is the nearest-2x upsampling of x_low (the `Upsample2D` pair): forward and dgrad run on x_low with 2 x 2 phase stencils
  // (GemmP::up2, 16 instead of 36 tap-pixels per output quad); the upsampled image is still produced, on the side stream, for the
  // weight gradient only.  knob 2 = 64: the plain path (A/B runs).
  Act* x_low = nullptr;
  Plan::GradDst* up_dx = nullptr;       // the UpsampleOp's input-gradient destination: this op writes it
  size_t weff_off = NONE, planar_off = NONE, dweff_off = NONE;
  size_t s2x_off = NONE;           // stride 2: the four phase planes of x (forward and weight gradient through the fast gather, GemmP::up2 == 3)
  size_t s2_planar_off = NONE;     // stride 2: the input gradient by output phase (GemmP::up2 == 2; knob 2 = 256: the general gather)
  int up_fs = 1, up_ds = 1, up_ws = 1;
  bool up_wg = false;       // the weight gradient too (needs whole 64-pixel reduction steps): the upsampled image is then not produced at all
  bool up2() const { return x_low != nullptr && KNOB(2) != 64; }
  void enable_up2(Plan& p, Act* xl, Plan::GradDst* udx) {
    if (stride != 1 || Cin % 64 || Cout % 64 || resid || rowvec) return;
    x_low = xl; up_dx = udx;
    const long plane = upconv_plane_rows(Bn, H / 2, W / 2);
    weff_off = p.alloc(sizeof(bf16) * (size_t)Cout * 16 * Cin);
    planar_off = p.alloc(sizeof(bf16) * (size_t)4 * plane * (Cout > Cin ? Cout : Cin));
    up_fs = gemm_pick_splitk_small((int)(4 * plane), Cout, 4 * Cin, 0);
    want_slab_main(p, (int)(4 * plane), Cout, up_fs);
    up_ds = gemm_pick_splitk_small(Bn * (H / 2) * (W / 2), Cin, 16 * Cout, 1);
    want_slab_main(p, Bn * (H / 2) * (W / 2), Cin, up_ds);
    up_wg = (Bn * (H / 2) * (W / 2)) % 64 == 0;
    if (up_wg) {
      dweff_off = p.alloc(sizeof(float) * (size_t)Cout * 16 * Cin);
      up_ws = pick_splitk(Cout, Cin, 16, (long)Bn * (H / 2) * (W / 2));
      want_slab(p, Cout, Cin, 16, up_ws);
      cs_part_off = p.alloc(sizeof(float) * colsum_part_floats(1, Bn * H * W, Cout));      // (rowvec excludes up2: one buffer serves either)
    }
  }
  ConvOp(Act* x_, Act* y_, PRef w_, PRef b_, int B_, int H_, int W_, int Cin_, int Cout_, int stride_, Act* resid_,
         Act* rowvec_)
      : x(x_), y(y_), resid(resid_), rowvec(rowvec_), w(w_), b(b_), Bn(B_), H(H_), W(W_), Cin(Cin_), Cout(Cout_),
        stride(stride_) {
    Ho = (H - 1) / stride + 1;
    Wo = (W - 1) / stride + 1;
  }
  int fwd(Plan& p, hipStream_t st) override {
#ifdef SDXL_DIAG
    if (s2x_off != NONE && KNOB(2) == 512)      // (knob 2 = 512, experiment: measured neutral in the step -- 114.0-114.4 on, 113.8-114.3 off -- the general strided gather stays)
      return launch_conv3x3_s2_fwd(p.P(x), p.eng->Wp(w), p.eng->Wp(b), (bf16*)p.F(s2x_off), p.P(y), Bn, H, W, Cin, Cout, st);
#endif
    if (up2())
      return launch_upconv3x3_fwd(p.P(x_low), p.eng->Wp(w), p.eng->Wp(b), (bf16*)p.F(weff_off), (bf16*)p.F(planar_off), p.P(y), Bn, H / 2,
                                  W / 2, Cin, Cout, up_fs, p.F(p.slab_main_off), st);
    GemmP g;
    gemm_defaults(&g);
    g.form = GEMM_NT;
    g.A = p.P(x); g.B = p.eng->Wp(w); g.C = p.P(y);
    g.M = Bn * Ho * Wo; g.N = Cout; g.K = Cin;
    g.lda = Cin; g.ldb = 9L * Cin; g.ldc = Cout;
    g.taps = 9; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.sm = stride; g.sd = 1;
    g.b_tap_stride = Cin;
    g.bias = p.eng->Wp(b);
    if (resid) { g.resid = p.P(resid); g.ldr = Cout; }
    if (rowvec) { g.rowvec = p.P(rowvec); g.ldv = rowvec->ld(); g.rows_per_batch = Ho * Wo; }
    if (fsplit > 1) { g.splitk = fsplit; g.slab = p.F(p.slab_main_off); }
    return launch_gemm(g, st);
  }
  void plan_bwd(Plan& p) override {
    dy_off = y->goff;
    if (resid) {
      resid_alias = p.grad_alias(resid, y) ? 1 : 0;
      if (!resid_alias) dres = p.grad_dst(resid);
    }
    if (rowvec) {
      drv = p.grad_dst(rowvec);     // (allocates the grouped projection's bf16 output gradient; written by its cast, not here)
      rv32_off = p.tp32_off + (size_t)rowvec->col0 * sizeof(float);
      rv32_ld = rowvec->ld();
      cs_part_off = p.alloc(sizeof(float) * colsum_part_floats(Bn, Ho * Wo, Cout));
    }
    if (x->need_grad) dx = p.grad_dst(x);
    if (SDXL_UP2_3 && stride == 2 && H % 2 == 0 && W % 2 == 0 && Cin % 64 == 0 && Cout % 8 == 0 && !resid && !rowvec && (Bn * Ho * Wo) % 64 == 0)
      s2x_off = p.alloc(sizeof(bf16) * (size_t)4 * upconv_plane_rows(Bn, Ho, Wo) * Cin);
    if (x->need_grad && stride == 2 && H % 2 == 0 && W % 2 == 0 && Cout % 64 == 0 && Cin % 8 == 0)
      s2_planar_off = p.alloc(sizeof(bf16) * (size_t)4 * upconv_plane_rows(Bn, Ho, Wo) * Cin);
    splitk = pick_splitk(Cout, Cin, 9, (long)Bn * Ho * Wo);
    if (conv_wgrad3_policy(Cout, Cin, (long)Bn * Ho * Wo, Wo, stride)) splitk = conv_wgrad3_pick_splitk(Cout, Cin, (long)Bn * Ho * Wo);
    want_slab(p, Cout, Cin, 9, splitk);
    if (stride == 1 && Cin % 64 == 0) {
      fsplit = gemm_pick_splitk_small(Bn * Ho * Wo, Cout, 9 * Cin, 0);
      want_slab_main(p, Bn * Ho * Wo, Cout, fsplit);
      if (x->need_grad && Cout % 64 == 0) { dsplit = gemm_pick_splitk_small(Bn * H * W, Cin, 9 * Cout, 1); want_slab_main(p, Bn * H * W, Cin, dsplit); }
    }
  }
  int bwd(Plan& p, hipStream_t st, bool first) override {
    const bf16* dy = p.GP(dy_off);
    const long Mo = (long)Bn * Ho * Wo;
    if (resid && !resid_alias) CHK(launch_add(p.GP(dres.addend), dy, p.GP(dres.out), Mo * Cout, st));
    const bool upw = up2() && up_wg && KNOB(2) != 128;       // (knob 2 = 128: weight gradient on the upsampled image, A/B runs)
    if (up2()) CHK(launch_pixel_shuffle2(dy, (bf16*)p.F(planar_off), Bn, H / 2, W / 2, Cout, 0, st));      // dy in its four phases: dgrad and weight gradient read it
#ifdef SDXL_DIAG
    if (s2x_off != NONE && KNOB(2) == 512) {
      CHK(on_side(p, st, [&](hipStream_t s2) -> int {
        return launch_conv3x3_s2_wgrad(dy, (const bf16*)p.F(s2x_off), p.eng->Gp(w), p.eng->Gp(b), p.eng->emit_base ? p.eng->emit_base + w.off : nullptr,
                                       p.eng->emit_scale, first ? 0 : 1, Bn, H, W, Cin, Cout, splitk, p.F(p.slab_off), s2);
      }));
    } else
#endif
    if (upw) {
      CHK(on_side(p, st, [&](hipStream_t s2) -> int {
        // (bias gradient: the four phase planes would add in launch order -- a fixed-order column sum of dy instead, 10 - 20 us on this stream)
        CHK(launch_colsum_f32_batched(dy, p.eng->Gp(b), 1, (int)Mo, Cout, Cout, 0, p.F(cs_part_off), s2));
        return launch_upconv3x3_wgrad((const bf16*)p.F(planar_off), p.P(x_low), p.F(dweff_off), p.eng->Gp(w), nullptr,
                                      p.eng->emit_base ? p.eng->emit_base + w.off : nullptr, p.eng->emit_scale, first ? 0 : 1, Bn, H / 2, W / 2,
                                      Cin, Cout, up_ws, p.F(p.slab_off), s2);
      }));
    } else
    CHK(on_side(p, st, [&](hipStream_t s2) -> int {
      GemmP g;
      gemm_defaults(&g);
      g.form = GEMM_TN;
      g.A = dy; g.B = p.P(x); g.C = p.eng->Gp(w);
      g.M = Cout; g.N = Cin; g.K = (int)Mo;
      g.lda = Cout; g.ldb = Cin; g.ldc = 9L * Cin;
      g.taps = 9; g.Hm = Ho; g.Wm = Wo; g.Hs = H; g.Ws = W; g.sm = stride; g.sd = 1;
      g.c_tap_stride = Cin;
      g.out_f32 = 1;
      g.splitk = splitk;
      g.slab = p.F(p.slab_off);
      g.accumulate = first ? 0 : 1;
      g.bias_grad = p.eng->Gp(b);
      if (p.eng->emit_base) { g.Cb = p.eng->emit_base + w.off; g.cb_scale = p.eng->emit_scale; }
      CHK(launch_gemm(g, s2));
      return 0;
    }));
    if (rowvec) {   // d(time-embedding projection)[b][c] = sum over the pixels of sample b of dy: one launch, into the fp32 slice
      if (drv.addend != NONE) { sdxl_set_error("conv: time-embedding row vector has another gradient writer"); return 3; }
      // (partial rows now, on this stream; the fixed-order reduce of ALL the resnets' sums runs once, right before the grouped projection's cast)
      const size_t k0 = p.eng->cs_pending.size();
      p.eng->cs_pending.resize(k0 + Bn);
      CHK(launch_colsum_partials(dy, p.F(rv32_off), Bn, Ho * Wo, Cout, Cout, rv32_ld, p.F(cs_part_off), &p.eng->cs_pending[k0], st));
    }
    if (x->need_grad && s2_planar_off != NONE && KNOB(2) != 256) {
      CHK(launch_conv3x3_s2_dgrad(dy, p.eng->Wp(w), (bf16*)p.F(s2_planar_off), p.GP(dx.out), dx.addend != NONE ? p.GP(dx.addend) : nullptr, Bn, H, W,
                                  Cin, Cout, KNOB(0), st));
    } else
    if (x->need_grad && up2()) {      // straight into the low-resolution gradient (the UpsampleOp's backward is a no-op then)
      CHK(launch_upconv3x3_dgrad(nullptr, (const bf16*)p.F(weff_off), (bf16*)p.F(planar_off), p.GP(up_dx->out),
                                 up_dx->addend != NONE ? p.GP(up_dx->addend) : nullptr, Bn, H / 2, W / 2, Cin, Cout, up_ds,
                                 p.F(p.slab_main_off), KNOB(0), st));
    } else
    if (x->need_grad) {
      GemmP g;
      gemm_defaults(&g);
      g.form = GEMM_NN;
      g.A = dy; g.B = p.eng->Wp(w); g.C = p.GP(dx.out);
      g.M = Bn * H * W; g.N = Cin; g.K = Cout;
      g.lda = Cout; g.ldb = 9L * Cin; g.ldc = Cin;
      g.taps = 9; g.Hm = H; g.Wm = W; g.Hs = Ho; g.Ws = Wo; g.sm = 1; g.sd = stride;
      g.flip = 1; g.b_tap_stride = Cin;
      if (dx.addend != NONE) { g.resid = p.GP(dx.addend); g.ldr = Cin; }
      if (dsplit > 1) { g.splitk = dsplit; g.slab = p.F(p.slab_main_off); }
      g.prio = KNOB(0);
      if (KNOB(7) > 0) g.cfg = KNOB(7);     // experiment: configuration of the conv dgrads
      CHK(launch_gemm(g, st));
    }
    return 0;
  }
};

struct GroupNormOp : Op {
  Act *x, *y;
  PRef gm, bt;
  int Bn, HW, C, G, silu;
  float eps;
  size_t stats_off, prow_off;
  size_t dy_off = NONE;
  Plan::GradDst dx;
  GroupNormOp(Plan& p, Act* x_, Act* y_, PRef g_, PRef b_, int B_, int HW_, int C_, int G_, float eps_, int silu_)
      : x(x_), y(y_), gm(g_), bt(b_), Bn(B_), HW(HW_), C(C_), G(G_), silu(silu_), eps(eps_) {
    stats_off = p.alloc(sizeof(float) * Bn * G * 2);
    prow_off = p.alloc(sizeof(float) * Bn * 2 * C);      // per-sample rows of dgamma | dbeta partial sums, folded by the segment's flush_ln_params
    size_t need = groupnorm_ws_floats(Bn, C, G);
    if (need > p.gn_ws_floats) p.gn_ws_floats = need;  // one scratch, sized for the widest norm, allocated in build()
  }
  int fwd(Plan& p, hipStream_t st) override {
    if (KNOB(27) & 2) return 0;      // (knob 27: timing knock-outs, wrong results -- kernels.h)
    if (KNOB(27) & 128) return launch_silu_fwd(p.P(x), p.P(y), (long)Bn * HW * C, st);      // (timing only: ONE elementwise launch, live data downstream)
    return launch_groupnorm_fwd(p.P(x), p.P(y), p.eng->Wp(gm), p.eng->Wp(bt), p.F(stats_off), p.F(p.gn_ws_off), Bn, HW, C, G,
                                eps, silu, st);
  }
  void plan_bwd(Plan& p) override { dy_off = y->goff; dx = p.grad_dst(x); }
  int bwd(Plan& p, hipStream_t st, bool) override {
    if (KNOB(27) & 1) return 0;
    LnRedEntry r;      // dgamma | dbeta: B partial rows, added in a fixed order by the batched reduce (no atomics: bitwise reproducible)
    r.part = p.F(prow_off); r.dgamma = p.eng->Gp(gm); r.dbeta = p.eng->Gp(bt); r.C = C; r.nblk = Bn;
    p.eng->ln_pending.push_back(r);
    return launch_groupnorm_bwd(p.P(x), p.GP(dy_off), p.eng->Wp(gm), p.eng->Wp(bt), p.F(stats_off), p.GP(dx.out),
                                p.GP(dx.addend), p.eng->Gp(gm), p.eng->Gp(bt), p.F(p.gn_ws_off), Bn, HW, C, G, silu, st, p.F(prow_off));
  }
};

struct LayerNormOp : Op {
  Act *x, *y;
  PRef gm, bt;
  int C;
  float eps;
  size_t stats_off, part_off = NONE;
  size_t dy_off = NONE;
  Plan::GradDst dx;
  int fused = 0;      // the consumer's dgrad epilogue produces dx (LinearOp::ln_mode); set by its plan_bwd, which runs first
  LayerNormOp(Plan& p, Act* x_, Act* y_, PRef g_, PRef b_, int C_, float eps_) : x(x_), y(y_), gm(g_), bt(b_), C(C_), eps(eps_) {
    stats_off = p.alloc(sizeof(float) * x->rows * 2);
  }
  int fwd(Plan& p, hipStream_t st) override {
    if (KNOB(27) & 4) return 0;
    if (KNOB(27) & 256) return launch_silu_fwd(p.P(x), p.P(y), (long)x->rows * C, st);      // (timing only: an elementwise launch with the same bytes)
    return launch_layernorm_fwd(p.P(x), p.P(y), p.eng->Wp(gm), p.eng->Wp(bt), p.F(stats_off), (int)x->rows, C, eps, st);
  }
  void plan_bwd(Plan& p) override {
    dy_off = y->goff;
    dx = p.grad_dst(x);
    part_off = p.alloc(sizeof(float) * layernorm_bwd_part_floats((int)x->rows, C));   // own buffer: reduced at the segment's end
  }
  int bwd(Plan& p, hipStream_t st, bool) override {
    // dx and the per-block dgamma | dbeta partial sums in one pass (norm.hip); the partials of all LayerNorms of the segment
    // are folded into the gradients by one launch at its end (Engine::flush_ln_params)
    // Default (again, since round 5): dx + per-block parameter partial sums in ONE pass over x, dy (180 VGPRs + 40 KiB LDS).  Rounds 3 - 4 ran
    // the lean dx kernel (120 VGPRs, no LDS) on the caller's stream + the parameter gradients as a leaf pass on the side stream (knob 10 = 1):
    // -0.7 ms then; with the side stream's passes costing the step their full duration now (210 x 5.8 us re-reading 21 MB each beside the GEMMs:
    // knob 25 = 1 is worth -1.2 ms) the one-pass form is -0.4 ms again, five alternations (profiles/r05w_ab_ln_fused.txt).
    if (fused) {      // dx is in place (or will be: this op runs after the consumer in the backward order); the parameter gradients:
      if (fused == 2) {
        const bf16* xp = p.P(x); const bf16* dyp = p.GP(dy_off); const float* sp = p.F(stats_off);
        float* dg = p.eng->Gp(gm); float* db = p.eng->Gp(bt); const int Mr = (int)x->rows, Cc = C;
        return side_leaf(p, st, [=](hipStream_t s2) -> int { return launch_layernorm_param_grads(xp, dyp, sp, dg, db, Mr, Cc, s2); });
      }
      LnRedEntry r;      // the epilogue's per-row-block column sums, folded at the segment's end
      r.part = p.F(part_off); r.dgamma = p.eng->Gp(gm); r.dbeta = p.eng->Gp(bt); r.C = C; r.nblk = gemm_ln_rowblocks((int)x->rows);
      p.eng->ln_pending.push_back(r);
      return 0;
    }
    // knob 25 (timing knock-outs, wrong gradients): 1 = no parameter-gradient pass, 2 = no dx pass either, 3 = dx pass only skipped
    if (KNOB(25) == 2) return 0;
    if (KNOB(10) == 1 || KNOB(10) == 3) {      // (A/B runs: the lean dx kernel + the parameter gradients as a leaf pass of their own -- the form of rounds 4 / 5a)
      if (KNOB(25) != 3)
      CHK(launch_layernorm_bwd(p.P(x), p.GP(dy_off), p.eng->Wp(gm), p.F(stats_off), p.GP(dx.out), p.GP(dx.addend), nullptr, nullptr, (int)x->rows, C, st));
      const bf16* xp = p.P(x); const bf16* dyp = p.GP(dy_off); const float* sp = p.F(stats_off);
      const int Mr = (int)x->rows, Cc = C;
      if (KNOB(25) == 1) return 0;
      if (KNOB(10) == 3) {      // (A/B runs: the column-sum pass with one atomic per column and block)
        float* dg = p.eng->Gp(gm); float* db = p.eng->Gp(bt);
        return side_leaf(p, st, [=](hipStream_t s2) -> int { return launch_layernorm_param_grads(xp, dyp, sp, dg, db, Mr, Cc, s2); });
      }
      // partial rows by plain stores into the op's own buffer; the segment's flush folds them (queued leaves run before it on the side stream)
      float* part = p.F(part_off);
      LnRedEntry r;
      r.part = part; r.dgamma = p.eng->Gp(gm); r.dbeta = p.eng->Gp(bt); r.C = C; r.nblk = layernorm_param_partial_rows(Mr, Cc);
      p.eng->ln_pending.push_back(r);
      return side_leaf(p, st, [=](hipStream_t s2) -> int { return launch_layernorm_param_partials(xp, dyp, sp, part, Mr, Cc, s2); });
    }
    LnRedEntry r;
    r.part = p.F(part_off); r.dgamma = p.eng->Gp(gm); r.dbeta = p.eng->Gp(bt); r.C = C;
    CHK(launch_layernorm_bwd(p.P(x), p.GP(dy_off), p.eng->Wp(gm), p.F(stats_off), p.GP(dx.out), p.GP(dx.addend),
                             p.F(part_off), &r.nblk, (int)x->rows, C, st));
    p.eng->ln_pending.push_back(r);
    return 0;
  }
};

static bool ln_fuse_plan(LayerNormOp* ln, Plan& p, int mode) {
  const int M = (int)ln->x->rows, C = ln->C;
  if (ln->x->ld() != C || ln->y->ld() != C) return false;
  if (gemm_ln_pcol_floats(M, C) > layernorm_bwd_part_floats(M, C)) return false;      // (the column partials live in the LayerNorm's own part buffer)
  ln->fused = mode;
  const size_t pf = gemm_ln_part_floats(M, C);
  if (pf > p.ln_part_floats) p.ln_part_floats = pf;
  return true;
}
static void ln_fuse_target(LayerNormOp* ln, Plan& p, GemmP& g, int mode) {
  g.ln_x = p.P(ln->x); g.ln_ldx = ln->x->ld();
  g.ln_stats = p.F(ln->stats_off);
  g.ln_gamma = p.eng->Wp(ln->gm);
  g.ln_dx = p.GP(ln->dx.out); g.ln_addend = p.GP(ln->dx.addend); g.ln_ldo = ln->x->ld();
  g.ln_part = p.F(p.ln_part_off);
  g.ln_epoch = ++p.ln_epoch;
  if (mode == 1) { g.ln_pcol = p.F(ln->part_off); g.C = nullptr; }      // dy itself is not stored
}

// self attention: qkv [B*N][3C]; cross attention: q [B*N][C], kv [B*ctx][2C]
struct AttnOp : Op {
  Act *q, *kv, *o;  // self: q == kv == qkv tensor
  int Bn, heads, Nq, Nk, C;
  bool self;
  int qsplit = 1;
  size_t lse_off, delta_off;
  AttnOp(Plan& p, Act* q_, Act* kv_, Act* o_, int B_, int heads_, int Nq_, int Nk_, int C_, bool self_)
      : q(q_), kv(kv_), o(o_), Bn(B_), heads(heads_), Nq(Nq_), Nk(Nk_), C(C_), self(self_) {
    lse_off = p.alloc(sizeof(float) * (size_t)Bn * heads * Nq);
    delta_off = p.alloc(sizeof(float) * (size_t)Bn * heads * Nq);
    qsplit = attn_pick_qsplit(Bn, heads, Nq, Nk);
    size_t need = attn_part_floats(Bn, heads, Nk, qsplit);
    // self-attention's dK / dV kernel runs on the caller's stream, cross-attention's on the side stream (bwd): one scratch each
    if (self) { if (need > p.apart_floats) p.apart_floats = need; }
    else if (need > p.apart_side_floats) p.apart_side_floats = need;
  }
  void fill(Plan& p, AttnP& a, bool grads) {
    memset(&a, 0, sizeof(a));
    a.B = Bn; a.H = heads; a.Nq = Nq; a.Nk = Nk;
    if (self) {
      a.Q = p.P(q); a.K = p.P(q) + C; a.V = p.P(q) + 2 * C;
      a.ldq = a.ldk = a.ldv = 3L * C;
    } else {
      a.Q = p.P(q); a.ldq = C;
      a.K = p.P(kv); a.V = p.P(kv) + C; a.ldk = a.ldv = kv->ld();   // (a column slice of the grouped K | V projection)
    }
    a.O = p.P(o); a.ldo = o->ld();
    a.LSE = p.F(lse_off);
    if (grads) {
      a.dO = p.GP(do_off); a.lddo = o->ld();
      a.Delta = p.F(delta_off);
      a.qsplit = qsplit;
      a.part = p.F(self ? p.apart_off : p.apart_side_off);
      if (self) {
        a.dQ = p.GP(dq.out); a.dK = a.dQ + C; a.dV = a.dQ + 2 * C;
        a.lddq = a.lddk = a.lddv = 3L * C;
      } else {
        a.dQ = p.GP(dq.out); a.lddq = C;
        a.dK = p.GP(dkv.out); a.dV = a.dK + C; a.lddk = a.lddv = kv->ld();
      }
    }
  }
  int fwd(Plan& p, hipStream_t st) override {
    if (KNOB(27) & (self ? 32 : 64)) return 0;
    AttnP a;
    fill(p, a, false);
    return launch_attn_fwd(a, st);
  }
  bool bad = false;
  size_t do_off = NONE;
  Plan::GradDst dq, dkv;
  void plan_bwd(Plan& p) override {
    do_off = o->goff;
    dq = p.grad_dst(q);
    if (dq.addend != NONE) bad = true;
    if (!self) { dkv = p.grad_dst(kv); if (dkv.addend != NONE) bad = true; }
  }
  bool delta_fused = false;     // the out-projection's dgrad writes Delta (LinearOp::delta_attn): no Delta pass in the backward
  int bwd(Plan& p, hipStream_t st, bool) override {
    if (bad) { sdxl_set_error("attention: operand gradient has another writer"); return 3; }
    if (KNOB(27) & (self ? 16 : 8)) return 0;
    AttnP a;
    fill(p, a, true);
    a.prio = KNOB(1);
    a.delta_ready = delta_fused ? 1 : 0;
    if (self) return launch_attn_bwd(a, st);
    // cross attention: dK | dV (this block's slice of the grouped projection's gradient) is read by nothing before that
    // projection's weight gradient, a leaf on the side stream -- so the dK / dV kernel (+ its partial reduce) goes there too,
    // behind the dQ kernel that produces Delta, and leaves the caller's stream (1.8 ms per step)
    CHK(launch_attn_bwd_dq(a, st));
    return side_leaf(p, st, [=](hipStream_t s2) -> int { return launch_attn_bwd_dkv(a, s2); });
  }
};

// level-2 self-attention (1280 channels at M = 4096: the out-projection's dgrad runs on 128 x 128 tiles, whose 64 x 64 wave tiles are whole
// heads): Delta from that GEMM's epilogue instead of a pass of its own (60 launches per step).  knob 21 = 1: off (A/B runs)
static bool attn_delta_wanted(const AttnOp* a) { return a->self && a->qsplit <= 1 && a->Nk >= 256 && a->C % 128 == 0 && a->delta_fused; }
static void attn_set_delta_fused(AttnOp* a, bool on) { a->delta_fused = on; }
static void attn_delta_target(AttnOp* a, Plan& p, GemmP& g) {
  g.delta_o = p.P(a->o); g.delta_ldo = a->o->ld();
  g.delta_out = p.F(a->delta_off);
  g.delta_nq = a->Nq; g.delta_heads = a->heads;
}

struct SiluOp : Op {
  Act *x, *y;
  size_t dy_off = NONE;
  Plan::GradDst dx;
  SiluOp(Act* x_, Act* y_) : x(x_), y(y_) {}
  int fwd(Plan& p, hipStream_t st) override { return launch_silu_fwd(p.P(x), p.P(y), x->rows * x->cols, st); }
  void plan_bwd(Plan& p) override { dy_off = y->goff; dx = p.grad_dst(x); }
  int bwd(Plan& p, hipStream_t st, bool) override {
    return launch_silu_bwd(p.P(x), p.GP(dy_off), p.GP(dx.out), p.GP(dx.addend), x->rows * x->cols, st);
  }
};

struct ConcatOp : Op {
  Act *a, *b, *o;
  size_t do_off = NONE;
  Plan::GradDst da, db;
  ConcatOp(Act* a_, Act* b_, Act* o_) : a(a_), b(b_), o(o_) {}
  int fwd(Plan& p, hipStream_t st) override { return launch_concat(p.P(a), a->cols, p.P(b), b->cols, p.P(o), a->rows, st); }
  void plan_bwd(Plan& p) override { do_off = o->goff; da = p.grad_dst(a); db = p.grad_dst(b); }
  int bwd(Plan& p, hipStream_t st, bool) override {
    return launch_split_add(p.GP(do_off), p.GP(da.out), a->cols, p.GP(da.addend), p.GP(db.out), b->cols, p.GP(db.addend),
                            a->rows, st);
  }
};

struct UpsampleOp : Op {
  Act *x, *y;
  int Bn, H, W, C;
  size_t dy_off = NONE;
  Plan::GradDst dx;
  ConvOp* fused = nullptr;     // the convolution that consumes y works on x directly (ConvOp::up2): y is its weight gradient's operand only
  UpsampleOp(Act* x_, Act* y_, int B_, int H_, int W_, int C_) : x(x_), y(y_), Bn(B_), H(H_), W(W_), C(C_) {}
  int fwd(Plan& p, hipStream_t st) override {
    if (fused && fused->up2() && fused->up_wg && KNOB(2) != 128) return 0;      // nothing reads y
    if (fused && fused->up2() && !p.eng->use_graphs) {   // off the critical stream: nothing reads y before the backward's side-stream weight gradient
                                                         // (not under graph capture: the forward's capture ends with nothing to join the side stream)
      const bf16* xp = p.P(x); bf16* yp = p.P(y);
      const int b = Bn, h = H, w = W, c = C;
      return on_side(p, st, [=](hipStream_t s2) -> int { return launch_upsample2x(xp, yp, b, h, w, c, s2); });
    }
    return launch_upsample2x(p.P(x), p.P(y), Bn, H, W, C, st);
  }
  void plan_bwd(Plan& p) override { dy_off = y->goff; dx = p.grad_dst(x); }
  int bwd(Plan& p, hipStream_t st, bool) override {
    if (fused && fused->up2() && y->need_grad) return 0;      // the convolution's dgrad wrote dx
    return launch_upsample2x_bwd(p.GP(dy_off), p.GP(dx.out), p.GP(dx.addend), Bn, H, W, C, st);
  }
};

// conditioning embeddings (inputs only: no gradient).  t -> sincos(320); time_ids -> sincos(256) x 6;
// aug_in = [pooled | time-id embedding]
struct EmbedInOp : Op {
  int fwd(Plan& p, hipStream_t st) override {
    const sdxl_unet_config& c = p.eng->cfg;
    const int B = p.B, ch0 = c.block_out_channels[0], ad = c.addition_time_embed_dim;
    CHK(launch_sincos(p.F(p.t_off), p.P(p.te_sin), B, ch0, ch0, st));
    CHK(launch_sincos(p.F(p.tid_off), p.P(p.tid_emb), B * 6, ad, ad, st));
    CHK(launch_copy_cols(p.P(p.tid_emb), 6L * ad, p.P(p.aug_in) + c.pooled_dim, c.pooled_dim + 6L * ad, B, 6 * ad, st));
    return 0;
  }
  void plan_bwd(Plan&) override {}
  int bwd(Plan&, hipStream_t, bool) override { return 0; }
};

// ================================================================================================
// Parameters
// ================================================================================================
PRef Engine::param(size_t numel, bool small) {
  if (registering) {
    PRef p;
    p.off = param_elems;
    p.numel = numel;
    param_elems = align_up(param_elems + numel, 64);
    natives.push_back(p);
    if (small) small_ranges.emplace_back(p.off, numel);
    return p;
  }
  PRef p = natives.at(native_cursor++);
  if (p.numel != numel) {
    fprintf(stderr, "sdxlstep: internal error: parameter walk mismatch (%zu vs %zu)\n", p.numel, numel);
    abort();
  }
  return p;
}

void Engine::map_src(const std::string& name, std::vector<long> shape, PRef p, int kind, size_t elem_off, int ci_pad) {
  if (!registering) return;
  SrcParam s;
  s.name = name;
  s.ndim = (int)shape.size();
  for (int i = 0; i < 4; ++i) s.shape[i] = i < s.ndim ? shape[i] : 1;
  s.native = p;
  s.kind = kind;
  s.elem_off = elem_off;
  s.ci_pad = ci_pad;
  src_index[name] = (int)src.size();
  src.push_back(s);
}

int Engine::seg_of(size_t elem_off) const {
  for (int s = 0; s < nseg; ++s)
    if (elem_off >= seg_begin[s] && elem_off < seg_end[s]) return s;
  return nseg - 1;
}

// ================================================================================================
// Builder: one walk that either registers parameters (bp == nullptr) or emits the plan's ops
// ================================================================================================
namespace {
struct Builder {
  Engine& e;
  Plan* pl;
  int B, H, W, ctx;
  int last_seg = 0;
  Act* emb_act = nullptr;
  Act* tp_all = nullptr;                     // [B][sum Cout]: time-embedding projection of every resnet
  std::map<std::string, int> tp_col;        // resnet prefix -> first column in tp_all
  long tp_total = 0;
  PRef wtp, btp;
  std::map<int, Act*> kv_all;                // per channel width C: [B*ctx][sum 2C] cross-attention K | V of every transformer block of that width
  std::map<std::string, int> kv_col;        // transformer block prefix -> first column in its group's tensor
  std::map<int, bool> kv_done;
  explicit Builder(Engine& e_, Plan* p_) : e(e_), pl(p_) {
    if (pl) { B = pl->B; H = pl->H; W = pl->W; ctx = pl->ctx; } else { B = H = W = ctx = 0; }
  }
  template <class T>
  T* tagseg(T* op, PRef first) {
    if (first.off != NONE) last_seg = e.seg_of(first.off);
    op->seg = last_seg;
    return op;
  }
  // ---- layers ----
  // kind 2: GEGLU first projection, rows (and bias) packed value | gate interleaved in groups of `group` (repack_kernel)
  Act* linear(const std::string& name, Act* x, int K, int N, bool bias, Act* resid, bool conv1x1 = false, int kind = 0,
              LinearOp** op_out = nullptr, int group = 0) {
    PRef w = e.param((size_t)N * K);
    if (conv1x1) e.map_src(name + ".weight", {N, K, 1, 1}, w, 0, 0, 0);
    else e.map_src(name + ".weight", {N, K}, w, kind, 0, group);
    PRef b;
    if (bias) { b = e.param(N, true); e.map_src(name + ".bias", {N}, b, kind, 0, group); }
    if (!pl) return nullptr;
    Act* y = kind == 2 ? wide_act(x->rows, N) : pl->new_act(x->rows, N);
    LinearOp* op = tagseg(pl->add<LinearOp>(x, y, w, b, K, N, resid), w);
    if (op_out) *op_out = op;
    ln_consumer(op);
    return y;
  }
  // LayerNorm outputs and the linear layers that read them: a LayerNorm whose output feeds exactly one linear layer has its backward run
  // in that layer's dgrad epilogue (LinearOp::ln_src)
  std::map<Act*, LayerNormOp*> ln_of;
  std::map<Act*, LinearOp*> ln_reader;
  void ln_consumer(LinearOp* op) {
    auto it = ln_of.find(op->x);
    if (it == ln_of.end()) return;
    auto rd = ln_reader.find(op->x);
    if (rd != ln_reader.end()) { rd->second->ln_src = nullptr; return; }      // a second reader: nobody fuses
    ln_reader[op->x] = op;
    op->ln_src = it->second;
  }
  // The feed-forward hidden tensors ([rows][8C] pre-activation, [rows][4C] activation): a row stride that is a multiple of 1 KiB puts the
  // 16 rows of a K-contiguous LDS-DMA piece (and the 32 k-rows of an N-contiguous one) on 1, 2 or 4 of the 16 L2 channels of an XCD
  // (256-byte interleave); 64 padding columns spread them (NN 4096 x 1280 x 10240: DMA-only loop 172 -> 133 us,
  // profiles/r04a_cr256_ld_sensitivity.txt).  The padded tensor is the parent of a column view, as the grouped K | V projection's slices are.
  Act* wide_act(long rows, int cols) {
    if (KNOB(18) == 1 || cols % 512) return pl->new_act(rows, cols);
    return pl->view(pl->new_act(rows, cols + 64), 0, cols);
  }
  // fused projection of several [Ni, K] source matrices into one [sum Ni, K] native matrix (no bias)
  Act* linear_fused(const std::vector<std::string>& names, Act* x, int K, int Neach) {
    const int n = (int)names.size();
    PRef w = e.param((size_t)n * Neach * K);
    for (int i = 0; i < n; ++i) e.map_src(names[i] + ".weight", {Neach, K}, w, 0, (size_t)i * Neach * K, 0);
    if (!pl) return nullptr;
    Act* y = pl->new_act(x->rows, n * Neach);
    ln_consumer(tagseg(pl->add<LinearOp>(x, y, w, PRef(), K, n * Neach, nullptr), w));
    return y;
  }
  Act* conv(const std::string& name, Act* x, int h, int w_, int cin, int cout, int stride, Act* resid, Act* rowvec,
            int cin_src = -1, int cout_src = -1) {
    if (cin_src < 0) cin_src = cin;
    if (cout_src < 0) cout_src = cout;
    PRef w = e.param((size_t)cout * 9 * cin);
    e.map_src(name + ".weight", {cout_src, cin_src, 3, 3}, w, 1, 0, cin);
    PRef b = e.param(cout, true);
    e.map_src(name + ".bias", {cout_src}, b, 0, 0, 0);
    if (!pl) return nullptr;
    int ho = (h - 1) / stride + 1, wo = (w_ - 1) / stride + 1;
    Act* y = pl->new_act((long)B * ho * wo, cout);
    tagseg(pl->add<ConvOp>(x, y, w, b, B, h, w_, cin, cout, stride, resid, rowvec), w);
    return y;
  }
  Act* groupnorm(const std::string& name, Act* x, int hw, int C, float eps, int silu) {
    PRef g = e.param(C, true);
    e.map_src(name + ".weight", {C}, g, 0, 0, 0);
    PRef b = e.param(C, true);
    e.map_src(name + ".bias", {C}, b, 0, 0, 0);
    if (!pl) return nullptr;
    Act* y = pl->new_act(x->rows, C);
    tagseg(pl->add<GroupNormOp>(*pl, x, y, g, b, B, hw, C, e.cfg.norm_num_groups, eps, silu), g);
    return y;
  }
  Act* layernorm(const std::string& name, Act* x, int C) {
    PRef g = e.param(C, true);
    e.map_src(name + ".weight", {C}, g, 0, 0, 0);
    PRef b = e.param(C, true);
    e.map_src(name + ".bias", {C}, b, 0, 0, 0);
    if (!pl) return nullptr;
    Act* y = pl->new_act(x->rows, C);
    ln_of[y] = tagseg(pl->add<LayerNormOp>(*pl, x, y, g, b, C, e.cfg.ln_eps), g);
    return y;
  }
  Act* silu(Act* x) {
    if (!pl) return nullptr;
    Act* y = pl->new_act(x->rows, x->cols);
    tagseg(pl->add<SiluOp>(x, y), PRef());
    return y;
  }
  Act* resnet(const std::string& p, Act* x, int h, int w_, int cin, int cout) {
    Act* n1 = groupnorm(p + ".norm1", x, h * w_, cin, e.cfg.resnet_eps, 1);
    // diffusers key order: norm1, conv1, time_emb_proj, norm2, conv2, conv_shortcut.  The time projection must
    // run before conv1 consumes it, so parameters are registered in execution order instead.
    // this resnet's columns of the grouped time-embedding projection (one GEMM for all 17 resnets, run())
    Act* tp = pl ? pl->view(tp_all, tp_col.at(p), cout) : nullptr;
    Act* c1 = conv(p + ".conv1", n1, h, w_, cin, cout, 1, nullptr, tp);
    Act* n2 = groupnorm(p + ".norm2", c1, h * w_, cout, e.cfg.resnet_eps, 1);
    Act* sc = x;
    if (cin != cout) sc = linear(p + ".conv_shortcut", x, cin, cout, true, nullptr, true);
    return conv(p + ".conv2", n2, h, w_, cout, cout, 1, sc, nullptr);
  }
  Act* tf_block(const std::string& b, Act* x, Act* ehs, int C, int N) {
    const int heads = C / e.cfg.head_dim, cross = e.cfg.cross_attention_dim;
    Act* l1 = layernorm(b + ".norm1", x, C);
    Act* qkv = linear_fused({b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"}, l1, C, C);
    Act* a1 = nullptr;
    AttnOp* sa = nullptr;
    if (pl) {
      a1 = pl->new_act(x->rows, C);
      sa = tagseg(pl->add<AttnOp>(*pl, qkv, qkv, a1, B, heads, N, N, C, true), PRef());
    }
    LinearOp* o1 = nullptr;
    Act* x1 = linear(b + ".attn1.to_out.0", a1, C, C, true, x, false, 0, &o1);
    // Delta of the self-attention backward from the out-projection's dgrad epilogue where that GEMM runs on 128 x 128 tiles (one-round
    // problems: M x C = 4096 x 1280; at M = 16384 x 640 the dgrad takes 160-column tiles, which split heads: Delta pass there)
    if (sa && o1 && KNOB(21) != 1 && C % 128 == 0 && (x->rows / 128) * (C / 160) <= 256 && sa->qsplit <= 1) { sa->delta_fused = true; o1->delta_attn = sa; }
    Act* l2 = layernorm(b + ".norm2", x1, C);
    Act* q = linear(b + ".attn2.to_q", l2, C, C, false, nullptr);
    // K | V of the prompt embeddings: this block's 2C columns of the grouped projection (one GEMM for all blocks, run())
    Act* kv = pl ? pl->view(kv_all.at(C), kv_col.at(b), 2 * C) : nullptr;
    (void)cross;
    Act* a2 = nullptr;
    if (pl) {
      a2 = pl->new_act(x->rows, C);
      tagseg(pl->add<AttnOp>(*pl, q, kv, a2, B, heads, N, ctx, C, false), PRef())->needs_hoisted = true;
    }
    Act* x2 = linear(b + ".attn2.to_out.0", a2, C, C, true, x1);
    Act* l3 = layernorm(b + ".norm3", x2, C);
    // feed-forward: GEGLU lives in the epilogues of the two projections (forward: value * gelu(gate) next to u;
    // backward: the second projection's dgrad writes dU directly), no separate activation pass
    // packing group 64: the forward projection then runs on the 256 x 256 kernel, whose register epilogue has value and gate of
    // a channel in one lane (131 vs 156 us at level 2: the 128-row kernel stages the fp32 tile through LDS, two barriers per 64
    // rows); the backward's dgrad epilogue takes any group.  (knob 5 = 80: round 2's packing for 160-column tiles, A/B runs)
    const int group = (KNOB(5) == 80 && (4 * C) % 80 == 0) ? 80 : 64;
    LinearOp *ff1 = nullptr, *ff2 = nullptr;
    Act* u = linear(b + ".ff.net.0.proj", l3, C, 8 * C, true, nullptr, false, 2, &ff1, group);
    Act* g = pl ? wide_act(x->rows, 4 * C) : nullptr;
    if (ff1) { ff1->gact = g; ff1->ggroup = group; }
    Act* y = linear(b + ".ff.net.2", g, 4 * C, C, true, x2, false, 0, &ff2);
    if (ff2) { ff2->gu = u; ff2->ggroup = group; }
    return y;
  }
  Act* transformer(const std::string& p, Act* x, Act* ehs, int h, int w_, int C, int depth) {
    Act* n = groupnorm(p + ".norm", x, h * w_, C, e.cfg.tf_gn_eps, 0);
    Act* t = linear(p + ".proj_in", n, C, C, true, nullptr);
    for (int k = 0; k < depth; ++k) t = tf_block(p + ".transformer_blocks." + std::to_string(k), t, ehs, C, h * w_);
    return linear(p + ".proj_out", t, C, C, true, x);
  }
  void run() {
    const sdxl_unet_config& c = e.cfg;
    const int* ch = c.block_out_channels;
    const int temb = ch[0] * 4;
    const int add_in = c.pooled_dim + 6 * c.addition_time_embed_dim;
    Act *x_in = nullptr, *ehs = nullptr, *te_sin = nullptr, *aug_in = nullptr;
    if (pl) {
      pl->x_in = x_in = pl->new_act((long)B * H * W, 8, false);
      pl->ehs = ehs = pl->new_act((long)B * ctx, c.cross_attention_dim, false, kv_pad_rows(B, ctx));
      pl->te_sin = te_sin = pl->new_act(B, ch[0], false);
      pl->tid_emb = pl->new_act((long)B * 6, c.addition_time_embed_dim, false);
      pl->aug_in = aug_in = pl->new_act(B, add_in, false);
      pl->t_off = pl->alloc(sizeof(float) * B);
      pl->tid_off = pl->alloc(sizeof(float) * B * 6);
      pl->loss_off = pl->alloc(sizeof(float) * 8);
      pl->loss_part_off = pl->alloc(sizeof(float) * loss_part_floats(B, H * W));
      pl->in_lat_off = pl->alloc(sizeof(float) * (size_t)B * 4 * H * W);
      pl->in_noise_off = pl->alloc(sizeof(float) * (size_t)B * 4 * H * W);
      pl->in_sig_off = pl->alloc(sizeof(float) * B);
      pl->in_tag_off = pl->alloc(sizeof(float) * B);
      tagseg(pl->add<EmbedInOp>(), PRef());
    }
    auto register_kv = [&](int Cgrp) {
      // Cross-attention K / V projections of all transformer blocks of one width as ONE GEMM: they share the A operand (the prompt
      // embeddings, [B*77][2048]) and depend on nothing else, so the 70 per-block launches (M = 308 rows: 119 TFLOP/s each) become
      // two problems on the side stream at the start of the forward -- [308] x [60 x 2560 = 153 600] x [2048] for the 1280-channel
      // blocks, [308] x [10 x 1280] x [2048] for the 640-channel ones -- and their 70 weight-gradient launches two TN GEMMs.
      // A group's weight is registered (and its op placed) right before the FIRST transformer block of its width: in the backward its
      // gradient is then complete as soon as the last cross-attention backward of that width has run.  One group per width, not one
      // for all: the 1280-channel group is 92 % of the weight (314 M parameters, a segment of its own: Engine::build) and complete after
      // down_blocks.2, with the whole 640-channel level's backward still to run under its exchange; as ONE weight its 681 MB bucket
      // became ready ~3 ms before the end of the backward and its exchange was the step's un-overlapped tail (bench.py
      // --exchange-shadow: 2.7 ms at 300 GB/s, profiles/r04c_exchange_shadow.txt).
      // Every block's attention reads / writes its column slice of its group's result.
      std::vector<std::pair<std::string, int>> blocks;
      auto add_tf = [&](const std::string& p, int C, int depth) {
        if (C != Cgrp) return;
        for (int k = 0; k < depth; ++k) blocks.emplace_back(p + ".transformer_blocks." + std::to_string(k), C);
      };
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < c.layers_per_block; ++j)
          if (c.transformer_layers[i] > 0) add_tf("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), ch[i], c.transformer_layers[i]);
      add_tf("mid_block.attentions.0", ch[2], c.transformer_layers[2]);
      for (int ui = 0; ui < 3; ++ui)
        for (int j = 0; j < c.layers_per_block + 1; ++j)
          if (c.transformer_layers[2 - ui] > 0) add_tf("up_blocks." + std::to_string(ui) + ".attentions." + std::to_string(j), ch[2 - ui], c.transformer_layers[2 - ui]);
      long ntot = 0;
      for (auto& b : blocks) ntot += 2L * b.second;
      const int cross = c.cross_attention_dim;
      PRef wkv = e.param((size_t)ntot * cross);
      long col = 0;
      for (auto& b : blocks) {
        e.map_src(b.first + ".attn2.to_k.weight", {b.second, cross}, wkv, 0, (size_t)col * cross, 0);
        e.map_src(b.first + ".attn2.to_v.weight", {b.second, cross}, wkv, 0, (size_t)(col + b.second) * cross, 0);
        kv_col[b.first] = (int)col;
        col += 2L * b.second;
      }
      kv_done[Cgrp] = true;
      if (pl) {
        Act* all = pl->new_act((long)B * ctx, (int)ntot, true, kv_pad_rows(B, ctx));
        kv_all[Cgrp] = all;
        LinearOp* op = tagseg(pl->add<LinearOp>(ehs, all, wkv, PRef(), cross, (int)ntot, nullptr), wkv);
        op->hoist_fwd = true;
      }
    };
    {
      // The 17 resnets' time_emb_proj (Linear(silu(emb)) [B][1280] -> [B][Cout]) as ONE GEMM: rows of all of them contiguous
      // in the arena (first segment, like the K | V weight: its gradient completes last), each resnet's conv1 reads its
      // column slice as the per-sample row vector of its epilogue and adds its gradient into the fp32 [B][sum Cout] buffer.
      std::vector<std::pair<std::string, int>> rs;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < c.layers_per_block; ++j) rs.emplace_back("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), ch[i]);
      rs.emplace_back("mid_block.resnets.0", ch[2]);
      rs.emplace_back("mid_block.resnets.1", ch[2]);
      for (int ui = 0; ui < 3; ++ui)
        for (int j = 0; j < c.layers_per_block + 1; ++j) rs.emplace_back("up_blocks." + std::to_string(ui) + ".resnets." + std::to_string(j), ch[2 - ui]);
      tp_total = 0;
      for (auto& r : rs) tp_total += r.second;
      wtp = e.param((size_t)tp_total * temb);
      btp = e.param((size_t)tp_total, true);
      long col = 0;
      for (auto& r : rs) {
        e.map_src(r.first + ".time_emb_proj.weight", {r.second, temb}, wtp, 0, (size_t)col * temb, 0);
        e.map_src(r.first + ".time_emb_proj.bias", {r.second}, btp, 0, (size_t)col, 0);
        tp_col[r.first] = (int)col;
        col += r.second;
      }
    }
    Act* t1 = linear("time_embedding.linear_1", te_sin, ch[0], temb, true, nullptr);
    Act* t1s = silu(t1);
    Act* t2 = linear("time_embedding.linear_2", t1s, temb, temb, true, nullptr);
    Act* a1 = linear("add_embedding.linear_1", aug_in, add_in, temb, true, nullptr);
    Act* a1s = silu(a1);
    Act* emb = linear("add_embedding.linear_2", a1s, temb, temb, true, t2);
    emb_act = silu(emb);
    if (pl) {
      tp_all = pl->new_act(B, (int)tp_total);
      pl->tp32_bytes = sizeof(float) * (size_t)B * tp_total;
      pl->tp32_off = pl->alloc(pl->tp32_bytes);
      LinearOp* op = tagseg(pl->add<LinearOp>(emb_act, tp_all, wtp, btp, temb, (int)tp_total, nullptr), wtp);
      op->dy32_off = pl->tp32_off;
    }

    Act* x = conv("conv_in", x_in, H, W, 8, ch[0], 1, nullptr, nullptr, c.in_channels, ch[0]);
    struct Skip { Act* a; int c; };
    std::vector<Skip> skips;
    skips.push_back({x, ch[0]});
    int h = H, w = W, prev = ch[0];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < c.layers_per_block; ++j) {
        std::string p = "down_blocks." + std::to_string(i);
        x = resnet(p + ".resnets." + std::to_string(j), x, h, w, prev, ch[i]);
        prev = ch[i];
        if (c.transformer_layers[i] > 0) {
          if (!kv_done[ch[i]]) register_kv(ch[i]);
          x = transformer(p + ".attentions." + std::to_string(j), x, ehs, h, w, ch[i], c.transformer_layers[i]);
        }
        skips.push_back({x, prev});
      }
      if (i < 2) {
        x = conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", x, h, w, prev, prev, 2, nullptr, nullptr);
        h = (h - 1) / 2 + 1;
        w = (w - 1) / 2 + 1;
        skips.push_back({x, prev});
      }
    }
    x = resnet("mid_block.resnets.0", x, h, w, prev, prev);
    if (!kv_done[prev]) register_kv(prev);
    x = transformer("mid_block.attentions.0", x, ehs, h, w, prev, c.transformer_layers[2]);
    x = resnet("mid_block.resnets.1", x, h, w, prev, prev);
    for (int ui = 0; ui < 3; ++ui) {
      const int lvl = 2 - ui;
      std::string p = "up_blocks." + std::to_string(ui);
      for (int j = 0; j < c.layers_per_block + 1; ++j) {
        Skip s = skips.back();
        skips.pop_back();
        Act* cat = nullptr;
        if (pl) {
          cat = pl->new_act(x->rows, prev + s.c);
          tagseg(pl->add<ConcatOp>(x, s.a, cat), PRef());
        }
        x = resnet(p + ".resnets." + std::to_string(j), cat, h, w, prev + s.c, ch[lvl]);
        prev = ch[lvl];
        if (c.transformer_layers[lvl] > 0)
          x = transformer(p + ".attentions." + std::to_string(j), x, ehs, h, w, prev, c.transformer_layers[lvl]);
      }
      if (ui < 2) {
        Act* up = nullptr;
        UpsampleOp* uop = nullptr;
        Act* xlow = x;
        if (pl) {
          up = pl->new_act((long)B * 4 * h * w, prev);
          uop = tagseg(pl->add<UpsampleOp>(x, up, B, h, w, prev), PRef());
        }
        h *= 2;
        w *= 2;
        x = conv(p + ".upsamplers.0.conv", up, h, w, prev, prev, 1, nullptr, nullptr);
        if (pl) {
          ConvOp* cop = dynamic_cast<ConvOp*>(pl->ops.back().get());
          if (cop) { cop->enable_up2(*pl, xlow, &uop->dx); if (cop->x_low) uop->fused = cop; }
        }
      }
    }
    Act* no = groupnorm("conv_norm_out", x, h * w, prev, c.resnet_eps, 1);
    Act* out = conv("conv_out", no, h, w, prev, 8, 1, nullptr, nullptr, prev, c.out_channels);
    if (pl) pl->pred = out;
  }
};
}  // namespace

void Engine::build(Plan* plan) {
  registering = plan == nullptr;
  native_cursor = 0;
  Builder b(*this, plan);
  b.run();
  if (registering) {
    // segments: contiguous parameter ranges of ~96M elements (forward order)
    const size_t target = 96u << 20;
    seg_begin.clear();
    seg_end.clear();
    size_t start = 0;
    for (size_t i = 0; i < natives.size(); ++i) {
      size_t end = i + 1 < natives.size() ? natives[i + 1].off : param_elems;
      if (natives[i].numel >= target && natives[i].off > start) {   // a parameter larger than the target (the grouped K | V
        seg_begin.push_back(start);                                  // weight) gets a segment of its own: close the open one
        seg_end.push_back(natives[i].off);
        start = natives[i].off;
      }
      if (end - start >= target || i + 1 == natives.size()) {
        seg_begin.push_back(start);
        seg_end.push_back(end);
        start = end;
      }
    }
    nseg = (int)seg_begin.size();
    registering = false;
  } else {
    plan->gn_ws_off = plan->alloc(sizeof(float) * plan->gn_ws_floats);
    // reverse planning: the loss writes d(pred)
    plan->grad_dst(plan->pred);
    for (int i = (int)plan->ops.size() - 1; i >= 0; --i) plan->ops[i]->plan_bwd(*plan);
    plan->slab_off = plan->alloc(sizeof(float) * (plan->slab_floats ? plan->slab_floats : 4));
    plan->slab_main_off = plan->alloc(sizeof(float) * (plan->slab_main_floats ? plan->slab_main_floats : 4));
    plan->apart_off = plan->alloc(sizeof(float) * (plan->apart_floats ? plan->apart_floats : 4));
    plan->apart_side_off = plan->alloc(sizeof(float) * (plan->apart_side_floats ? plan->apart_side_floats : 4));
    plan->ln_part_off = plan->alloc(sizeof(float) * (plan->ln_part_floats ? plan->ln_part_floats : 4));
    plan->seg_first_op.assign(nseg, -1);
    plan->seg_last_op.assign(nseg, -2);
    for (int i = 0; i < (int)plan->ops.size(); ++i) {
      int s = plan->ops[i]->seg;
      if (plan->seg_first_op[s] < 0) plan->seg_first_op[s] = i;
      plan->seg_last_op[s] = i;
    }
  }
}

// ================================================================================================
// Parameter import / export (PyTorch layout <-> packed native layout)
// ================================================================================================
template <typename TS, typename TD>
__global__ void repack_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n, int kind, int co, int ci,
                              int ci_pad, int to_native) {
  // kind 0: flat copy.  kind 1: src [co][ci][3][3] <-> native [co][9][ci_pad]
  // kind 2: GEGLU projection, src rows [value 0..C4) | gate 0..C4)] <-> native rows interleaved in groups of G = ci_pad
  //         (64 or 80): channel c -> value row (c/G)*2G + c%G, gate row +G   (co = 2*C4 rows of ci elements; bias: ci = 1)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long s = i, d = i;
    if (kind == 1) {
      long o = i / ((long)ci * 9);
      long rem = i - o * (long)ci * 9;
      int c = (int)(rem / 9), tap = (int)(rem - (long)c * 9);
      long nat = (o * 9 + tap) * ci_pad + c;
      if (to_native) { s = i; d = nat; } else { s = nat; d = i; }
    } else if (kind == 2) {
      const long r = i / ci, k = i - r * ci;
      const int c4 = co / 2, half = (int)(r / c4), c = (int)(r - (long)half * c4), G = ci_pad;
      const long nat = ((long)(c / G) * (2 * G) + half * G + (c % G)) * ci + k;
      if (to_native) { s = i; d = nat; } else { s = nat; d = i; }
    }
    dst[d] = (TD)(float)src[s];
  }
}

template <typename TS, typename TD>
static int repack_launch(const TS* src, TD* dst, long n, const SrcParam& sp, int to_native, hipStream_t st) {
  long g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL((repack_kernel<TS, TD>), dim3((int)g), dim3(256), 0, st, src, dst, n, sp.kind, (int)sp.shape[0],
                     (int)sp.shape[1], sp.ci_pad, to_native);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static const SrcParam* find_src(Engine& e, const char* name) {
  auto it = e.src_index.find(name);
  if (it == e.src_index.end()) {
    sdxl_set_error("unknown parameter '%s'", name);
    return nullptr;
  }
  return &e.src[it->second];
}

int engine_load_weight(Engine& e, const char* name, const void* srcp, int dtype, hipStream_t st) {
  const SrcParam* sp = find_src(e, name);
  if (!sp) return 1;
  ARG_CHECK(e.weights, "weights are not bound");
  long n = 1;
  for (int i = 0; i < sp->ndim; ++i) n *= sp->shape[i];
  bf16* dst = e.weights + sp->native.off + sp->elem_off;
  if (dtype == 0) return repack_launch((const float*)srcp, dst, n, *sp, 1, st);
  if (dtype == 1) return repack_launch((const bf16*)srcp, dst, n, *sp, 1, st);
  ARG_CHECK(false, "dtype %d not supported (0 = fp32, 1 = bf16)", dtype);
}

int engine_export(Engine& e, const char* name, void* dstp, int dtype, bool grad, hipStream_t st) {
  const SrcParam* sp = find_src(e, name);
  if (!sp) return 1;
  long n = 1;
  for (int i = 0; i < sp->ndim; ++i) n *= sp->shape[i];
  size_t off = sp->native.off + sp->elem_off;
  if (grad) {
    ARG_CHECK(e.grads, "grads are not bound");
    const float* s = e.grads + off;
    if (dtype == 0) return repack_launch(s, (float*)dstp, n, *sp, 0, st);
    if (dtype == 1) return repack_launch(s, (bf16*)dstp, n, *sp, 0, st);
  } else {
    ARG_CHECK(e.weights, "weights are not bound");
    const bf16* s = e.weights + off;
    if (dtype == 0) return repack_launch(s, (float*)dstp, n, *sp, 0, st);
    if (dtype == 1) return repack_launch(s, (bf16*)dstp, n, *sp, 0, st);
  }
  ARG_CHECK(false, "dtype %d not supported (0 = fp32, 1 = bf16)", dtype);
}

// ================================================================================================
// Hardware layout probe: what ds_read_b64_tr_b16 and the 16x16x32 MFMA really do on this chip.
// out[0..1023]   : transpose-read result, lane l element j (as float) from an LDS image holding value = index
// out[1024..1279]: C[16][16] of A.B with A[i][k] = (i==k), B[k][j] = k*16+j  (k < 16), via the slot convention
// ================================================================================================
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
__global__ void probe_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) bf16 t[64 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 16; i += 64) t[i] = (bf16)(float)(i & 255);
  __syncthreads();
  // four 16-lane groups; group g reads the [4][16] block starting at row 4g of a 16-column image
  const int l16 = l & 15, g = l >> 4;
  const bf16* p0 = t + (g * 4 + (l16 >> 2)) * 16 + (l16 & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p0);
  union { s16x4 s; bf16x4 b; } u;
  u.s = v;
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)u.b[j];
  // MFMA probe
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = g * 8 + j;
    a[j] = (bf16)((k == l16) ? 1.f : 0.f);                 // A[i=l16][k]
    b[j] = (bf16)((k < 16) ? (float)(k * 16 + l16) : 0.f);  // B[k][j=l16]
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[1024 + l * 4 + r] = c[r];
}
int probe_layout(void* out, hipStream_t st) {
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, st, (float*)out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
