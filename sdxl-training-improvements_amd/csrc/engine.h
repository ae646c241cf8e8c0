// Static execution plan ("tape") of the SDXL UNet training step.
//
// Engine  = configuration + packed parameters (bf16 weight arena, fp32 gradient arena) shared by all plans.
// Plan    = one bucket shape (B, H, W): every activation, gradient, statistic and scratch buffer has a fixed
//           offset in one workspace arena, every op holds resolved offsets, and forward / backward are plain
//           in-order / reverse-order walks that only launch kernels (no allocation, no host sync, no autograd).
// Gradient buffers are assigned while walking the tape in reverse: the first writer of a tensor's gradient
// overwrites, later writers accumulate, and a residual connection's gradient is an alias of its output's.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/sdxlstep.h"
#include "kernels.h"

static const size_t NONE = (size_t)-1;

struct Act {           // bf16 activation [rows][cols] in the workspace
  size_t off = NONE;   // byte offset of data
  size_t goff = NONE;  // byte offset of gradient (assigned in reverse planning)
  long rows = 0;
  int cols = 0;
  int pad_rows = 0;    // rows allocated beyond `rows` in the data and the gradient buffer (a weight-gradient GEMM over the rows can
                       // then run a whole number of 64-row reduction steps: the op zeroes them before it reads them)
  bool need_grad = true;
  // column-slice view of a wider tensor (one block's K | V columns of the grouped cross-attention projection):
  // data / gradient live inside the parent's buffers, row stride = the parent's width
  Act* parent = nullptr;
  int col0 = 0;
  long ld() const { return parent ? parent->cols : cols; }
};

struct PRef {  // native parameter: element offset into the bf16 weight arena and the fp32 grad arena
  size_t off = NONE;
  size_t numel = 0;
};

struct SrcParam {  // one diffusers state-dict tensor and where it lives in the native arena
  std::string name;
  int ndim;
  long shape[4];
  PRef native;
  int kind;  // 0: rows copied at elem_off ; 1: conv3x3 [co][ci][3][3] -> [co][9][ci_pad]
  size_t elem_off;
  int ci_pad;
};

struct Engine;
struct Plan;

struct Op {
  int seg = 0;
  bool hoist_fwd = false;   // forward depends on the inputs only (cross-attention K/V projections of the prompt
                            // embeddings): launched on the side stream at the start of the forward
  bool needs_hoisted = false;  // consumes a hoisted op's output: main stream waits for the side stream first
  virtual ~Op() {}
  virtual int fwd(Plan& p, hipStream_t st) = 0;
  virtual void plan_bwd(Plan& p) = 0;
  virtual int bwd(Plan& p, hipStream_t st, bool first_micro) = 0;
};

struct Plan {
  Engine* eng = nullptr;
  int B = 0, H = 0, W = 0, ctx = 0;
  size_t ws_bytes = 0;   // bytes needed
  size_t cursor = 0;
  std::vector<std::unique_ptr<Act>> acts;
  std::vector<std::unique_ptr<Op>> ops;
  std::vector<int> seg_first_op, seg_last_op;  // op index ranges per segment (forward order)
  size_t tp32_off = NONE, tp32_bytes = 0;   // fp32 [B][sum Cout] time-embedding-projection gradient (zeroed at the start of a backward;
                                            // every resnet's conv1 backward adds its per-sample column sums into its slice)
  // well-known buffers
  Act *x_in = nullptr, *pred = nullptr, *ehs = nullptr, *aug_in = nullptr, *te_sin = nullptr, *tid_emb = nullptr;
  size_t t_off = NONE, tid_off = NONE, loss_off = NONE, loss_part_off = NONE;
  size_t in_lat_off = NONE, in_noise_off = NONE, in_sig_off = NONE, in_tag_off = NONE;   // staged copies of the step's inputs (fixed
                                                                                          // addresses for the captured graphs)
  size_t gn_ws_off = NONE, gn_ws_floats = 0;  // GroupNorm scratch shared by all (stream-ordered) norm ops
  size_t slab_off = NONE, slab_floats = 0;    // split-K partial slabs of the wgrad GEMMs (shared, stream-ordered)
  size_t slab_main_off = NONE, slab_main_floats = 0;   // the same for split forward / dgrad launches (caller's stream)
  size_t apart_off = NONE, apart_floats = 0;  // query-split partials of the self-attention dK / dV kernel (caller's stream)
  size_t apart_side_off = NONE, apart_side_floats = 0;   // ... of the cross-attention dK / dV kernel (side stream): never shared
  // LayerNorm backward inside the dgrad epilogue of the linear layer its output feeds (GemmP::ln_x): the tagged per-row partial sums the
  // column tiles of a row block exchange; shared, stream-ordered; zeroed at the start of a backward, whose fused launches count the epoch up
  size_t ln_part_off = NONE, ln_part_floats = 0;
  int ln_epoch = 0;

  Act* new_act(long rows, int cols, bool need_grad = true, int pad_rows = 0);
  Act* view(Act* parent, int col0, int cols);   // columns [col0, col0 + cols) of parent
  size_t alloc(size_t bytes);
  // reverse planning.  Gradient buffers are WRITE-ONCE: the first writer of a tensor's gradient gets a fresh buffer,
  // every later writer gets another fresh buffer plus the previous one as addend (out = addend + contribution), so a
  // buffer is never modified after it has been produced -- deferred / concurrent readers (the wgrad GEMMs on the
  // side stream) need no ordering against later accumulations.
  struct GradDst { size_t out = NONE, addend = NONE; };
  GradDst grad_dst(Act* a);
  bool grad_alias(Act* x, Act* y);  // x.g := y.g when x has none yet (residual pass-through, no copy)
  bf16* GP(size_t off) const;       // workspace pointer of a gradient offset (nullptr for NONE)
  template <class T, class... A>
  T* add(A&&... a);
  // pointers (valid once a workspace is bound)
  bf16* P(const Act* a) const;
  bf16* G(const Act* a) const;
  float* F(size_t off) const;
};

struct Engine {
  sdxl_unet_config cfg;
  int device = 0;
  // parameters
  std::vector<PRef> natives;  // in creation (= forward) order
  std::vector<SrcParam> src;
  std::map<std::string, int> src_index;
  size_t param_elems = 0;
  bf16* weights = nullptr;
  float* grads = nullptr;
  bool own_weights = false, own_grads = false;
  // segments: contiguous native-parameter ranges (forward order)
  int nseg = 1;
  std::vector<size_t> seg_begin, seg_end;
  // plans
  std::map<std::tuple<int, int, int, int>, std::unique_ptr<Plan>> plans;
  Plan* cur = nullptr;
  char* ws = nullptr;
  size_t ws_cap = 0;
  bool own_ws = false;
  // backward concurrency: weight-gradient GEMMs (and bias column sums) run on a side stream, off the dgrad chain
  hipStream_t side = nullptr;
  hipEvent_t ev_join = nullptr, ev_hoist = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  bool use_side = true;
  bf16* emit_base = nullptr;     // sdxl_set_grad_emit: weight-gradient GEMMs write their final value as bf16 into this arena (same
  float emit_scale = 1.f;        // element offsets as `grads`) instead of the fp32 arena -- the exchange micro-step under data parallelism
  bool join_last_only = false;   // sdxl_set_join_mode 1 / 2: main waits for the side stream at the last segment only
  bool seg_on_side = false;      // sdxl_set_join_mode 2: at every segment end the SIDE stream waits for main instead; the caller
                                 // enqueues the segment's cast + collective on the side stream (sdxl_side_stream)
  hipEvent_t ev_seg = nullptr;
  hipEvent_t next_event();
  std::vector<std::vector<GemmP>> wg_pending;   // weight gradients waiting for a grouped launch, one bucket per shape
  int defer_wgrad(Plan& p, hipStream_t main, const GemmP& g, int grp);
  int flush_wgrads(Plan& p, hipStream_t main);
  // leaf launches for the side stream that may start any time after the point they were queued at (their inputs are write-once
  // buffers produced by launches already enqueued on the caller's stream): they ride on the NEXT fork event instead of paying for
  // one of their own (LayerNorm parameter gradients, cross-attention dK / dV: 280 events per step)
  std::vector<std::function<int(hipStream_t)>> side_leaves;
  std::vector<LnRedEntry> ln_pending;     // LayerNorm backward launches whose dgamma | dbeta partials are not reduced yet
  std::vector<LnRedEntry> cs_pending;     // per-sample column sums (time-embedding row vectors) whose partial rows are not reduced yet: the consumer's cast flushes them
  int flush_ln_params(Plan& p, hipStream_t main);
  // Everything queued for a later launch holds raw workspace pointers of the plan and step it was queued in: dropped at the start of a
  // forward / of a backward, when a backward segment fails half way, and when the current plan changes (never launched against another
  // step's or plan's buffers).
  void drop_pending() { side_leaves.clear(); wg_pending.clear(); ln_pending.clear(); cs_pending.clear(); }
  bool side_dirty = false;       // the side stream has work the caller's stream has not joined yet
  // hipGraph replay of the step (sdxl_set_graph_mode, OFF by default): forward (+ loss) and backward are captured once per
  // (plan, configuration) -- both streams, every event edge -- and replayed with one hipGraphLaunch, on an engine-owned
  // stream (the caller's may be the legacy default stream, which cannot be captured) fenced by two events.  Measured on
  // ROCm 7.2 / MI355X (r02, same box): B=4 1024^2 eager 124.1 ms/step, forward graph only 124.6, backward graph 142.8, both
  // 143.9; B=1 512^2 (3,400 launches in 61 ms) eager 61.5, forward only 62.5, both 67.5 -- the runtime replays a
  // two-branch graph with LESS overlap between the branches than two streams give, and no cheaper per kernel.  Kept as an
  // opt-in for runtimes where that changes; the step's gaps are the dispatcher's, not this host loop's.
  bool use_graphs = false;
  hipStream_t gstream = nullptr;
  hipEvent_t ev_gin = nullptr, ev_gout = nullptr;
  struct GraphKey {
    const void* plan; int kind, k, first, join; unsigned scale_bits, cfg;
    bool operator<(const GraphKey& o) const {
      return std::tie(plan, kind, k, first, join, scale_bits, cfg) < std::tie(o.plan, o.kind, o.k, o.first, o.join, o.scale_bits, o.cfg);
    }
  };
  struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; };
  std::map<GraphKey, GraphEntry> graphs;
  void clear_graphs() {
    for (auto& kv : graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    graphs.clear();
  }
  // builder state
  bool registering = true;
  size_t native_cursor = 0;
  Plan* bp = nullptr;  // plan being built (nullptr while registering parameters)

  PRef param(size_t numel, bool small = false);   // small: bias / norm vectors, accumulated with atomics -> zeroed per cycle
  std::vector<std::pair<size_t, size_t>> small_ranges;   // (elem offset, elems) of the small parameters
  unsigned long long* small_ranges_dev = nullptr;         // same list on the device (pairs), for the zeroing kernel
  void map_src(const std::string& name, std::vector<long> shape, PRef p, int kind, size_t elem_off, int ci_pad);
  int seg_of(size_t elem_off) const;
  void build(Plan* plan);  // registers parameters (plan == nullptr) or builds a plan
  bf16* Wp(PRef p) const { return weights + p.off; }
  float* Gp(PRef p) const { return grads + p.off; }
};

inline bf16* Plan::P(const Act* a) const { return (bf16*)(eng->ws + a->off); }
inline bf16* Plan::G(const Act* a) const { return a->goff == NONE ? nullptr : (bf16*)(eng->ws + a->goff); }
inline float* Plan::F(size_t off) const { return (float*)(eng->ws + off); }
inline bf16* Plan::GP(size_t off) const { return off == NONE ? nullptr : (bf16*)(eng->ws + off); }
template <class T, class... A>
T* Plan::add(A&&... a) {
  T* o = new T(std::forward<A>(a)...);
  ops.emplace_back(o);
  return o;
}

int engine_load_weight(Engine& e, const char* name, const void* src, int dtype, hipStream_t st);
int engine_export(Engine& e, const char* name, void* dst, int dtype, bool grad, hipStream_t st);
int probe_layout(void* out, hipStream_t st);
