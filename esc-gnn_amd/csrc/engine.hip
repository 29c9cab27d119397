// engine.hip — one host call = one NestedGIN_eff training step (forward + L1 + backward) or one
// eval-mode forward.  Orchestrates the kernels of this library from C++ so that the per-op host
// overhead of the Python/autograd path disappears, and applies the cross-op fusions a per-op
// interface cannot express:
//   * BatchNorm(+ReLU) is applied by the CONSUMER: esc_bn_stats emits (scale, shift) and the next
//     GEMM applies relu(x*scale+shift) while staging its A (or, for weight gradients, B) operand.
//     z_emb (E x H) and the hidden activation of every MLP are never written to memory; the backward
//     recomputes the ReLU mask from the pre-BN value.
//   * layer outputs are materialised directly into their slice of the [N,(L+1)H] concatenation buffer
//     (reference: torch.cat(xs, dim=1), run_graphcount.py:181), and d(cat) slices are consumed in place;
//     the aggregate backward accumulates dx into the previous layer's slice.
//   * d(z_emb) = sum_l d_e_l * W_lin_l accumulates inside the dX GEMM epilogue.
// Model composition followed: /root/reference/run_graphcount.py:134-194 (graph_pred=False, dropout=0,
// use_cycle=True — the configuration run_graphcount.py:465 instantiates).
#include "common.h"

#include <cstring>
#include <map>
#include <vector>
#include <cstdlib>

namespace esc {

struct Arena {
  float* base;
  int64_t off;
  float* take(int64_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~63LL;          // 256-byte granules keep every buffer float4-aligned
    return p;
  }
};

struct BnWs { float *mean, *invstd, *scale, *shift, *nglob, *coef; };     // nglob: rows over all ranks (SyncBN); coef: float2[C] of the backward
struct MlpWs { float *Y0, *Y1; BnWs b0, b1; float* A1 = nullptr; };   // A1: materialised hidden activation (ELU models)

struct Layout {
  int64_t N, E, Z, H, L, C0, W;   // W = (L+1)*H
  // forward state
  float *Zb, *Yz; BnWs zb0, zb1;
  float *A0, *Zemb;               // relu(BN(Zb)) and z_emb = relu(BN(Yz)), materialised when g_materialise_edge_act
  float* e[ESC_MAX_LAYERS]; float* agg[ESC_MAX_LAYERS]; MlpWs conv[ESC_MAX_LAYERS];
  int64_t ld_e[ESC_MAX_LAYERS];   // leading dimension of e[l]: C, or (L-1)*H when the H-wide edge terms are column blocks of ONE matrix
  float *e_cat, *w_cat;           // g_edge_batched: [E, (L-1)*H] and the packed [(L-1)*H + (L-1), H] weights ++ biases
  MlpWs xemb; float *cat, *Yl; BnWs bl; float *pred, *dpred;
  float *Ypart;                   // readout: lin1 over the concat slices that are final before the last layer (g_readout_split)
  // backward scratch
  float *dcat, *dAl, *dT1, *dT2, *dagg, *dZemb, *dAz, *deps_part;
  float *dT1_l[ESC_MAX_LAYERS], *dT2_l[ESC_MAX_LAYERS];   // per-layer copies: read by weight-gradient tiles that run behind the node chain (g_wgrad_stream)
  float* d_e[ESC_MAX_LAYERS];      // one per GINE layer: the edge stream consumes d_e[l] while the node chain moves on
  float *bn_scratch, *bag_scratch, *slabs;
  float *col_stats;               // GEMM-epilogue BatchNorm partials: float2[ceil(rows/32)][H]
  float *col_stats_b;             // second set: a folded BatchNorm's partials live until its consumer has run
  float *bst_part;                // BatchNorm-backward column sums left by a dX epilogue / an aggregate backward: float2[slots][H]
  float *bst_part_e;              // ... of the edge pipeline (z_embedding's first BatchNorm, from its Linear's dX epilogue)
  // private scratch of the x_embedding branch (runs on a side stream next to the z/conv chain)
  float *bn_scratch_x, *dT1x, *dT2x, *slabs_x;
  float *bn_scratch_e, *col_stats_e;   // the edge stream's own BatchNorm scratch / GEMM-epilogue partials
  // ZINC variant: node features from a table, [z_emb | edge type] edge-term input, pooled readout
  float *cat_scale, *cat_shift;     // (scale, shift) of the BatchNorm that produced each H-wide slice of cat: [(L+1)*H]
  float *X0, *dX0, *Zcat, *dZcat, *pooled, *dpool, *Al;
  int64_t G, D, Wz;                 // graphs, type-embedding width, Wz = H + D
  int64_t total;
};

// SURVEY section 7 step 6, built to be measured: the H-wide edge terms e_1 .. e_{L-1} = lin_l(z_emb) of ALL layers as ONE GEMM
// [E, H] x [H, (L-1)*H] over packed weights (one launch of 119 x 6 tiles instead of three of 119 x 2), their outputs column blocks of
// one matrix.  ESC_EDGE_BATCHED=1; see DESIGN.md for what it measured.
// The readout Linear reduces over all (L+1)*H concat columns (K = 1280: 152 workgroups walking 40 K-steps, 33 us — the longest kernel of
// the node forward).  L of its L+1 slices are final before the LAST layer starts: their share is computed on the idle edge stream while
// that layer runs, and the launch on the node chain starts from it and only reduces over the last slice (esc_linear_fwd_from).
static int g_l1_head = getenv("ESC_L1_HEAD") ? atoi(getenv("ESC_L1_HEAD")) : 1;     // see train_step_impl()
static int g_ogb_bnb = getenv("ESC_OGB_BNB") ? atoi(getenv("ESC_OGB_BNB")) : 0;
static int g_readout_split = getenv("ESC_READOUT_SPLIT") ? atoi(getenv("ESC_READOUT_SPLIT")) : 0;    // measured neutral (1.012-1.016 ms either way): off
static int g_skip_waits = getenv("ESC_SKIP_WAITS") ? atoi(getenv("ESC_SKIP_WAITS")) : 1;   // -7 us of step time
static int g_edge_batched = getenv("ESC_EDGE_BATCHED") ? atoi(getenv("ESC_EDGE_BATCHED")) : 1;
static BnWs take_bn(Arena& a, int64_t C) { BnWs w; w.mean = a.take(C); w.invstd = a.take(C); w.scale = a.take(C); w.shift = a.take(C); w.nglob = a.take(16); w.coef = a.take(2 * C); return w; }

static Layout plan_layout(const esc_nested_gin_t* m, int64_t N, int64_t E, int64_t Z, float* base, bool train) {
  Layout y{};
  Arena a{base, 0};
  const int64_t H = m->hidden, L = m->num_layers, C0 = m->in_dim;
  y.N = N; y.E = E; y.Z = Z; y.H = H; y.L = L; y.C0 = C0; y.W = (L + 1) * H;
  y.Zb = a.take(E * H); y.Yz = a.take(E * H); y.zb0 = take_bn(a, H); y.zb1 = take_bn(a, H);
  y.A0 = a.take(E * H); y.Zemb = a.take(E * H);
  const bool batched = g_edge_batched && L >= 3;            // (two or more H-wide edge terms)
  if (batched) { y.e_cat = a.take(E * (L - 1) * H); y.w_cat = a.take(((L - 1) * H + (L - 1)) * H); }
  for (int l = 0; l < L; ++l) {
    const int64_t C = l == 0 ? C0 : H;
    if (batched && l >= 1) { y.e[l] = base ? y.e_cat + (int64_t)(l - 1) * H : nullptr; y.ld_e[l] = (L - 1) * H; }
    else { y.e[l] = a.take(E * C); y.ld_e[l] = C; }
    y.agg[l] = a.take(N * C);
    y.conv[l].Y0 = a.take(N * H); y.conv[l].Y1 = a.take(N * H);
    y.conv[l].b0 = take_bn(a, H); y.conv[l].b1 = take_bn(a, H);
  }
  y.xemb.Y0 = a.take(N * H); y.xemb.Y1 = a.take(N * H); y.xemb.b0 = take_bn(a, H); y.xemb.b1 = take_bn(a, H);
  y.cat = a.take(N * y.W); y.Yl = a.take(N * H); y.bl = take_bn(a, H);
  y.Ypart = a.take(N * H);
  y.cat_scale = a.take(y.W); y.cat_shift = a.take(y.W);       // the slices' BatchNorm coefficients side by side (readout prologue)
  if (base) {
    y.xemb.b1.scale = y.cat_scale; y.xemb.b1.shift = y.cat_shift;
    for (int l = 0; l < L; ++l) { y.conv[l].b1.scale = y.cat_scale + (int64_t)(l + 1) * H; y.conv[l].b1.shift = y.cat_shift + (int64_t)(l + 1) * H; }
  }
  y.pred = a.take(N); y.dpred = a.take(N);
  y.bn_scratch = a.take(esc_bn_scratch(H));
  y.bn_scratch_x = a.take(esc_bn_scratch(H));
  y.col_stats = a.take(2 * ((E > N ? E : N) / 32 + 1) * H);
  y.col_stats_b = a.take(2 * (N / 32 + 1) * H);
  y.bn_scratch_e = a.take(esc_bn_scratch(H));
  y.col_stats_e = a.take(2 * (E / 32 + 1) * H);
  if (train) {
    y.bst_part = a.take(2 * (N / 4 + 2) * H);
    y.bst_part_e = a.take(2 * (E / 64 + 2) * H);
    y.dT1x = a.take(N * H); y.dT2x = a.take(N * H);
    y.slabs_x = a.take(esc_linear_bwd_weight_scratch(N, H, H));
    y.dcat = a.take(N * y.W); y.dAl = a.take(N * H); y.dT1 = a.take(N * H); y.dT2 = a.take(N * H);
    for (int l = 0; l < L; ++l) { y.dT1_l[l] = a.take(N * H); y.dT2_l[l] = a.take(N * H); }
    y.dagg = a.take(N * H); y.dZemb = a.take(E * H); y.dAz = a.take(E * H);
    for (int l = 0; l < L; ++l) y.d_e[l] = a.take(E * (l == 0 ? C0 : H));
    y.deps_part = a.take(2 * N * (L > 0 ? L : 1));        // one vector per GINE layer, summed together at the end
    y.bag_scratch = a.take(esc_bag_bwd_scratch(Z, H));
    // one private slab region per weight gradient: their ordered reduces are deferred to ONE launch at the end
    int64_t sl = esc_linear_bwd_weight_scratch(E, H, H) + 64;                        // zlin
    for (int l = 0; l < L; ++l) {
      const int64_t C = l == 0 ? C0 : H;
      sl += esc_linear_bwd_weight_scratch(E, C, H) + esc_linear_bwd_weight_scratch(N, H, H) +
            esc_linear_bwd_weight_scratch(N, H, C) + 3 * 64;                           // conv.lin, nn.lin1, nn.lin0
    }
    sl += esc_linear_bwd_weight_scratch(N, H, H) + esc_linear_bwd_weight_scratch(N, H, C0) + 2 * 64;   // x_embedding
    sl += esc_linear_bwd_weight_scratch(N, H, y.W) + esc_linear_bwd_weight_scratch(N, 1, H) + 2 * 64;  // lin1, lin2
    sl += esc_linear_bwd_weight_scratch(N, H, 1) + 64;      // lin1 is reduced as two column blocks with a job each
    y.slabs = a.take(sl);
  }
  y.total = a.off;
  return y;
}

#define ESC_TRY(call)            \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != ESC_OK) return rc__; \
  } while (0)

struct Ctx {
  const esc_nested_gin_t* m;
  const esc_batch_t* b;
  Layout y;
  void* s;
  bool train;
  std::vector<esc_reduce_job>* jobs = nullptr;   // deferred weight-gradient reduces (main chain only)
  float** slab_cursor = nullptr;
  bool on_edge_stream = false;
  int act = 1;                                   // 1 ReLU (counting model), 2 ELU (ZINC): materialised activations
  void* wgrad = nullptr;                         // != NULL: the node chain's weight-gradient tiles go to this stream (backward())
  const float* l1_target = nullptr;              // train_step: the prediction head also leaves d L1 / d pred (forward(), g_l1_head)
  int64_t l1_denom = 0;
};
// While it lives, node-sized Linear backwards launched for `c` put their dW tiles on c.wgrad (esc_linear_bwd_set_wgrad_stream)
struct WgradScope {
  explicit WgradScope(const Ctx& c) : on_(c.wgrad != nullptr && !c.on_edge_stream && c.jobs != nullptr) { if (on_) (void)esc_linear_bwd_set_wgrad_stream(c.wgrad); }
  ~WgradScope() { if (on_) (void)esc_linear_bwd_set_wgrad_stream(nullptr); }
  bool on_;
};

// dX + dW tiles now, slab reduce deferred (or immediate when the context has no job list)
static int g_edge_ahead = 1;      // edge terms one layer ahead of the node chain (bit 5 of esc_engine_set_side_stream: two)
static int g_cap_tail = 0;        // bit 4: ... and the z_embedding GEMM of the backward tail
static int g_cap_forward = 1;     // the occupancy cap also applies to the forward's edge GEMMs (bit 3: backward only)
struct LdsFloorGuard {            // occupancy cap for the GEMMs launched while it lives (edge stream only)
  explicit LdsFloorGuard(bool on) : on_(on && edge_lds_floor() > 0) { if (on_) set_gemm_lds_floor(edge_lds_floor()); }
  ~LdsFloorGuard() { if (on_) set_gemm_lds_floor(0); }
  bool on_;
};

// Where the node chain's GEMM workgroups land.  An edge-stream GEMM (128x128 tile, 96 KB of LDS, one workgroup per CU) leaves 64 KB
// of its CU's LDS free, so a 54-KB node workgroup co-resides with it and the two share the CU's MFMA pipe.  A node workgroup that
// ASKS for 66 KB (dynamic LDS it does not use) is only placed on CUs without an edge workgroup: in the backward — dual launches of 392
// workgroups beside 57-us edge launches of 478 — that is worth 15 us of the node chain (phase marks: node backward 477 -> 462 us,
// step 0.995 -> 0.980 ms; bench 1.048 -> 1.031 ms); 60 000 bytes (still co-resident) changes nothing, 82 000 (one node workgroup per
// CU) costs 50 us, and in the forward (152-workgroup launches) the same request costs 8 us.  ESC_NODE_LDS_FLOOR_FWD / _BWD, bytes.
static int g_node_floor_fwd = getenv("ESC_NODE_LDS_FLOOR_FWD") ? atoi(getenv("ESC_NODE_LDS_FLOOR_FWD")) : 0;
static int g_node_floor_bwd = getenv("ESC_NODE_LDS_FLOOR_BWD") ? atoi(getenv("ESC_NODE_LDS_FLOOR_BWD")) : 66 * 1024;
struct NodeFloorGuard {
  explicit NodeFloorGuard(int bytes) : on_(bytes > 0) { if (on_) set_node_lds_floor(bytes); }
  ~NodeFloorGuard() { if (on_) set_node_lds_floor(0); }
  bool on_;
};

static int linear_backward(const Ctx& c, const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* sc,
                           const float* sh, const esc_linear_t& lin, int64_t M, float* dX, int64_t ld_dx, int accumulate) {
  const LdsFloorGuard cap(c.on_edge_stream);
  const WgradScope side(c);
  const int64_t N = lin.out_dim, K = lin.in_dim;
  if (c.jobs == nullptr)
    return esc_linear_bwd_both(dY, ld_dy, X, ld_x, sc, sh, lin.w, K, M, N, K, dX, ld_dx, accumulate, lin.dw, K, lin.db,
                               c.y.slabs, c.s);
  float* slabs = *c.slab_cursor;
  *c.slab_cursor += (esc_linear_bwd_weight_scratch(M, N, K) + 63) & ~63LL;
  c.jobs->emplace_back();
  return esc_linear_bwd_both_deferred(dY, ld_dy, X, ld_x, sc, sh, lin.w, K, M, N, K, dX, ld_dx, accumulate, lin.dw, K, lin.db,
                                      slabs, &c.jobs->back(), c.s);
}

// ---- BatchNorm(+ReLU) backward folded into the Linear backward behind it (r03; esc_linear_bwd_both_bn) ------------------
// The node chain's backward was, per Linear -> BatchNorm -> ReLU pair, partial -> finalize -> apply -> dX+dW: four dependent
// launches.  The apply now happens while the GEMM stages its dY operand (bit 0) and the column sums of an MLP's FIRST
// BatchNorm come out of the dX epilogue of its second Linear (bit 1): partial/finalize -> dX+dW -> finalize -> dX+dW.
// ESC_BN_FUSE_BWD=0 restores the elementwise launches (A/B runs, bisecting).
static int g_bn_fuse_bwd = getenv("ESC_BN_FUSE_BWD") ? atoi(getenv("ESC_BN_FUSE_BWD")) : 11;      // bit 2: the edge tail, see backward() (measured: no gain, off)
static int g_bn_fuse_elu = getenv("ESC_BN_FUSE_ELU") ? atoi(getenv("ESC_BN_FUSE_ELU")) : 0;
static esc_bn_bwd_fused bn_fused(const float* x, int64_t ld_x, const BnWs& w, int relu) {
  return esc_bn_bwd_fused{x, ld_x, w.mean, w.invstd, w.scale, w.shift, w.coef, relu};
}
static int linear_backward_bn(const Ctx& c, const float* dOut, int64_t ld_dout, const esc_bn_bwd_fused& f, const float* X, int64_t ld_x,
                              const float* sc, const float* sh, const esc_linear_t& lin, int64_t M, float* dX, int64_t ld_dx,
                              int accumulate, const esc_bn_bwd_next* next) {
  const LdsFloorGuard cap(c.on_edge_stream);
  const WgradScope side(c);
  const int64_t N = lin.out_dim, K = lin.in_dim;
  if (c.jobs == nullptr)
    return esc_linear_bwd_both_bn(dOut, ld_dout, &f, X, ld_x, sc, sh, lin.w, K, M, N, K, dX, ld_dx, accumulate, lin.dw, K, lin.db, c.y.slabs,
                                  nullptr, next, c.s);
  float* slabs = *c.slab_cursor;
  *c.slab_cursor += (esc_linear_bwd_weight_scratch(M, N, K) + 63) & ~63LL;
  c.jobs->emplace_back();
  return esc_linear_bwd_both_bn(dOut, ld_dout, &f, X, ld_x, sc, sh, lin.w, K, M, N, K, dX, ld_dx, accumulate, lin.dw, K, lin.db, slabs,
                                &c.jobs->back(), next, c.s);
}

// The x_embedding MLP depends only on x (forward) / on d(cat)[:, 0:H] (backward): five to eight small,
// latency-bound launches that overlap perfectly with the edge-sized work of the main chain.  They run on a
// second HIP stream with their own scratch, forked and joined with events (a capturable fork/join).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork_f = nullptr, join_f = nullptr, fork_b = nullptr, join_b = nullptr;
  bool ok = false;
};
// Edge-sized activations: the affine+ReLU prologue costs the edge-row GEMMs ~25 % (VALU-bound staging), more
// than the two extra elementwise passes that materialise them once; node-sized MLP activations stay fused.
static int g_materialise_edge_act = 1;
static int g_use_side_stream = 0;     // esc_engine_set_side_stream(); measured neutral-to-negative on MI355X r01

// The edge-sized conv.lin GEMMs (e_l = lin_l(z_emb) forward; dX/dW of lin_l backward) depend on the node chain only
// through e_l / d_e_l: they run on a second HIP stream and fill the CUs that the latency-bound node-sized launches
// (152 workgroups on 256 CUs) leave idle.  One event per dependency, no host synchronisation.
struct EdgeStream {
  hipStream_t stream = nullptr;
  hipEvent_t z_ready = nullptr, joined = nullptr, lin1_fork = nullptr, lin1_rest = nullptr, tail_dz = nullptr, e_ready[ESC_MAX_LAYERS] = {}, de_ready[ESC_MAX_LAYERS] = {},
             agg_done[ESC_MAX_LAYERS] = {};
  bool ok = false;
};
// 1: the weight gradient of the LAST conv.lin backward (l == 0, in the tail of the step) runs on the node stream.  Measured
// neutral (1.038 vs 1.031-1.044 ms by the phase marks; so was doing the same for z_embedding's Linear): the node stream's
// own reductions then become the end of the step.  Off.
static int g_split_last_lin = getenv("ESC_SPLIT_LAST_LIN") ? atoi(getenv("ESC_SPLIT_LAST_LIN")) : 1;
static int g_edge_priority_low = 1;
static int g_use_edge_stream = 1;     // esc_engine_set_side_stream() bit 1
static int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}
static EdgeStream& edge_stream() {
  static thread_local std::map<int, EdgeStream> per_device;   // streams / events belong to the device that was current when they were made
  static thread_local EdgeStream off;   // ok == false
  if (!g_use_edge_stream) return off;
  EdgeStream& es = per_device[current_device()];
  if (!es.ok && es.stream == nullptr) {
    // lowest priority: when workgroup slots free up, the latency-critical node chain is served first
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    // ESC_EDGE_CU_MASK (experiment, DESIGN.md *Step engine, r03* (vi)): the edge stream on a subset of the CUs.  "low:N" = the first N
    // mask bits (the driver deals mask bits round-robin over the 8 XCDs: N/8 CUs of every XCD), "xcd:K" = the CUs of XCDs 0..K-1.
    bool good = false;
    if (const char* mk = getenv("ESC_EDGE_CU_MASK")) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int n = atoi(strchr(mk, ':') ? strchr(mk, ':') + 1 : "0");
      for (int i = 0; i < 256; ++i) {
        const bool on = strncmp(mk, "xcd", 3) == 0 ? (i % 8) < n : i < n;
        if (on) mask[i / 32] |= 1u << (i % 32);
      }
      good = n > 0 && hipExtStreamCreateWithCUMask(&es.stream, 8, mask) == hipSuccess;
    } else {
      good = hipStreamCreateWithPriority(&es.stream, hipStreamNonBlocking, g_edge_priority_low ? least : greatest) == hipSuccess;
    }
    good = good && hipEventCreateWithFlags(&es.z_ready, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&es.joined, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&es.lin1_fork, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&es.lin1_rest, hipEventDisableTiming) == hipSuccess;
    good = good && hipEventCreateWithFlags(&es.tail_dz, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l < ESC_MAX_LAYERS; ++l) {
      good = good && hipEventCreateWithFlags(&es.e_ready[l], hipEventDisableTiming) == hipSuccess;
      good = good && hipEventCreateWithFlags(&es.de_ready[l], hipEventDisableTiming) == hipSuccess;
      good = good && hipEventCreateWithFlags(&es.agg_done[l], hipEventDisableTiming) == hipSuccess;
    }
    es.ok = good;
  }
  return es;
}
// The engines use the second stream only when the batch has enough edges for the edge-sized kernels to matter:
// ZINC at bs=128 (6 400 edges) measured 1.24 ms on one stream and 1.35 ms on two (the events cost more than the overlap
// returns); ogbg-molhiv at bs=256 (20 000 edges, emb 300) 5.44 -> 5.18 ms.
static int64_t g_two_stream_min_edges = 12000;       // esc_engine_set_two_stream_min_edges()
static EdgeStream& edge_stream_for(int64_t edges) {
  static thread_local EdgeStream off;   // ok == false
  return edges >= g_two_stream_min_edges ? edge_stream() : off;
}
// `waiter` continues only after everything queued on `src` so far
static int chain(hipEvent_t ev, hipStream_t src, hipStream_t waiter) {
  if (hipEventRecord(ev, src) != hipSuccess || hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) {
    set_error("esc_engine: stream event failed");
    return ESC_ELAUNCH;
  }
  return ESC_OK;
}

static SideStream& side_stream() {
  static thread_local std::map<int, SideStream> per_device;
  static thread_local SideStream off;   // ok == false
  if (!g_use_side_stream) return off;
  SideStream& ss = per_device[current_device()];
  if (!ss.ok && ss.stream == nullptr) {
    bool good = hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t* evs[4] = {&ss.fork_f, &ss.join_f, &ss.fork_b, &ss.join_b};
    for (auto e : evs) good = good && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    ss.ok = good;
  }
  return ss;
}
// The node chain's backward waits for a Linear's dX only; its dW tiles (the larger half of the dual launch: 240 of 392 workgroups
// at 2400 rows) are needed by the optimiser.  With this on they run on a third, lowest-priority stream behind the chain
// (esc_linear_bwd_set_wgrad_stream), the scratch rows they read are per-layer copies, and the node-side slab reduce joins it.
static int g_wgrad_stream = getenv("ESC_WGRAD_STREAM") ? atoi(getenv("ESC_WGRAD_STREAM")) : 0;
struct WgradStream {
  hipStream_t stream = nullptr;
  hipEvent_t joined = nullptr;
  bool ok = false;
};
static WgradStream& wgrad_stream() {
  static thread_local std::map<int, WgradStream> per_device;
  static thread_local WgradStream off;   // ok == false
  if (!g_wgrad_stream) return off;
  WgradStream& ws = per_device[current_device()];
  if (!ws.ok && ws.stream == nullptr) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    bool good = hipStreamCreateWithPriority(&ws.stream, hipStreamNonBlocking, least) == hipSuccess;
    good = good && hipEventCreateWithFlags(&ws.joined, hipEventDisableTiming) == hipSuccess;
    ws.ok = good;
  }
  return ws;
}
static Ctx edge_ctx(const Ctx& c, hipStream_t edge) {       // same job list / slab cursor: host-side bookkeeping only
  Ctx x = c;
  x.s = edge;
  x.on_edge_stream = true;
  x.wgrad = nullptr;
  x.y.bn_scratch = c.y.bn_scratch_e;
  x.y.col_stats = c.y.col_stats_e;
  return x;
}
static Ctx side_ctx(const Ctx& c, hipStream_t side) {
  Ctx x = c;
  x.s = side;
  x.y.bn_scratch = c.y.bn_scratch_x;
  x.y.dT1 = c.y.dT1x; x.y.dT2 = c.y.dT2x; x.y.slabs = c.y.slabs_x;
  x.jobs = nullptr; x.slab_cursor = nullptr; x.wgrad = nullptr;
  return x;
}

// ---- phase marks (diagnostics, ESC_PHASE_TIMING=1): timing-enabled events at fixed points of the two pipelines, kept in a
// ring and read only by esc_engine_phase_times() after the caller has synchronised — no host/device sync inside the loop.
enum { PH_START = 0, PH_EDGE_FWD_DONE, PH_NODE_FWD_DONE, PH_NODE_BWD_DONE, PH_EDGE_BWD_DONE, PH_END, PH_COUNT };
struct PhaseRing {
  static constexpr int RING = 128;
  hipEvent_t ev[RING][PH_COUNT] = {};
  int steps = 0;
  bool on = getenv("ESC_PHASE_TIMING") != nullptr;
};
static PhaseRing& phases() { static PhaseRing p; return p; }
static void mark(int which, void* stream) {
  PhaseRing& p = phases();
  if (!p.on) return;
  hipEvent_t& e = p.ev[p.steps % PhaseRing::RING][which];
  if (e == nullptr && hipEventCreate(&e) != hipSuccess) { p.on = false; return; }
  (void)hipEventRecord(e, (hipStream_t)stream);
}

static int g_bag_stats = getenv("ESC_BAG_STATS") ? atoi(getenv("ESC_BAG_STATS")) : 1;   // BatchNorm partials from the bag kernel's epilogue
static int g_e0_early = getenv("ESC_E0_EARLY") ? atoi(getenv("ESC_E0_EARLY")) : 1;   // see forward()
static int g_fuse_finalize = 1; // ... and their merge by the GEMM's last workgroup (no bn_finalize launch)
static int g_gemm_stats = 1;   // BatchNorm statistics from the producing GEMM's epilogue (no extra pass over Y)
// Node-sized BatchNorms: partials merged in the consumer's prologue instead of a finalize launch (esc_engine_set_gemm_stats
// bit 2 / ESC_BN_FOLD=1).  Measured on MI355X (cfg1, untraced, same box): 1.154 ms with it vs 1.144 ms without — with the
// host running ahead a finalize launch costs the node chain ~5 us, and so does the redundant merge in every consumer
// workgroup (77 KB of partials + 38 fp64 merges: GEMM 8.5 -> 13.3 us, affine pass 4.4 -> 10 us).  Off by default.
// The LAST BatchNorm+ReLU of every node MLP is not materialised: the Linear writes its pre-BatchNorm rows straight into the
// concat slice and the consumers apply relu(x*scale+shift) as they read them — the next layer's aggregate (forward and
// backward, esc_gine_aggregate_*_affine) and the readout GEMM (prologue over all (L+1)*H columns).  One elementwise launch
// less per layer on the dependent node chain.  ESC_FUSE_NODE_ACT=0 / esc_engine_set_gemm_stats bit 4 switch it off.
static int g_ogb_prologue = getenv("ESC_OGB_PROLOGUE") ? atoi(getenv("ESC_OGB_PROLOGUE")) : 1;        // OGB node MLP: BN+ReLU of the hidden layer in lin1's GEMM prologue
static int g_ogb_split_tail = getenv("ESC_OGB_SPLIT_TAIL") ? atoi(getenv("ESC_OGB_SPLIT_TAIL")) : 0;    // OGB engine: weight gradients of the tail's two edge-row Linears on the node stream
static int g_ogb_bonds_on_node = getenv("ESC_OGB_BONDS_ON_NODE") ? atoi(getenv("ESC_OGB_BONDS_ON_NODE")) : 1;   // 4.34 -> 4.31 ms/step
static int g_fuse_drop_bwd = getenv("ESC_FUSE_DROP_BWD") ? atoi(getenv("ESC_FUSE_DROP_BWD")) : 1;     // dropout backward inside the BatchNorm backward (OGB engine)
static int g_fuse_node_act = getenv("ESC_FUSE_NODE_ACT") ? atoi(getenv("ESC_FUSE_NODE_ACT")) : 1;
static int g_fold = getenv("ESC_BN_FOLD") ? atoi(getenv("ESC_BN_FOLD")) : 0;     // 1: both BatchNorms of an MLP merged by their consumers; 2: only the last one (by the affine pass)
static bool fuse_node_act(const Ctx& c) { return g_fuse_node_act && !g_fold && c.act == 1 && c.y.H >= 64 && c.y.H % 4 == 0 && c.y.cat_scale != nullptr; }

static esc_bn_fold make_fold(const float* partials, int64_t rows, int64_t block_rows, int64_t C, const esc_bn_t& bn, const BnWs& w) {
  return esc_bn_fold{partials, rows, block_rows, C, bn.eps, bn.momentum, bn.gamma, bn.beta, w.mean, w.invstd, w.scale, w.shift,
                     bn.running_mean, bn.running_var};
}

// ---- SyncBN (esc_engine_set_collective): statistics over all ranks of a graph-sharded step ----------------------------
struct Collective {
  esc_allreduce_fn fn = nullptr; void* user = nullptr;
  int rank = 0, world = 1;
  float *buf_node = nullptr, *buf_edge = nullptr; int64_t cap = 0;
};
static Collective g_coll;
static bool sync_on(const Ctx& c) { return c.train && g_coll.fn != nullptr && g_coll.world > 1; }
static float* sync_buf(const Ctx& c) {
  EdgeStream& es = edge_stream();
  return (es.ok && c.s == (void*)es.stream) ? g_coll.buf_edge : g_coll.buf_node;
}
static int sync_allreduce(const Ctx& c, float* buf, int64_t n) {
  ESC_REQUIRE(buf != nullptr && n <= g_coll.cap, "esc_engine: SyncBN exchange buffer too small (%ld > %ld floats)", (long)n, (long)g_coll.cap);
  const int rc = g_coll.fn(buf, n, c.s, g_coll.user);
  if (rc != 0) { set_error("esc_engine: the collective provider failed (%d)", rc); return ESC_ELAUNCH; }
  return ESC_OK;
}
// w.mean / w.invstd hold THIS rank's statistics over its M rows: replace them (and the consumer-side coefficients, the
// running statistics) by the statistics over all ranks' rows
static int bn_sync_forward(const Ctx& c, int64_t M, const esc_bn_t& bn, const BnWs& w, int64_t C) {
  float* buf = sync_buf(c);
  ESC_TRY(esc_bn_sync_pack(w.mean, w.invstd, M, bn.eps, C, g_coll.rank, g_coll.world, buf, c.s));
  ESC_TRY(sync_allreduce(c, buf, (int64_t)g_coll.world * 3 * C));
  return esc_bn_sync_finalize(buf, g_coll.world, C, bn.eps, bn.momentum, w.mean, w.invstd, bn.running_mean, bn.running_var,
                              bn.gamma, bn.beta, w.scale, w.shift, w.nglob, c.s);
}
// BatchNorm(+ReLU) backward; d(gamma), d(beta) stay this rank's sums (the gradient all-reduce adds them up)
static int bn_backward(const Ctx& c, const float* X, int64_t ldx, const float* Y, int64_t ldy, const float* dY, int64_t lddy,
                       int64_t M, const BnWs& w, const esc_bn_t& bn, float* dX, int64_t lddx, float* scratch, int64_t width = 0) {
  const int64_t C = width > 0 ? width : c.y.H;
  if (!sync_on(c))
    return esc_bn_bwd(X, ldx, Y, ldy, dY, lddy, M, C, w.mean, w.invstd, bn.gamma, bn.beta, c.act, dX, lddx, bn.dgamma, bn.dbeta,
                      scratch, c.s);
  float* buf = sync_buf(c);
  ESC_TRY(esc_bn_bwd_sums(X, ldx, Y, ldy, dY, lddy, M, C, w.mean, w.invstd, bn.gamma, bn.beta, c.act, buf, bn.dgamma, bn.dbeta,
                          scratch, c.s));
  ESC_TRY(sync_allreduce(c, buf, 2 * C));
  ESC_TRY(esc_bn_sync_coef(buf, C, w.nglob, c.s));
  return esc_bn_bwd_apply(X, ldx, Y, ldy, dY, lddy, M, C, w.mean, w.invstd, bn.gamma, bn.beta, c.act, buf, dX, lddx, c.s);
}

// BatchNorm(+ReLU) backward next to a dropout: on_output == 0: dY = grad of dropout(act(bn(X))) (mask applied to dY first);
// on_output == 1: X = dropout(input) (mask applied to the result).  One fused sequence when the operands allow it
// (esc_bn_bwd_dropout), else the dropout backward as its own pass.  ReLU / no activation only (c.act in {0, 1}).
static int bn_backward_drop(const Ctx& c, const float* X, int64_t ldx, const float* dY, int64_t lddy, int64_t M, const BnWs& w,
                            const esc_bn_t& bn, const uint8_t* mask, float p, int on_output, float* dX, int64_t lddx,
                            float* scratch, int64_t width = 0) {
  const int64_t C = width > 0 ? width : c.y.H;
  if (p <= 0.f) return bn_backward(c, X, ldx, nullptr, 0, dY, lddy, M, w, bn, dX, lddx, scratch, width);
  const bool fused = !sync_on(c) && c.act <= 1 && ldx == C && lddy == C && lddx == C && esc_bn_bwd_dropout_ok(C, ldx, lddy, lddx) &&
                     ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(dX) |
                       reinterpret_cast<uintptr_t>(w.mean) | reinterpret_cast<uintptr_t>(w.invstd) | reinterpret_cast<uintptr_t>(bn.gamma) |
                       reinterpret_cast<uintptr_t>(bn.beta) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(mask) & 3) == 0 && g_fuse_drop_bwd;
  if (fused)
    return esc_bn_bwd_dropout(X, ldx, dY, lddy, M, C, w.mean, w.invstd, bn.gamma, bn.beta, c.act, mask, p, on_output, dX, lddx,
                              bn.dgamma, bn.dbeta, scratch, c.s);
  if (!on_output) {
    ESC_TRY(esc_dropout_bwd(dY, lddy, M, C, p, mask, nullptr, 0, dX, lddx, c.s));
    return bn_backward(c, X, ldx, nullptr, 0, dX, lddx, M, w, bn, dX, lddx, scratch, width);
  }
  ESC_TRY(bn_backward(c, X, ldx, nullptr, 0, dY, lddy, M, w, bn, dX, lddx, scratch, width));
  return esc_dropout_bwd(dX, lddx, M, C, p, mask, nullptr, 0, dX, lddx, c.s);
}

// Y = X*W^T + b followed by BatchNorm coefficient computation (training: batch statistics; eval: running ones)
static int linear_bn(const Ctx& c, const float* X, int64_t ld_x, const esc_linear_t& lin, const float* sc, const float* sh,
                     int64_t M, float* Y, const esc_bn_t& bn, const BnWs& w, int64_t ld_y = 0) {
  const LdsFloorGuard cap(c.on_edge_stream && g_cap_forward);
  const int64_t H = lin.out_dim, K = lin.in_dim;          // (the BatchNorm is as wide as the Linear's output)
  if (ld_y <= 0) ld_y = H;
  const bool sync = sync_on(c);     // statistics over all ranks: local ones first (no running update, no coefficients), then the exchange
  const bool fused = c.train && g_gemm_stats && H > 32 && c.jobs != nullptr;   // main chain only (col_stats is shared scratch)
  if (fused && g_fuse_finalize && M > 1 && !sync) {    // statistics AND their merge ride on the GEMM launch
    esc_bn_fuse f{bn.eps, bn.momentum, w.mean, w.invstd, bn.running_mean, bn.running_var, bn.gamma, bn.beta, w.scale, w.shift};
    return esc_linear_bn_fwd(X, ld_x, lin.w, K, lin.b, sc, sh, M, H, K, Y, ld_y, c.y.col_stats, &f, c.s);
  }
  ESC_TRY(esc_linear_fwd(X, ld_x, lin.w, K, lin.b, sc, sh, M, H, K, Y, ld_y, fused ? c.y.col_stats : nullptr, c.s));
  if (fused) {
    ESC_TRY(esc_bn_stats_from_partials_rows(c.y.col_stats, M, H, esc_linear_stats_block_rows(X, ld_x, lin.w, K, M, H, K), bn.eps,
                                            bn.momentum, w.mean, w.invstd, sync ? nullptr : bn.running_mean,
                                            sync ? nullptr : bn.running_var, bn.gamma, bn.beta, sync ? nullptr : w.scale,
                                            sync ? nullptr : w.shift, c.s));
    return sync ? bn_sync_forward(c, M, bn, w, H) : ESC_OK;
  }
  if (c.train) {
    ESC_TRY(esc_bn_stats(Y, ld_y, M, H, bn.eps, bn.momentum, w.mean, w.invstd, sync ? nullptr : bn.running_mean,
                         sync ? nullptr : bn.running_var, bn.gamma, bn.beta, sync ? nullptr : w.scale, sync ? nullptr : w.shift,
                         c.y.bn_scratch, c.s));
    return sync ? bn_sync_forward(c, M, bn, w, H) : ESC_OK;
  }
  return esc_bn_eval_coef(bn.running_mean, bn.running_var, bn.gamma, bn.beta, bn.eps, H, w.scale, w.shift, c.s);
}

static int bn_coeffs(const Ctx& c, const float* X, int64_t ld, int64_t M, const esc_bn_t& bn, const BnWs& w, int64_t width = 0) {
  const int64_t C = width > 0 ? width : c.y.H;
  if (c.train) {
    const bool sync = sync_on(c);
    ESC_TRY(esc_bn_stats(X, ld, M, C, bn.eps, bn.momentum, w.mean, w.invstd, sync ? nullptr : bn.running_mean,
                         sync ? nullptr : bn.running_var, bn.gamma, bn.beta, sync ? nullptr : w.scale, sync ? nullptr : w.shift,
                         c.y.bn_scratch, c.s));
    return sync ? bn_sync_forward(c, M, bn, w, C) : ESC_OK;
  }
  return esc_bn_eval_coef(bn.running_mean, bn.running_var, bn.gamma, bn.beta, bn.eps, C, w.scale, w.shift, c.s);
}

// Linear, BN, ReLU, Linear, BN, ReLU  (reference :65-73, :78-87) -> out (materialised, ld_out)
// node-sized training-mode MLPs: no finalize launches — each BatchNorm's partials are merged by its consumer
static bool fold_ok(const Ctx& c, int64_t M) {
  const int64_t H = c.y.H;
  return c.act == 1 && c.train && g_fold && !sync_on(c) && g_gemm_stats && esc_linear_fold_available() && c.jobs != nullptr && !c.on_edge_stream && M > 1 && M <= 4096 && H % 32 == 0 &&
         H > 32 && H <= 1024;
}

static int mlp_forward(const Ctx& c, const esc_mlp_t& p, const MlpWs& w, const float* A, int64_t ld_a, int64_t M,
                       float* out, int64_t ld_out, bool pre_out = false) {
  const int64_t H = c.y.H;
  if (pre_out) {                              // `out` receives the PRE-BatchNorm rows of lin1; w.b1 holds the coefficients
    ESC_TRY(linear_bn(c, A, ld_a, p.lin0, nullptr, nullptr, M, w.Y0, p.bn0, w.b0));
    return linear_bn(c, w.Y0, H, p.lin1, w.b0.scale, w.b0.shift, M, out, p.bn1, w.b1, ld_out);
  }
  if (fold_ok(c, M) && g_fold == 2) {         // only the MLP's last BatchNorm: its finalize launch folds into the affine pass
    ESC_TRY(linear_bn(c, A, ld_a, p.lin0, nullptr, nullptr, M, w.Y0, p.bn0, w.b0));
    ESC_TRY(esc_linear_fwd(w.Y0, H, p.lin1.w, H, p.lin1.b, w.b0.scale, w.b0.shift, M, H, H, w.Y1, H, c.y.col_stats_b, c.s));
    const esc_bn_fold f1 = make_fold(c.y.col_stats_b, M, esc_linear_stats_block_rows(w.Y0, H, p.lin1.w, H, M, H, H), H, p.bn1, w.b1);
    return esc_affine_act_fold(w.Y1, H, M, H, &f1, 1, out, ld_out, c.s);
  }
  if (fold_ok(c, M)) {
    const int64_t K0 = p.lin0.in_dim;
    ESC_TRY(esc_linear_fwd(A, ld_a, p.lin0.w, K0, p.lin0.b, nullptr, nullptr, M, H, K0, w.Y0, H, c.y.col_stats, c.s));
    const esc_bn_fold f0 = make_fold(c.y.col_stats, M, esc_linear_stats_block_rows(A, ld_a, p.lin0.w, K0, M, H, K0), H, p.bn0, w.b0);
    ESC_TRY(esc_linear_fwd_fold(w.Y0, H, p.lin1.w, H, p.lin1.b, &f0, M, H, H, w.Y1, H, c.y.col_stats_b, c.s));
    const esc_bn_fold f1 = make_fold(c.y.col_stats_b, M, esc_linear_stats_block_rows(w.Y0, H, p.lin1.w, H, M, H, H), H, p.bn1, w.b1);
    return esc_affine_act_fold(w.Y1, H, M, H, &f1, 1, out, ld_out, c.s);
  }
  ESC_TRY(linear_bn(c, A, ld_a, p.lin0, nullptr, nullptr, M, w.Y0, p.bn0, w.b0));
  if (w.A1) {                                 // activation other than ReLU: the hidden activation is written once
    ESC_TRY(esc_affine_act(w.Y0, H, M, H, w.b0.scale, w.b0.shift, c.act, w.A1, H, c.s));
    ESC_TRY(linear_bn(c, w.A1, H, p.lin1, nullptr, nullptr, M, w.Y1, p.bn1, w.b1));
    return esc_affine_act(w.Y1, H, M, H, w.b1.scale, w.b1.shift, c.act, out, ld_out, c.s);
  }
  ESC_TRY(linear_bn(c, w.Y0, H, p.lin1, w.b0.scale, w.b0.shift, M, w.Y1, p.bn1, w.b1));
  return esc_affine_act(w.Y1, H, M, H, w.b1.scale, w.b1.shift, 1, out, ld_out, c.s);
}

// given dOut (grad of the materialised output `out`), produce parameter grads and, if dA != NULL, dA
// Which of an MLP's two Linear backwards take the fused form (see g_bn_fuse_bwd)?  ReLU MLPs whose hidden activation was never
// materialised (counting model) and ELU MLPs with a materialised one (ZINC), statistics of this rank only, shapes the fused
// kernels serve; lin0 only together with lin1.
struct MlpFuse { bool lin1 = false, lin0 = false; };
static MlpFuse mlp_fuse_plan(const Ctx& c, const esc_mlp_t& p, const MlpWs& w, const float* A, int64_t ld_a, int64_t M, const float* out,
                             int64_t ld_out, const float* dOut, int64_t ld_dout, const float* dA, int64_t ld_da, bool pre_out) {
  const Layout& y = c.y;
  const int64_t H = y.H;
  MlpFuse f;
  if (!((g_bn_fuse_bwd & 1) && c.train && !sync_on(c) && y.bst_part != nullptr && p.lin1.in_dim == H && p.lin1.out_dim == H)) return f;
  // (ELU: built and tested, but the exp in the operand staging costs more than the two launches it saves — ZINC config 4: 1.14 ms
  // fused against 1.10 ms per step; ESC_BN_FUSE_ELU=1 enables it)
  if (!((c.act == 1 && w.A1 == nullptr) || (c.act == 2 && w.A1 != nullptr && !pre_out && g_bn_fuse_elu))) return f;
  const esc_bn_bwd_fused f1 = bn_fused(pre_out ? out : w.Y1, pre_out ? ld_out : H, w.b1, c.act), f0 = bn_fused(w.Y0, H, w.b0, c.act);
  const esc_bn_bwd_next n0{y.bst_part, w.Y0, H, w.b0.mean, w.b0.invstd, w.b0.scale, w.b0.shift, c.act};
  const float* slab_probe = c.jobs ? *c.slab_cursor : y.slabs;
  const float* xh = w.A1 ? w.A1 : w.Y0;                    // lin1's input: the materialised activation, or its pre-BatchNorm rows
  f.lin1 = esc_linear_bwd_both_bn_ok(dOut, ld_dout, &f1, xh, H, p.lin1.w, H, M, H, H, y.dT2, H, slab_probe, (g_bn_fuse_bwd & 2) ? &n0 : nullptr) != 0;
  f.lin0 = f.lin1 && esc_linear_bwd_both_bn_ok(y.dT2, H, &f0, A, ld_a, p.lin0.w, p.lin0.in_dim, M, H, p.lin0.in_dim, dA, ld_da, slab_probe, nullptr) != 0;
  return f;
}
// (callers that can leave BatchNorm 1's column sums ask whether they will be consumed)
static bool mlp_backward_fused(const Ctx& c, const esc_mlp_t& p, const MlpWs& w, const float* A, int64_t ld_a, int64_t M, const float* out,
                               int64_t ld_out, const float* dOut, int64_t ld_dout, const float* dA, int64_t ld_da, bool pre_out) {
  return mlp_fuse_plan(c, p, w, A, ld_a, M, out, ld_out, dOut, ld_dout, dA, ld_da, pre_out).lin1;
}
// have_slots > 0: the column sums of BatchNorm 1's backward are already in c.y.bst_part (float2[have_slots][H], left by the
// producer of dOut: the readout's dX epilogue or the next layer's aggregate backward)
static int mlp_backward(const Ctx& c, const esc_mlp_t& p, const MlpWs& w, const float* A, int64_t ld_a, int64_t M,
                        const float* out, int64_t ld_out, const float* dOut, int64_t ld_dout, float* dA,
                        int64_t ld_da, bool pre_out = false, int64_t have_slots = 0) {
  const Layout& y = c.y;
  const int64_t H = y.H;
  const MlpFuse fuse = mlp_fuse_plan(c, p, w, A, ld_a, M, out, ld_out, dOut, ld_dout, dA, ld_da, pre_out);
  if (fuse.lin1) {
    const float* x1 = pre_out ? out : w.Y1;                  // what BatchNorm 1 normalised
    const int64_t ld_x1 = pre_out ? ld_out : H;
    const esc_bn_bwd_fused f1 = bn_fused(x1, ld_x1, w.b1, c.act), f0 = bn_fused(w.Y0, H, w.b0, c.act);
    esc_bn_bwd_next n0{y.bst_part, w.Y0, H, w.b0.mean, w.b0.invstd, w.b0.scale, w.b0.shift, c.act};
    const bool stats = (g_bn_fuse_bwd & 2) != 0;
    // (activation derivatives are recomputed from the pre-BatchNorm rows everywhere — Y == NULL — so that the sums and the
    // fused apply see the same values)
    if (have_slots > 0)
      ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part, have_slots, M, H, w.b1.coef, p.bn1.dgamma, p.bn1.dbeta, c.s));
    else
      ESC_TRY(esc_bn_bwd_coef(x1, ld_x1, nullptr, 0, dOut, ld_dout, M, H, w.b1.mean, w.b1.invstd, p.bn1.gamma, p.bn1.beta, c.act, w.b1.coef,
                              p.bn1.dgamma, p.bn1.dbeta, y.bn_scratch, c.s));
    if (w.A1) ESC_TRY(linear_backward_bn(c, dOut, ld_dout, f1, w.A1, H, nullptr, nullptr, p.lin1, M, y.dT2, H, 0, stats ? &n0 : nullptr));
    else      ESC_TRY(linear_backward_bn(c, dOut, ld_dout, f1, w.Y0, H, w.b0.scale, w.b0.shift, p.lin1, M, y.dT2, H, 0, stats ? &n0 : nullptr));
    if (stats)
      ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part, cdiv(M, esc_linear_bwd_bn_block_rows(M, H, H)), M, H, w.b0.coef, p.bn0.dgamma, p.bn0.dbeta, c.s));
    else
      ESC_TRY(esc_bn_bwd_coef(w.Y0, H, nullptr, 0, y.dT2, H, M, H, w.b0.mean, w.b0.invstd, p.bn0.gamma, p.bn0.beta, c.act, w.b0.coef,
                              p.bn0.dgamma, p.bn0.dbeta, y.bn_scratch, c.s));
    if (fuse.lin0) return linear_backward_bn(c, y.dT2, H, f0, A, ld_a, nullptr, nullptr, p.lin0, M, dA, ld_da, 0, nullptr);
    // lin0's shape is not served by the fused kernels (e.g. a 32-wide input): the apply as its own pass, then the plain backward
    ESC_TRY(esc_bn_bwd_apply(w.Y0, H, nullptr, 0, y.dT2, H, M, H, w.b0.mean, w.b0.invstd, p.bn0.gamma, p.bn0.beta, c.act, w.b0.coef, y.dT2, H, c.s));
    return linear_backward(c, y.dT2, H, A, ld_a, nullptr, nullptr, p.lin0, M, dA, ld_da, 0);
  }
  if (pre_out) ESC_TRY(bn_backward(c, out, ld_out, nullptr, 0, dOut, ld_dout, M, w.b1, p.bn1, y.dT1, H, y.bn_scratch));   // `out` = pre-BN rows
  else         ESC_TRY(bn_backward(c, w.Y1, H, out, ld_out, dOut, ld_dout, M, w.b1, p.bn1, y.dT1, H, y.bn_scratch));
  // bit 1 alone: the elementwise apply launches stay, only the column sums of BatchNorm 0 come out of lin1's dX epilogue
  if ((g_bn_fuse_bwd & 3) == 2 && c.train && c.act == 1 && w.A1 == nullptr && !sync_on(c) && y.bst_part != nullptr && p.lin1.in_dim == H && p.lin1.out_dim == H) {
    esc_bn_bwd_next n0{y.bst_part, w.Y0, H, w.b0.mean, w.b0.invstd, w.b0.scale, w.b0.shift, 1};
    const float* slab_probe = c.jobs ? *c.slab_cursor : y.slabs;
    if (esc_linear_bwd_both_bn_ok(y.dT1, H, nullptr, w.Y0, H, p.lin1.w, H, M, H, H, y.dT2, H, slab_probe, &n0)) {
      const LdsFloorGuard cap(c.on_edge_stream);
      float* slabs = y.slabs;
      esc_reduce_job* job = nullptr;
      if (c.jobs) {
        slabs = *c.slab_cursor;
        *c.slab_cursor += (esc_linear_bwd_weight_scratch(M, H, H) + 63) & ~63LL;
        c.jobs->emplace_back();
        job = &c.jobs->back();
      }
      ESC_TRY(esc_linear_bwd_both_bn(y.dT1, H, nullptr, w.Y0, H, w.b0.scale, w.b0.shift, p.lin1.w, H, M, H, H, y.dT2, H, 0, p.lin1.dw, H, p.lin1.db,
                                     slabs, job, &n0, c.s));
      ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part, cdiv(M, esc_linear_bwd_bn_block_rows(M, H, H)), M, H, w.b0.coef, p.bn0.dgamma, p.bn0.dbeta, c.s));
      ESC_TRY(esc_bn_bwd_apply(w.Y0, H, nullptr, 0, y.dT2, H, M, H, w.b0.mean, w.b0.invstd, p.bn0.gamma, p.bn0.beta, 1, w.b0.coef, y.dT2, H, c.s));
      return linear_backward(c, y.dT2, H, A, ld_a, nullptr, nullptr, p.lin0, M, dA, ld_da, 0);
    }
  }
  if (w.A1) {
    ESC_TRY(linear_backward(c, y.dT1, H, w.A1, H, nullptr, nullptr, p.lin1, M, y.dT2, H, 0));
    ESC_TRY(bn_backward(c, w.Y0, H, w.A1, H, y.dT2, H, M, w.b0, p.bn0, y.dT2, H, y.bn_scratch));
    return linear_backward(c, y.dT2, H, A, ld_a, nullptr, nullptr, p.lin0, M, dA, ld_da, 0);
  }
  ESC_TRY(linear_backward(c, y.dT1, H, w.Y0, H, w.b0.scale, w.b0.shift, p.lin1, M, y.dT2, H, 0));
  ESC_TRY(bn_backward(c, w.Y0, H, nullptr, 0, y.dT2, H, M, w.b0, p.bn0, y.dT2, H, y.bn_scratch));
  return linear_backward(c, y.dT2, H, A, ld_a, nullptr, nullptr, p.lin0, M, dA, ld_da, 0);
}

static int forward(const Ctx& c) {
  const esc_nested_gin_t* m = c.m;
  const esc_batch_t* b = c.b;
  const Layout& y = c.y;
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, W = y.W;
  // ---- edge pipeline: ESC bag, z_embedding (reference :155-156) and the edge terms e_l = lin_l(z_emb) of ALL layers
  // (:161,:169 inside GINEConv).  It touches the node chain only through e_l, so it runs on the edge stream — for batches of at
  // least g_two_stream_min_edges edges: below that (the per-rank slices of a strong-scaling run: 16 graphs, 1 900 edges) the
  // edge-sized kernels are as latency-bound as the node chain and the events cost more than the overlap returns (bs 16: 0.68 ms on
  // one stream, 0.84 on two; bs 64: 0.85 / 0.89; bs 128: 1.25 / 1.04).
  EdgeStream& es = edge_stream_for(E);
  const NodeFloorGuard node_floor(es.ok && c.train ? g_node_floor_fwd : 0);
  Ctx ce = c;
  if (es.ok) {
    ESC_TRY(chain(es.z_ready, (hipStream_t)c.s, es.stream));       // the batch arrays were produced on the caller's stream
    ce = edge_ctx(c, es.stream);
  }
  // ESC bag (LDS-staged table slices); in training mode its epilogue leaves the BatchNorm partials of z_embedding's first
  // BatchNorm, so the statistics pass over the E x H output is one finalize launch
  const int64_t bag_block = (c.train && !sync_on(ce) && g_bag_stats) ? esc_bag_fwd_stats_block_rows(m->z_table, m->z_rows, H, y.Zb, H, E) : 0;
  if (bag_block > 0) {
    ESC_TRY(esc_bag_fwd_rows(m->z_table, m->z_rows, H, b->row_ptr, b->bag_idx, b->bag_val, E, y.Zb, H, 0, ce.y.col_stats, ce.s));
    ESC_TRY(esc_bn_stats_from_partials_rows(ce.y.col_stats, E, H, bag_block, m->zbn0.eps, m->zbn0.momentum, y.zb0.mean, y.zb0.invstd,
                                            m->zbn0.running_mean, m->zbn0.running_var, m->zbn0.gamma, m->zbn0.beta, y.zb0.scale, y.zb0.shift, ce.s));
  } else {
    ESC_TRY(esc_bag_fwd_rows(m->z_table, m->z_rows, H, b->row_ptr, b->bag_idx, b->bag_val, E, y.Zb, H, 0, nullptr, ce.s));
    ESC_TRY(bn_coeffs(ce, y.Zb, H, E, m->zbn0, y.zb0));
  }
  const bool mat = g_materialise_edge_act != 0;
  if (mat) {
    ESC_TRY(esc_affine_act(y.Zb, H, E, H, y.zb0.scale, y.zb0.shift, 1, y.A0, H, ce.s));
    ESC_TRY(linear_bn(ce, y.A0, H, m->zlin, nullptr, nullptr, E, y.Yz, m->zbn1, y.zb1));
  } else {
    ESC_TRY(linear_bn(ce, y.Zb, H, m->zlin, y.zb0.scale, y.zb0.shift, E, y.Yz, m->zbn1, y.zb1));
  }                                                                       // z_emb = relu(Yz*scale+shift)
  // The first edge term is narrow (in_dim columns: a bandwidth pass over z_emb): it applies z_embedding's last BatchNorm+ReLU to
  // its operand itself and runs BEFORE the pass that materialises z_emb for the wide layers — the node chain's first
  // aggregate waits for e_0 only (same fmaf + max per element: e_0 is bit-identical either way)
  const bool e0_early = mat && g_e0_early && y.C0 <= 32 && L >= 1;
  if (e0_early) {
    const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
    ESC_TRY(esc_linear_fwd(y.Yz, H, m->conv[0].lin.w, H, m->conv[0].lin.b, y.zb1.scale, y.zb1.shift, E, y.C0, H, y.e[0], y.ld_e[0], nullptr, ce.s));
    if (es.ok && hipEventRecord(es.e_ready[0], es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
  }
  if (mat) ESC_TRY(esc_affine_act(y.Yz, H, E, H, y.zb1.scale, y.zb1.shift, 1, y.Zemb, H, ce.s));
  // Per-layer launches (ESC_EDGE_BATCHED=0, L < 3, or activations not materialised): the edge terms run one layer ahead of the node
  // chain — e_{l+1} is queued behind the aggregate of layer l and overlaps that layer's MLP, so that a bandwidth-bound aggregate never
  // shares the HBM with an edge-sized GEMM (r01: 1.255 -> 1.225 ms against two layers ahead).  The default since r03 is the batched
  // launch below.
  auto edge_term = [&](int l) -> int {
    const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? y.C0 : H;
    if (mat) ESC_TRY(esc_linear_fwd(y.Zemb, H, cv.lin.w, H, cv.lin.b, nullptr, nullptr, E, C, H, y.e[l], y.ld_e[l], nullptr, ce.s));
    else     ESC_TRY(esc_linear_fwd(y.Yz, H, cv.lin.w, H, cv.lin.b, y.zb1.scale, y.zb1.shift, E, C, H, y.e[l], y.ld_e[l], nullptr, ce.s));
    if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    return ESC_OK;
  };
  mark(PH_START, c.s);
  const bool batched = y.e_cat != nullptr && mat;
  if (batched) {                                            // e_1 .. e_{L-1} in one launch over the packed weights
    if (!e0_early) ESC_TRY(edge_term(0));
    esc_table_list tl{};
    tl.count = 2 * ((int)L - 1);
    for (int l = 1; l < (int)L; ++l) {
      tl.rows[l - 1] = (int32_t)H; tl.w[l - 1] = m->conv[l].lin.w;
      tl.rows[(L - 1) + l - 1] = 1; tl.w[(L - 1) + l - 1] = m->conv[l].lin.b;
    }
    ESC_TRY(esc_table_pack(&tl, H, y.w_cat, ce.s));
    {
      const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
      ESC_TRY(esc_linear_fwd(y.Zemb, H, y.w_cat, H, y.w_cat + (L - 1) * H * H, nullptr, nullptr, E, (L - 1) * H, H, y.e_cat, (L - 1) * H, nullptr, ce.s));
    }
    for (int l = 1; l < (int)L; ++l)
      if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
  }
  const int ahead = batched ? 0 : (es.ok ? g_edge_ahead : (int)L);          // one stream: all of them up front, in layer order
  for (int l = e0_early ? 1 : 0; l < (int)L && l < ahead; ++l) ESC_TRY(edge_term(l));
  // ---- node pipeline (first, while it would otherwise wait for the first edge term: the chunk schedule of the bag
  // gradient, which depends on the batch's index arrays only)
  if (c.train) ESC_TRY(esc_bag_bwd_classify(b->col_row, y.Z, H, E, y.bag_scratch, c.s));
  // xs[0] = x_embedding(x) (reference :166) — side stream
  SideStream& ss = side_stream();
  if (ss.ok) {
    if (hipEventRecord(ss.fork_f, (hipStream_t)c.s) != hipSuccess || hipStreamWaitEvent(ss.stream, ss.fork_f, 0) != hipSuccess) {
      set_error("esc_engine: side-stream fork failed");
      return ESC_ELAUNCH;
    }
    const Ctx cx = side_ctx(c, ss.stream);
    ESC_TRY(mlp_forward(cx, m->xemb, y.xemb, b->x, y.C0, N, y.cat, W, fuse_node_act(c)));
    (void)hipEventRecord(ss.join_f, ss.stream);
  } else {
    ESC_TRY(mlp_forward(c, m->xemb, y.xemb, b->x, y.C0, N, y.cat, W, fuse_node_act(c)));
  }
  // (readout split: see g_readout_split; the conditions are linear_bn's for statistics from the GEMM epilogue + a finalize launch)
  const bool ro_split = g_readout_split && es.ok && !ss.ok && c.train && L >= 2 && fuse_node_act(c) && !sync_on(c) && g_gemm_stats && c.jobs != nullptr &&
                        !(fold_ok(c, N) && H <= 256) && L * H <= 1280 &&
                        esc_linear_fwd_from_ok(y.cat + L * H, W, m->lin1.w + L * H, W, N, H, H, 1) != 0;
  // GINE layers (reference :161, :167-175): xs[l+1] -> cat[:, (l+1)H : (l+2)H]
  for (int l = 0; l < L; ++l) {
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? y.C0 : H;
    const float* hin = l == 0 ? b->x : y.cat + (int64_t)l * H;
    const int64_t ld_h = l == 0 ? y.C0 : W;
    // (batched edge terms: e_1 .. e_{L-1} come out of one launch, the wait in front of layer 1 covers the later layers)
    if (es.ok && !(batched && l >= 2 && g_skip_waits) && hipStreamWaitEvent((hipStream_t)c.s, es.e_ready[l], 0) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    if (fuse_node_act(c) && l > 0)          // hin = pre-BatchNorm rows of the previous layer: relu(x*scale+shift) on the fly
      ESC_TRY(esc_gine_aggregate_fwd_affine(hin, ld_h, y.cat_scale + (int64_t)l * H, y.cat_shift + (int64_t)l * H, y.e[l], y.ld_e[l], b->in_ptr,
                                            b->in_edge, b->in_src, cv.eps, N, C, y.agg[l], C, c.s));
    else
      ESC_TRY(esc_gine_aggregate_fwd(hin, ld_h, y.e[l], y.ld_e[l], b->in_ptr, b->in_edge, b->in_src, cv.eps, N, C, y.agg[l], C, c.s));
    if (!batched && es.ok && l + ahead < (int)L) {
      ESC_TRY(chain(es.agg_done[l], (hipStream_t)c.s, es.stream));
      ESC_TRY(edge_term(l + ahead));
    }
    ESC_TRY(mlp_forward(c, cv.nn, y.conv[l], y.agg[l], C, N, y.cat + (int64_t)(l + 1) * H, W, fuse_node_act(c)));
    if (ro_split && l == (int)L - 2) {        // slices 0 .. L-1 of cat (and their BatchNorm coefficients) are final: their share of lin1
      ESC_TRY(chain(es.lin1_fork, (hipStream_t)c.s, es.stream));
      ESC_TRY(esc_linear_fwd(y.cat, W, m->lin1.w, W, nullptr, y.cat_scale, y.cat_shift, N, H, L * H, y.Ypart, H, nullptr, es.stream));
      if (hipEventRecord(es.lin1_rest, es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    }
  }
  if (c.train) mark(PH_EDGE_FWD_DONE, ce.s);
  // readout (reference :183-189) needs every slice of cat, including the side stream's
  if (ss.ok && hipStreamWaitEvent((hipStream_t)c.s, ss.join_f, 0) != hipSuccess) {
    set_error("esc_engine: side-stream join failed");
    return ESC_ELAUNCH;
  }
  if (fold_ok(c, N) && H <= 256) {          // lin2 (a wave per row, H <= 256) merges bn_lin1's partials itself
    ESC_TRY(esc_linear_fwd(y.cat, W, m->lin1.w, W, m->lin1.b, nullptr, nullptr, N, H, W, y.Yl, H, c.y.col_stats, c.s));
    const esc_bn_fold f = make_fold(c.y.col_stats, N, esc_linear_stats_block_rows(y.cat, W, m->lin1.w, W, N, H, W), H, m->bn_lin1, y.bl);
    return esc_linear_fwd_fold(y.Yl, H, m->lin2.w, H, m->lin2.b, &f, N, 1, H, y.pred, 1, nullptr, c.s);
  }
  const bool fa = fuse_node_act(c);       // cat holds pre-BatchNorm rows: the readout GEMM applies every slice's BatchNorm+ReLU itself
  if (ro_split) {
    const int64_t K0 = L * H;
    if (hipStreamWaitEvent((hipStream_t)c.s, es.lin1_rest, 0) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    ESC_TRY(esc_linear_fwd_from(y.Ypart, H, y.cat + K0, W, m->lin1.w + K0, W, m->lin1.b, y.cat_scale + K0, y.cat_shift + K0, N, H, H, y.Yl, H,
                                c.y.col_stats, c.s));
    ESC_TRY(esc_bn_stats_from_partials_rows(c.y.col_stats, N, H, esc_linear_stats_block_rows(y.cat + K0, W, m->lin1.w + K0, W, N, H, H), m->bn_lin1.eps,
                                            m->bn_lin1.momentum, y.bl.mean, y.bl.invstd, m->bn_lin1.running_mean, m->bn_lin1.running_var,
                                            m->bn_lin1.gamma, m->bn_lin1.beta, y.bl.scale, y.bl.shift, c.s));
  } else
  ESC_TRY(linear_bn(c, y.cat, W, m->lin1, fa ? y.cat_scale : nullptr, fa ? y.cat_shift : nullptr, N, y.Yl, m->bn_lin1, y.bl));
  // train_step: the head's launch leaves the L1 gradient of every prediction too, so the backward starts behind it and the loss
  // launch (a single workgroup walking all predictions, 5-7 us) leaves the chain
  if (c.l1_target != nullptr && esc_linear_fwd_l1_ok(y.Yl, H, m->lin2.w, H, y.bl.scale, y.bl.shift))
    return esc_linear_fwd_l1(y.Yl, H, m->lin2.w, m->lin2.b, y.bl.scale, y.bl.shift, N, H, c.l1_target, c.l1_denom, 1.0f, y.pred, y.dpred, c.s);
  return esc_linear_fwd(y.Yl, H, m->lin2.w, H, m->lin2.b, y.bl.scale, y.bl.shift, N, 1, H, y.pred, 1, nullptr, c.s);
}

// What esc_engine_train_step_begin leaves for esc_engine_train_step_end: the join with the edge stream and the
// edge-side weight-gradient reductions.  Work the caller enqueues between the two calls (the next batch's collate)
// runs on the node stream while the edge pipeline is still finishing the step.
struct Pending {
  bool open = false;
  bool join = false;
  hipStream_t stream = nullptr;
  std::vector<esc_reduce_job> edge_jobs;
};
static Pending& pending() {
  static thread_local Pending p;
  return p;
}
static int finish_pending(Pending& p) {
  if (!p.open) return ESC_OK;
  p.open = false;
  EdgeStream& es = edge_stream();
  if (p.join && hipStreamWaitEvent(p.stream, es.joined, 0) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
  if (!p.edge_jobs.empty()) ESC_TRY(esc_slab_reduce_jobs(p.edge_jobs.data(), (int)p.edge_jobs.size(), p.stream));
  p.edge_jobs.clear();
  mark(PH_END, p.stream);
  if (phases().on) ++phases().steps;
  return ESC_OK;
}

static int backward(const Ctx& c_in, Pending* defer) {
  Ctx c = c_in;
  EdgeStream& es = edge_stream_for(c.y.E);
  const NodeFloorGuard node_floor(es.ok ? g_node_floor_bwd : 0);
  WgradStream& ws = wgrad_stream();
  if (ws.ok && es.ok && c.jobs != nullptr && c.train && c.y.dT1_l[0] != nullptr) c.wgrad = ws.stream;
  const esc_nested_gin_t* m = c.m;
  const esc_batch_t* b = c.b;
  const Layout& y = c.y;
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, W = y.W;
  // lin2 <- dpred
  ESC_TRY(linear_backward(c, y.dpred, 1, y.Yl, H, y.bl.scale, y.bl.shift, m->lin2, N, y.dAl, H, 0));
  // bn_lin1 backward: folded into lin1's backward when the fused kernels serve its column blocks (see g_bn_fuse_bwd)
  const bool split_lin1 = es.ok && c.jobs != nullptr && L >= 1;
  const esc_bn_bwd_fused fl = bn_fused(y.Yl, H, y.bl, 1);
  const bool fa_l = fuse_node_act(c);
  // the last layer's output gradient d(cat)[:, L*H:] is final once lin1's node-side block has written it: its dX tiles
  // leave the column sums of that layer's last BatchNorm backward (bit 1)
  const MlpWs& wl = y.conv[L > 0 ? L - 1 : 0];
  esc_bn_bwd_next nl{y.bst_part, y.cat + L * H, W, wl.b1.mean, wl.b1.invstd, wl.b1.scale, wl.b1.shift, 1};
  auto lin1_ok = [&](int64_t col0, int64_t ncols, const esc_bn_bwd_next* nx) {
    return esc_linear_bwd_both_bn_ok(y.dAl, H, &fl, y.cat + col0, W, m->lin1.w + col0, W, N, H, ncols, y.dcat + col0, W,
                                     c.jobs ? *c.slab_cursor : y.slabs, nx) != 0;
  };
  const bool fuse_l = (g_bn_fuse_bwd & 1) && c.act == 1 && !sync_on(c) && y.bst_part != nullptr &&
                      (split_lin1 ? (lin1_ok(0, L * H, nullptr) && lin1_ok(L * H, H, nullptr)) : lin1_ok(0, W, nullptr));
  int64_t last_slots = 0;          // > 0: slots of bst_part that hold the last layer's BatchNorm-backward sums
  if (fuse_l)
    ESC_TRY(esc_bn_bwd_coef(y.Yl, H, nullptr, 0, y.dAl, H, N, H, y.bl.mean, y.bl.invstd, m->bn_lin1.gamma, m->bn_lin1.beta, 1, y.bl.coef,
                            m->bn_lin1.dgamma, m->bn_lin1.dbeta, y.bn_scratch, c.s));
  else
    ESC_TRY(bn_backward(c, y.Yl, H, nullptr, 0, y.dAl, H, N, y.bl, m->bn_lin1, y.dAl, H, y.bn_scratch));
  // lin1 backward.  The node chain needs d(cat)[:, L*H:] (the last layer's output gradient) at once and the other
  // slices only when the first aggregate backward accumulates into them ~60 us later: with an edge stream the last
  // column block (dX slice + its dW columns) is computed here and the other L blocks over there, concurrently.
  if (split_lin1) {
    const int64_t K0 = L * H;                                 // columns [0, K0) go to the edge stream
    auto part = [&](void* stream, int64_t col0, int64_t ncols, float* db, const esc_bn_bwd_next* nx) -> int {
      float* slabs = *c.slab_cursor;
      *c.slab_cursor += (esc_linear_bwd_weight_scratch(N, H, ncols) + 63) & ~63LL;
      c.jobs->emplace_back();
      const bool fa = fa_l;
      Ctx cs = c;
      if (stream != c.s) cs.wgrad = nullptr;
      const WgradScope side(cs);
      if (fuse_l)
        return esc_linear_bwd_both_bn(y.dAl, H, &fl, y.cat + col0, W, fa ? y.cat_scale + col0 : nullptr, fa ? y.cat_shift + col0 : nullptr, m->lin1.w + col0, W,
                                      N, H, ncols, y.dcat + col0, W, 0, m->lin1.dw + col0, W, db, slabs, &c.jobs->back(), nx, stream);
      return esc_linear_bwd_both_deferred(y.dAl, H, y.cat + col0, W, fa ? y.cat_scale + col0 : nullptr, fa ? y.cat_shift + col0 : nullptr, m->lin1.w + col0, W, N, H, ncols,
                                          y.dcat + col0, W, 0, m->lin1.dw + col0, W, db, slabs, &c.jobs->back(), stream);
    };
    ESC_TRY(chain(es.lin1_fork, (hipStream_t)c.s, es.stream));
    {
      const LdsFloorGuard cap(true);
      ESC_TRY(part(es.stream, 0, K0, nullptr, nullptr));
    }
    if (hipEventRecord(es.lin1_rest, es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    // (the last layer's MLP backward follows at once: does it take the fused form, and can this block leave its sums?)
    const bool leave = fuse_l && fa_l && (g_bn_fuse_bwd & 2) && lin1_ok(K0, H, &nl) &&
                       mlp_backward_fused(c, m->conv[L - 1].nn, y.conv[L - 1], y.agg[L - 1], L == 1 ? y.C0 : H, N, y.cat + L * H, W, y.dcat + L * H, W,
                                          y.dagg, L == 1 ? y.C0 : H, true);
    ESC_TRY(part(c.s, K0, H, m->lin1.db, leave ? &nl : nullptr));
    if (leave) last_slots = cdiv(N, esc_linear_bwd_bn_block_rows(N, H, H));
  } else if (fuse_l) {
    ESC_TRY(linear_backward_bn(c, y.dAl, H, fl, y.cat, W, fa_l ? y.cat_scale : nullptr, fa_l ? y.cat_shift : nullptr, m->lin1, N, y.dcat, W, 0, nullptr));
  } else {
    ESC_TRY(linear_backward(c, y.dAl, H, y.cat, W, fa_l ? y.cat_scale : nullptr, fa_l ? y.cat_shift : nullptr, m->lin1, N, y.dcat, W, 0));
  }
  // x_embedding backward (input x needs no gradient): only reads d(cat)[:, 0:H] -> side stream
  SideStream& ss = side_stream();
  if (ss.ok) {
    if (hipEventRecord(ss.fork_b, (hipStream_t)c.s) != hipSuccess || hipStreamWaitEvent(ss.stream, ss.fork_b, 0) != hipSuccess) {
      set_error("esc_engine: side-stream fork failed");
      return ESC_ELAUNCH;
    }
    if (split_lin1 && hipStreamWaitEvent(ss.stream, es.lin1_rest, 0) != hipSuccess) {      // d(cat)[:, 0:H] comes from there
      set_error("esc_engine: stream event failed");
      return ESC_ELAUNCH;
    }
    const Ctx cx = side_ctx(c, ss.stream);
    ESC_TRY(mlp_backward(cx, m->xemb, y.xemb, b->x, y.C0, N, y.cat, W, y.dcat, W, nullptr, 0, fuse_node_act(c)));
    (void)hipEventRecord(ss.join_b, ss.stream);
  }
  // GINE layers, last to first (the eps gradients are only needed by the optimiser: one reduce launch at the end)
  Ctx ce = es.ok ? edge_ctx(c, es.stream) : c;
  std::vector<esc_reduce_job> edge_jobs;     // the edge pipeline's weight gradients are reduced after the join,
  edge_jobs.reserve(ESC_MAX_REDUCE_JOBS);    // the node pipeline's while the edge tail is still running
  if (es.ok && c.jobs) ce.jobs = &edge_jobs;
  std::vector<esc_sum_job> eps_jobs;
  int64_t agg_slots = 0;            // > 0: the aggregate backward of the layer above left this layer's BatchNorm sums in bst_part
  for (int l = (int)L - 1; l >= 0; --l) {
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? y.C0 : H;
    const float* hin = l == 0 ? b->x : y.cat + (int64_t)l * H;
    const int64_t ld_h = l == 0 ? y.C0 : W;
    Ctx cl = c;
    if (c.wgrad) { cl.y.dT1 = y.dT1_l[l]; cl.y.dT2 = y.dT2_l[l]; }     // still read by this layer's dW tiles when the next layer writes its own
    ESC_TRY(mlp_backward(cl, cv.nn, y.conv[l], y.agg[l], C, N, y.cat + (int64_t)(l + 1) * H, W,
                         y.dcat + (int64_t)(l + 1) * H, W, y.dagg, C, fuse_node_act(c), l == (int)L - 1 ? last_slots : agg_slots));
    agg_slots = 0;
    float* dx = l == 0 ? nullptr : y.dcat + (int64_t)l * H;            // accumulate into the previous slice
    if (split_lin1 && l == (int)L - 1 &&                               // ... which the edge stream's lin1 blocks fill
        hipStreamWaitEvent((hipStream_t)c.s, es.lin1_rest, 0) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    // (bit 3) d(cat)[:, l*H:(l+1)*H] is final once this launch has added its share: it leaves the column sums of the PREVIOUS layer's
    // last BatchNorm backward, and that layer's MLP backward starts with the finalize
    const bool stats_here = fuse_node_act(c) && l > 0 && (g_bn_fuse_bwd & 8) && esc_gine_aggregate_bwd_deps_slots(C) == 1 && C <= 2048 &&
                            mlp_backward_fused(c, m->conv[l - 1].nn, y.conv[l - 1], y.agg[l - 1], l - 1 == 0 ? y.C0 : H, N, y.cat + (int64_t)l * H, W,
                                               y.dcat + (int64_t)l * H, W, y.dagg, l - 1 == 0 ? y.C0 : H, true);
    if (stats_here) {
      ESC_TRY(esc_gine_aggregate_bwd_affine_stats(hin, ld_h, y.cat_scale + (int64_t)l * H, y.cat_shift + (int64_t)l * H, y.conv[l - 1].b1.mean,
                                                  y.conv[l - 1].b1.invstd, y.e[l], y.ld_e[l], y.dagg, C, b->out_ptr, b->out_edge, b->out_dst, cv.eps, N, C,
                                                  y.d_e[l], C, dx, W, 1, y.deps_part + (int64_t)l * 2 * N, y.bst_part, c.s));
      agg_slots = esc_gine_aggregate_bwd_stats_slots(N);
    } else if (fuse_node_act(c) && l > 0)
      ESC_TRY(esc_gine_aggregate_bwd_affine(hin, ld_h, y.cat_scale + (int64_t)l * H, y.cat_shift + (int64_t)l * H, y.e[l], y.ld_e[l], y.dagg, C,
                                            b->out_ptr, b->out_edge, b->out_dst, cv.eps, N, C, y.d_e[l], C, dx, W, 1,
                                            y.deps_part + (int64_t)l * 2 * N, c.s));
    else
      ESC_TRY(esc_gine_aggregate_bwd(hin, ld_h, y.e[l], y.ld_e[l], y.dagg, C, b->out_ptr, b->out_edge, b->out_dst, cv.eps, N, C,
                                     y.d_e[l], C, dx, W, 1, y.deps_part + (int64_t)l * 2 * N, c.s));
    eps_jobs.push_back(esc_sum_job{y.deps_part + (int64_t)l * 2 * N, N * esc_gine_aggregate_bwd_deps_slots(C), cv.deps});
    if (es.ok) ESC_TRY(chain(es.de_ready[l], (hipStream_t)c.s, es.stream));   // lin_l backward: edge stream
    const float* zin = g_materialise_edge_act ? y.Zemb : y.Yz;
    const float* zsc = g_materialise_edge_act ? nullptr : y.zb1.scale;
    const float* zsh = g_materialise_edge_act ? nullptr : y.zb1.shift;
    if (l == 0 && es.ok && c.jobs && g_split_last_lin) {
      // the LAST lin backward sits in the tail of the step: only its input gradient (which completes d(z_emb)) stays on
      // the edge stream; the weight gradient runs on the node stream, which has nothing left to do
      {
        const LdsFloorGuard cap(true);
        ESC_TRY(esc_linear_bwd_input(y.d_e[l], C, cv.lin.w, H, E, C, H, y.dZemb, H, l == (int)L - 1 ? 0 : 1, es.stream));
      }
      float* slabs = *c.slab_cursor;
      *c.slab_cursor += (esc_linear_bwd_weight_scratch(E, C, H) + 63) & ~63LL;
      c.jobs->emplace_back();
      ESC_TRY(esc_linear_bwd_both_deferred(y.d_e[l], C, zin, H, zsc, zsh, cv.lin.w, H, E, C, H, nullptr, 0, 0, cv.lin.dw, H, cv.lin.db,
                                           slabs, &c.jobs->back(), c.s));
    } else {
      ESC_TRY(linear_backward(ce, y.d_e[l], C, zin, H, zsc, zsh, cv.lin, E, y.dZemb, H, l == (int)L - 1 ? 0 : 1));
    }
    if (es.ok && !edge_jobs.empty() && l > 0) {      // the edge stream now idles until d_e of the next layer: reduce these slabs there
      ESC_TRY(esc_slab_reduce_jobs(edge_jobs.data(), (int)edge_jobs.size(), es.stream));   // (l == 0: the tail of the step follows at
      edge_jobs.clear();                                                                    // once — its slabs wait for the final reduce)
    }
  }
  if (!ss.ok) {
    Ctx cx = c;
    if (c.wgrad) { cx.y.dT1 = y.dT1x; cx.y.dT2 = y.dT2x; }
    ESC_TRY(mlp_backward(cx, m->xemb, y.xemb, b->x, y.C0, N, y.cat, W, y.dcat, W, nullptr, 0, fuse_node_act(c)));
  }
  // z_embedding + bag: the tail of the edge pipeline (d(z_emb) is complete in edge-stream order); it overlaps the
  // x_embedding backward queued above on the node stream
  const bool mat = g_materialise_edge_act != 0;
  // (the ReLU mask is recomputed from the pre-BN value even when the activation was materialised: one array less to read)
  Ctx ct = ce;
  ct.on_edge_stream = ce.on_edge_stream && g_cap_tail;    // the tail is the critical path: its GEMM runs at full occupancy
  // The tail is serial edge-sized work behind the last d_e: both of z_embedding's BatchNorm backwards lose a pass when they
  // ride on its Linear's backward (bit 2 of g_bn_fuse_bwd) — BatchNorm 1's apply on the staged dY, BatchNorm 0's column sums
  // from the dX epilogue: coef (2 launches), dX+dW, finalize, apply instead of 3 + 1 + 3
  if ((g_bn_fuse_bwd & 4) && mat && c.act == 1 && !sync_on(c) && y.bst_part_e != nullptr) {
    const esc_bn_bwd_fused f1 = bn_fused(y.Yz, H, y.zb1, 1);
    const esc_bn_bwd_next n0{y.bst_part_e, y.Zb, H, y.zb0.mean, y.zb0.invstd, y.zb0.scale, y.zb0.shift, 1};
    const float* slab_probe = ct.jobs ? *ct.slab_cursor : y.slabs;
    if (esc_linear_bwd_both_bn_ok(y.dZemb, H, &f1, y.A0, H, m->zlin.w, H, E, H, H, y.dAz, H, slab_probe, &n0)) {
      ESC_TRY(esc_bn_bwd_coef(y.Yz, H, nullptr, 0, y.dZemb, H, E, H, y.zb1.mean, y.zb1.invstd, m->zbn1.gamma, m->zbn1.beta, 1, y.zb1.coef,
                              m->zbn1.dgamma, m->zbn1.dbeta, ce.y.bn_scratch, ce.s));
      ESC_TRY(linear_backward_bn(ct, y.dZemb, H, f1, y.A0, H, nullptr, nullptr, m->zlin, E, y.dAz, H, 0, &n0));
      ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part_e, cdiv(E, esc_linear_bwd_bn_block_rows(E, H, H)), E, H, y.zb0.coef, m->zbn0.dgamma,
                                            m->zbn0.dbeta, ce.s));
      ESC_TRY(esc_bn_bwd_apply(y.Zb, H, nullptr, 0, y.dAz, H, E, H, y.zb0.mean, y.zb0.invstd, m->zbn0.gamma, m->zbn0.beta, 1, y.zb0.coef, y.dAz, H, ce.s));
      goto tail_bag;
    }
  }
  ESC_TRY(bn_backward(ce, y.Yz, H, nullptr, 0, y.dZemb, H, E, y.zb1, m->zbn1, y.dZemb, H, ce.y.bn_scratch));
  if (mat) {
    ESC_TRY(linear_backward(ct, y.dZemb, H, y.A0, H, nullptr, nullptr, m->zlin, E, y.dAz, H, 0));
  } else {
    ESC_TRY(linear_backward(ct, y.dZemb, H, y.Zb, H, y.zb0.scale, y.zb0.shift, m->zlin, E, y.dAz, H, 0));
  }
  ESC_TRY(bn_backward(ce, y.Zb, H, nullptr, 0, y.dAz, H, E, y.zb0, m->zbn0, y.dAz, H, ce.y.bn_scratch));
tail_bag:
  ESC_TRY(esc_bag_bwd_table_rows(y.dAz, H, H, b->col_ptr, b->col_row, b->col_val, b->col_col, y.Z, m->z_rows, E,
                                 1, m->dz_table, y.bag_scratch, ce.s));
  mark(PH_NODE_BWD_DONE, c.s);
  mark(PH_EDGE_BWD_DONE, ce.s);
  // node-side reductions first (with an edge stream they overlap its tail), then join, then the edge-side ones
  if (!eps_jobs.empty()) ESC_TRY(esc_reduce_sum_jobs(eps_jobs.data(), (int)eps_jobs.size(), c.s));
  if (c.wgrad) ESC_TRY(chain(ws.joined, ws.stream, (hipStream_t)c.s));          // the slabs of the dW tiles that ran behind the chain
  if (c.jobs && !c.jobs->empty()) ESC_TRY(esc_slab_reduce_jobs(c.jobs->data(), (int)c.jobs->size(), c.s));
  if (ss.ok && hipStreamWaitEvent((hipStream_t)c.s, ss.join_b, 0) != hipSuccess) {
    set_error("esc_engine: side-stream join failed");
    return ESC_ELAUNCH;
  }
  if (es.ok) {
    if (hipEventRecord(es.joined, es.stream) != hipSuccess) { set_error("esc_engine: stream event failed"); return ESC_ELAUNCH; }
    Pending local;
    Pending& p = defer ? *defer : local;
    p.open = true; p.join = true; p.stream = (hipStream_t)c.s;
    p.edge_jobs.swap(edge_jobs);
    if (!defer) return finish_pending(p);
  }
  return ESC_OK;
}

// =====================================================================================================================
// ZINC variant (zinc_models.py:504-611): one stream, ELU, materialised activations.  Reuses the helpers above through a
// Ctx whose count-model pointers are null.
// =====================================================================================================================
static Layout plan_layout_zinc(const esc_zinc_gin_t* m, int64_t N, int64_t E, int64_t Z, int64_t G, float* base, bool train) {
  Layout y{};
  Arena a{base, 0};
  const int64_t H = m->hidden, L = m->num_layers, D = m->edge_emb.dim, C0 = m->node_emb.dim;
  y.N = N; y.E = E; y.Z = Z; y.H = H; y.L = L; y.C0 = C0; y.W = L * H; y.G = G; y.D = D; y.Wz = H + D;
  y.X0 = a.take(N * C0);
  y.Zb = a.take(E * H); y.Yz = a.take(E * H); y.zb0 = take_bn(a, H); y.zb1 = take_bn(a, H);
  y.A0 = a.take(E * H); y.Zcat = a.take(E * y.Wz);
  const int64_t lb = C0 == H ? 0 : 1, nb = L - lb;            // the H-wide edge terms: layers lb .. L-1, batched into one GEMM (g_edge_batched)
  const bool batched = g_edge_batched && nb >= 2;
  if (batched) { y.e_cat = a.take(E * nb * H); y.w_cat = a.take(nb * H * y.Wz + nb * H); }
  for (int l = 0; l < L; ++l) {
    const int64_t C = l == 0 ? C0 : H;
    if (batched && l >= lb) { y.e[l] = base ? y.e_cat + (int64_t)(l - lb) * H : nullptr; y.ld_e[l] = nb * H; }
    else { y.e[l] = a.take(E * C); y.ld_e[l] = C; }
    y.agg[l] = a.take(N * C);
    y.conv[l].Y0 = a.take(N * H); y.conv[l].Y1 = a.take(N * H); y.conv[l].A1 = a.take(N * H);
    y.conv[l].b0 = take_bn(a, H); y.conv[l].b1 = take_bn(a, H);
  }
  y.cat = a.take(N * y.W); y.pooled = a.take(G * y.W); y.Yl = a.take(G * H); y.Al = a.take(G * H); y.bl = take_bn(a, H);
  y.pred = a.take(G); y.dpred = a.take(G);
  y.bn_scratch = a.take(esc_bn_scratch(H));
  y.col_stats = a.take(2 * ((E > N ? E : N) / 32 + 1) * H);
  y.col_stats_b = a.take(2 * (N / 32 + 1) * H);
  y.bn_scratch_e = a.take(esc_bn_scratch(H));
  y.col_stats_e = a.take(2 * (E / 32 + 1) * H);
  if (train) {
    y.dcat = a.take(N * y.W); y.dpool = a.take(G * y.W); y.dAl = a.take(G * H);
    y.bst_part = a.take(2 * (N / 4 + E / 64 + 4) * H);
    y.dT1 = a.take(N * H); y.dT2 = a.take(N * H); y.dagg = a.take(N * H); y.dX0 = a.take(N * C0);
    y.dZcat = a.take(E * y.Wz); y.dZemb = a.take(E * H); y.dAz = a.take(E * H);
    for (int l = 0; l < L; ++l) y.d_e[l] = a.take(E * (l == 0 ? C0 : H));
    y.deps_part = a.take(2 * N * (L > 0 ? L : 1));
    y.bag_scratch = a.take(esc_bag_bwd_scratch(Z, H));
    int64_t sl = esc_linear_bwd_weight_scratch(E, H, H) + 64;                                    // zlin
    for (int l = 0; l < L; ++l) {
      const int64_t C = l == 0 ? C0 : H;
      sl += esc_linear_bwd_weight_scratch(E, C, y.Wz) + esc_linear_bwd_weight_scratch(N, H, H) +
            esc_linear_bwd_weight_scratch(N, H, C) + 3 * 64;                                       // conv.lin, nn.lin1, nn.lin0
    }
    sl += esc_linear_bwd_weight_scratch(G, H, y.W) + esc_linear_bwd_weight_scratch(G, 1, H) + 2 * 64;   // lin1, lin2
    y.slabs = a.take(sl);
  }
  y.total = a.off;
  return y;
}

struct ZincCtx {
  const esc_zinc_gin_t* m;
  const esc_mol_batch_t* b;
  Ctx c;                       // helpers' view: layout, stream, job list, act = ELU
};

static int forward_zinc(const ZincCtx& z) {
  const esc_zinc_gin_t* m = z.m;
  const esc_mol_batch_t* b = z.b;
  const Ctx& c = z.c;
  const Layout& y = c.y;
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, W = y.W, G = y.G, D = y.D, Wz = y.Wz, C0 = y.C0;
  const int act = c.act;
  // ---- edge pipeline on the second stream (as in the counting engine): bag, z_embedding, edge-term input, edge terms
  EdgeStream& es = edge_stream_for(E);
  Ctx ce = c;
  if (es.ok) {
    ESC_TRY(chain(es.z_ready, (hipStream_t)c.s, es.stream));
    ce = edge_ctx(c, es.stream);
  }
  // z_emb = z_embedding(ESC bag) (:589-590), written into the first H columns of the edge-term input; the last D
  // columns are edge_type_embedding(edge_attr) (:591)
  ESC_TRY(esc_bag_fwd_rows(m->z_table, m->z_rows, H, b->row_ptr, b->bag_idx, b->bag_val, E, y.Zb, H, 0, nullptr, ce.s));
  if (c.train) ESC_TRY(esc_bag_bwd_classify(b->col_row, y.Z, H, E, y.bag_scratch, ce.s));
  ESC_TRY(bn_coeffs(ce, y.Zb, H, E, m->zbn0, y.zb0));
  ESC_TRY(esc_affine_act(y.Zb, H, E, H, y.zb0.scale, y.zb0.shift, act, y.A0, H, ce.s));
  ESC_TRY(linear_bn(ce, y.A0, H, m->zlin, nullptr, nullptr, E, y.Yz, m->zbn1, y.zb1));
  ESC_TRY(esc_affine_act(y.Yz, H, E, H, y.zb1.scale, y.zb1.shift, act, y.Zcat, Wz, ce.s));
  ESC_TRY(esc_embed_fwd(m->edge_emb.w, m->edge_emb.rows, D, b->edge_type, E, y.Zcat + H, Wz, nullptr, ce.s));
  auto edge_term = [&](int l) -> int {
    const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? C0 : H;
    ESC_TRY(esc_linear_fwd(y.Zcat, Wz, cv.lin.w, Wz, cv.lin.b, nullptr, nullptr, E, C, Wz, y.e[l], y.ld_e[l], nullptr, ce.s));
    if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_zinc: stream event failed"); return ESC_ELAUNCH; }
    return ESC_OK;
  };
  const bool batched = y.e_cat != nullptr;
  if (batched) {                      // every H-wide edge term in one launch over the packed weights (rows of Wz floats) ++ biases
    const int lb = C0 == H ? 0 : 1, nb = (int)L - lb;
    if (lb == 1) ESC_TRY(edge_term(0));
    // (two packs: the weight rows are Wz wide, the biases H wide)
    esc_table_list tw{}, tb{};
    tw.count = nb; tb.count = nb;
    for (int l = lb; l < (int)L; ++l) {
      tw.rows[l - lb] = (int32_t)H; tw.w[l - lb] = m->conv[l].lin.w;
      tb.rows[l - lb] = 1; tb.w[l - lb] = m->conv[l].lin.b;
    }
    ESC_TRY(esc_table_pack(&tw, Wz, y.w_cat, ce.s));
    ESC_TRY(esc_table_pack(&tb, H, y.w_cat + (int64_t)nb * H * Wz, ce.s));
    {
      const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
      ESC_TRY(esc_linear_fwd(y.Zcat, Wz, y.w_cat, Wz, y.w_cat + (int64_t)nb * H * Wz, nullptr, nullptr, E, nb * H, Wz, y.e_cat, nb * H, nullptr, ce.s));
    }
    for (int l = lb; l < (int)L; ++l)
      if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_zinc: stream event failed"); return ESC_ELAUNCH; }
  }
  const int ahead = batched ? 0 : (es.ok ? g_edge_ahead : (int)L);
  for (int l = 0; l < (int)L && l < ahead; ++l) ESC_TRY(edge_term(l));
  // ---- node pipeline: x = node_type_embedding(data.x) (:581)
  ESC_TRY(esc_embed_fwd(m->node_emb.w, m->node_emb.rows, C0, b->node_type, N, y.X0, C0, nullptr, c.s));
  // GINE layers (:593-598): xs[l] -> cat[:, l*H : (l+1)*H]
  for (int l = 0; l < (int)L; ++l) {
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? C0 : H;
    const float* hin = l == 0 ? y.X0 : y.cat + (int64_t)(l - 1) * H;
    const int64_t ld_h = l == 0 ? C0 : W;
    if (es.ok && hipStreamWaitEvent((hipStream_t)c.s, es.e_ready[l], 0) != hipSuccess) { set_error("esc_zinc: stream event failed"); return ESC_ELAUNCH; }
    ESC_TRY(esc_gine_aggregate_fwd(hin, ld_h, y.e[l], y.ld_e[l], b->in_ptr, b->in_edge, b->in_src, cv.eps, N, C, y.agg[l], C, c.s));
    if (!batched && es.ok && l + ahead < (int)L) {
      ESC_TRY(chain(es.agg_done[l], (hipStream_t)c.s, es.stream));
      ESC_TRY(edge_term(l + ahead));
    }
    ESC_TRY(mlp_forward(c, cv.nn, y.conv[l], y.agg[l], C, N, y.cat + (int64_t)l * H, W));
  }
  // readout (:601-609): global_add_pool -> lin1 -> bn_lin1 -> ELU -> lin2
  ESC_TRY(esc_segment_pool_fwd(y.cat, W, b->graph_ptr, G, W, 0, y.pooled, W, c.s));
  ESC_TRY(linear_bn(c, y.pooled, W, m->lin1, nullptr, nullptr, G, y.Yl, m->bn_lin1, y.bl));
  ESC_TRY(esc_affine_act(y.Yl, H, G, H, y.bl.scale, y.bl.shift, act, y.Al, H, c.s));
  ESC_TRY(esc_linear_fwd(y.Al, H, m->lin2.w, H, m->lin2.b, nullptr, nullptr, G, 1, H, y.pred, 1, nullptr, c.s));
  if (es.ok) ESC_TRY(chain(es.joined, es.stream, (hipStream_t)c.s));    // forward-only calls return ordered behind the edge stream
  return ESC_OK;
}

static int backward_zinc(const ZincCtx& z) {
  const esc_zinc_gin_t* m = z.m;
  const esc_mol_batch_t* b = z.b;
  const Ctx& c = z.c;
  const Layout& y = c.y;
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, W = y.W, G = y.G, D = y.D, Wz = y.Wz, C0 = y.C0;
  EdgeStream& es = edge_stream_for(E);
  Ctx ce = es.ok ? edge_ctx(c, es.stream) : c;
  std::vector<esc_reduce_job> edge_jobs;
  edge_jobs.reserve(ESC_MAX_REDUCE_JOBS);
  if (es.ok && c.jobs) ce.jobs = &edge_jobs;
  if (es.ok) ESC_TRY(chain(es.z_ready, (hipStream_t)c.s, es.stream));
  ESC_TRY(linear_backward(c, y.dpred, 1, y.Al, H, nullptr, nullptr, m->lin2, G, y.dAl, H, 0));
  ESC_TRY(bn_backward(c, y.Yl, H, y.Al, H, y.dAl, H, G, y.bl, m->bn_lin1, y.dAl, H, y.bn_scratch));
  ESC_TRY(linear_backward(c, y.dAl, H, y.pooled, W, nullptr, nullptr, m->lin1, G, y.dpool, W, 0));
  ESC_TRY(esc_segment_pool_bwd(y.dpool, W, b->graph_ptr, G, W, 0, y.dcat, W, c.s));
  std::vector<esc_sum_job> eps_jobs;
  for (int l = (int)L - 1; l >= 0; --l) {
    const esc_conv_t& cv = m->conv[l];
    const int64_t C = l == 0 ? C0 : H;
    const float* hin = l == 0 ? y.X0 : y.cat + (int64_t)(l - 1) * H;
    const int64_t ld_h = l == 0 ? C0 : W;
    ESC_TRY(mlp_backward(c, cv.nn, y.conv[l], y.agg[l], C, N, y.cat + (int64_t)l * H, W, y.dcat + (int64_t)l * H, W, y.dagg, C));
    float* dx = l == 0 ? y.dX0 : y.dcat + (int64_t)(l - 1) * H;
    ESC_TRY(esc_gine_aggregate_bwd(hin, ld_h, y.e[l], y.ld_e[l], y.dagg, C, b->out_ptr, b->out_edge, b->out_dst, cv.eps, N, C,
                                   y.d_e[l], C, dx, l == 0 ? C0 : W, l == 0 ? 0 : 1, y.deps_part + (int64_t)l * 2 * N, c.s));
    eps_jobs.push_back(esc_sum_job{y.deps_part + (int64_t)l * 2 * N, N * esc_gine_aggregate_bwd_deps_slots(C), cv.deps});
    if (es.ok) ESC_TRY(chain(es.de_ready[l], (hipStream_t)c.s, es.stream));      // conv.lin backward: edge stream
    ESC_TRY(linear_backward(ce, y.d_e[l], C, y.Zcat, Wz, nullptr, nullptr, cv.lin, E, y.dZcat, Wz, l == (int)L - 1 ? 0 : 1));
  }
  ESC_TRY(esc_embed_bwd(y.dX0, C0, b->node_type, N, m->node_emb.rows, C0, m->node_emb.dw, c.s));
  // edge pipeline tail: edge-type table, z_embedding, bag
  ESC_TRY(esc_embed_bwd(y.dZcat + H, Wz, b->edge_type, E, m->edge_emb.rows, D, m->edge_emb.dw, ce.s));
  // z_embedding's two BatchNorms ride on its Linear's backward when the fused kernels serve the shape (molecule batches: a few
  // thousand edge rows, the node-sized tile): coef, dX+dW (apply on the staged dY, the first BatchNorm's sums from the dX epilogue),
  // finalize, apply — instead of 3 + 1 + 3 launches
  bool z_fused = false;
  if ((g_bn_fuse_bwd & 3) == 3 && g_bn_fuse_elu && !sync_on(c) && y.bst_part != nullptr) {
    const esc_bn_bwd_fused f1 = bn_fused(y.Yz, H, y.zb1, c.act);
    const esc_bn_bwd_next n0{y.bst_part, y.Zb, H, y.zb0.mean, y.zb0.invstd, y.zb0.scale, y.zb0.shift, c.act};
    const float* slab_probe = ce.jobs ? *ce.slab_cursor : y.slabs;
    if (esc_linear_bwd_both_bn_ok(y.dZcat, Wz, &f1, y.A0, H, m->zlin.w, H, E, H, H, y.dAz, H, slab_probe, &n0)) {
      ESC_TRY(esc_bn_bwd_coef(y.Yz, H, nullptr, 0, y.dZcat, Wz, E, H, y.zb1.mean, y.zb1.invstd, m->zbn1.gamma, m->zbn1.beta, c.act, y.zb1.coef,
                              m->zbn1.dgamma, m->zbn1.dbeta, ce.y.bn_scratch, ce.s));
      ESC_TRY(linear_backward_bn(ce, y.dZcat, Wz, f1, y.A0, H, nullptr, nullptr, m->zlin, E, y.dAz, H, 0, &n0));
      ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part, cdiv(E, esc_linear_bwd_bn_block_rows(E, H, H)), E, H, y.zb0.coef, m->zbn0.dgamma,
                                            m->zbn0.dbeta, ce.s));
      ESC_TRY(esc_bn_bwd_apply(y.Zb, H, nullptr, 0, y.dAz, H, E, H, y.zb0.mean, y.zb0.invstd, m->zbn0.gamma, m->zbn0.beta, c.act, y.zb0.coef, y.dAz, H, ce.s));
      z_fused = true;
    }
  }
  if (!z_fused) {
    ESC_TRY(bn_backward(ce, y.Yz, H, y.Zcat, Wz, y.dZcat, Wz, E, y.zb1, m->zbn1, y.dZemb, H, ce.y.bn_scratch));
    ESC_TRY(linear_backward(ce, y.dZemb, H, y.A0, H, nullptr, nullptr, m->zlin, E, y.dAz, H, 0));
    ESC_TRY(bn_backward(ce, y.Zb, H, y.A0, H, y.dAz, H, E, y.zb0, m->zbn0, y.dAz, H, ce.y.bn_scratch));
  }
  ESC_TRY(esc_bag_bwd_table_rows(y.dAz, H, H, b->col_ptr, b->col_row, b->col_val, b->col_col, y.Z, m->z_rows, E, 1,
                                 m->dz_table, y.bag_scratch, ce.s));
  if (es.ok && !edge_jobs.empty()) ESC_TRY(esc_slab_reduce_jobs(edge_jobs.data(), (int)edge_jobs.size(), es.stream));
  if (!eps_jobs.empty()) ESC_TRY(esc_reduce_sum_jobs(eps_jobs.data(), (int)eps_jobs.size(), c.s));
  if (c.jobs && !c.jobs->empty()) ESC_TRY(esc_slab_reduce_jobs(c.jobs->data(), (int)c.jobs->size(), c.s));
  if (es.ok) ESC_TRY(chain(es.joined, es.stream, (hipStream_t)c.s));
  return ESC_OK;
}

static int check_zinc(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, const float* ws, bool train, bool need_y) {
  ESC_REQUIRE(m && b && ws, "esc_zinc: null pointer");
  ESC_REQUIRE(m->num_layers >= 1 && m->num_layers <= ESC_MAX_LAYERS, "esc_zinc: %ld layers unsupported", (long)m->num_layers);
  ESC_REQUIRE(m->hidden > 0 && m->hidden % 4 == 0, "esc_zinc: hidden must be a multiple of 4");
  ESC_REQUIRE(m->node_emb.dim > 0 && m->node_emb.dim % 4 == 0 && m->edge_emb.dim > 0 && m->edge_emb.dim % 4 == 0,
              "esc_zinc: embedding widths must be multiples of 4");
  ESC_REQUIRE(m->conv[0].lin.in_dim == m->hidden + m->edge_emb.dim && m->conv[0].lin.out_dim == m->node_emb.dim,
              "esc_zinc: conv1.lin must map hidden + edge width to the node width");
  ESC_REQUIRE(b->N >= 2 && b->E >= 2 && b->Z >= 0 && b->G >= 2, "esc_zinc: batch needs >= 2 nodes, edges and graphs (BatchNorm statistics)");
  ESC_REQUIRE(b->node_type && b->edge_type && b->graph_ptr && b->in_ptr && b->row_ptr &&
              (!train || ((b->y || !need_y) && b->out_ptr && b->col_ptr)), "esc_zinc: null batch arrays");
  ESC_REQUIRE(aligned16(ws), "esc_zinc: workspace must be 16-byte aligned");
  return ESC_OK;
}

static ZincCtx make_zinc(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* ws, void* stream, bool train) {
  ZincCtx z{m, b, Ctx{nullptr, nullptr, plan_layout_zinc(m, b->N, b->E, b->Z, b->G, ws, train), stream, train}};
  z.c.act = 2;
  return z;
}

// =====================================================================================================================
// OGB molecule variant (ogb_mol_gnn.py: GNN(gnn_type='gin_eff'), virtual node, JK last): one stream, ReLU,
// materialised activations, dropout with its own counter-based stream.
// =====================================================================================================================
struct OgbLayer {
  float *hin, *e, *agg, *Y0, *A1, *hc, *hb;       // hin = h + vn[batch]; Y0/A1 [N,2H]; hc pre-BN; hb after BN(+ReLU)
  BnWs b0, bn;
  unsigned char* mask_h;
  float *tmp, *V0, *VA, *V1, *VB, *vn;             // virtual-node update (rows G); vn = the embedding ENTERING layer l
  BnWs vb0, vb1;
  unsigned char* mask_v;
  float* d_e;
};
struct OgbLayout {
  int64_t N, E, Z, G, H, L, T;
  float *Tcat, *dTcat, *h0, *hL;
  float *Zb, *Zd, *A0, *Yz, *Yzd, *Zemb; BnWs zb0, zb1;
  unsigned char *mask_z0, *mask_z1;
  OgbLayer l[ESC_MAX_LAYERS];
  float *e_cat, *w_cat; int64_t ld_e;     // g_edge_batched: every layer's edge term is a column block of ONE [E, L*H] matrix
  float *h[ESC_MAX_LAYERS + 1];
  float *pooled, *logits, *dlogits;
  // backward
  float *dpooled, *dHa, *dHb, *dT, *dA1, *dagg, *dZemb, *dYz, *dA0, *dvn_a, *dvn_b, *dG1, *dG2, *dtmp, *poolG;
  float *deps_part, *bag_scratch, *emb_scratch, *emb_scratch_n, *slabs, *bn_scratch, *col_stats, *bn_scratch_e, *col_stats_e;
  float *bst_part;                 // column sums of a node MLP's first BatchNorm backward, left by its second Linear's dX epilogue
  int64_t total;
};

static unsigned char* take_bytes(Arena& a, int64_t n) { return reinterpret_cast<unsigned char*>(a.take((n + 3) / 4)); }

static OgbLayout plan_layout_ogb(const esc_ogb_gnn_t* m, int64_t N, int64_t E, int64_t Z, int64_t G, int64_t atom_entries,
                                 int64_t bond_entries, float* base, bool train) {
  OgbLayout y{};
  Arena a{base, 0};
  const int64_t H = m->hidden, L = m->num_layers, T = m->num_tasks, H2 = 2 * H;
  const bool drop = train && m->drop_ratio > 0.f;
  y.N = N; y.E = E; y.Z = Z; y.G = G; y.H = H; y.L = L; y.T = T;
  const int64_t rows_total = m->atom_rows + L * m->bond_rows;
  y.Tcat = a.take(rows_total * H);
  y.h0 = a.take(N * H); y.h[0] = y.h0;
  y.Zb = a.take(E * H); y.Zd = drop ? a.take(E * H) : y.Zb; y.A0 = a.take(E * H);
  y.Yz = a.take(E * H); y.Yzd = drop ? a.take(E * H) : y.Yz; y.Zemb = a.take(E * H);
  y.zb0 = take_bn(a, H); y.zb1 = take_bn(a, H);
  if (drop) { y.mask_z0 = take_bytes(a, E * H); y.mask_z1 = take_bytes(a, E * H); }
  // see plan_layout(): the edge terms of layers 1 .. L-1 from one GEMM (layer 0's keeps its own launch: the first aggregate waits for it)
  const bool batched = g_edge_batched >= 2 && L >= 3 && H % 4 == 0;
  y.ld_e = batched ? (L - 1) * H : H;
  if (batched) { y.e_cat = a.take(E * (L - 1) * H); y.w_cat = a.take((L - 1) * H * H + (L - 1) * H); }
  for (int l = 0; l < L; ++l) {
    OgbLayer& q = y.l[l];
    q.vn = a.take(G * H);
    q.hin = a.take(N * H);
    q.e = (batched && l >= 1) ? y.e_cat + (int64_t)(l - 1) * H : a.take(E * H);
    q.agg = a.take(N * H);
    q.Y0 = a.take(N * H2); q.A1 = a.take(N * H2); q.hc = a.take(N * H); q.hb = nullptr;         // (hb is never materialised: BN -> ReLU -> dropout in one pass)
    q.b0 = take_bn(a, H2); q.bn = take_bn(a, H);
    y.h[l + 1] = a.take(N * H);
    if (drop) q.mask_h = take_bytes(a, N * H);
    if (l < L - 1) {
      q.tmp = a.take(G * H); q.V0 = a.take(G * H2); q.VA = a.take(G * H2); q.V1 = a.take(G * H); q.VB = nullptr;
      q.vb0 = take_bn(a, H2); q.vb1 = take_bn(a, H);
      if (drop) q.mask_v = take_bytes(a, G * H);
    }
  }
  y.hL = y.h[L];
  y.pooled = a.take(G * H); y.logits = a.take(G * T); y.dlogits = a.take(G * T);
  y.bn_scratch = a.take(esc_bn_scratch(H2));
  y.col_stats = a.take(2 * ((E > N ? E : N) / 32 + 1) * H2);
  y.bn_scratch_e = a.take(esc_bn_scratch(H2));               // the edge pipeline's own BatchNorm scratch / GEMM-epilogue partials
  y.col_stats_e = a.take(2 * (E / 32 + 1) * H2);
  if (train) {
    y.dTcat = a.take(rows_total * H);
    y.bst_part = a.take(2 * (N / 64 + 2) * H2);
    y.dpooled = a.take(G * H); y.dHa = a.take(N * H); y.dHb = a.take(N * H); y.dT = a.take(N * H);
    y.dA1 = a.take(N * H2); y.dagg = a.take(N * H);
    y.dZemb = a.take(E * H); y.dYz = a.take(E * H); y.dA0 = a.take(E * H);
    y.dvn_a = a.take(G * H); y.dvn_b = a.take(G * H); y.dG1 = a.take(G * H2); y.dG2 = a.take(G * H); y.dtmp = a.take(G * H);
    y.poolG = a.take(G * H);
    for (int l = 0; l < L; ++l) y.l[l].d_e = a.take(E * H);
    y.deps_part = a.take(2 * N * (L > 0 ? L : 1));
    y.bag_scratch = a.take(esc_bag_bwd_scratch(Z, H));
    const int64_t ent = atom_entries > bond_entries ? atom_entries : bond_entries;
    y.emb_scratch = a.take(esc_bag_bwd_scratch(ent, H));
    y.emb_scratch_n = a.take(esc_bag_bwd_scratch(ent, H));
    int64_t sl = esc_linear_bwd_weight_scratch(E, H, H) + 64;                                   // zlin
    for (int l = 0; l < L; ++l) {
      sl += esc_linear_bwd_weight_scratch(E, H, H) + esc_linear_bwd_weight_scratch(N, H2, H) +
            esc_linear_bwd_weight_scratch(N, H, H2) + 3 * 64;                                     // pos, lin0, lin1
      if (l < L - 1) sl += esc_linear_bwd_weight_scratch(G, H2, H) + esc_linear_bwd_weight_scratch(G, H, H2) + 2 * 64;
    }
    sl += esc_linear_bwd_weight_scratch(G, T, H) + 64;                                          // head
    y.slabs = a.take(sl);
  }
  y.total = a.off;
  return y;
}

struct OgbCtx {
  const esc_ogb_gnn_t* m;
  const esc_ogb_batch_t* b;
  OgbLayout o;
  Ctx c;                // helpers' view (H, col_stats, bn_scratch, slabs, stream, jobs); act = ReLU
  float p;              // dropout probability of this call (0 in eval mode)
};

static uint64_t drop_seed(const OgbCtx& z, int which) { return z.b->seed * 0x2545F4914F6CDD1Dull + (uint64_t)(which + 1) * 0xD1342543DE82EF95ull; }

static int forward_ogb(const OgbCtx& z) {
  const esc_ogb_gnn_t* m = z.m;
  const esc_ogb_batch_t* b = z.b;
  const OgbLayout& y = z.o;
  const Ctx& c = z.c;
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, G = y.G, T = y.T, H2 = 2 * H;
  const float p = z.p;
  // encoders' tables in one buffer (both pipelines read it)
  ESC_TRY(esc_table_pack(&m->tables, H, y.Tcat, c.s));
  // ---- edge pipeline (second stream, like the counting engine): ESC bag, z_embedding, the edge terms of all layers
  EdgeStream& es = edge_stream_for(E);
  Ctx ce = c;
  if (es.ok) {
    ESC_TRY(chain(es.z_ready, (hipStream_t)c.s, es.stream));
    ce = edge_ctx(c, es.stream);
  }
  // z_emb = z_embedding(ESC bag): Dropout BN ReLU Linear Dropout BN ReLU (:638-645)
  ESC_TRY(esc_bag_fwd_rows(m->z_table, m->z_rows, H, b->row_ptr, b->bag_idx, b->bag_val, E, y.Zb, H, 0, nullptr, ce.s));
  if (c.train) ESC_TRY(esc_bag_bwd_classify(b->col_row, y.Z, H, E, y.bag_scratch, ce.s));
  if (p > 0.f) ESC_TRY(esc_dropout_fwd(y.Zb, H, E, H, p, drop_seed(z, 0), nullptr, 0, y.Zd, H, y.mask_z0, ce.s));
  ESC_TRY(bn_coeffs(ce, y.Zd, H, E, m->zbn0, y.zb0));
  ESC_TRY(esc_affine_act(y.Zd, H, E, H, y.zb0.scale, y.zb0.shift, 1, y.A0, H, ce.s));
  if (p > 0.f) {
    ESC_TRY(esc_linear_fwd(y.A0, H, m->zlin.w, H, m->zlin.b, nullptr, nullptr, E, H, H, y.Yz, H, nullptr, ce.s));
    ESC_TRY(esc_dropout_fwd(y.Yz, H, E, H, p, drop_seed(z, 1), nullptr, 0, y.Yzd, H, y.mask_z1, ce.s));
    ESC_TRY(bn_coeffs(ce, y.Yzd, H, E, m->zbn1, y.zb1));
  } else {
    ESC_TRY(linear_bn(ce, y.A0, H, m->zlin, nullptr, nullptr, E, y.Yz, m->zbn1, y.zb1));
  }
  ESC_TRY(esc_affine_act(y.Yzd, H, E, H, y.zb1.scale, y.zb1.shift, 1, y.Zemb, H, ce.s));
  // edge term of layer l = BondEncoder(edge_attr) + edge_encoder_pos(z_emb) (:352)
  auto edge_term = [&](int l) -> int {
    const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
    const esc_ogb_layer_t& q = m->layer[l];
    const int64_t ld = (l == 0 || y.ld_e == H) ? H : y.ld_e;
    ESC_TRY(esc_linear_fwd(y.Zemb, H, q.pos.w, H, q.pos.b, nullptr, nullptr, E, H, H, y.l[l].e, ld, nullptr, ce.s));
    ESC_TRY(esc_bag_fwd_rows(y.Tcat + q.bond_row0 * H, m->bond_rows, H, b->bonds.row_ptr, b->bonds.idx, b->bonds.ones, E, y.l[l].e, ld, 1, nullptr, ce.s));
    if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_ogb: stream event failed"); return ESC_ELAUNCH; }
    return ESC_OK;
  };
  const bool batched = y.ld_e != H;
  if (batched) {                        // layer 0's edge term first, then edge_encoder_pos of layers 1 .. L-1 in one launch + their bond sums
    ESC_TRY(edge_term(0));
    const int nb = (int)L - 1;
    esc_table_list tl{};
    tl.count = 2 * nb;
    for (int l = 1; l < (int)L; ++l) {
      tl.rows[l - 1] = (int32_t)H; tl.w[l - 1] = m->layer[l].pos.w;
      tl.rows[nb + l - 1] = 1; tl.w[nb + l - 1] = m->layer[l].pos.b;
    }
    ESC_TRY(esc_table_pack(&tl, H, y.w_cat, ce.s));
    {
      const LdsFloorGuard cap(ce.on_edge_stream && g_cap_forward);
      ESC_TRY(esc_linear_fwd(y.Zemb, H, y.w_cat, H, y.w_cat + (int64_t)nb * H * H, nullptr, nullptr, E, nb * H, H, y.e_cat, nb * H, nullptr, ce.s));
    }
    for (int l = 1; l < (int)L; ++l) {
      ESC_TRY(esc_bag_fwd_rows(y.Tcat + m->layer[l].bond_row0 * H, m->bond_rows, H, b->bonds.row_ptr, b->bonds.idx, b->bonds.ones, E, y.l[l].e, y.ld_e, 1, nullptr, ce.s));
      if (es.ok && hipEventRecord(es.e_ready[l], es.stream) != hipSuccess) { set_error("esc_ogb: stream event failed"); return ESC_ELAUNCH; }
    }
  }
  const int ahead = batched ? 0 : (es.ok ? g_edge_ahead : (int)L);
  for (int l = 0; l < (int)L && l < ahead; ++l) ESC_TRY(edge_term(l));
  // ---- node pipeline: h0 = AtomEncoder(x) (:264-282); vn_0 = virtualnode_embedding(0) per graph (:701)
  ESC_TRY(esc_bag_fwd_rows(y.Tcat, m->atom_rows, H, b->atoms.row_ptr, b->atoms.idx, b->atoms.ones, N, y.h0, H, 0, nullptr, c.s));
  ESC_TRY(esc_embed_fwd(m->vn_w, 1, H, b->zero_idx, G, y.l[0].vn, H, nullptr, c.s));
  for (int l = 0; l < (int)L; ++l) {
    const esc_ogb_layer_t& q = m->layer[l];
    const OgbLayer& w = y.l[l];
    ESC_TRY(esc_segment_broadcast_add(y.h[l], H, w.vn, H, b->graph_ptr, G, N, H, w.hin, H, c.s));                 // :739
    if (es.ok && hipStreamWaitEvent((hipStream_t)c.s, es.e_ready[l], 0) != hipSuccess) { set_error("esc_ogb: stream event failed"); return ESC_ELAUNCH; }
    ESC_TRY(esc_gine_aggregate_fwd(w.hin, H, w.e, l == 0 ? H : y.ld_e, b->in_ptr, b->in_edge, b->in_src, q.eps, N, H, w.agg, H, c.s));
    if (!batched && es.ok && l + ahead < (int)L) {
      ESC_TRY(chain(es.agg_done[l], (hipStream_t)c.s, es.stream));
      ESC_TRY(edge_term(l + ahead));
    }
    ESC_TRY(linear_bn(c, w.agg, H, q.lin0, nullptr, nullptr, N, w.Y0, q.bn0, w.b0));
    if (g_ogb_prologue) {                       // relu(bn(Y0)) applied to the staged operand of lin1: A1 is never written
      ESC_TRY(linear_bn(c, w.Y0, H2, q.lin1, w.b0.scale, w.b0.shift, N, w.hc, q.bn, w.bn));
    } else {
      ESC_TRY(esc_affine_act(w.Y0, H2, N, H2, w.b0.scale, w.b0.shift, 1, w.A1, H2, c.s));
      ESC_TRY(linear_bn(c, w.A1, H2, q.lin1, nullptr, nullptr, N, w.hc, q.bn, w.bn));                              // + batch_norms[l] statistics
    }
    // batch_norm -> ReLU (not after the last layer) -> dropout (+ residual), :744-755, as one pass over hc
    ESC_TRY(esc_affine_act_dropout_fwd(w.hc, H, N, H, w.bn.scale, w.bn.shift, l == (int)L - 1 ? 0 : 1, p, drop_seed(z, 2 + 2 * l),
                                       m->residual ? w.hin : nullptr, H, y.h[l + 1], H, w.mask_h, c.s));
    if (l < (int)L - 1) {                                                                                            // :757-783
      ESC_TRY(esc_segment_pool_fwd(w.hin, H, b->graph_ptr, G, H, 0, w.tmp, H, c.s));
      ESC_TRY(esc_dropout_fwd(w.tmp, H, G, H, 0.f, 0, w.vn, H, w.tmp, H, nullptr, c.s));                            // + vn
      ESC_TRY(linear_bn(c, w.tmp, H, q.vlin0, nullptr, nullptr, G, w.V0, q.vbn0, w.vb0));
      ESC_TRY(esc_affine_act(w.V0, H2, G, H2, w.vb0.scale, w.vb0.shift, 1, w.VA, H2, c.s));
      ESC_TRY(linear_bn(c, w.VA, H2, q.vlin1, nullptr, nullptr, G, w.V1, q.vbn1, w.vb1));
      ESC_TRY(esc_affine_act_dropout_fwd(w.V1, H, G, H, w.vb1.scale, w.vb1.shift, 1, p, drop_seed(z, 3 + 2 * l),
                                         m->residual ? w.vn : nullptr, H, y.l[l + 1].vn, H, w.mask_v, c.s));
    }
  }
  ESC_TRY(esc_segment_pool_fwd(y.hL, H, b->graph_ptr, G, H, m->mean_pool, y.pooled, H, c.s));
  ESC_TRY(esc_linear_fwd(y.pooled, H, m->head.w, H, m->head.b, nullptr, nullptr, G, T, H, y.logits, T, nullptr, c.s));
  // a forward-only call (predict / forward_train) returns with the caller's stream ordered behind the edge stream too
  if (es.ok) ESC_TRY(chain(es.joined, es.stream, (hipStream_t)c.s));
  return ESC_OK;
}

static int backward_ogb(const OgbCtx& z) {
  const esc_ogb_gnn_t* m = z.m;
  const esc_ogb_batch_t* b = z.b;
  const OgbLayout& y = z.o;
  const Ctx& c = z.c;
  Ctx c0 = c; c0.act = 0;                        // the last layer's batch_norm has no ReLU
  const int64_t N = y.N, E = y.E, H = y.H, L = y.L, G = y.G, T = y.T, H2 = 2 * H;
  const float p = z.p;
  ESC_TRY(linear_backward(c, y.dlogits, T, y.pooled, H, nullptr, nullptr, m->head, G, y.dpooled, H, 0));
  float* dH = y.dHa;                             // d h_{l+1}
  float* dHin = y.dHb;                           // d (h_l + vn_l[batch]) under construction
  ESC_TRY(esc_segment_pool_bwd(y.dpooled, H, b->graph_ptr, G, H, m->mean_pool, dH, H, c.s));
  float* dvn_next = nullptr;                     // d vn_{l+1}
  float* dvn_cur = y.dvn_a;
  std::vector<esc_sum_job> eps_jobs;
  // the edge pipeline's backward (bond tables, edge_encoder_pos, z_embedding, bag) runs on the second stream behind d_e[l]
  EdgeStream& es = edge_stream_for(E);
  Ctx ce = es.ok ? edge_ctx(c, es.stream) : c;
  std::vector<esc_reduce_job> edge_jobs;
  edge_jobs.reserve(ESC_MAX_REDUCE_JOBS);
  if (es.ok && c.jobs) ce.jobs = &edge_jobs;
  if (es.ok) ESC_TRY(chain(es.z_ready, (hipStream_t)c.s, es.stream));      // (a backward called on its own: order behind the caller's stream)
  for (int l = (int)L - 1; l >= 0; --l) {
    const esc_ogb_layer_t& q = m->layer[l];
    const OgbLayer& w = y.l[l];
    const bool last = l == (int)L - 1;
    // h_{l+1} = dropout(hb) (+ hin)
    ESC_TRY(bn_backward_drop(last ? c0 : c, w.hc, H, dH, H, N, w.bn, q.bn, w.mask_h, p, 0, y.dT, H, y.bn_scratch));
    // The hidden BatchNorm's backward loses its partial-sum pass (2H-wide rows: the most expensive of its three launches): the
    // column sums come out of lin1's dX epilogue (esc_linear_bwd_both_bn with bn == NULL), then finalize + apply
    bool hidden_done = false, hidden_fused = false;
    if (g_ogb_prologue && (g_bn_fuse_bwd & 2) && !sync_on(c) && y.bst_part != nullptr) {
      const esc_bn_bwd_next n0{y.bst_part, w.Y0, H2, w.b0.mean, w.b0.invstd, w.b0.scale, w.b0.shift, 1};
      const float* slab_probe = c.jobs ? *c.slab_cursor : c.y.slabs;
      if (esc_linear_bwd_both_bn_ok(y.dT, H, nullptr, w.Y0, H2, q.lin1.w, H2, N, H, H2, y.dA1, H2, slab_probe, &n0)) {
        float* slabs = c.y.slabs;
        esc_reduce_job* job = nullptr;
        if (c.jobs) {
          slabs = *c.slab_cursor;
          *c.slab_cursor += (esc_linear_bwd_weight_scratch(N, H, H2) + 63) & ~63LL;
          c.jobs->emplace_back();
          job = &c.jobs->back();
        }
        ESC_TRY(esc_linear_bwd_both_bn(y.dT, H, nullptr, w.Y0, H2, w.b0.scale, w.b0.shift, q.lin1.w, H2, N, H, H2, y.dA1, H2, 0, q.lin1.dw, H2,
                                       q.lin1.db, slabs, job, &n0, c.s));
        ESC_TRY(esc_bn_bwd_coef_from_partials(y.bst_part, cdiv(N, esc_linear_bwd_bn_block_rows(N, H, H2)), N, H2, w.b0.coef, q.bn0.dgamma,
                                              q.bn0.dbeta, c.s));
        // the apply of the hidden BatchNorm's backward rides on lin0's dX+dW launch (operand staging) when the fused kernels serve the shape
        // (ESC_OGB_BNB=1; measured below)
        {
          const esc_bn_bwd_fused f0 = bn_fused(w.Y0, H2, w.b0, 1);
          const float* probe = c.jobs ? *c.slab_cursor : c.y.slabs;
          if (g_ogb_bnb && esc_linear_bwd_both_bn_ok(y.dA1, H2, &f0, w.agg, H, q.lin0.w, H, N, H2, H, y.dagg, H, probe, nullptr)) {
            ESC_TRY(linear_backward_bn(c, y.dA1, H2, f0, w.agg, H, nullptr, nullptr, q.lin0, N, y.dagg, H, 0, nullptr));
            hidden_fused = true;
          } else {
            ESC_TRY(esc_bn_bwd_apply(w.Y0, H2, nullptr, 0, y.dA1, H2, N, H2, w.b0.mean, w.b0.invstd, q.bn0.gamma, q.bn0.beta, 1, w.b0.coef, y.dA1, H2, c.s));
          }
        }
        hidden_done = true;
      }
    }
    if (!hidden_done) {
      if (g_ogb_prologue) ESC_TRY(linear_backward(c, y.dT, H, w.Y0, H2, w.b0.scale, w.b0.shift, q.lin1, N, y.dA1, H2, 0));
      else                ESC_TRY(linear_backward(c, y.dT, H, w.A1, H2, nullptr, nullptr, q.lin1, N, y.dA1, H2, 0));
      ESC_TRY(bn_backward(c, w.Y0, H2, nullptr, 0, y.dA1, H2, N, w.b0, q.bn0, y.dA1, H2, y.bn_scratch, H2));      // (ReLU mask from the pre-BatchNorm rows: A1 is not re-read)
    }
    if (!hidden_fused) ESC_TRY(linear_backward(c, y.dA1, H2, w.agg, H, nullptr, nullptr, q.lin0, N, y.dagg, H, 0));
    // virtual-node update of this layer: vn_{l+1} = dropout(mlp(add_pool(hin) + vn_l)) (+ vn_l)
    bool have_dhin = false;
    if (!last) {
      ESC_TRY(bn_backward_drop(c, w.V1, H, dvn_next, H, G, w.vb1, q.vbn1, w.mask_v, p, 0, y.dG2, H, y.bn_scratch));
      ESC_TRY(linear_backward(c, y.dG2, H, w.VA, H2, nullptr, nullptr, q.vlin1, G, y.dG1, H2, 0));
      ESC_TRY(bn_backward(c, w.V0, H2, nullptr, 0, y.dG1, H2, G, w.vb0, q.vbn0, y.dG1, H2, y.bn_scratch, H2));
      ESC_TRY(linear_backward(c, y.dG1, H2, w.tmp, H, nullptr, nullptr, q.vlin0, G, y.dtmp, H, 0));
      // d vn_l = d tmp (+ d vn_{l+1} through the residual); d hin = broadcast(d tmp) (+ d h_{l+1} through the residual)
      ESC_TRY(esc_dropout_bwd(y.dtmp, H, G, H, 0.f, nullptr, m->residual ? dvn_next : nullptr, H, dvn_cur, H, c.s));
      ESC_TRY(esc_segment_broadcast_add(m->residual ? dH : nullptr, H, y.dtmp, H, b->graph_ptr, G, N, H, dHin, H, c.s));
      have_dhin = true;
    } else if (m->residual) {
      float* t = dH; dH = dHin; dHin = t;          // d hin starts as d h_{l+1}: accumulate into that buffer
      have_dhin = true;
    }
    ESC_TRY(esc_gine_aggregate_bwd(w.hin, H, w.e, l == 0 ? H : y.ld_e, y.dagg, H, b->out_ptr, b->out_edge, b->out_dst, q.eps, N, H, w.d_e, H,
                                   dHin, H, have_dhin ? 1 : 0, y.deps_part + (int64_t)l * 2 * N, c.s));
    eps_jobs.push_back(esc_sum_job{y.deps_part + (int64_t)l * 2 * N, N * esc_gine_aggregate_bwd_deps_slots(H), q.deps});
    // edge term: bond tables and edge_encoder_pos — edge stream
    if (es.ok) ESC_TRY(chain(es.de_ready[l], (hipStream_t)c.s, es.stream));
    // bond tables: on the edge stream — except for the LAST layer processed (l == 0), whose edge-term backward opens the tail of
    // the step: there the table gradient runs on the node stream (d_e is its own product; it has the atom tables' scratch to
    // itself until the encoders below) and the edge stream goes straight to the Linear that completes d(z_emb)
    const bool bonds_on_node = l == 0 && es.ok && g_ogb_bonds_on_node;
    ESC_TRY(esc_bag_bwd_table(w.d_e, H, H, b->bonds.col_ptr, b->bonds.c_row, b->bonds.ones, b->bonds.c_col, b->bonds.n_entries,
                              m->bond_rows, y.dTcat + q.bond_row0 * H, bonds_on_node ? y.emb_scratch_n : y.emb_scratch,
                              bonds_on_node ? c.s : ce.s));
    if (l == 0 && es.ok && c.jobs && g_ogb_split_tail) {
      // the LAST edge-term backward starts the tail of the step: only its input gradient (which completes d(z_emb)) stays on
      // the edge stream; the weight gradient runs on the node stream, which has little left to do (d_e is its own product)
      {
        const LdsFloorGuard cap(true);
        ESC_TRY(esc_linear_bwd_input(w.d_e, H, q.pos.w, H, E, H, H, y.dZemb, H, last ? 0 : 1, es.stream));
      }
      float* slabs = *c.slab_cursor;
      *c.slab_cursor += (esc_linear_bwd_weight_scratch(E, H, H) + 63) & ~63LL;
      c.jobs->emplace_back();
      ESC_TRY(esc_linear_bwd_both_deferred(w.d_e, H, y.Zemb, H, nullptr, nullptr, q.pos.w, H, E, H, H, nullptr, 0, 0, q.pos.dw, H, q.pos.db,
                                           slabs, &c.jobs->back(), c.s));
    } else {
      ESC_TRY(linear_backward(ce, w.d_e, H, y.Zemb, H, nullptr, nullptr, q.pos, E, y.dZemb, H, last ? 0 : 1));
    }
    // d hin_l is complete: d h_l = d hin_l, d vn_l += add_pool(d hin_l)
    ESC_TRY(esc_segment_pool_fwd(dHin, H, b->graph_ptr, G, H, 0, y.poolG, H, c.s));
    ESC_TRY(esc_dropout_bwd(y.poolG, H, G, H, 0.f, nullptr, last ? nullptr : dvn_cur, H, dvn_cur, H, c.s));
    dvn_next = dvn_cur;
    dvn_cur = dvn_cur == y.dvn_a ? y.dvn_b : y.dvn_a;
    { float* t = dH; dH = dHin; dHin = t; }       // dH = d h_l
  }
  // encoders
  ESC_TRY(esc_bag_bwd_table(dH, H, H, b->atoms.col_ptr, b->atoms.c_row, b->atoms.ones, b->atoms.c_col, b->atoms.n_entries,
                            m->atom_rows, y.dTcat, y.emb_scratch_n, c.s));
  ESC_TRY(esc_embed_bwd(dvn_next, H, b->zero_idx, G, 1, H, m->vn_dw, c.s));
  // z_embedding + bag: the tail of the edge pipeline (d(z_emb) is complete in edge-stream order)
  ESC_TRY(bn_backward_drop(ce, y.Yzd, H, y.dZemb, H, E, y.zb1, m->zbn1, y.mask_z1, p, 1, y.dYz, H, ce.y.bn_scratch));
  if (es.ok && c.jobs && g_ogb_split_tail) {      // same split for z_embedding's Linear: dX continues the tail, dW on the node stream
    ESC_TRY(chain(es.tail_dz, es.stream, (hipStream_t)c.s));           // d(Yz) is complete here in edge-stream order
    {
      const LdsFloorGuard cap(true);
      ESC_TRY(esc_linear_bwd_input(y.dYz, H, m->zlin.w, H, E, H, H, y.dA0, H, 0, es.stream));
    }
    float* slabs = *c.slab_cursor;
    *c.slab_cursor += (esc_linear_bwd_weight_scratch(E, H, H) + 63) & ~63LL;
    c.jobs->emplace_back();
    ESC_TRY(esc_linear_bwd_both_deferred(y.dYz, H, y.A0, H, nullptr, nullptr, m->zlin.w, H, E, H, H, nullptr, 0, 0, m->zlin.dw, H, m->zlin.db,
                                         slabs, &c.jobs->back(), c.s));
  } else {
    ESC_TRY(linear_backward(ce, y.dYz, H, y.A0, H, nullptr, nullptr, m->zlin, E, y.dA0, H, 0));
  }
  ESC_TRY(bn_backward_drop(ce, y.Zd, H, y.dA0, H, E, y.zb0, m->zbn0, y.mask_z0, p, 1, y.dA0, H, ce.y.bn_scratch));
  ESC_TRY(esc_bag_bwd_table_rows(y.dA0, H, H, b->col_ptr, b->col_row, b->col_val, b->col_col, y.Z, m->z_rows, E, 1,
                                 m->dz_table, y.bag_scratch, ce.s));
  if (es.ok && !edge_jobs.empty()) ESC_TRY(esc_slab_reduce_jobs(edge_jobs.data(), (int)edge_jobs.size(), es.stream));
  // node-side reductions overlap the edge tail; then join and hand the packed table gradients back to the tables
  if (!eps_jobs.empty()) ESC_TRY(esc_reduce_sum_jobs(eps_jobs.data(), (int)eps_jobs.size(), c.s));
  if (c.jobs && !c.jobs->empty()) ESC_TRY(esc_slab_reduce_jobs(c.jobs->data(), (int)c.jobs->size(), c.s));
  if (es.ok) ESC_TRY(chain(es.joined, es.stream, (hipStream_t)c.s));
  return esc_table_unpack_grad(&m->tables, H, y.dTcat, c.s);
}

static int check_ogb(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, const float* ws, bool train, bool need_y) {
  ESC_REQUIRE(m && b && ws, "esc_ogb: null pointer");
  ESC_REQUIRE(m->num_layers >= 1 && m->num_layers <= ESC_MAX_LAYERS, "esc_ogb: %ld layers unsupported", (long)m->num_layers);
  ESC_REQUIRE(m->hidden > 0 && m->hidden % 4 == 0 && m->num_tasks >= 1, "esc_ogb: hidden must be a multiple of 4");
  ESC_REQUIRE(m->drop_ratio >= 0.f && m->drop_ratio < 1.f, "esc_ogb: drop_ratio %g outside [0,1)", (double)m->drop_ratio);
  ESC_REQUIRE(m->tables.count > 0 && m->tables.count <= ESC_MAX_TABLES && m->atom_rows > 0 && m->bond_rows > 0 && m->vn_w,
              "esc_ogb: encoder tables missing");
  ESC_REQUIRE(b->N >= 2 && b->E >= 2 && b->Z >= 0 && b->G >= 2, "esc_ogb: batch needs >= 2 nodes, edges and graphs (BatchNorm statistics)");
  ESC_REQUIRE(b->graph_ptr && b->zero_idx && b->in_ptr && b->row_ptr && b->atoms.row_ptr && b->atoms.idx && b->atoms.ones &&
              b->bonds.row_ptr && b->bonds.idx && b->bonds.ones, "esc_ogb: null batch arrays");
  ESC_REQUIRE(!train || ((b->y || !need_y) && b->out_ptr && b->col_ptr && b->atoms.col_ptr && b->atoms.c_row && b->atoms.c_col &&
                         b->bonds.col_ptr && b->bonds.c_row && b->bonds.c_col && m->vn_dw), "esc_ogb: null training arrays");
  ESC_REQUIRE(aligned16(ws), "esc_ogb: workspace must be 16-byte aligned");
  return ESC_OK;
}

static OgbCtx make_ogb(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* ws, void* stream, bool train) {
  OgbCtx z{m, b, plan_layout_ogb(m, b->N, b->E, b->Z, b->G, b->atoms.n_entries, b->bonds.n_entries, ws, train),
           Ctx{nullptr, nullptr, Layout{}, stream, train}, train ? m->drop_ratio : 0.f};
  z.c.y.H = m->hidden; z.c.y.N = b->N; z.c.y.E = b->E;
  z.c.y.col_stats = z.o.col_stats; z.c.y.bn_scratch = z.o.bn_scratch; z.c.y.slabs = z.o.slabs;
  z.c.y.col_stats_e = z.o.col_stats_e; z.c.y.bn_scratch_e = z.o.bn_scratch_e;
  z.c.act = 1;
  return z;
}

static int check(const esc_nested_gin_t* m, const esc_batch_t* b, const float* ws, bool train, bool need_y = true) {
  ESC_REQUIRE(m && b && ws, "esc_engine: null pointer");
  ESC_REQUIRE(m->num_layers >= 1 && m->num_layers <= ESC_MAX_LAYERS, "esc_engine: %ld layers unsupported", (long)m->num_layers);
  ESC_REQUIRE(m->hidden > 0 && m->hidden % 4 == 0 && m->in_dim > 0, "esc_engine: hidden must be a multiple of 4");
  ESC_REQUIRE(b->N >= 2 && b->E >= 2 && b->Z >= 0, "esc_engine: batch needs >= 2 nodes and edges (BatchNorm statistics)");
  ESC_REQUIRE(b->x && b->in_ptr && b->row_ptr && (!train || ((b->y || !need_y) && b->out_ptr && b->col_ptr)), "esc_engine: null batch arrays");
  ESC_REQUIRE(aligned16(ws), "esc_engine: workspace must be 16-byte aligned");
  return ESC_OK;
}

}  // namespace esc

using namespace esc;

extern "C" {

int esc_engine_set_two_stream_min_edges(int64_t edges) {
  g_two_stream_min_edges = edges < 0 ? 0 : edges;
  return ESC_OK;
}

int esc_engine_set_side_stream(int on) {
  g_use_edge_stream = (on & 2) != 0;
  g_cap_forward = (on & 8) == 0;
  g_cap_tail = (on & 16) != 0;
  g_edge_ahead = (on & 32) ? 2 : 1;
  g_edge_priority_low = (on & 4) == 0;      // bit 2: give the edge stream the HIGHEST priority instead (experiments)
  g_use_side_stream = (on & 1) != 0;
  return ESC_OK;
}

/* diagnostics: mean milliseconds between the phase marks of the last recorded steps (ESC_PHASE_TIMING=1); call after a
 * device synchronise.  out[0..5]: start->edge-forward done, start->node-forward done, node-forward done->node-backward done,
 * node-backward done->edge-backward done (negative: the edge pipeline finished first), start->end of step, end->next start */
int esc_engine_phase_times(double* out, int skip) {
  PhaseRing& p = phases();
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
  if (!p.on || p.steps < 2) return 0;
  const int first = p.steps > PhaseRing::RING ? p.steps - PhaseRing::RING + 1 : 0;
  int n = 0;
  for (int sidx = first + skip; sidx < p.steps; ++sidx) {
    hipEvent_t* e = p.ev[sidx % PhaseRing::RING];
    bool ok = true;
    for (int k = 0; k < PH_COUNT; ++k) ok = ok && e[k] != nullptr;
    if (!ok) continue;
    float t[6] = {};
    if (hipEventElapsedTime(&t[0], e[PH_START], e[PH_EDGE_FWD_DONE]) != hipSuccess) continue;
    (void)hipEventElapsedTime(&t[1], e[PH_START], e[PH_NODE_FWD_DONE]);
    (void)hipEventElapsedTime(&t[2], e[PH_NODE_FWD_DONE], e[PH_NODE_BWD_DONE]);
    if (hipEventElapsedTime(&t[3], e[PH_NODE_BWD_DONE], e[PH_EDGE_BWD_DONE]) != hipSuccess) {
      (void)hipEventElapsedTime(&t[3], e[PH_EDGE_BWD_DONE], e[PH_NODE_BWD_DONE]); t[3] = -t[3];
    }
    (void)hipEventElapsedTime(&t[4], e[PH_START], e[PH_END]);
    if (sidx + 1 < p.steps) (void)hipEventElapsedTime(&t[5], e[PH_END], p.ev[(sidx + 1) % PhaseRing::RING][PH_START]);
    for (int i = 0; i < 6; ++i) out[i] += t[i];
    ++n;
  }
  for (int i = 0; i < 6; ++i) out[i] /= n > 0 ? n : 1;
  return n;
}

int esc_engine_set_collective(esc_allreduce_fn fn, void* user, int rank, int world, float* buf_node, float* buf_edge,
                              int64_t cap) {
  ESC_REQUIRE(fn == nullptr || (world >= 1 && rank >= 0 && rank < world && buf_node && buf_edge && cap > 0),
              "esc_engine_set_collective: bad argument");
  g_coll = Collective{fn, user, rank, world < 1 ? 1 : world, buf_node, buf_edge, cap};
  return ESC_OK;
}

int esc_engine_set_gemm_stats(int on) {
  g_fuse_finalize = (on & 2) == 0;       // bit 1: keep the statistics epilogue but finalize in a separate launch
  g_fuse_node_act = (on & 16) == 0;      // bit 4: materialise the node MLPs' output activations again (one more launch per layer)
  g_fold = (on & 8) ? 2 : ((on & 4) != 0);   // bit 2: node-sized BatchNorms are merged by their consumers; bit 3: only an MLP's last one
  g_gemm_stats = (on & 1) != 0;
  return ESC_OK;
}

int esc_engine_set_materialise_edge_act(int on) {
  g_materialise_edge_act = on != 0;
  return ESC_OK;
}

int64_t esc_engine_workspace_floats(const esc_nested_gin_t* m, int64_t N, int64_t E, int64_t Z) {
  if (!m || N < 0 || E < 0 || Z < 0) return -1;
  return plan_layout(m, N, E, Z, nullptr, true).total;
}

static int train_step_impl(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                           int64_t loss_denom, float* loss, float* pred, void* stream, Pending* defer);

int esc_engine_train_step(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                          int64_t loss_denom, float* loss, float* pred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  return train_step_impl(m, b, workspace, loss_denom, loss, pred, stream, nullptr);
}

int esc_engine_train_step_begin(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                                int64_t loss_denom, float* loss, float* pred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  return train_step_impl(m, b, workspace, loss_denom, loss, pred, stream, &pending());
}

int esc_engine_train_step_end(void) { return finish_pending(pending()); }

static int train_step_impl(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                           int64_t loss_denom, float* loss, float* pred, void* stream, Pending* defer) {
  int rc = check(m, b, workspace, true);
  if (rc) return rc;
  ESC_REQUIRE(loss, "esc_engine_train_step: null loss pointer");
  Ctx c{m, b, plan_layout(m, b->N, b->E, b->Z, workspace, true), stream, true};
  std::vector<esc_reduce_job> jobs;
  jobs.reserve(ESC_MAX_REDUCE_JOBS);
  float* cursor = c.y.slabs;
  if (3 * m->num_layers + 7 <= ESC_MAX_REDUCE_JOBS) { c.jobs = &jobs; c.slab_cursor = &cursor; }
  const int64_t denom = loss_denom > 0 ? loss_denom : b->N;
  EdgeStream& es = edge_stream_for(b->E);
  const bool head_l1 = g_l1_head && es.ok && m->hidden <= 1024 &&
                       esc_linear_fwd_l1_ok(c.y.Yl, m->hidden, m->lin2.w, m->hidden, c.y.bl.scale, c.y.bl.shift) != 0;
  if (head_l1) { c.l1_target = b->y; c.l1_denom = denom; }
  ESC_TRY(forward(c));
  if (head_l1) {              // the loss VALUE: on the edge stream (idle here), ordered behind the head; the step's join covers it
    ESC_TRY(chain(es.lin1_fork, (hipStream_t)stream, es.stream));
    ESC_TRY(esc_l1_loss(c.y.pred, b->y, b->N, denom, 1.0f, loss, nullptr, es.stream));
  } else {
    ESC_TRY(esc_l1_loss(c.y.pred, b->y, b->N, denom, 1.0f, loss, c.y.dpred, stream));
  }
  mark(PH_NODE_FWD_DONE, stream);
  if (pred) {
    if (hipMemcpyAsync(pred, c.y.pred, sizeof(float) * (size_t)b->N, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      set_error("esc_engine_train_step: prediction copy failed");
      return ESC_ELAUNCH;
    }
  }
  return backward(c, defer);
}

// The step as an autograd node: forward in training mode (batch statistics, activations kept in `workspace`), then —
// with the gradient of ANY loss w.r.t. the predictions — the backward.  The workspace must be left untouched in
// between.  This is what lets `NestedGIN_eff.forward` itself run on the engine inside a user's own training loop.
int esc_engine_forward_train(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, float* pred,
                             void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(pred, "esc_engine_forward_train: null output");
  Ctx c{m, b, plan_layout(m, b->N, b->E, b->Z, workspace, true), stream, true};
  std::vector<esc_reduce_job> jobs;            // only marks the main chain (statistics from the GEMM epilogues)
  float* cursor = c.y.slabs;
  c.jobs = &jobs; c.slab_cursor = &cursor;
  ESC_TRY(forward(c));
  if (hipMemcpyAsync(pred, c.y.pred, sizeof(float) * (size_t)b->N, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_error("esc_engine_forward_train: prediction copy failed");
    return ESC_ELAUNCH;
  }
  return ESC_OK;
}

int esc_engine_backward(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, const float* dpred,
                        void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(dpred, "esc_engine_backward: null gradient");
  Ctx c{m, b, plan_layout(m, b->N, b->E, b->Z, workspace, true), stream, true};
  std::vector<esc_reduce_job> jobs;
  jobs.reserve(ESC_MAX_REDUCE_JOBS);
  float* cursor = c.y.slabs;
  if (3 * m->num_layers + 7 <= ESC_MAX_REDUCE_JOBS) { c.jobs = &jobs; c.slab_cursor = &cursor; }
  if (hipMemcpyAsync(c.y.dpred, dpred, sizeof(float) * (size_t)b->N, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_error("esc_engine_backward: gradient copy failed");
    return ESC_ELAUNCH;
  }
  return backward(c, nullptr);
}

int esc_engine_predict(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, float* pred,
                       void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check(m, b, workspace, false);
  if (rc) return rc;
  ESC_REQUIRE(pred, "esc_engine_predict: null output");
  Ctx c{m, b, plan_layout(m, b->N, b->E, b->Z, workspace, false), stream, false};
  ESC_TRY(forward(c));
  if (hipMemcpyAsync(pred, c.y.pred, sizeof(float) * (size_t)b->N, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_error("esc_engine_predict: prediction copy failed");
    return ESC_ELAUNCH;
  }
  return ESC_OK;
}

// ---- ZINC variant ------------------------------------------------------------------------------------------------------
int64_t esc_zinc_workspace_floats(const esc_zinc_gin_t* m, int64_t N, int64_t E, int64_t Z, int64_t G) {
  if (!m || N < 0 || E < 0 || Z < 0 || G < 0) return -1;
  return plan_layout_zinc(m, N, E, Z, G, nullptr, true).total + 64;
}

static int copy_floats(float* dst, const float* src, int64_t n, void* stream, const char* what) {
  if (hipMemcpyAsync(dst, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_error("%s: copy failed", what);
    return ESC_ELAUNCH;
  }
  return ESC_OK;
}

int esc_zinc_train_step(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, int64_t loss_denom,
                        float* loss, float* pred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_zinc(m, b, workspace, true, true);
  if (rc) return rc;
  ESC_REQUIRE(loss, "esc_zinc_train_step: null loss pointer");
  ZincCtx z = make_zinc(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;
  jobs.reserve(ESC_MAX_REDUCE_JOBS);
  float* cursor = z.c.y.slabs;
  if (3 * m->num_layers + 3 <= ESC_MAX_REDUCE_JOBS) { z.c.jobs = &jobs; z.c.slab_cursor = &cursor; }
  ESC_TRY(forward_zinc(z));
  ESC_TRY(esc_l1_loss(z.c.y.pred, b->y, b->G, loss_denom > 0 ? loss_denom : b->G, 1.0f, loss, z.c.y.dpred, stream));
  if (pred) ESC_TRY(copy_floats(pred, z.c.y.pred, b->G, stream, "esc_zinc_train_step"));
  return backward_zinc(z);
}

int esc_zinc_forward_train(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, float* pred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_zinc(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(pred, "esc_zinc_forward_train: null output");
  ZincCtx z = make_zinc(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;            // only marks the main chain (statistics from the GEMM epilogues)
  float* cursor = z.c.y.slabs;
  z.c.jobs = &jobs; z.c.slab_cursor = &cursor;
  ESC_TRY(forward_zinc(z));
  return copy_floats(pred, z.c.y.pred, b->G, stream, "esc_zinc_forward_train");
}

int esc_zinc_backward(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, const float* dpred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_zinc(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(dpred, "esc_zinc_backward: null gradient");
  ZincCtx z = make_zinc(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;
  jobs.reserve(ESC_MAX_REDUCE_JOBS);
  float* cursor = z.c.y.slabs;
  if (3 * m->num_layers + 3 <= ESC_MAX_REDUCE_JOBS) { z.c.jobs = &jobs; z.c.slab_cursor = &cursor; }
  ESC_TRY(copy_floats(z.c.y.dpred, dpred, b->G, stream, "esc_zinc_backward"));
  return backward_zinc(z);
}

int esc_zinc_predict(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, float* pred, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_zinc(m, b, workspace, false, false);
  if (rc) return rc;
  ESC_REQUIRE(pred, "esc_zinc_predict: null output");
  ZincCtx z = make_zinc(m, b, workspace, stream, false);
  ESC_TRY(forward_zinc(z));
  return copy_floats(pred, z.c.y.pred, b->G, stream, "esc_zinc_predict");
}

// ---- OGB molecule variant ------------------------------------------------------------------------------------------------
int64_t esc_ogb_workspace_floats(const esc_ogb_gnn_t* m, int64_t N, int64_t E, int64_t Z, int64_t G, int64_t atom_entries,
                                 int64_t bond_entries) {
  if (!m || N < 0 || E < 0 || Z < 0 || G < 0 || atom_entries < 0 || bond_entries < 0) return -1;
  return plan_layout_ogb(m, N, E, Z, G, atom_entries, bond_entries, nullptr, true).total + 64;
}

static void ogb_jobs(OgbCtx& z, std::vector<esc_reduce_job>& jobs, float*& cursor) {
  jobs.reserve(ESC_MAX_REDUCE_JOBS);
  cursor = z.o.slabs;
  if (5 * z.m->num_layers + 2 <= ESC_MAX_REDUCE_JOBS) { z.c.jobs = &jobs; z.c.slab_cursor = &cursor; }
}

int esc_ogb_train_step(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, int64_t loss_denom, float* loss,
                       float* logits, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_ogb(m, b, workspace, true, true);
  if (rc) return rc;
  ESC_REQUIRE(loss, "esc_ogb_train_step: null loss pointer");
  OgbCtx z = make_ogb(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;
  float* cursor = nullptr;
  ogb_jobs(z, jobs, cursor);
  ESC_TRY(forward_ogb(z));
  ESC_TRY(esc_bce_logits_loss(z.o.logits, b->y, b->G * m->num_tasks, loss_denom, loss, z.o.dlogits, stream));
  if (logits) ESC_TRY(copy_floats(logits, z.o.logits, b->G * m->num_tasks, stream, "esc_ogb_train_step"));
  return backward_ogb(z);
}

int esc_ogb_forward_train(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, float* logits, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_ogb(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(logits, "esc_ogb_forward_train: null output");
  OgbCtx z = make_ogb(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;
  float* cursor = z.o.slabs;
  z.c.jobs = &jobs; z.c.slab_cursor = &cursor;
  ESC_TRY(forward_ogb(z));
  return copy_floats(logits, z.o.logits, b->G * m->num_tasks, stream, "esc_ogb_forward_train");
}

int esc_ogb_backward(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, const float* dlogits, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_ogb(m, b, workspace, true, false);
  if (rc) return rc;
  ESC_REQUIRE(dlogits, "esc_ogb_backward: null gradient");
  OgbCtx z = make_ogb(m, b, workspace, stream, true);
  std::vector<esc_reduce_job> jobs;
  float* cursor = nullptr;
  ogb_jobs(z, jobs, cursor);
  ESC_TRY(copy_floats(z.o.dlogits, dlogits, b->G * m->num_tasks, stream, "esc_ogb_backward"));
  return backward_ogb(z);
}

int esc_ogb_predict(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, float* logits, void* stream) {
  ESC_TRY(finish_pending(pending()));
  int rc = check_ogb(m, b, workspace, false, false);
  if (rc) return rc;
  ESC_REQUIRE(logits, "esc_ogb_predict: null output");
  OgbCtx z = make_ogb(m, b, workspace, stream, false);
  ESC_TRY(forward_ogb(z));
  return copy_floats(logits, z.o.logits, b->G * m->num_tasks, stream, "esc_ogb_predict");
}

}  // extern "C"
