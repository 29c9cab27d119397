// runtime.hip — error reporting + HIP-event kernel-family profiler of libescgnn_hip.so
#include "common.h"
#include <algorithm>
#include <cstdlib>

#include <mutex>
#include <vector>

namespace esc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct ProfState {
  bool on = false;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
static ProfState g_prof[ESC_K_COUNT];
static std::mutex g_prof_mu;

bool prof_slot(int k, hipEvent_t* start, hipEvent_t* stop) {
  if (k == ESC_K_GEMM_EDGE && !g_prof[k].on) k = ESC_K_LINEAR;       // the edge-row tiles are Linear launches too
  if (k < 0 || k >= ESC_K_COUNT || !g_prof[k].on) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfState& p = g_prof[k];
  if (p.used == p.start.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    p.start.push_back(a);
    p.stop.push_back(b);
  }
  *start = p.start[p.used];
  *stop = p.stop[p.used];
  ++p.used;
  return true;
}

// in-kernel execution windows of a profiled kernel family (esc_prof_span_*): per launch ESC_SPAN_WGS start slots (one per workgroup)
// and ESC_SPAN_WAVES end slots (one per wave), written by kernels that take a `span` argument
struct SpanState { unsigned long long* dev = nullptr; size_t cap = 0; };
static SpanState g_span[ESC_K_COUNT];
unsigned long long* prof_span_next(int k) {
  if (k < 0 || k >= ESC_K_COUNT) return nullptr;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof[k].on || g_span[k].dev == nullptr || g_prof[k].used >= g_span[k].cap) return nullptr;
  return g_span[k].dev + g_prof[k].used * ESC_SPAN_STRIDE;
}
__global__ void span_init_kernel(unsigned long long* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (i % ESC_SPAN_STRIDE) < ESC_SPAN_WGS ? ~0ULL : 0ULL;
}

// ticket counters for grid_last_block (common.h): one zeroed pool per device, handed out as rotating windows
constexpr int TICKET_POOL = 4096;
static unsigned* g_tickets[64] = {};
static int g_ticket_pos[64] = {};
static std::mutex g_ticket_mu;

static int g_last_block = 0;
bool last_block_finalize() { return g_last_block != 0; }
void set_last_block_finalize(int on) { g_last_block = on != 0; }

static thread_local int g_gemm_lds_floor = 0;
int gemm_lds_floor() { return g_gemm_lds_floor; }
void set_gemm_lds_floor(int bytes) { g_gemm_lds_floor = bytes < 0 ? 0 : (bytes > 160 * 1024 ? 160 * 1024 : bytes); }
static int g_edge_lds_floor = 52 * 1024;      // 3 edge-GEMM workgroups per CU: a wave slot per SIMD stays free for the node stream
int edge_lds_floor() { return g_edge_lds_floor; }
static thread_local int g_node_lds_floor = 0;
int node_lds_floor() { return g_node_lds_floor; }
void set_node_lds_floor(int bytes) { g_node_lds_floor = bytes < 0 ? 0 : (bytes > 160 * 1024 ? 160 * 1024 : bytes); }
void set_edge_lds_floor(int bytes) { g_edge_lds_floor = bytes < 0 ? 0 : bytes; }

static const bool g_trace_launch = getenv("ESC_TRACE_LAUNCH") != nullptr && atoi(getenv("ESC_TRACE_LAUNCH")) != 0;
bool trace_launch() { return g_trace_launch; }
static int g_bn_bwd_one = getenv("ESC_BN_BWD_ONE") ? atoi(getenv("ESC_BN_BWD_ONE")) : 0;
bool bn_bwd_one_launch() { return g_bn_bwd_one != 0; }
void set_bn_bwd_one_launch(int on) { g_bn_bwd_one = on != 0; }
static int g_bn_bwd_fold = 0;      // measured on MI355X r02: 1.298 ms with it vs 1.215 ms without (the 32-block partial pass is slower than what the finalize launch costs)
bool bn_bwd_fold() { return g_bn_bwd_fold != 0; }
void set_bn_bwd_fold(int on) { g_bn_bwd_fold = on != 0; }
static int g_norm_rowblock_cap = 256;
int norm_rowblock_cap() { return g_norm_rowblock_cap; }
void set_norm_rowblock_cap(int v) { g_norm_rowblock_cap = v < 1 ? 1 : (v > 512 ? 512 : v); }

unsigned* tickets(int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || n <= 0 || n > TICKET_POOL) return nullptr;
  std::lock_guard<std::mutex> lk(g_ticket_mu);
  if (g_tickets[dev] == nullptr) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned) * TICKET_POOL) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(unsigned) * TICKET_POOL) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_tickets[dev] = p;
  }
  if (g_ticket_pos[dev] + n > TICKET_POOL) g_ticket_pos[dev] = 0;
  unsigned* out = g_tickets[dev] + g_ticket_pos[dev];
  g_ticket_pos[dev] += n;
  return out;
}

}  // namespace esc

extern "C" {

int esc_abi_version(void) { return 3; }   // 2: esc_features_* take sum_nodes_sq; esc_zinc_*, esc_embed_*; 3: esc_collate_args grew (edge_attr, x_long, graph_ptr), esc_embed_plan
const char* esc_last_error(void) { return esc::g_err; }

int esc_prof_enable(int kind, int on) {
  ESC_REQUIRE(kind >= 0 && kind < ESC_K_COUNT, "esc_prof_enable: bad kind %d", kind);
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::g_prof[kind].on = on != 0;
  return ESC_OK;
}

int esc_prof_reset(int kind) {
  ESC_REQUIRE(kind >= 0 && kind < ESC_K_COUNT, "esc_prof_reset: bad kind %d", kind);
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::g_prof[kind].used = 0;
  return ESC_OK;
}

int esc_prof_read(int kind, int64_t* launches, double* total_ms) {
  ESC_REQUIRE(kind >= 0 && kind < ESC_K_COUNT && launches && total_ms, "esc_prof_read: bad argument");
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::ProfState& p = esc::g_prof[kind];
  double tot = 0;
  for (size_t i = 0; i < p.used; ++i) {
    if (hipEventSynchronize(p.stop[i]) != hipSuccess) {
      esc::set_error("esc_prof_read: event sync failed");
      return ESC_ELAUNCH;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
    tot += ms;
  }
  *launches = (int64_t)p.used;
  *total_ms = tot;
  return ESC_OK;
}

int esc_prof_span_arm(int kind, int64_t launches, void* stream) {
  ESC_REQUIRE(kind >= 0 && kind < ESC_K_COUNT && launches > 0 && launches <= 4096, "esc_prof_span_arm: 1..4096 launches");
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::SpanState& sp = esc::g_span[kind];
  if (sp.cap < (size_t)launches) {
    if (sp.dev) (void)hipFree(sp.dev);
    sp.dev = nullptr; sp.cap = 0;
    if (hipMalloc(&sp.dev, (size_t)launches * ESC_SPAN_STRIDE * sizeof(unsigned long long)) != hipSuccess) { esc::set_error("esc_prof_span_arm: allocation failed"); return ESC_ELAUNCH; }
    sp.cap = (size_t)launches;
  }
  const size_t n = sp.cap * ESC_SPAN_STRIDE;
  hipLaunchKernelGGL(esc::span_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sp.dev, n);
  return ESC_OK;
}

/* execution windows (us) of the first launches of `kind` recorded since esc_prof_reset; the caller has synchronised */
int64_t esc_prof_span_read(int kind, double* us_out, int64_t cap) {
  if (kind < 0 || kind >= ESC_K_COUNT || us_out == nullptr || cap <= 0) return 0;
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::SpanState& sp = esc::g_span[kind];
  const size_t n = std::min<size_t>(std::min<size_t>(esc::g_prof[kind].used, sp.cap), (size_t)cap);
  if (n == 0 || sp.dev == nullptr) return 0;
  std::vector<unsigned long long> h(n * ESC_SPAN_STRIDE);
  if (hipMemcpy(h.data(), sp.dev, n * ESC_SPAN_STRIDE * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  int dev = 0, khz = 100000;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  for (size_t i = 0; i < n; ++i) {
    unsigned long long lo = ~0ULL, hi = 0ULL;
    const unsigned long long* q = h.data() + i * ESC_SPAN_STRIDE;
    for (int w = 0; w < ESC_SPAN_WGS; ++w) lo = std::min(lo, q[w]);
    for (int w = 0; w < ESC_SPAN_WAVES; ++w) hi = std::max(hi, q[ESC_SPAN_WGS + w]);
    us_out[i] = (hi > lo && lo != ~0ULL) ? (double)(hi - lo) * 1e3 / (double)khz : 0.0;
  }
  return (int64_t)n;
}

/* per-launch durations (ms) of the recorded launches of `kind`, in launch order; returns how many were written */
int64_t esc_prof_read_all(int kind, double* ms_out, int64_t cap) {
  if (kind < 0 || kind >= ESC_K_COUNT || ms_out == nullptr || cap <= 0) return 0;
  std::lock_guard<std::mutex> lk(esc::g_prof_mu);
  esc::ProfState& p = esc::g_prof[kind];
  int64_t n = 0;
  for (size_t i = 0; i < p.used && n < cap; ++i) {
    if (hipEventSynchronize(p.stop[i]) != hipSuccess) break;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
    ms_out[n++] = ms;
  }
  return n;
}

}  // extern "C"
