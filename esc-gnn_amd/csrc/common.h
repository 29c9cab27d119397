// common.h — shared helpers for the gfx950 kernels of libescgnn_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/escgnn_hip.h"

namespace esc {

constexpr int WAVE = 64;  // CDNA4 wavefront

// Instruction-issue priority of every kernel except the edge-sized GEMM tiles.  The step engine runs two pipelines on
// two streams; the edge pipeline's 128x128 GEMM workgroups hold every CU for 20-50 us, and a co-resident kernel of the
// latency-critical node chain loses the (age-ordered) issue arbitration to them: measured 25 us for an 8 us node GEMM.
// Raising the priority of the short kernels lets them cut through; the big tiles soak up what is left.
#ifndef ESC_NODE_PRIO
#define ESC_NODE_PRIO 2
#endif
#define ESC_PRIO() __builtin_amdgcn_s_setprio(ESC_NODE_PRIO)

void set_error(const char* fmt, ...);

#define ESC_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      esc::set_error(__VA_ARGS__);                \
      return ESC_EINVAL;                          \
    }                                             \
  } while (0)

#define ESC_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t err__ = hipGetLastError();                                   \
    if (err__ != hipSuccess) {                                              \
      esc::set_error("%s: launch failed: %s", name, hipGetErrorString(err__)); \
      return ESC_ELAUNCH;                                                   \
    }                                                                       \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// per-launch kernel timing (include/escgnn_hip.h "profiling hook").  When a kernel family is being
// profiled its launches go through hipExtLaunchKernelGGL with a start/stop event pair, which stamps
// the dispatch packet itself — the same begin/end a rocprofv3 kernel trace reports — instead of
// bracketing the launch with stream events (that adds ~3 us of queue latency to a 7 us kernel).
bool prof_slot(int kind, hipEvent_t* start, hipEvent_t* stop);
// the `span` argument of the NEXT profiled launch of `kind` (esc_prof_span_arm), or NULL
unsigned long long* prof_span_next(int kind);
#define ESC_SPAN_WGS 2048          /* per-launch slots: one start per workgroup ... */
#define ESC_SPAN_WAVES 8192        /* ... one end per wave */
#define ESC_SPAN_STRIDE (ESC_SPAN_WGS + ESC_SPAN_WAVES)
// ESC_TRACE_LAUNCH=1 (debugging a fault or a hang): every launch is announced on stderr and waited for
bool trace_launch();

template <typename... KArgs, typename... Args>
inline void launch(int kind, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s,
                   Args... args) {
  hipEvent_t a = nullptr, b = nullptr;
  if (prof_slot(kind, &a, &b))
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, a, b, 0, static_cast<KArgs>(args)...);
  else
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, static_cast<KArgs>(args)...);
  if (trace_launch()) {
    const char* name = hipKernelNameRefByPtr(reinterpret_cast<const void*>(kernel), s);
    fprintf(stderr, "[esc] %s grid (%u,%u,%u) block %u lds %zu stream %p ...", name ? name : "?", grid.x, grid.y, grid.z, block.x, lds, (void*)s);
    fflush(stderr);
    const hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
  }
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-uniform value into an SGPR so that dependent loads become scalar loads
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- "last workgroup finishes the reduction" ------------------------------------------------------------------
// A per-device pool of zeroed ticket counters (runtime.hip).  A kernel whose workgroups each publish a partial
// result takes a ticket after an agent-scope release fence; the workgroup drawing the last ticket sees every
// partial (acquire fence), finishes the reduction and re-zeroes the counter — this replaces a separate
// few-hundred-thread "finalize" launch (a dependent launch costs ~4.5 us on MI355X whatever its size).
// `tickets(n)` hands out a rotating window of n counters, so kernels in flight on different streams never share.
unsigned* tickets(int n);
// Measured on MI355X (cfg1 step, r01): the fused tail costs what the finalize launch did — the last workgroup pays
// an atomic round trip, an L2 invalidate and cold reads of partials that were written through to memory:
// 1.712 ms with it vs 1.692 ms without.  Kept as an option (esc_tune_set(8, 1)), off by default.
// Dynamic-LDS floor of the GEMM launches of the calling thread: the step engine can raise it around the edge stream's
// GEMMs so that they occupy 3 instead of 4 workgroups per CU (esc_tune_set(10, bytes); 0 = off).
int gemm_lds_floor();
void set_gemm_lds_floor(int bytes);
int edge_lds_floor();
int node_lds_floor();          // least dynamic LDS of the 64-row tile launches (bytes): set by the step engine around its node chain
void set_node_lds_floor(int bytes);
void set_edge_lds_floor(int bytes);
int norm_rowblock_cap();              // workgroups per column block of the BatchNorm reduction kernels (esc_tune_set(9, v))
void set_norm_rowblock_cap(int v);
bool bn_bwd_one_launch();              // node-sized BatchNorm backward as ONE launch with a grid barrier (esc_tune_set(13, v))
void set_bn_bwd_one_launch(int on);
bool bn_bwd_fold();                    // node-sized BatchNorm backward: the apply kernel adds the partial slots itself (esc_tune_set(12, v))
void set_bn_bwd_fold(int on);
bool last_block_finalize();
void set_last_block_finalize(int on);

// Partials that the last workgroup will read are written with agent-scope (write-through, `sc1`) stores: they
// become visible device-wide when the store is acknowledged, so the publishing workgroups need NO L2 write-back /
// invalidate (a full `__threadfence()` per workgroup — buffer_wbl2 + buffer_inv — cost the edge-row GEMMs +25 %).
__device__ __forceinline__ void store_agent(float2* p, float2 v) {
  const unsigned long long bits = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true in every thread of the workgroup that arrives last among the `nblocks` sharing `counter`.
// All threads of the workgroup must call it (barriers inside); partials must have been written with store_agent.
__device__ __forceinline__ bool grid_last_block(unsigned* counter, unsigned nblocks) {
  __shared__ unsigned s_ticket;
  __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this thread's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const bool last = s_ticket == nblocks - 1;
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop stale lines: the partials are read from memory
    if (threadIdx.x == 0) *counter = 0;                  // ready for the next launch (kernel boundary orders it)
  }
  return last;
}

// Chan/Welford merge of two (count, mean, M2) partials
__device__ __forceinline__ void chan_merge(double& n, double& mu, double& m2, double nb, double mub, double m2b) {
  if (nb == 0.0) return;
  const double tot = n + nb;
  const double delta = mub - mu;
  mu += delta * nb / tot;
  m2 += m2b + delta * delta * n * nb / tot;
  n = tot;
}


// ---- BatchNorm statistics consumed in place --------------------------------------------------------------------------
// A forward GEMM leaves per-row-block (mean, M2) partials of its output columns; a separate finalize launch costs the
// latency-critical node chain ~6 us per BatchNorm (13 per training step).  Instead the CONSUMER of the normalised value
// merges the partials in its own prologue — every workgroup redundantly, in the same fixed order, so all of them see
// bit-identical coefficients — and workgroup 0 stores what the backward pass and the running statistics need.
struct BnFoldDev {
  const float2* partials;     // [P][C]; partial p covers rows [p*block_rows, min(M, (p+1)*block_rows))
  int P, block_rows, M, C;
  float eps, momentum;
  const float* gamma; const float* beta;            // may be null
  float* mean; float* invstd; float* scale; float* shift; float* running_mean; float* running_var;   // outputs (writer only)
};
// Division-free merge of the equal-size groups around the first group's mean (fp64: no cancellation), then one Chan
// merge with the ragged last group.  Sequential in p: a fixed order.
__device__ __forceinline__ void bn_fold_column(const BnFoldDev& f, int col, bool writer, float& sc, float& sh) {
  const int full = f.M / f.block_rows;
  double n = 0.0, mu = 0.0, m2 = 0.0;
  if (full > 0) {
    const double pivot = (double)f.partials[col].x;
    double S1 = 0.0, S2 = 0.0, SM = 0.0;
    // the partials were written by other workgroups: every load is an L2 / Infinity-Cache round trip, so ALL of a
    // chunk's loads are issued before the first is consumed (38 partials for 2 400 rows: one round trip, not five)
    constexpr int CH = 40;
    for (int p0 = 0; p0 < full; p0 += CH) {
      float2 v[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) v[u] = f.partials[(size_t)min(p0 + u, full - 1) * f.C + col];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (p0 + u < full) { const double d = (double)v[u].x - pivot; S1 += d; S2 += d * d; SM += (double)v[u].y; }
      }
    }
    n = (double)f.block_rows * full;
    mu = pivot + S1 / full;
    m2 = SM + (double)f.block_rows * (S2 - S1 * S1 / full);
    if (m2 < 0.0) m2 = 0.0;
  }
  if (f.M > f.block_rows * full) {
    const float2 v = f.partials[(size_t)full * f.C + col];
    chan_merge(n, mu, m2, (double)(f.M - f.block_rows * full), (double)v.x, (double)v.y);
  }
  const float is = (float)(1.0 / sqrt(m2 / (double)f.M + (double)f.eps));
  sc = (f.gamma ? f.gamma[col] : 1.f) * is;
  sh = (f.beta ? f.beta[col] : 0.f) - (float)mu * sc;
  if (writer) {
    f.mean[col] = (float)mu;
    f.invstd[col] = is;
    if (f.scale) { f.scale[col] = sc; f.shift[col] = sh; }
    if (f.running_mean) f.running_mean[col] = (1.f - f.momentum) * f.running_mean[col] + f.momentum * (float)mu;
    if (f.running_var) f.running_var[col] = (1.f - f.momentum) * f.running_var[col] + f.momentum * (float)(m2 / (double)(f.M - 1));
  }
}

// BatchNorm(+ReLU) BACKWARD applied to the A operand while it is staged (PRO bit 2): A holds the gradient dOut of the
// BatchNorm's output, `x` the rows the BatchNorm normalised (same shape as A), and the operand the MFMAs see is
//   dY = scale * (g - k1 - (x - mean) * invstd * k2),   g = dOut * [fmaf(x, scale, shift) > 0]   (relu; g = dOut otherwise)
// with (k1, k2) = coef = (sum g, sum g*xhat) / rows — the value esc_bn_bwd_apply would have written to memory for the
// GEMM to read back.  The elementwise launch between the finalize and the GEMM disappears from the dependent chain.
struct BnbDev {
  const float* x; int ldx;
  const float* mean; const float* invstd; const float* scale; const float* shift;
  const float2* coef;
  int relu;                         // activation behind the BatchNorm: 0 none, 1 ReLU, 2 ELU (the fused-activation codes of norm.hip)
};
// ... and the column sums (sum g, sum g*xhat) of the NEXT BatchNorm backward, taken over the rows of every output tile in
// the epilogue (PRO bit 3): C is the gradient of that BatchNorm's output, `x` what it normalised; partial[tile_m][col].
struct BnStatDev {
  float2* partial;
  const float* x; int ldx;
  const float* mean; const float* invstd; const float* scale; const float* shift;
  int relu;
};

// one element: gradient `gv` of the BatchNorm output, its input `xv`, channel coefficients mu = mean, a = scale,
// (ms, mh) = the mask's affine (0, 1 without activation), k1 = sum g / rows, k2 = invstd * sum g*xhat / rows
__device__ __forceinline__ float bnb_apply(float gv, float xv, float mu, float a, float ms, float mh, float k1, float k2) {
  const float gm = fmaf(xv, ms, mh) > 0.f ? gv : 0.f;
  return a * fmaf(mu - xv, k2, gm - k1);                               // = a * (g - k1 - (x - mu) * invstd * k2)
}
// ... with ELU(alpha = 1) behind the BatchNorm (zinc_models.py:513-522): d act / d v = v > 0 ? 1 : exp(v), v = the pre-activation
// the forward formed (fmaf(x, scale, shift)); `elu` is wave-uniform
__device__ __forceinline__ float bnb_apply_elu(float gv, float xv, float mu, float a, float ms, float mh, float k1, float k2) {
  const float v = fmaf(xv, ms, mh);
  const float gm = v > 0.f ? gv : gv * expf(v);
  return a * fmaf(mu - xv, k2, gm - k1);
}
// (the flag is tested ONCE per call site on a wave-uniform value: callers branch around whole fragments, so that the ReLU path
// carries no trace of the exp — folding the flag into the select cost the counting step 70 us)
__device__ __forceinline__ float bnb_apply_act(float gv, float xv, float mu, float a, float ms, float mh, float k1, float k2, bool elu) {
  return elu ? bnb_apply_elu(gv, xv, mu, a, ms, mh, k1, k2) : bnb_apply(gv, xv, mu, a, ms, mh, k1, k2);
}

}  // namespace esc
