// aggregate.hip — GINE neighbour aggregation (the "scatter-add" of NestedGIN_eff) without atomics.
//
//   forward   out[i,:] = (1+eps)*x[i,:] + sum_{k : dst_k = i, ascending k} relu(x[src_k,:] + e[k,:])
//   backward  d_e[k,:] = [x[src_k,:]+e[k,:] > 0] * g[dst_k,:]
//             dx[i,:]  = (1+eps)*g[i,:] + sum_{k : src_k = i} d_e[k,:]
//             deps     = sum_{i,c} g[i,c]*x[i,c]
//
// Replaces PyG 2.0.4 GINEConv.propagate (index_select gather + torch_scatter scatter-add) reached
// from /root/reference/run_graphcount.py:161,169; semantics as restated in
// /root/reference/GraphGPS/graphgps/layer/gine_conv_layer.py:56-84 (minus r_ij); hand-rolled twin
// /root/reference/ogb_mol_gnn.py:346-358.
//
// Design (HBM/L2-bound, no MFMA): global float atomics run at ~1.3 TB/s chip-wide on gfx950, so the
// scatter is turned into a *segmented gather-reduce* over a destination-sorted CSR view of the
// edge list (stable => the per-node sum runs in ascending edge order = the order a sequential
// scatter_add_ uses => bitwise reproducible and bit-identical to the CPU oracle).  One wave owns
// one node row; with C=256 each lane holds a float4, so every x/e row access is one coalesced
// 1 KiB wave read.  x (N*C*4 = 2.4 MB @cfg1) stays L2-resident; e rows stream once.
// Algorithmic bytes/launch: 2*E*C*4 + 2*N*C*4 + E*8 + (N+1)*4.
#include "common.h"

#include <cstdlib>

// Bitwise contract with the sequential CPU scatter: this file is compiled with -ffp-contract=off
// (see Makefile) so a*b+c is never fused behind our back; explicit fmaf() calls still emit FMAs
// where the order is free.

namespace esc {

// ---- wide rows: one wave per node, VEC floats per lane per pass ---------------------------------
// Edges are consumed in batches of AGG_BATCH with every row read of the batch issued before the first
// use (2*AGG_BATCH loads in flight per wave); a short tail batch is padded by clamping the edge slot to
// the node's last edge and masking its contribution, so the typical in-degree (6-7 for the counting
// graphs) costs ONE round of memory latency instead of one per leftover edge.
constexpr int AGG_BATCH = 8;

// VEC consecutive floats of a row <-> registers (16-, 8- or 4-byte accesses)
template <int VEC> __device__ __forceinline__ void row_load(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
  else if constexpr (VEC == 2) { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[1] = a.y; }
  else v[0] = *p;
}
template <int VEC> __device__ __forceinline__ void row_store(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  else *p = v[0];
}

// AFF: x holds PRE-activation rows of a BatchNorm(+ReLU) whose output was never materialised; every row read applies
// relu(x*xa_scale + xa_shift) on the fly (same fmaf + max as esc_affine_act, so the sums are bit-identical to the
// materialised path) — the step engine's node chain loses one elementwise launch per layer.
template <int VEC, bool AFF = false, bool SPAN = false>
__global__ __launch_bounds__(256) void agg_fwd_wave(const float* __restrict__ x, int64_t ld_x,
                                                    const float* __restrict__ e, int64_t ld_e,
                                                    const int* __restrict__ in_ptr,
                                                    const int* __restrict__ in_edge,
                                                    const int* __restrict__ in_src,
                                                    const float* __restrict__ eps_p, int N, int C,
                                                    float* __restrict__ out, int64_t ld_out,
                                                    const float* __restrict__ xa_scale, const float* __restrict__ xa_shift, int split,
                                                    unsigned long long* __restrict__ span) {
  ESC_PRIO();
  // span != NULL (esc_prof_span_arm, diagnostics): the launch's execution window on the device's wall clock — every workgroup stores
  // the time of its first instruction, every wave the time of its last one (plain stores into per-launch slots: atomics on shared
  // slots cost this kernel 20 us); the host takes min / max
  // (SPAN is its own instantiation: the production kernel carries none of this)
  if constexpr (SPAN) {
    if (span != nullptr && threadIdx.x == 0 && blockIdx.x < ESC_SPAN_WGS) span[blockIdx.x] = (unsigned long long)wall_clock64();
  }
  struct SpanEnd {
    unsigned long long* p;
    __device__ ~SpanEnd() {
      if constexpr (SPAN) {
        const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (p != nullptr && lane_id() == 0 && w < ESC_SPAN_WAVES) { __builtin_amdgcn_s_waitcnt(0); p[ESC_SPAN_WGS + w] = (unsigned long long)wall_clock64(); }
      }
    }
  } span_end{span};
  (void)span_end;
  // split > 1: `split` waves share one destination row, each owning C / split consecutive columns.  A wave is three
  // dependent memory round trips (segment pointers -> edge ids -> rows) for one row's worth of bytes: 2 400 one-shot waves
  // on 1 024 SIMDs cannot hide them; twice the waves at half the bytes each can.  Every column still sums its edges in
  // ascending order, so the result stays bit-identical to the sequential scatter.
  const int wid = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  const int node = wid / split;
  if (node >= N) return;
  const int cw = C / split, c_lo = (wid - node * split) * cw, c_hi = c_lo + cw;
  const int lane = lane_id();
  const int beg = uniform(in_ptr[node]);
  const int end = uniform(in_ptr[node + 1]);
  const bool has_self = eps_p != nullptr;           // nullptr: plain neighbour sum (GINE+ per-distance terms)
  const bool has_e = e != nullptr;                  // nullptr: message = relu(x_j)
  const float one_eps = has_self ? __fadd_rn(1.0f, *eps_p) : 0.f;
  for (int c = c_lo + lane * VEC; c < c_hi; c += WAVE * VEC) {
    float acc[VEC], self[VEC], asc[VEC], ash[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) { acc[t] = 0.f; asc[t] = 1.f; ash[t] = 0.f; }
    if constexpr (AFF) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) { asc[t] = xa_scale[c + t]; ash[t] = xa_shift[c + t]; }
    }
    {
      row_load<VEC>(x + (size_t)node * ld_x + c, self);
      if constexpr (AFF) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) self[t] = fmaxf(fmaf(self[t], asc[t], ash[t]), 0.f);
      }
    }
    for (int j = beg; j < end; j += AGG_BATCH) {
      float xv[AGG_BATCH][VEC], ev[AGG_BATCH][VEC];
#pragma unroll
      for (int u = 0; u < AGG_BATCH; ++u) {
        const int jj = min(j + u, end - 1);                 // wave-uniform clamp (tail padding)
        const int k = uniform(in_edge[jj]);
        const int s = uniform(in_src[jj]);
        const float* px = x + (size_t)s * ld_x + c;
        const float* pe = has_e ? e + (size_t)k * ld_e + c : px;
        row_load<VEC>(px, xv[u]);
        if (has_e) row_load<VEC>(pe, ev[u]);
        else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) ev[u][t] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < AGG_BATCH; ++u) {
        if (j + u < end) {                                  // wave-uniform: ascending-edge order is preserved
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            const float xs = AFF ? fmaxf(fmaf(xv[u][t], asc[t], ash[t]), 0.f) : xv[u][t];
            acc[t] = __fadd_rn(acc[t], fmaxf(has_e ? __fadd_rn(xs, ev[u][t]) : xs, 0.f));
          }
        }
      }
    }
    float* po = out + (size_t)node * ld_out + c;
    if (has_self) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = __fadd_rn(acc[t], __fmul_rn(one_eps, self[t]));
    }
    row_store<VEC>(po, acc);
  }
}

// rows of exactly 256 floats (the hidden width of the reference, run_graphcount.py:465) may be split over two waves of 128
// columns, 8 bytes per lane (see agg_fwd_wave).  Measured on a config-1 batch (profiles/r03_kernel_roofline.txt): forward
// 7.2 -> 6.5 us with cold operands, 6.4 either way cache-resident: ON (ESC_AGG_SPLIT=1 switches it off); backward 11.8 -> 11.0
// cold but 8.2 -> 9.3 with the operands the step has just produced: OFF (ESC_AGG_SPLIT_BWD=2 switches it on).  Non-temporal loads
// of the once-streamed edge-term rows (so that they would not evict the gathered x rows from the L2) measured SLOWER
// (cache-resident 6.4 -> 7.4 us, cold 6.5 -> 7.2) and were removed again.
static int g_agg_split = getenv("ESC_AGG_SPLIT") ? atoi(getenv("ESC_AGG_SPLIT")) : 2;
static int g_agg_split_bwd = getenv("ESC_AGG_SPLIT_BWD") ? atoi(getenv("ESC_AGG_SPLIT_BWD")) : 1;
static inline int agg_split(int64_t C) { return (g_agg_split == 2 && C == 256) ? 2 : 1; }
static inline int agg_split_bwd(int64_t C) { return (g_agg_split_bwd == 2 && C == 256) ? 2 : 1; }

// ---- narrow rows (C < 64, e.g. the 10-wide first layer): one thread per (node, channel) ----------
__global__ __launch_bounds__(256) void agg_fwd_elem(const float* __restrict__ x, int64_t ld_x,
                                                    const float* __restrict__ e, int64_t ld_e,
                                                    const int* __restrict__ in_ptr,
                                                    const int* __restrict__ in_edge,
                                                    const int* __restrict__ in_src,
                                                    const float* __restrict__ eps_p, int N, int C,
                                                    float* __restrict__ out, int64_t ld_out) {
  ESC_PRIO();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)N * C) return;
  const int node = (int)(t / C), c = (int)(t % C);
  float acc = 0.f;
  for (int j = in_ptr[node]; j < in_ptr[node + 1]; ++j) {
    const float xs = x[(size_t)in_src[j] * ld_x + c];
    const float v = e ? __fadd_rn(xs, e[(size_t)in_edge[j] * ld_e + c]) : xs;
    acc = __fadd_rn(acc, fmaxf(v, 0.f));
  }
  if (eps_p) acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(1.0f, *eps_p), x[(size_t)node * ld_x + c]));
  out[(size_t)node * ld_out + c] = acc;
}

// ---- backward: one wave per SOURCE node (out-CSR) -------------------------------------------------
// each edge has exactly one source, so d_e[k] is written exactly once and dx[i] needs no atomics.
// BST (with AFF, dx != NULL, accumulate or not): dx is then the FINAL gradient of the BatchNorm(+ReLU) output x' whose
// pre-activation rows are x, so the column sums of that BatchNorm's backward — (sum g, sum g*xhat), g = dx * [x' > 0],
// xhat = (x - mean) * invstd — are taken here, per workgroup (4 source rows), and the separate partial-sum pass over
// dx and x before the finalize disappears from the node chain: partial[blockIdx.x][c] (float2).
template <int VEC, bool AFF = false, bool BST = false>
__global__ __launch_bounds__(256) void agg_bwd_wave(const float* __restrict__ x, int64_t ld_x,
                                                    const float* __restrict__ e, int64_t ld_e,
                                                    const float* __restrict__ g, int64_t ld_g,
                                                    const int* __restrict__ out_ptr,
                                                    const int* __restrict__ out_edge,
                                                    const int* __restrict__ out_dst,
                                                    const float* __restrict__ eps_p, int N, int C,
                                                    float* __restrict__ d_e, int64_t ld_de,
                                                    float* __restrict__ dx, int64_t ld_dx, int accumulate_dx,
                                                    float* __restrict__ deps_part,
                                                    const float* __restrict__ xa_scale, const float* __restrict__ xa_shift, int split,
                                                    const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd,
                                                    float2* __restrict__ bn_partial) {
  ESC_PRIO();
  static_assert(!BST || AFF, "the BatchNorm sums need the pre-activation rows");
  extern __shared__ __attribute__((aligned(16))) float2 bst_sh[];        // BST: [4 waves][C]
  // split > 1: as in agg_fwd_wave — `split` waves share a source row, C / split columns each; deps_part then holds
  // N * split partial dot products (wave w writes deps_part[w])
  const int wid = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  const int node = wid / split;
  if constexpr (BST) {
    if (node >= N) {                                  // an idle wave of the last workgroup still feeds the workgroup's sum (zeros)
      for (int c = lane_id(); c < C; c += WAVE) bst_sh[(threadIdx.x >> 6) * C + c] = make_float2(0.f, 0.f);
      __syncthreads();                                // ... and takes its share of the columns in the final pass
      for (int c = threadIdx.x; c < C; c += 256) {
        float2 a = bst_sh[c];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float2 b = bst_sh[w * C + c]; a.x += b.x; a.y += b.y; }
        bn_partial[(size_t)blockIdx.x * C + c] = a;
      }
      return;
    }
  } else {
    if (node >= N) return;
  }
  const int cw = C / split, c_lo = (wid - node * split) * cw, c_hi = c_lo + cw;
  const int lane = lane_id();
  const int beg = uniform(out_ptr[node]);
  const int end = uniform(out_ptr[node + 1]);
  const bool has_self = eps_p != nullptr, has_e = e != nullptr;
  const float one_eps = has_self ? 1.0f + *eps_p : 0.f;
  float dot = 0.f;
  for (int c = c_lo + lane * VEC; c < c_hi; c += WAVE * VEC) {
    float xi[VEC], gi[VEC], acc[VEC], xraw[VEC];
    {
      row_load<VEC>(x + (size_t)node * ld_x + c, xi);
      row_load<VEC>(g + (size_t)node * ld_g + c, gi);
#pragma unroll
      for (int t = 0; t < VEC; ++t) xraw[t] = xi[t];
      if constexpr (AFF) {                                  // x is the pre-activation row: the layer input is relu(x*scale+shift)
#pragma unroll
        for (int t = 0; t < VEC; ++t) xi[t] = fmaxf(fmaf(xi[t], xa_scale[c + t], xa_shift[c + t]), 0.f);
      }
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) { acc[t] = 0.f; dot = fmaf(xi[t], gi[t], dot); }
    for (int j = beg; j < end; j += AGG_BATCH) {          // all 2*AGG_BATCH row reads of a batch in flight
      float ev[AGG_BATCH][VEC], gv[AGG_BATCH][VEC];
      int kk[AGG_BATCH];
#pragma unroll
      for (int u = 0; u < AGG_BATCH; ++u) {
        const int jj = min(j + u, end - 1);
        kk[u] = uniform(out_edge[jj]);
        const int d = uniform(out_dst[jj]);
        const float* pg = g + (size_t)d * ld_g + c;
        const float* pe = has_e ? e + (size_t)kk[u] * ld_e + c : pg;
        if (has_e) row_load<VEC>(pe, ev[u]);
        else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) ev[u][t] = 0.f;
        }
        row_load<VEC>(pg, gv[u]);
      }
#pragma unroll
      for (int u = 0; u < AGG_BATCH; ++u) {
        if (j + u < end) {
          float o[VEC];
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            o[t] = ((has_e ? __fadd_rn(xi[t], ev[u][t]) : xi[t]) > 0.f) ? gv[u][t] : 0.f;
            acc[t] += o[t];
          }
          if (d_e != nullptr) row_store<VEC>(d_e + (size_t)kk[u] * ld_de + c, o);
        }
      }
    }
    if (dx != nullptr) {
      float* po = dx + (size_t)node * ld_dx + c;
      float o[VEC], prev[VEC];
      if (accumulate_dx) row_load<VEC>(po, prev);
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        o[t] = fmaf(one_eps, gi[t], acc[t]);
        if (accumulate_dx) o[t] += prev[t];
      }
      row_store<VEC>(po, o);
      if constexpr (BST) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          const float gm = xi[t] > 0.f ? o[t] : 0.f;                    // x' = relu(.) > 0  <=>  the pre-activation is > 0
          bst_sh[(threadIdx.x >> 6) * C + c + t] = make_float2(gm, gm * ((xraw[t] - bn_mean[c + t]) * bn_invstd[c + t]));
        }
      }
    }
  }
  if (deps_part != nullptr) {
    dot = wave_sum(dot);
    if (lane == 0) deps_part[wid] = dot;
  }
  if constexpr (BST) {            // the four rows of the workgroup, added in wave order
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float2 a = bst_sh[c];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const float2 b = bst_sh[w * C + c]; a.x += b.x; a.y += b.y; }
      bn_partial[(size_t)blockIdx.x * C + c] = a;
    }
  }
}

// ---- graph readout: global_add_pool / global_mean_pool over the (sorted) node->graph vector -----------
// (reference run_graphcount.py:179 graph_pred=True; zinc_models.py:602).  One wave per graph walks its
// contiguous node range in order => bit-identical to a sequential index_add_.
template <int VEC>
__global__ __launch_bounds__(256) void segment_pool_fwd(const float* __restrict__ x, int64_t ld_x,
                                                        const int* __restrict__ seg_ptr, int G, int C, int mean,
                                                        float* __restrict__ out, int64_t ld_out) {
  ESC_PRIO();
  const int gidx = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  if (gidx >= G) return;
  const int lane = lane_id();
  const int beg = uniform(seg_ptr[gidx]), end = uniform(seg_ptr[gidx + 1]);
  const float cnt = (float)max(end - beg, 1);
  for (int c = lane * VEC; c < C; c += WAVE * VEC) {
    float acc[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[t] = 0.f;
#pragma unroll 4
    for (int r = beg; r < end; ++r) {
      const float* p = x + (size_t)r * ld_x + c;
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        acc[0] = __fadd_rn(acc[0], q.x); acc[1] = __fadd_rn(acc[1], q.y);
        acc[2] = __fadd_rn(acc[2], q.z); acc[3] = __fadd_rn(acc[3], q.w);
      } else {
        acc[0] = __fadd_rn(acc[0], *p);
      }
    }
    if (mean) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = acc[t] / cnt;
    }
    float* o = out + (size_t)gidx * ld_out + c;
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else *o = acc[0];
  }
}

// dx[i,:] = g[graph(i),:] (/ count) — one wave per graph broadcasts its gradient row to its nodes
template <int VEC>
__global__ __launch_bounds__(256) void segment_pool_bwd(const float* __restrict__ g, int64_t ld_g,
                                                        const int* __restrict__ seg_ptr, int G, int C, int mean,
                                                        float* __restrict__ dx, int64_t ld_dx) {
  ESC_PRIO();
  const int gidx = uniform((int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6));
  if (gidx >= G) return;
  const int lane = lane_id();
  const int beg = uniform(seg_ptr[gidx]), end = uniform(seg_ptr[gidx + 1]);
  const float cnt = (float)max(end - beg, 1);
  for (int c = lane * VEC; c < C; c += WAVE * VEC) {
    float v[VEC];
    const float* p = g + (size_t)gidx * ld_g + c;
    if constexpr (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4*>(p);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = *p;
    }
    if (mean) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) v[t] = v[t] / cnt;
    }
    for (int r = beg; r < end; ++r) {
      float* o = dx + (size_t)r * ld_dx + c;
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      else *o = v[0];
    }
  }
}

// deterministic single-block sum of n floats -> out[0] (fp64 accumulation)
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ v, int64_t n,
                                                          float* __restrict__ out) {
  ESC_PRIO();
  __shared__ double sh[16];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += (double)v[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    out[0] = (float)t;
  }
}

// several independent sums in one launch: workgroup j reduces job j exactly like reduce_sum_kernel
struct SumJobs { esc_sum_job job[ESC_MAX_SUM_JOBS]; };
__global__ __launch_bounds__(1024) void reduce_sum_jobs_kernel(SumJobs t) {
  ESC_PRIO();
  __shared__ double sh[16];
  const esc_sum_job& q = t.job[blockIdx.x];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < q.n; i += blockDim.x) acc += (double)q.v[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    q.out[0] = (float)s;
  }
}

}  // namespace esc

extern "C" {

int esc_gine_aggregate_fwd(const float* x, int64_t ld_x, const float* e, int64_t ld_e,
                           const int32_t* in_ptr, const int32_t* in_edge, const int32_t* in_src,
                           const float* eps, int64_t N, int64_t C, float* out, int64_t ld_out,
                           void* stream) {
  ESC_REQUIRE(x && in_ptr && out, "esc_gine_aggregate_fwd: null pointer");
  ESC_REQUIRE(N >= 0 && C > 0 && ld_x >= C && (!e || ld_e >= C) && ld_out >= C, "esc_gine_aggregate_fwd: bad sizes N=%ld C=%ld", (long)N, (long)C);
  ESC_REQUIRE(N < (1LL << 31) / 64, "esc_gine_aggregate_fwd: N too large");
  if (N == 0) return ESC_OK;
  ESC_REQUIRE(in_edge && in_src, "esc_gine_aggregate_fwd: null edge arrays");
  hipStream_t s = (hipStream_t)stream;
  if (C >= 64) {
    const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (!e || ld_e % 4 == 0) && (ld_out % 4 == 0) &&
                     esc::aligned16(x) && (!e || esc::aligned16(e)) && esc::aligned16(out);
    const int64_t blocks = esc::cdiv(N, 4);
    if (vec && esc::agg_split(C) == 2)
      esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<2, false>, dim3((unsigned)esc::cdiv(2 * N, 4)), dim3(256), 0, s, x, ld_x, e, ld_e, in_ptr, in_edge, in_src, eps, (int)N, (int)C, out, ld_out, (const float*)nullptr, (const float*)nullptr, 2, (unsigned long long*)nullptr);
    else if (vec)
      esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<4, false>, dim3(blocks), dim3(256), 0, s, x, ld_x, e, ld_e, in_ptr, in_edge, in_src, eps, (int)N, (int)C, out, ld_out, (const float*)nullptr, (const float*)nullptr, 1, (unsigned long long*)nullptr);
    else
      esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<1, false>, dim3(blocks), dim3(256), 0, s, x, ld_x, e, ld_e, in_ptr, in_edge, in_src, eps, (int)N, (int)C, out, ld_out, (const float*)nullptr, (const float*)nullptr, 1, (unsigned long long*)nullptr);
  } else {
    const int64_t blocks = esc::cdiv(N * C, 256);
    esc::launch(-1, esc::agg_fwd_elem, dim3(blocks), dim3(256), 0, s, x, ld_x, e, ld_e, in_ptr, in_edge, in_src, eps, (int)N, (int)C, out, ld_out);
  }
  ESC_CHECK_LAUNCH("esc_gine_aggregate_fwd");
  return ESC_OK;
}

/* deps_part entries per node the backward writes for rows of C floats (1, or 2 when the row is split over two waves) */
int esc_gine_aggregate_bwd_deps_slots(int64_t C) { return esc::agg_split_bwd(C); }

int esc_gine_aggregate_bwd(const float* x, int64_t ld_x, const float* e, int64_t ld_e,
                           const float* g, int64_t ld_g, const int32_t* out_ptr,
                           const int32_t* out_edge, const int32_t* out_dst, const float* eps,
                           int64_t N, int64_t C, float* d_e, int64_t ld_de, float* dx,
                           int64_t ld_dx, int accumulate_dx, float* deps_part, void* stream) {
  ESC_REQUIRE(x && g && out_ptr, "esc_gine_aggregate_bwd: null pointer");
  ESC_REQUIRE((e == nullptr) == (d_e == nullptr) || e != nullptr, "esc_gine_aggregate_bwd: d_e without e");
  ESC_REQUIRE(N >= 0 && C > 0 && ld_x >= C && (!e || ld_e >= C) && ld_g >= C && (!d_e || ld_de >= C) && (!dx || ld_dx >= C),
              "esc_gine_aggregate_bwd: bad sizes");
  ESC_REQUIRE(N < (1LL << 31) / 64, "esc_gine_aggregate_bwd: N too large");
  if (N == 0) return ESC_OK;
  ESC_REQUIRE(out_edge && out_dst, "esc_gine_aggregate_bwd: null edge arrays");
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (!e || ld_e % 4 == 0) && (ld_g % 4 == 0) && (!d_e || ld_de % 4 == 0) &&
                   (!dx || ld_dx % 4 == 0) && esc::aligned16(x) && (!e || esc::aligned16(e)) && esc::aligned16(g) &&
                   (!d_e || esc::aligned16(d_e)) && (!dx || esc::aligned16(dx));
  const int64_t blocks = esc::cdiv(N, 4);
  if (vec && esc_gine_aggregate_bwd_deps_slots(C) == 2)
    esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<2, false>, dim3((unsigned)esc::cdiv(2 * N, 4)), dim3(256), 0, s, x, ld_x, e, ld_e, g, ld_g, out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, (const float*)nullptr, (const float*)nullptr, 2, (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  else if (vec)
    esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<4, false>, dim3(blocks), dim3(256), 0, s, x, ld_x, e, ld_e, g, ld_g, out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, (const float*)nullptr, (const float*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  else
    esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<1, false>, dim3(blocks), dim3(256), 0, s, x, ld_x, e, ld_e, g, ld_g, out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, (const float*)nullptr, (const float*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  ESC_CHECK_LAUNCH("esc_gine_aggregate_bwd");
  return ESC_OK;
}

// The same two passes with the layer input given as the PRE-activation rows of a BatchNorm+ReLU that was never written:
// x' = relu(x*x_scale + x_shift) is applied to every row as it is read (forward: the gathered and the self rows; backward:
// the own row, for the ReLU mask of x'+e and for deps).  dx stays the gradient with respect to x'.  Wide rows only.
int esc_gine_aggregate_fwd_affine(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* e,
                                  int64_t ld_e, const int32_t* in_ptr, const int32_t* in_edge, const int32_t* in_src,
                                  const float* eps, int64_t N, int64_t C, float* out, int64_t ld_out, void* stream) {
  ESC_REQUIRE(x && x_scale && x_shift && in_ptr && out, "esc_gine_aggregate_fwd_affine: null pointer");
  ESC_REQUIRE(N >= 0 && C >= 64 && C % 4 == 0 && ld_x >= C && ld_x % 4 == 0 && (!e || (ld_e >= C && ld_e % 4 == 0)) && ld_out >= C && ld_out % 4 == 0 &&
              N < (1LL << 31) / 64, "esc_gine_aggregate_fwd_affine: needs C >= 64, C and the leading dimensions multiples of 4 (N=%ld C=%ld)", (long)N, (long)C);
  ESC_REQUIRE(esc::aligned16(x) && (!e || esc::aligned16(e)) && esc::aligned16(out), "esc_gine_aggregate_fwd_affine: pointers must be 16-byte aligned");
  if (N == 0) return ESC_OK;
  ESC_REQUIRE(in_edge && in_src, "esc_gine_aggregate_fwd_affine: null edge arrays");
  unsigned long long* span = esc::prof_span_next(ESC_K_AGG_FWD);      // non-NULL only for launches armed by esc_prof_span_arm
  if (esc::agg_split(C) == 2 && span != nullptr)
    esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<2, true, true>, dim3((unsigned)esc::cdiv(2 * N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, e, ld_e, in_ptr,
                in_edge, in_src, eps, (int)N, (int)C, out, ld_out, x_scale, x_shift, 2, span);
  else if (esc::agg_split(C) == 2)
    esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<2, true>, dim3((unsigned)esc::cdiv(2 * N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, e, ld_e, in_ptr,
                in_edge, in_src, eps, (int)N, (int)C, out, ld_out, x_scale, x_shift, 2, (unsigned long long*)nullptr);
  else
    esc::launch(ESC_K_AGG_FWD, esc::agg_fwd_wave<4, true>, dim3((unsigned)esc::cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, e, ld_e, in_ptr,
                in_edge, in_src, eps, (int)N, (int)C, out, ld_out, x_scale, x_shift, 1, (unsigned long long*)nullptr);
  ESC_CHECK_LAUNCH("esc_gine_aggregate_fwd_affine");
  return ESC_OK;
}

int esc_gine_aggregate_bwd_affine(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* e,
                                  int64_t ld_e, const float* g, int64_t ld_g, const int32_t* out_ptr, const int32_t* out_edge,
                                  const int32_t* out_dst, const float* eps, int64_t N, int64_t C, float* d_e, int64_t ld_de,
                                  float* dx, int64_t ld_dx, int accumulate_dx, float* deps_part, void* stream) {
  ESC_REQUIRE(x && x_scale && x_shift && g && out_ptr, "esc_gine_aggregate_bwd_affine: null pointer");
  ESC_REQUIRE((e == nullptr) == (d_e == nullptr) || e != nullptr, "esc_gine_aggregate_bwd_affine: d_e without e");
  ESC_REQUIRE(N >= 0 && C >= 64 && C % 4 == 0 && ld_x >= C && ld_x % 4 == 0 && (!e || (ld_e >= C && ld_e % 4 == 0)) && ld_g >= C && ld_g % 4 == 0 &&
              (!d_e || (ld_de >= C && ld_de % 4 == 0)) && (!dx || (ld_dx >= C && ld_dx % 4 == 0)) && N < (1LL << 31) / 64,
              "esc_gine_aggregate_bwd_affine: needs C >= 64, C and the leading dimensions multiples of 4");
  ESC_REQUIRE(esc::aligned16(x) && (!e || esc::aligned16(e)) && esc::aligned16(g) && (!d_e || esc::aligned16(d_e)) && (!dx || esc::aligned16(dx)),
              "esc_gine_aggregate_bwd_affine: pointers must be 16-byte aligned");
  if (N == 0) return ESC_OK;
  ESC_REQUIRE(out_edge && out_dst, "esc_gine_aggregate_bwd_affine: null edge arrays");
  if (esc_gine_aggregate_bwd_deps_slots(C) == 2)
    esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<2, true>, dim3((unsigned)esc::cdiv(2 * N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, e, ld_e, g, ld_g,
                out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, x_scale, x_shift, 2, (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  else
    esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<4, true>, dim3((unsigned)esc::cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, e, ld_e, g, ld_g,
                out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, x_scale, x_shift, 1, (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  ESC_CHECK_LAUNCH("esc_gine_aggregate_bwd_affine");
  return ESC_OK;
}

/* esc_gine_aggregate_bwd_affine that also leaves the column sums of the BatchNorm backward in front of it: dx (required) is
 * then the complete gradient of x' = relu(BN(x)), and partial[slot][C] (float2, slot = 4 consecutive source rows,
 * esc_gine_aggregate_bwd_stats_slots(N) of them) holds (sum g, sum g*xhat) for esc_bn_bwd_coef_from_partials. */
int64_t esc_gine_aggregate_bwd_stats_slots(int64_t N) { return esc::cdiv(N, 4); }

int esc_gine_aggregate_bwd_affine_stats(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* bn_mean,
                                        const float* bn_invstd, const float* e, int64_t ld_e, const float* g, int64_t ld_g,
                                        const int32_t* out_ptr, const int32_t* out_edge, const int32_t* out_dst, const float* eps, int64_t N,
                                        int64_t C, float* d_e, int64_t ld_de, float* dx, int64_t ld_dx, int accumulate_dx, float* deps_part,
                                        float* partial, void* stream) {
  ESC_REQUIRE(x && x_scale && x_shift && bn_mean && bn_invstd && g && out_ptr && dx && partial, "esc_gine_aggregate_bwd_affine_stats: null pointer");
  ESC_REQUIRE((e == nullptr) == (d_e == nullptr) || e != nullptr, "esc_gine_aggregate_bwd_affine_stats: d_e without e");
  ESC_REQUIRE(N >= 0 && C >= 64 && C % 4 == 0 && C <= 2048 && ld_x >= C && ld_x % 4 == 0 && (!e || (ld_e >= C && ld_e % 4 == 0)) && ld_g >= C && ld_g % 4 == 0 &&
              (!d_e || (ld_de >= C && ld_de % 4 == 0)) && ld_dx >= C && ld_dx % 4 == 0 && N < (1LL << 31) / 64,
              "esc_gine_aggregate_bwd_affine_stats: needs 64 <= C <= 2048, C and the leading dimensions multiples of 4");
  ESC_REQUIRE(esc::aligned16(x) && (!e || esc::aligned16(e)) && esc::aligned16(g) && (!d_e || esc::aligned16(d_e)) && esc::aligned16(dx) && esc::aligned16(partial),
              "esc_gine_aggregate_bwd_affine_stats: pointers must be 16-byte aligned");
  if (N == 0) return ESC_OK;
  ESC_REQUIRE(out_edge && out_dst, "esc_gine_aggregate_bwd_affine_stats: null edge arrays");
  esc::launch(ESC_K_AGG_BWD, esc::agg_bwd_wave<4, true, true>, dim3((unsigned)esc::cdiv(N, 4)), dim3(256), (size_t)4 * C * sizeof(float2), (hipStream_t)stream,
              x, ld_x, e, ld_e, g, ld_g, out_ptr, out_edge, out_dst, eps, (int)N, (int)C, d_e, ld_de, dx, ld_dx, accumulate_dx, deps_part, x_scale,
              x_shift, 1, bn_mean, bn_invstd, reinterpret_cast<float2*>(partial));
  ESC_CHECK_LAUNCH("esc_gine_aggregate_bwd_affine_stats");
  return ESC_OK;
}

int esc_segment_pool_fwd(const float* x, int64_t ld_x, const int32_t* seg_ptr, int64_t G, int64_t C, int mean,
                         float* out, int64_t ld_out, void* stream) {
  ESC_REQUIRE(x && seg_ptr && out, "esc_segment_pool_fwd: null pointer");
  ESC_REQUIRE(G > 0 && C > 0 && ld_x >= C && ld_out >= C && G < (1LL << 31) / 64, "esc_segment_pool_fwd: bad sizes");
  const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_out % 4 == 0) && esc::aligned16(x) && esc::aligned16(out);
  if (vec) esc::launch(-1, esc::segment_pool_fwd<4>, dim3((unsigned)esc::cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, seg_ptr, (int)G, (int)C, mean, out, ld_out);
  else     esc::launch(-1, esc::segment_pool_fwd<1>, dim3((unsigned)esc::cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, x, ld_x, seg_ptr, (int)G, (int)C, mean, out, ld_out);
  ESC_CHECK_LAUNCH("esc_segment_pool_fwd");
  return ESC_OK;
}

int esc_segment_pool_bwd(const float* g, int64_t ld_g, const int32_t* seg_ptr, int64_t G, int64_t C, int mean,
                         float* dx, int64_t ld_dx, void* stream) {
  ESC_REQUIRE(g && seg_ptr && dx, "esc_segment_pool_bwd: null pointer");
  ESC_REQUIRE(G > 0 && C > 0 && ld_g >= C && ld_dx >= C && G < (1LL << 31) / 64, "esc_segment_pool_bwd: bad sizes");
  const bool vec = (C % 4 == 0) && (ld_g % 4 == 0) && (ld_dx % 4 == 0) && esc::aligned16(g) && esc::aligned16(dx);
  if (vec) esc::launch(-1, esc::segment_pool_bwd<4>, dim3((unsigned)esc::cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, g, ld_g, seg_ptr, (int)G, (int)C, mean, dx, ld_dx);
  else     esc::launch(-1, esc::segment_pool_bwd<1>, dim3((unsigned)esc::cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, g, ld_g, seg_ptr, (int)G, (int)C, mean, dx, ld_dx);
  ESC_CHECK_LAUNCH("esc_segment_pool_bwd");
  return ESC_OK;
}

int esc_reduce_sum(const float* v, int64_t n, float* out, void* stream) {
  ESC_REQUIRE(out && (v || n == 0) && n >= 0, "esc_reduce_sum: bad argument");
  esc::launch(-1, esc::reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, out);
  ESC_CHECK_LAUNCH("esc_reduce_sum");
  return ESC_OK;
}

int esc_reduce_sum_jobs(const esc_sum_job* jobs, int count, void* stream) {
  ESC_REQUIRE(count >= 0 && (jobs || count == 0), "esc_reduce_sum_jobs: bad argument");
  for (int first = 0; first < count; first += ESC_MAX_SUM_JOBS) {
    esc::SumJobs t{};
    const int n = count - first < ESC_MAX_SUM_JOBS ? count - first : ESC_MAX_SUM_JOBS;
    for (int j = 0; j < n; ++j) {
      ESC_REQUIRE(jobs[first + j].out && (jobs[first + j].v || jobs[first + j].n == 0) && jobs[first + j].n >= 0, "esc_reduce_sum_jobs: bad job %d", first + j);
      t.job[j] = jobs[first + j];
    }
    esc::launch(-1, esc::reduce_sum_jobs_kernel, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, t);
  }
  ESC_CHECK_LAUNCH("esc_reduce_sum_jobs");
  return ESC_OK;
}

}  // extern "C"
