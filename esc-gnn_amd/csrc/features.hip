// features.hip — edge-rooted h-hop ego-net structural encoding (the ESC-GNN pre_transform) on the GPU.
//
// Replaces create_subgraphs / k_hop_subgraph of /root/reference/utils_edge_efficient.py:20-152,201-294,
// a python loop over every directed edge with two tensor-mask BFS's, a relabel, a dense one_hot
// histogram and a scipy pinv per edge (0.2-0.9 s per 10..30-node graph on the CPU).
//
// Semantics restated (quirks kept; validated spec in SURVEY.md Appendix A):
//   * self_loop: drop every (a,a), append (i,i) i<n at the END (:33-36)
//   * BFS walks target->source (:210,:222-226); hop > h is reported as h+1 (:56-61)
//   * sub-edges = edges induced by S_u UNION edges induced by S_v — not induced on the union (:55)
//   * sub-degree = out-degree over sub-edges, self loops count (:86)
//   * an edge with u==v carries a phantom isolated duplicate of the root: deg 0, z=(0,0), rd 0 (:52-54,:66)
//   * rd = pinv(L)[r,r]+pinv(L)[i,i]-pinv(L)[r,i]-pinv(L)[i,r], L = D_in - A without self loops
//     (scipy csgraph.laplacian), fp64 -> fp32 -> trunc (:92-107,:131)
//   * histogram layout [0,200) degree | [200,300) d(u,.) | [300,400) d(v,.) | [400,500) rd |
//     500.. (400.. without rd) 216 z_s0 + 36 z_s1 + 6 z_t0 + z_t1 over non-loop sub-edges (:129-138)
//   * sparse form: ascending bin index per edge, pos_batch = graph-local edge id (:140-143)
//
// GPU design: one wave per output edge, everything per-edge lives in LDS (hop labels, degrees, the
// 1800-bin histogram, and for rd the m x m fp64 matrices).  Both BFS's run level-synchronously over
// the graph's (L2-resident) edge list; histogram updates are LDS integer atomics; the pseudo-inverse
// is a one-sided Jacobi SVD (Hestenes) in fp64 — it handles the rank deficiency of a Laplacian and
// non-symmetric L of directed inputs alike — parallelised over the round-robin column pairs of a
// sweep with row-split sub-lanes.  Two passes (count nnz -> exclusive scan -> fill) because the
// sparse output size is data dependent; integer work, bound by LDS/L2 latency, not HBM.
#include "common.h"

namespace esc {

constexpr int HIST_BINS = 1800;
constexpr int MAX_SUBGRAPH = 96;        // rd matrices: 2 * m^2 * 8 B of LDS  (m <= 96 -> 147 KB)
constexpr unsigned char HOP_INF = 255;

// ---- exclusive scan of int32 counts into int64 offsets (single workgroup, chunked with carry) -----
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int* __restrict__ cnt, int64_t n,
                                                           int64_t* __restrict__ out) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    int64_t v = (i < n) ? (int64_t)cnt[i] : 0;
    int64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const int64_t carry = carry_s;
    if (i < n) out[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry_s;
}

// ---- pass 0: per-graph edge-list normalisation (one wave per graph) -------------------------------
// work edge list of graph g lives at offset edge_ptr[g] + node_ptr[g] (room for all edges + n loops)
__global__ __launch_bounds__(64) void feat_prepare_kernel(const int64_t* __restrict__ node_ptr,
                                                          const int64_t* __restrict__ edge_ptr,
                                                          const int64_t* __restrict__ src,
                                                          const int64_t* __restrict__ dst, int G, int self_loop,
                                                          int* __restrict__ w_src, int* __restrict__ w_dst,
                                                          int* __restrict__ w_in, int* __restrict__ e_out,
                                                          int* __restrict__ status) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int lane = threadIdx.x;
  const int64_t e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
  const int64_t n = node_ptr[g + 1] - node_ptr[g];
  const int64_t wo = e0 + node_ptr[g];
  int base = 0, bad = 0;
  for (int64_t j0 = e0; j0 < e1; j0 += 64) {
    const int64_t j = j0 + lane;
    int64_t s = 0, d = 0;
    bool keep = false;
    if (j < e1) {
      s = src[j]; d = dst[j];
      if (s < 0 || s >= n || d < 0 || d >= n) bad = 1;
      keep = !(self_loop && s == d);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int p = base + __popcll(m & ((1ull << lane) - 1ull));
      w_src[wo + p] = (int)s; w_dst[wo + p] = (int)d; w_in[wo + p] = (int)(j - e0);
    }
    base += __popcll(m);
  }
  if (self_loop) {
    for (int64_t i = lane; i < n; i += 64) {
      w_src[wo + base + i] = (int)i; w_dst[wo + base + i] = (int)i; w_in[wo + base + i] = -1;
    }
    base += (int)n;
  }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) {
    status[g] = bad ? ESC_ERANGE : 0;
    e_out[g] = bad ? 0 : base;
  }
}

// ---- one-sided Jacobi SVD based pseudo-inverse probe, one wave, matrices in LDS (column-major) -----
// On entry Gm = L (m x m), Vm = I.  On exit Gm = L*V with mutually orthogonal columns.
// Returns the squared-norm threshold below which a column counts as numerically null:
// 4 (m eps)^2 ||L||_F^2, i.e. scipy.linalg.pinv's default cutoff max(M,N)*eps*sigma_max with a
// small margin (||L||_F >= sigma_max).  Null columns are neither rotated nor inverted.
__device__ double jacobi_orthogonalise(double* __restrict__ Gm, double* __restrict__ Vm, int m, int ld) {
  const int lane = threadIdx.x;
  double fro2 = 0.0;
  for (int idx = lane; idx < m * m; idx += 64) {
    const double v = Gm[(idx / m) * ld + (idx % m)];
    fro2 += v * v;
  }
  fro2 = wave_sum(fro2);
  const double meps = (double)m * 2.220446049250313e-16;
  const double null2 = 4.0 * meps * meps * fro2;
  if (m < 2 || fro2 == 0.0) return null2;
  const int M = (m + 1) & ~1;                  // round-robin needs an even player count (last = dummy)
  const int npairs = M / 2;
  int S = 1;                                   // sub-lanes per pair (power of two)
  while (npairs * S * 2 <= 64) S *= 2;
  const int pairs_per_pass = 64 / S;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int rotated = 0;
    for (int r = 0; r < M - 1; ++r) {
      for (int p0 = 0; p0 < npairs; p0 += pairs_per_pass) {
        const int pi = p0 + lane / S, sub = lane % S;
        int p = -1, q = -1;
        if (pi < npairs) {
          const int a = pi, b = M - 1 - pi;
          p = (a == 0) ? 0 : 1 + ((a - 1 + r) % (M - 1));
          q = 1 + ((b - 1 + r) % (M - 1));
          if (p > q) { const int t = p; p = q; q = t; }
          if (q >= m) p = -1;                  // paired with the dummy player
        }
        double al = 0.0, be = 0.0, ga = 0.0;
        if (p >= 0) {
          const double* gp = Gm + p * ld;
          const double* gq = Gm + q * ld;
          for (int i = sub; i < m; i += S) {
            const double x = gp[i], y = gq[i];
            al += x * x; be += y * y; ga += x * y;
          }
        }
        for (int o = 1; o < S; o <<= 1) {
          al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64);
        }
        bool rot = false;
        double c = 1.0, s = 0.0;
        if (p >= 0 && al > null2 && be > null2 && fabs(ga) > 1e-15 * sqrt(al * be)) {
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          c = 1.0 / sqrt(1.0 + t * t);
          s = c * t;
          rot = true;
        }
        if (rot) {
          double* gp = Gm + p * ld; double* gq = Gm + q * ld;
          double* vp = Vm + p * ld; double* vq = Vm + q * ld;
          for (int i = sub; i < m; i += S) {
            const double x = gp[i], y = gq[i];
            gp[i] = c * x - s * y; gq[i] = s * x + c * y;
            const double u = vp[i], w = vq[i];
            vp[i] = c * u - s * w; vq[i] = s * u + c * w;
          }
        }
        rotated |= __any(rot) ? 1 : 0;
        __syncthreads();
      }
    }
    if (!rotated) break;
  }
  return null2;
}

// ---- pass 1/2: encode every output edge (one wave per edge) ---------------------------------------
struct EncodeArgs {
  const int64_t* node_ptr; const int64_t* edge_ptr;
  const int* w_src; const int* w_dst; const int* e_out;
  const int64_t* out_edge_ptr;   // [G+1]
  const int* status_in;
  int G, h, use_rd, n_cap, m_cap;
  int* nnz_cnt;                  // count pass: per output edge
  int* status;                   // per graph, sticky error
  const int64_t* nnz_ptr;        // fill pass
  int64_t* pos_enc; int64_t* pos_index; int64_t* pos_batch;
};

template <bool FILL>
__global__ __launch_bounds__(64) void feat_encode_kernel(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t k_glob = blockIdx.x;
  const int64_t total = a.out_edge_ptr[a.G];
  if (k_glob >= total) return;
  // graph of this edge: last g with out_edge_ptr[g] <= k_glob
  int lo = 0, hi = a.G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.out_edge_ptr[mid] <= k_glob) lo = mid; else hi = mid;
  }
  const int g = lo;
  const int k = (int)(k_glob - a.out_edge_ptr[g]);
  const int n = (int)(a.node_ptr[g + 1] - a.node_ptr[g]);
  const int Eg = a.e_out[g];
  const int64_t wo = a.edge_ptr[g] + a.node_ptr[g];
  const int* __restrict__ es = a.w_src + wo;
  const int* __restrict__ ed = a.w_dst + wo;
  const int h = a.h, far = a.h + 1;

  // LDS carve-up
  int* hist = reinterpret_cast<int*>(smem);                         // [1800]
  int* deg = hist + HIST_BINS;                                      // [n_cap]
  short* loc = reinterpret_cast<short*>(deg + a.n_cap);             // [n_cap]
  unsigned char* hop_u = reinterpret_cast<unsigned char*>(loc + a.n_cap);   // [n_cap]
  unsigned char* hop_v = hop_u + a.n_cap;                           // [n_cap]
  size_t off = (size_t)(hop_v + a.n_cap - smem);
  off = (off + 15) & ~(size_t)15;
  double* Gm = reinterpret_cast<double*>(smem + off);               // [m_cap*m_cap] column-major
  double* Vm = Gm + (size_t)a.m_cap * a.m_cap;
  double* inv_s2 = Vm + (size_t)a.m_cap * a.m_cap;                  // [m_cap]

  const int u = es[k], v = ed[k];
  for (int i = lane; i < HIST_BINS; i += 64) hist[i] = 0;
  for (int x = lane; x < n; x += 64) { hop_u[x] = HOP_INF; hop_v[x] = HOP_INF; deg[x] = 0; }
  __syncthreads();
  if (lane == 0) { hop_u[u] = 0; hop_v[v] = 0; }
  __syncthreads();
  // level-synchronous BFS from both roots, stepping target -> source
  for (int d = 1; d <= h; ++d) {
    for (int j = lane; j < Eg; j += 64) {
      const int s = es[j], t = ed[j];
      if (hop_u[t] == d - 1 && hop_u[s] == HOP_INF) hop_u[s] = (unsigned char)d;
      if (hop_v[t] == d - 1 && hop_v[s] == HOP_INF) hop_v[s] = (unsigned char)d;
    }
    __syncthreads();
  }
  // local indices of S_u ∪ S_v (any order; rd is order independent given the root)
  int m = 0;
  for (int x0 = 0; x0 < n; x0 += 64) {
    const int x = x0 + lane;
    const bool in = x < n && (hop_u[x] != HOP_INF || hop_v[x] != HOP_INF);
    const unsigned long long msk = __ballot(in);
    if (x < n) loc[x] = in ? (short)(m + __popcll(msk & ((1ull << lane) - 1ull))) : (short)-1;
    m += __popcll(msk);
  }
  const bool do_rd = a.use_rd != 0;
  bool bad = false;
  if (do_rd) {
    if (m > a.m_cap) {
      bad = true;
    } else {
      for (int idx = lane; idx < m * m; idx += 64) {
        const int c = idx / m, r = idx % m;
        Gm[c * a.m_cap + r] = 0.0;
        Vm[c * a.m_cap + r] = (c == r) ? 1.0 : 0.0;
      }
    }
  }
  __syncthreads();
  const int off_edge = do_rd ? 500 : 400;
  // sub-edges: union of the two induced edge sets
  for (int j = lane; j < Eg; j += 64) {
    const int s = es[j], t = ed[j];
    const unsigned char us = hop_u[s], ut = hop_u[t], vs = hop_v[s], vt = hop_v[t];
    const bool in_sub = (us != HOP_INF && ut != HOP_INF) || (vs != HOP_INF && vt != HOP_INF);
    if (!in_sub) continue;
    atomicAdd(&deg[s], 1);
    if (s != t) {
      const int z0s = us == HOP_INF ? far : us, z1s = vs == HOP_INF ? far : vs;
      const int z0t = ut == HOP_INF ? far : ut, z1t = vt == HOP_INF ? far : vt;
      const int code = 216 * z0s + 36 * z1s + 6 * z0t + z1t;
      if (code >= 1300) { bad = true; } else { atomicAdd(&hist[off_edge + code], 1); }
      if (do_rd && !bad) {
        const int ls = loc[s], lt = loc[t];
        atomicAdd(&Gm[lt * a.m_cap + ls], -1.0);        // L[s][t] -= 1   (column-major: [col t][row s])
        atomicAdd(&Gm[lt * a.m_cap + lt], 1.0);         // L[t][t] += 1   (in-degree, loops excluded)
      }
    }
  }
  __syncthreads();
  // node terms
  for (int x = lane; x < n; x += 64) {
    const unsigned char hu = hop_u[x], hv = hop_v[x];
    if (hu == HOP_INF && hv == HOP_INF) continue;
    const int dg = deg[x];
    if (dg >= 200) { bad = true; continue; }
    atomicAdd(&hist[dg], 1);
    atomicAdd(&hist[200 + (hu == HOP_INF ? far : hu)], 1);
    atomicAdd(&hist[300 + (hv == HOP_INF ? far : hv)], 1);
  }
  const bool phantom = (u == v);
  if (phantom && lane == 0) {
    atomicAdd(&hist[0], 1); atomicAdd(&hist[200], 1); atomicAdd(&hist[300], 1);
    if (do_rd) atomicAdd(&hist[400], 1);                // rd of the isolated duplicate = 0
  }
  bad = __any(bad);
  if (do_rd && !bad) {
    __syncthreads();
    const double null2 = jacobi_orthogonalise(Gm, Vm, m, a.m_cap);
    __syncthreads();
    // squared singular values = squared column norms; numerically-null ones are dropped (pinv cutoff)
    for (int j = lane; j < m; j += 64) {
      double s2 = 0.0;
      for (int i = 0; i < m; ++i) { const double t = Gm[j * a.m_cap + i]; s2 += t * t; }
      inv_s2[j] = (s2 > null2) ? 1.0 / s2 : 0.0;
    }
    __syncthreads();
    const int r = loc[u];
    double prr = 0.0;
    if (!phantom)
      for (int j = 0; j < m; ++j) prr += Vm[j * a.m_cap + r] * Gm[j * a.m_cap + r] * inv_s2[j];
    for (int x = lane; x < n; x += 64) {
      const int i = loc[x];
      if (i < 0) continue;
      double pii = 0.0, pri = 0.0, pir = 0.0;
      for (int j = 0; j < m; ++j) {
        const double w = inv_s2[j];
        const double vi = Vm[j * a.m_cap + i], gi = Gm[j * a.m_cap + i];
        pii += vi * gi * w;
        if (!phantom) {
          pri += Vm[j * a.m_cap + r] * gi * w;            // P[r][i] = sum_j V[r,j] G[i,j] / s_j^2
          pir += vi * Gm[j * a.m_cap + r] * w;            // P[i][r]
        }
      }
      const double rd64 = phantom ? pii : (((prr + pii) - pri) - pir);
      const float rd32 = (float)rd64;
      if (!(rd32 > -1.0f && rd32 < 100.0f)) { bad = true; continue; }
      atomicAdd(&hist[400 + (int)rd32], 1);               // (int) truncates toward zero like .long()
    }
    bad = __any(bad);
  }
  __syncthreads();
  if (bad) {
    if (lane == 0) a.status[g] = ESC_ERANGE;
    if (!FILL) { if (lane == 0) a.nnz_cnt[k_glob] = 0; }
    return;
  }
  // sparse form, ascending bin index
  int64_t base = FILL ? a.nnz_ptr[k_glob] : 0;
  int cnt = 0;
  for (int i0 = 0; i0 < HIST_BINS; i0 += 64) {
    const int i = i0 + lane;
    const int val = (i < HIST_BINS) ? hist[i] : 0;
    const unsigned long long msk = __ballot(val != 0);
    if (FILL && val != 0) {
      const int64_t p = base + cnt + __popcll(msk & ((1ull << lane) - 1ull));
      a.pos_enc[p] = val; a.pos_index[p] = i; a.pos_batch[p] = k;
    }
    cnt += __popcll(msk);
  }
  if (!FILL && lane == 0) a.nnz_cnt[k_glob] = cnt;
}

__global__ __launch_bounds__(256) void feat_edges_out_kernel(const int64_t* __restrict__ node_ptr,
                                                             const int64_t* __restrict__ edge_ptr,
                                                             const int* __restrict__ w_src,
                                                             const int* __restrict__ w_dst,
                                                             const int* __restrict__ w_in,
                                                             const int64_t* __restrict__ out_edge_ptr, int G,
                                                             int64_t* __restrict__ out_src,
                                                             int64_t* __restrict__ out_dst,
                                                             int64_t* __restrict__ in_edge_of_out) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int64_t wo = edge_ptr[g] + node_ptr[g];
  const int64_t o0 = out_edge_ptr[g], cnt = out_edge_ptr[g + 1] - o0;
  for (int64_t j = threadIdx.x; j < cnt; j += blockDim.x) {
    out_src[o0 + j] = w_src[wo + j];
    out_dst[o0 + j] = w_dst[wo + j];
    if (in_edge_of_out) in_edge_of_out[o0 + j] = w_in[wo + j] < 0 ? -1 : (edge_ptr[g] + w_in[wo + j]);
  }
}

struct WorkLayout {
  int* w_src; int* w_dst; int* w_in; int* e_out; int* nnz_cnt;
};
static WorkLayout carve(void* work, int64_t G, int64_t total_nodes, int64_t total_in_edges) {
  const int64_t cap = total_in_edges + total_nodes;
  int* p = reinterpret_cast<int*>(work);
  WorkLayout w;
  w.w_src = p; p += cap;
  w.w_dst = p; p += cap;
  w.w_in = p; p += cap;
  w.e_out = p; p += G;
  w.nnz_cnt = p;
  return w;
}

static size_t encode_lds_bytes(int n_cap, int m_cap, int use_rd) {
  size_t b = (size_t)HIST_BINS * 4 + (size_t)n_cap * 4 + (size_t)n_cap * 2 + (size_t)n_cap * 2;
  b = (b + 15) & ~(size_t)15;
  if (use_rd) b += ((size_t)2 * m_cap * m_cap + m_cap) * 8;
  return b;
}

}  // namespace esc

using namespace esc;

extern "C" {

int64_t esc_features_scratch_bytes(int64_t G, int64_t total_nodes, int64_t total_in_edges) {
  const int64_t cap = total_in_edges + total_nodes;
  return (4 * cap + G + 16) * 4;
}

static int check_common(const int64_t* node_ptr, const int64_t* edge_ptr, int64_t G, int h, int64_t max_nodes) {
  ESC_REQUIRE(node_ptr && edge_ptr, "esc_features: null pointer");
  ESC_REQUIRE(G > 0 && G < (1LL << 31), "esc_features: bad graph count %ld", (long)G);
  if (h < 1 || h > 4) {
    set_error("esc_features: h=%d outside 1..4 (edge codes need hop labels <= 5, utils_edge_efficient.py:137)", h);
    return ESC_ERANGE;
  }
  if (max_nodes < 1 || max_nodes > 4096) {
    set_error("esc_features: max_nodes=%ld outside 1..4096", (long)max_nodes);
    return ESC_ERANGE;
  }
  return ESC_OK;
}

int esc_features_count(const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* src,
                       const int64_t* dst, int64_t G, int64_t total_nodes, int64_t total_in_edges,
                       int64_t max_nodes, int h, int use_rd, int self_loop, int64_t* out_edge_ptr,
                       int64_t* nnz_ptr, int32_t* status, void* work, void* stream) {
  int rc = check_common(node_ptr, edge_ptr, G, h, max_nodes);
  if (rc) return rc;
  ESC_REQUIRE((src && dst) || total_in_edges == 0, "esc_features_count: null edge arrays");
  ESC_REQUIRE(out_edge_ptr && nnz_ptr && status && work, "esc_features_count: null output");
  ESC_REQUIRE(total_nodes >= 0 && total_in_edges >= 0 && total_nodes + total_in_edges < (1LL << 31) - 1,
              "esc_features_count: too many nodes+edges in one call");
  hipStream_t s = (hipStream_t)stream;
  WorkLayout w = carve(work, G, total_nodes, total_in_edges);
  esc::launch(ESC_K_FEATURES, feat_prepare_kernel, dim3((unsigned)G), dim3(64), 0, s, node_ptr, edge_ptr, src, dst, (int)G,
                     self_loop, w.w_src, w.w_dst, w.w_in, w.e_out, status);
  ESC_CHECK_LAUNCH("esc_features_count.prepare");
  esc::launch(ESC_K_FEATURES, scan_counts_kernel, dim3(1), dim3(1024), 0, s, w.e_out, G, out_edge_ptr);
  ESC_CHECK_LAUNCH("esc_features_count.scan_edges");
  const int64_t cap_edges = total_in_edges + (self_loop ? total_nodes : 0);
  if (cap_edges == 0) {
    (void)hipMemsetAsync(nnz_ptr, 0, sizeof(int64_t), s);
    return ESC_OK;
  }
  EncodeArgs a{};
  a.node_ptr = node_ptr; a.edge_ptr = edge_ptr; a.w_src = w.w_src; a.w_dst = w.w_dst; a.e_out = w.e_out;
  a.out_edge_ptr = out_edge_ptr; a.G = (int)G; a.h = h; a.use_rd = use_rd;
  a.n_cap = (int)((max_nodes + 3) & ~3LL);
  a.m_cap = (int)(max_nodes < MAX_SUBGRAPH ? ((max_nodes + 1) & ~1LL) : MAX_SUBGRAPH);
  a.nnz_cnt = w.nnz_cnt; a.status = status;
  // nnz_cnt lives in `work` after e_out: needs cap_edges ints — covered by esc_features_scratch_bytes
  (void)hipMemsetAsync(w.nnz_cnt, 0, sizeof(int) * (size_t)cap_edges, s);
  const size_t lds = encode_lds_bytes(a.n_cap, a.m_cap, use_rd);
  ESC_REQUIRE(lds <= 160 * 1024, "esc_features_count: graph too large for the LDS encoder (%zu B)", lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)feat_encode_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  esc::launch(ESC_K_FEATURES, feat_encode_kernel<false>, dim3((unsigned)cap_edges), dim3(64), lds, s, a);
  ESC_CHECK_LAUNCH("esc_features_count.encode");
  esc::launch(ESC_K_FEATURES, scan_counts_kernel, dim3(1), dim3(1024), 0, s, w.nnz_cnt, cap_edges, nnz_ptr);
  ESC_CHECK_LAUNCH("esc_features_count.scan_nnz");
  return ESC_OK;
}

int esc_features_fill(const int64_t* node_ptr, const int64_t* edge_ptr, int64_t G, int64_t total_nodes,
                      int64_t total_in_edges, int64_t max_nodes, int h, int use_rd, int self_loop,
                      const int64_t* out_edge_ptr, const int64_t* nnz_ptr, int64_t total_out_edges,
                      int64_t* out_src, int64_t* out_dst, int64_t* in_edge_of_out, int64_t* pos_enc,
                      int64_t* pos_index, int64_t* pos_batch, int32_t* status, void* work, void* stream) {
  int rc = check_common(node_ptr, edge_ptr, G, h, max_nodes);
  if (rc) return rc;
  ESC_REQUIRE(out_edge_ptr && nnz_ptr && status && work, "esc_features_fill: null pointer");
  ESC_REQUIRE(total_out_edges >= 0, "esc_features_fill: bad edge total");
  if (total_out_edges == 0) return ESC_OK;
  ESC_REQUIRE(out_src && out_dst && pos_enc && pos_index && pos_batch, "esc_features_fill: null output");
  hipStream_t s = (hipStream_t)stream;
  WorkLayout w = carve(work, G, total_nodes, total_in_edges);
  esc::launch(ESC_K_FEATURES, feat_edges_out_kernel, dim3((unsigned)G), dim3(256), 0, s, node_ptr, edge_ptr, w.w_src, w.w_dst,
                     w.w_in, out_edge_ptr, (int)G, out_src, out_dst, in_edge_of_out);
  ESC_CHECK_LAUNCH("esc_features_fill.edges");
  EncodeArgs a{};
  a.node_ptr = node_ptr; a.edge_ptr = edge_ptr; a.w_src = w.w_src; a.w_dst = w.w_dst; a.e_out = w.e_out;
  a.out_edge_ptr = out_edge_ptr; a.G = (int)G; a.h = h; a.use_rd = use_rd;
  a.n_cap = (int)((max_nodes + 3) & ~3LL);
  a.m_cap = (int)(max_nodes < MAX_SUBGRAPH ? ((max_nodes + 1) & ~1LL) : MAX_SUBGRAPH);
  a.status = status; a.nnz_ptr = nnz_ptr; a.pos_enc = pos_enc; a.pos_index = pos_index; a.pos_batch = pos_batch;
  const size_t lds = encode_lds_bytes(a.n_cap, a.m_cap, use_rd);
  ESC_REQUIRE(lds <= 160 * 1024, "esc_features_fill: graph too large for the LDS encoder (%zu B)", lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)feat_encode_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  esc::launch(ESC_K_FEATURES, feat_encode_kernel<true>, dim3((unsigned)total_out_edges), dim3(64), lds, s, a);
  ESC_CHECK_LAUNCH("esc_features_fill.encode");
  return ESC_OK;
}

}  // extern "C"
