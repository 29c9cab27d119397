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
// GPU design.  The reference redoes everything per directed edge; the work that does not depend on the edge is shared
// (SURVEY.md App. A (i),(iv)):
//   hop tables   one BFS per ROOT NODE (one wave each, level-synchronous over the graph's L2-resident edge list) ->
//                hop[g][r][x] bytes; an edge (u,v) reads rows u and v instead of running two BFS's
//   rd classes   the Laplacian of an edge's ego-net depends only on the two node SETS {S_u, S_v}.  Roots with equal
//                reach sets get one canonical id (64-bit hash prefilter, then an exact row compare), an edge's class is
//                the unordered canonical pair, and ONE pseudo-inverse is computed per class present in the graph (for
//                the counting graphs at h=3 most ego-nets are the whole graph: ~10 classes instead of ~125 edges).
//                The pinv is a one-sided Jacobi SVD (Hestenes) in fp64 — it handles the rank deficiency of a Laplacian
//                and the non-symmetric L of directed inputs alike.  Classes are bucketed by ego-net size: m <= 32 / 64 /
//                96 keep both m x m matrices in LDS (16 / 66 / 147 KB), larger ones run in a 256-thread workgroup on a
//                global-memory slab (no size limit below the 4096-node cap).  The class wave then walks the graph's
//                edges, and for each member edge bins rd against ITS root into a 100-bin per-edge row.
//   encode       one wave per output edge: sub-degrees and edge codes from the two hop rows, the rd row added in, the
//                1800-bin histogram in LDS (integer atomics) compacted to ascending sparse form.  Two passes (count nnz
//                -> exclusive scan -> fill) because the sparse size is data dependent; everything shared above is
//                computed once, in the count pass, and kept in `work` for the fill pass.
// Integer work bound by LDS/L2 latency, not HBM.
#include "common.h"

namespace esc {

constexpr int HIST_BINS = 1800;
constexpr int RD_BINS = 100;
constexpr int LDS_SUBGRAPH = 96;        // largest ego-net whose two m x m fp64 matrices fit LDS (147 KB)
constexpr unsigned char HOP_INF = 255;
constexpr int N_BUCKETS = 4;            // ego-net size classes: <=32, <=64, <=96 (LDS), larger (global slab)

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

// prefix sums of n_g^2: where graph g's hop table (and its class-flag bits) start
__global__ __launch_bounds__(1024) void feat_sq_scan_kernel(const int64_t* __restrict__ node_ptr, int64_t G,
                                                            int64_t* __restrict__ out) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = 0; base < G; base += 1024) {
    const int64_t i = base + threadIdx.x;
    int64_t v = 0;
    if (i < G) { const int64_t n = node_ptr[i + 1] - node_ptr[i]; v = n * n; }
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
    if (i < G) out[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[G] = carry_s;
}

// last g with ptr[g] <= i
__device__ __forceinline__ int find_segment(const int64_t* __restrict__ ptr, int G, int64_t i) {
  int lo = 0, hi = G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- hop tables: one wave per root node ------------------------------------------------------------
// hop[sq_ptr[g] + r*n + x] = d(r,x) walking target -> source, HOP_INF beyond h; root_hash = hash of the reach SET
__global__ __launch_bounds__(64) void feat_bfs_kernel(const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
                                                      const int* __restrict__ w_src, const int* __restrict__ w_dst,
                                                      const int* __restrict__ e_out, const int64_t* __restrict__ sq_ptr,
                                                      int G, int h, int64_t total_nodes,
                                                      unsigned char* __restrict__ hop_tab,
                                                      unsigned long long* __restrict__ root_hash) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t r_glob = blockIdx.x;
  if (r_glob >= total_nodes) return;
  const int g = find_segment(node_ptr, G, r_glob);
  const int n = (int)(node_ptr[g + 1] - node_ptr[g]);
  const int r = (int)(r_glob - node_ptr[g]);
  const int Eg = e_out[g];
  const int64_t wo = edge_ptr[g] + node_ptr[g];
  const int* __restrict__ es = w_src + wo;
  const int* __restrict__ ed = w_dst + wo;
  unsigned char* hop = smem;
  for (int x = lane; x < n; x += 64) hop[x] = HOP_INF;
  __syncthreads();
  if (lane == 0) hop[r] = 0;
  __syncthreads();
  for (int d = 1; d <= h; ++d) {
    int grew = 0;
    for (int j = lane; j < Eg; j += 64) {
      const int s = es[j], t = ed[j];
      if (hop[t] == d - 1 && hop[s] == HOP_INF) { hop[s] = (unsigned char)d; grew = 1; }
    }
    __syncthreads();
    if (!__any(grew)) break;
  }
  unsigned char* __restrict__ row = hop_tab + sq_ptr[g] + (int64_t)r * n;
  unsigned long long hsh = 0x9E3779B97F4A7C15ull;
  for (int x0 = 0; x0 < n; x0 += 64) {
    const int x = x0 + lane;
    const unsigned char v = x < n ? hop[x] : HOP_INF;
    if (x < n) row[x] = v;
    const unsigned long long msk = __ballot(v != HOP_INF);
    hsh = (hsh ^ msk) * 0xFF51AFD7ED558CCDull;
    hsh ^= hsh >> 33;
  }
  if (lane == 0) root_hash[r_glob] = hsh;
}

// ---- canonical root of every reach set: smallest r' with S_r' == S_r (exact; the hash only prefilters) ---------
__global__ __launch_bounds__(64) void feat_canon_kernel(const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ sq_ptr,
                                                        int G, int64_t total_nodes,
                                                        const unsigned char* __restrict__ hop_tab,
                                                        const unsigned long long* __restrict__ root_hash,
                                                        int* __restrict__ canon) {
  const int lane = threadIdx.x;
  const int64_t r_glob = blockIdx.x;
  if (r_glob >= total_nodes) return;
  const int g = find_segment(node_ptr, G, r_glob);
  const int64_t n0 = node_ptr[g];
  const int n = (int)(node_ptr[g + 1] - n0);
  const int r = (int)(r_glob - n0);
  const unsigned char* __restrict__ tab = hop_tab + sq_ptr[g];
  const unsigned char* __restrict__ mine = tab + (int64_t)r * n;
  const unsigned long long hsh = root_hash[r_glob];
  int found = r;
  for (int c0 = 0; c0 < r && found == r; c0 += 64) {
    const int c = c0 + lane;
    unsigned long long cand = __ballot(c < r && root_hash[n0 + c] == hsh);
    while (cand) {
      const int q = c0 + __ffsll((long long)cand) - 1;
      cand &= cand - 1;
      const unsigned char* __restrict__ other = tab + (int64_t)q * n;
      int diff = 0;
      for (int x = lane; x < n; x += 64) diff |= ((mine[x] != HOP_INF) != (other[x] != HOP_INF));
      if (!__any(diff)) { found = q; break; }
    }
  }
  if (lane == 0) canon[r_glob] = found;
}

// ---- rd classes present in each graph: one wave per output edge --------------------------------------
struct ClassArgs {
  const int64_t* node_ptr; const int64_t* edge_ptr; const int64_t* sq_ptr; const int64_t* out_edge_ptr;
  const int* w_src; const int* w_dst; const int* e_out; const int* canon;
  const unsigned char* hop_tab;
  unsigned* cls_flag; int* cls_count; int* cls_list;     // [N_BUCKETS] counters, [N_BUCKETS][cap] representative edges
  short* rd_rows;                                        // [cap][RD_BINS]
  double* slabs;                                         // HUGE: gridDim.x slabs of 2*ld*ld doubles
  int* status;
  int64_t cap;
  int G, ld, bucket, n_cap;
};

__global__ __launch_bounds__(64) void feat_class_mark_kernel(ClassArgs a) {
  const int lane = threadIdx.x;
  const int64_t k_glob = blockIdx.x;
  if (k_glob >= a.out_edge_ptr[a.G]) return;
  const int g = find_segment(a.out_edge_ptr, a.G, k_glob);
  const int k = (int)(k_glob - a.out_edge_ptr[g]);
  const int64_t n0 = a.node_ptr[g];
  const int n = (int)(a.node_ptr[g + 1] - n0);
  const int64_t wo = a.edge_ptr[g] + n0;
  const int u = a.w_src[wo + k], v = a.w_dst[wo + k];
  const unsigned char* __restrict__ hu = a.hop_tab + a.sq_ptr[g] + (int64_t)u * n;
  const unsigned char* __restrict__ hv = a.hop_tab + a.sq_ptr[g] + (int64_t)v * n;
  int m = 0;
  for (int x0 = 0; x0 < n; x0 += 64) {
    const int x = x0 + lane;
    m += __popcll(__ballot(x < n && (hu[x] != HOP_INF || hv[x] != HOP_INF)));
  }
  if (lane == 0) {
    const int cu = a.canon[n0 + u], cv = a.canon[n0 + v];
    const int lo = cu < cv ? cu : cv, hi = cu < cv ? cv : cu;
    const int64_t slot = a.sq_ptr[g] + (int64_t)lo * n + hi;
    const unsigned bit = 1u << (slot & 31);
    const unsigned old = atomicOr(&a.cls_flag[slot >> 5], bit);
    if (!(old & bit)) {
      const int b = m <= 32 ? 0 : m <= 64 ? 1 : m <= LDS_SUBGRAPH ? 2 : 3;
      const int idx = atomicAdd(&a.cls_count[b], 1);
      a.cls_list[(int64_t)b * a.cap + idx] = (int)k_glob;
    }
  }
}

// ---- one-sided Jacobi SVD based pseudo-inverse probe, NT threads, column-major matrices (LDS or global) -----
// On entry Gm = L (m x m), Vm = I.  On exit Gm = L*V with mutually orthogonal columns.
// Returns the squared-norm threshold below which a column counts as numerically null:
// 4 (m eps)^2 ||L||_F^2, i.e. scipy.linalg.pinv's default cutoff max(M,N)*eps*sigma_max with a
// small margin (||L||_F >= sigma_max).  Null columns are neither rotated nor inverted.
template <int NT>
__device__ double jacobi_orthogonalise(double* __restrict__ Gm, double* __restrict__ Vm, int m, int ld, double* red) {
  const int tid = threadIdx.x;
  double fro2 = 0.0;
  for (int idx = tid; idx < m * m; idx += NT) {
    const double v = Gm[(size_t)(idx / m) * ld + (idx % m)];
    fro2 += v * v;
  }
  fro2 = wave_sum(fro2);
  if (NT > 64) {                                  // fixed order over the waves: the threshold is the same in every thread
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = fro2;
    __syncthreads();
    fro2 = 0.0;
    for (int w = 0; w < NT / 64; ++w) fro2 += red[w];
    __syncthreads();
  }
  const double meps = (double)m * 2.220446049250313e-16;
  const double null2 = 4.0 * meps * meps * fro2;
  if (m < 2 || fro2 == 0.0) return null2;
  const int M = (m + 1) & ~1;                  // round-robin needs an even player count (last = dummy)
  const int npairs = M / 2;
  int S = 1;                                   // sub-lanes per pair (power of two, inside one wave)
  while (S < 64 && npairs * S * 2 <= NT) S *= 2;
  if (NT > 64 && S < 8) S = 8;                 // global-memory matrices: 8 consecutive rows per pair = one 64-B segment
  const int pairs_per_pass = NT / S;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int rotated = 0;
    for (int r = 0; r < M - 1; ++r) {
      for (int p0 = 0; p0 < npairs; p0 += pairs_per_pass) {
        const int pi = p0 + tid / S, sub = tid % S;
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
          const double* gp = Gm + (size_t)p * ld;
          const double* gq = Gm + (size_t)q * ld;
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
          double* gp = Gm + (size_t)p * ld; double* gq = Gm + (size_t)q * ld;
          double* vp = Vm + (size_t)p * ld; double* vq = Vm + (size_t)q * ld;
          for (int i = sub; i < m; i += S) {
            const double x = gp[i], y = gq[i];
            gp[i] = c * x - s * y; gq[i] = s * x + c * y;
            const double u = vp[i], w = vq[i];
            vp[i] = c * u - s * w; vq[i] = s * u + c * w;
          }
        }
        rotated |= __syncthreads_or(rot ? 1 : 0);
      }
    }
    if (!rotated) break;
  }
  return null2;
}

// ---- one pseudo-inverse per class, then the rd row of every member edge ----------------------------------
// HUGE = false: one wave, matrices in LDS with leading dimension a.ld; HUGE = true: 256 threads, matrices on this
// workgroup's global slab.  Both stride over the bucket's class list (its length is only known on the device).
template <bool HUGE>
__global__ __launch_bounds__(HUGE ? 256 : 64) void feat_rd_class_kernel(ClassArgs a) {
  constexpr int NT = HUGE ? 256 : 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int ld = a.ld;
  // LDS carve-up: [Gm Vm (LDS variant)] inv_s2[ld] vrow[ld] red[8] rdh[RD_BINS] loc[n_cap] memb[n_cap]
  double* fp = reinterpret_cast<double*>(smem);
  double* Gm; double* Vm;
  if (HUGE) {
    Gm = a.slabs + (size_t)blockIdx.x * 2 * ld * ld; Vm = Gm + (size_t)ld * ld;
  } else {
    Gm = fp; Vm = Gm + (size_t)ld * ld; fp = Vm + (size_t)ld * ld;
  }
  double* inv_s2 = fp;
  double* vrow = inv_s2 + ld;
  double* red = vrow + ld;
  int* rdh = reinterpret_cast<int*>(red + 8);
  short* loc = reinterpret_cast<short*>(rdh + RD_BINS);
  unsigned char* memb = reinterpret_cast<unsigned char*>(loc + a.n_cap);
  __shared__ int bad_s;

  const int n_cls = a.cls_count[a.bucket];
  for (int c = blockIdx.x; c < n_cls; c += gridDim.x) {
    const int64_t k_rep = a.cls_list[(int64_t)a.bucket * a.cap + c];
    const int g = find_segment(a.out_edge_ptr, a.G, k_rep);
    const int64_t n0 = a.node_ptr[g];
    const int n = (int)(a.node_ptr[g + 1] - n0);
    const int Eg = a.e_out[g];
    const int64_t wo = a.edge_ptr[g] + n0;
    const int* __restrict__ es = a.w_src + wo;
    const int* __restrict__ ed = a.w_dst + wo;
    const int* __restrict__ cn = a.canon + n0;
    int ca, cb;
    {
      const int k = (int)(k_rep - a.out_edge_ptr[g]);
      const int cu = cn[es[k]], cv = cn[ed[k]];
      ca = cu < cv ? cu : cv; cb = cu < cv ? cv : cu;
    }
    const unsigned char* __restrict__ ha = a.hop_tab + a.sq_ptr[g] + (int64_t)ca * n;
    const unsigned char* __restrict__ hb = a.hop_tab + a.sq_ptr[g] + (int64_t)cb * n;
    __syncthreads();                                          // previous class is done with the LDS arrays
    if (tid == 0) bad_s = 0;
    // membership bits and local indices of S_a ∪ S_b (ascending node id; rd is order independent given the root)
    int m = 0;
    for (int x0 = 0; x0 < n; x0 += NT) {
      const int x = x0 + tid;
      int mb = 0;
      if (x < n) { mb = (ha[x] != HOP_INF ? 1 : 0) | (hb[x] != HOP_INF ? 2 : 0); memb[x] = (unsigned char)mb; }
      if (HUGE) {
        // ranks across 4 waves: per-wave ballots exchanged through red[] (as ints)
        const unsigned long long msk = __ballot(mb != 0);
        int* wcnt = reinterpret_cast<int*>(red);
        __syncthreads();
        if ((tid & 63) == 0) wcnt[tid >> 6] = __popcll(msk);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < NT / 64; ++w) { if (w < (tid >> 6)) before += wcnt[w]; all += wcnt[w]; }
        if (x < n) loc[x] = mb ? (short)(m + before + __popcll(msk & ((1ull << (tid & 63)) - 1ull))) : (short)-1;
        m += all;
      } else {
        const unsigned long long msk = __ballot(mb != 0);
        if (x < n) loc[x] = mb ? (short)(m + __popcll(msk & ((1ull << tid) - 1ull))) : (short)-1;
        m += __popcll(msk);
      }
    }
    for (int idx = tid; idx < m * m; idx += NT) {
      const int cc = idx / m, rr = idx % m;
      Gm[(size_t)cc * ld + rr] = 0.0;
      Vm[(size_t)cc * ld + rr] = (cc == rr) ? 1.0 : 0.0;
    }
    __syncthreads();
    // L = D_in - A over the non-loop sub-edges (union of the two induced edge sets)
    for (int j = tid; j < Eg; j += NT) {
      const int s = es[j], t = ed[j];
      if (s == t) continue;
      if (!(memb[s] & memb[t])) continue;                     // both in S_a, or both in S_b
      const int ls = loc[s], lt = loc[t];
      atomicAdd(&Gm[(size_t)lt * ld + ls], -1.0);             // L[s][t] -= 1   (column-major: [col t][row s])
      atomicAdd(&Gm[(size_t)lt * ld + lt], 1.0);              // L[t][t] += 1   (in-degree, loops excluded)
    }
    __syncthreads();
    const double null2 = jacobi_orthogonalise<NT>(Gm, Vm, m, ld, red);
    __syncthreads();
    // squared singular values = squared column norms; numerically-null ones are dropped (pinv cutoff)
    for (int j = tid; j < m; j += NT) {
      double s2 = 0.0;
      for (int i = 0; i < m; ++i) { const double t = Gm[(size_t)j * ld + i]; s2 += t * t; }
      inv_s2[j] = (s2 > null2) ? 1.0 / s2 : 0.0;
    }
    __syncthreads();
    // P = V diag(1/s^2) G^T, written over V row by row: P[i][j'] = sum_j V[i][j] G[j'][j] / s_j^2
    for (int i = 0; i < m; ++i) {
      for (int j = tid; j < m; j += NT) vrow[j] = Vm[(size_t)j * ld + i] * inv_s2[j];
      __syncthreads();
      for (int jp = tid; jp < m; jp += NT) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc += vrow[j] * Gm[(size_t)j * ld + jp];
        Vm[(size_t)jp * ld + i] = acc;
      }
      __syncthreads();
    }
    const double* __restrict__ P = Vm;                        // P[i][j] at P[j*ld + i]
    // member edges of this class, each against its own root
    for (int k0 = 0; k0 < Eg; k0 += 64) {
      unsigned long long members;
      {
        const int kk = k0 + (tid & 63);
        bool mine = false;
        if (kk < Eg) {
          const int cu = cn[es[kk]], cv = cn[ed[kk]];
          mine = (cu < cv ? cu : cv) == ca && (cu < cv ? cv : cu) == cb;
        }
        members = __ballot(mine);                             // identical in every wave of the workgroup
      }
      while (members) {
        const int kk = k0 + __ffsll((long long)members) - 1;
        members &= members - 1;
        const int u = es[kk], v = ed[kk];
        const bool phantom = (u == v);
        const int r = loc[u];
        for (int b = tid; b < RD_BINS; b += NT) rdh[b] = 0;
        __syncthreads();
        const double prr = phantom ? 0.0 : P[(size_t)r * ld + r];
        for (int x = tid; x < n; x += NT) {
          const int i = loc[x];
          if (i < 0) continue;
          const double pii = P[(size_t)i * ld + i];
          const double rd64 = phantom ? pii : (((prr + pii) - P[(size_t)i * ld + r]) - P[(size_t)r * ld + i]);
          const float rd32 = (float)rd64;
          if (!(rd32 > -1.0f && rd32 < 100.0f)) { bad_s = 1; continue; }
          atomicAdd(&rdh[(int)rd32], 1);                      // (int) truncates toward zero like .long()
        }
        __syncthreads();
        short* __restrict__ row = a.rd_rows + (a.out_edge_ptr[g] + kk) * RD_BINS;
        for (int b = tid; b < RD_BINS; b += NT) row[b] = (short)rdh[b];
        __syncthreads();
      }
    }
    if (tid == 0 && bad_s) a.status[g] = ESC_ERANGE;
  }
}

// ---- pass 1/2: encode every output edge (one wave per edge) ---------------------------------------
struct EncodeArgs {
  const int64_t* node_ptr; const int64_t* edge_ptr; const int64_t* sq_ptr;
  const int* w_src; const int* w_dst; const int* e_out;
  const int64_t* out_edge_ptr;   // [G+1]
  const unsigned char* hop_tab;
  const short* rd_rows;
  int G, h, use_rd, n_cap;
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
  const int g = find_segment(a.out_edge_ptr, a.G, k_glob);
  const int k = (int)(k_glob - a.out_edge_ptr[g]);
  const int n = (int)(a.node_ptr[g + 1] - a.node_ptr[g]);
  const int Eg = a.e_out[g];
  const int64_t wo = a.edge_ptr[g] + a.node_ptr[g];
  const int* __restrict__ es = a.w_src + wo;
  const int* __restrict__ ed = a.w_dst + wo;
  const int far = a.h + 1;

  int* hist = reinterpret_cast<int*>(smem);                         // [1800]
  int* deg = hist + HIST_BINS;                                      // [n_cap]
  unsigned char* hop_u = reinterpret_cast<unsigned char*>(deg + a.n_cap);   // [n_cap]
  unsigned char* hop_v = hop_u + a.n_cap;                           // [n_cap]

  const int u = es[k], v = ed[k];
  const unsigned char* __restrict__ row_u = a.hop_tab + a.sq_ptr[g] + (int64_t)u * n;
  const unsigned char* __restrict__ row_v = a.hop_tab + a.sq_ptr[g] + (int64_t)v * n;
  for (int i = lane; i < HIST_BINS; i += 64) hist[i] = 0;
  for (int x = lane; x < n; x += 64) { hop_u[x] = row_u[x]; hop_v[x] = row_v[x]; deg[x] = 0; }
  __syncthreads();
  const bool do_rd = a.use_rd != 0;
  bool bad = false;
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
  if (do_rd) {
    const short* __restrict__ row = a.rd_rows + k_glob * RD_BINS;
    for (int b = lane; b < RD_BINS; b += 64) {
      const int cnt = row[b];
      if (cnt) atomicAdd(&hist[400 + b], cnt);
    }
  }
  if (u == v && lane == 0) {                                // the phantom duplicate of the root
    atomicAdd(&hist[0], 1); atomicAdd(&hist[200], 1); atomicAdd(&hist[300], 1);
    if (do_rd) atomicAdd(&hist[400], 1);                    // rd of the isolated duplicate = 0
  }
  bad = __any(bad);
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

// ---- layout of `work` (the caller's scratch, shared by the count and the fill pass) -------------------------
struct WorkLayout {
  int* w_src; int* w_dst; int* w_in; int* e_out; int* nnz_cnt;
  int64_t* sq_ptr; unsigned long long* root_hash; int* canon; int* cls_count; int* cls_list;
  unsigned* cls_flag; int64_t flag_words; unsigned char* hop_tab; short* rd_rows; double* slabs;
  int n_slabs; int ld_huge;
  int64_t bytes;
};
static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static WorkLayout carve(void* work, int64_t G, int64_t total_nodes, int64_t total_in_edges, int64_t sum_sq,
                        int64_t max_nodes, int use_rd) {
  const int64_t cap = total_in_edges + total_nodes;
  unsigned char* base = reinterpret_cast<unsigned char*>(work);
  int64_t off = 0;
  auto take = [&](int64_t bytes) { unsigned char* p = base + off; off = align_up(off + bytes, 16); return p; };
  WorkLayout w{};
  w.w_src = (int*)take(cap * 4); w.w_dst = (int*)take(cap * 4); w.w_in = (int*)take(cap * 4);
  w.e_out = (int*)take(G * 4); w.nnz_cnt = (int*)take((cap + 1) * 4);
  w.sq_ptr = (int64_t*)take((G + 1) * 8);
  w.root_hash = (unsigned long long*)take(total_nodes * 8);
  w.canon = (int*)take(total_nodes * 4);
  w.hop_tab = take(sum_sq);
  if (use_rd) {
    w.cls_count = (int*)take(N_BUCKETS * 4);
    w.cls_list = (int*)take(N_BUCKETS * cap * 4);
    w.flag_words = (sum_sq + 31) / 32;
    w.cls_flag = (unsigned*)take(w.flag_words * 4);
    w.rd_rows = (short*)take(cap * RD_BINS * 2);
    if (max_nodes > LDS_SUBGRAPH) {
      w.ld_huge = (int)((max_nodes + 1) & ~1LL);
      const int64_t slab = (int64_t)2 * w.ld_huge * w.ld_huge * 8;
      int64_t ns = (1LL << 30) / slab;
      w.n_slabs = (int)(ns < 1 ? 1 : ns > 64 ? 64 : ns);
      w.slabs = (double*)take(slab * w.n_slabs);
    }
  }
  w.bytes = off;
  return w;
}

static size_t encode_lds_bytes(int n_cap) { return (size_t)HIST_BINS * 4 + (size_t)n_cap * 4 + (size_t)n_cap * 2; }
static size_t class_lds_bytes(int ld, int n_cap, bool huge) {
  size_t b = ((size_t)2 * ld + 8) * 8 + (size_t)RD_BINS * 4 + (size_t)n_cap * 3;
  if (!huge) b += (size_t)2 * ld * ld * 8;
  return (b + 15) & ~(size_t)15;
}

template <class K>
static void allow_lds(K kernel, size_t lds) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace esc

using namespace esc;

extern "C" {

int64_t esc_features_scratch_bytes(int64_t G, int64_t total_nodes, int64_t total_in_edges, int64_t sum_nodes_sq,
                                   int64_t max_nodes, int use_rd) {
  if (G < 0 || total_nodes < 0 || total_in_edges < 0 || sum_nodes_sq < 0 || max_nodes < 0) return -1;
  return carve(nullptr, G, total_nodes, total_in_edges, sum_nodes_sq, max_nodes, use_rd).bytes + 64;
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

static EncodeArgs encode_args(const WorkLayout& w, const int64_t* node_ptr, const int64_t* edge_ptr,
                              const int64_t* out_edge_ptr, int64_t G, int64_t max_nodes, int h, int use_rd, int32_t* status) {
  EncodeArgs a{};
  a.node_ptr = node_ptr; a.edge_ptr = edge_ptr; a.sq_ptr = w.sq_ptr; a.w_src = w.w_src; a.w_dst = w.w_dst; a.e_out = w.e_out;
  a.out_edge_ptr = out_edge_ptr; a.hop_tab = w.hop_tab; a.rd_rows = w.rd_rows;
  a.G = (int)G; a.h = h; a.use_rd = use_rd; a.n_cap = (int)((max_nodes + 3) & ~3LL);
  a.status = status;
  return a;
}

int esc_features_count(const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* src,
                       const int64_t* dst, int64_t G, int64_t total_nodes, int64_t total_in_edges,
                       int64_t sum_nodes_sq, int64_t max_nodes, int h, int use_rd, int self_loop,
                       int64_t* out_edge_ptr, int64_t* nnz_ptr, int32_t* status, void* work, void* stream) {
  int rc = check_common(node_ptr, edge_ptr, G, h, max_nodes);
  if (rc) return rc;
  ESC_REQUIRE((src && dst) || total_in_edges == 0, "esc_features_count: null edge arrays");
  ESC_REQUIRE(out_edge_ptr && nnz_ptr && status && work, "esc_features_count: null output");
  ESC_REQUIRE(total_nodes >= 0 && total_in_edges >= 0 && total_nodes + total_in_edges < (1LL << 31) - 1,
              "esc_features_count: too many nodes+edges in one call");
  ESC_REQUIRE(sum_nodes_sq >= total_nodes && sum_nodes_sq <= total_nodes * max_nodes,
              "esc_features_count: sum_nodes_sq=%ld is not the sum of squared graph sizes", (long)sum_nodes_sq);
  hipStream_t s = (hipStream_t)stream;
  const WorkLayout w = carve(work, G, total_nodes, total_in_edges, sum_nodes_sq, max_nodes, use_rd);
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
  const int64_t cap = total_in_edges + total_nodes;
  const int n_cap = (int)((max_nodes + 3) & ~3LL);
  // hop tables and canonical roots
  esc::launch(ESC_K_FEATURES, feat_sq_scan_kernel, dim3(1), dim3(1024), 0, s, node_ptr, G, w.sq_ptr);
  esc::launch(ESC_K_FEATURES, feat_bfs_kernel, dim3((unsigned)total_nodes), dim3(64), (size_t)n_cap, s, node_ptr, edge_ptr,
              (const int*)w.w_src, (const int*)w.w_dst, (const int*)w.e_out, (const int64_t*)w.sq_ptr, (int)G, h, total_nodes,
              w.hop_tab, w.root_hash);
  ESC_CHECK_LAUNCH("esc_features_count.bfs");
  if (use_rd) {
    esc::launch(ESC_K_FEATURES, feat_canon_kernel, dim3((unsigned)total_nodes), dim3(64), 0, s, node_ptr,
                (const int64_t*)w.sq_ptr, (int)G, total_nodes, (const unsigned char*)w.hop_tab,
                (const unsigned long long*)w.root_hash, w.canon);
    ESC_CHECK_LAUNCH("esc_features_count.canon");
    if (hipMemsetAsync(w.cls_count, 0, sizeof(int) * N_BUCKETS, s) != hipSuccess ||
        hipMemsetAsync(w.cls_flag, 0, sizeof(unsigned) * (size_t)w.flag_words, s) != hipSuccess ||
        hipMemsetAsync(w.rd_rows, 0, sizeof(short) * (size_t)cap * RD_BINS, s) != hipSuccess) {
      set_error("esc_features_count: memset failed");
      return ESC_ELAUNCH;
    }
    ClassArgs c{};
    c.node_ptr = node_ptr; c.edge_ptr = edge_ptr; c.sq_ptr = w.sq_ptr; c.out_edge_ptr = out_edge_ptr;
    c.w_src = w.w_src; c.w_dst = w.w_dst; c.e_out = w.e_out; c.canon = w.canon; c.hop_tab = w.hop_tab;
    c.cls_flag = w.cls_flag; c.cls_count = w.cls_count; c.cls_list = w.cls_list; c.rd_rows = w.rd_rows; c.slabs = w.slabs;
    c.status = status; c.cap = cap; c.G = (int)G; c.n_cap = n_cap;
    esc::launch(ESC_K_FEATURES, feat_class_mark_kernel, dim3((unsigned)cap_edges), dim3(64), 0, s, c);
    ESC_CHECK_LAUNCH("esc_features_count.class_mark");
    const int m_even = (int)((max_nodes + 1) & ~1LL);
    const int lds_ld[3] = {m_even < 32 ? m_even : 32, 64, LDS_SUBGRAPH};
    for (int b = 0; b < 3; ++b) {
      if (b > 0 && max_nodes <= (b == 1 ? 32 : 64)) break;           // no ego-net can land in this bucket
      c.bucket = b; c.ld = lds_ld[b];
      const size_t lds = class_lds_bytes(c.ld, n_cap, false);
      ESC_REQUIRE(lds <= 160 * 1024, "esc_features_count: class working set %zu B exceeds LDS", lds);
      allow_lds(feat_rd_class_kernel<false>, lds);
      const int64_t per_cu = (160 * 1024) / (int64_t)lds;
      int64_t grid = 256 * (per_cu < 1 ? 1 : per_cu > 16 ? 16 : per_cu);
      if (grid > cap_edges) grid = cap_edges;
      esc::launch(ESC_K_FEATURES, feat_rd_class_kernel<false>, dim3((unsigned)grid), dim3(64), lds, s, c);
      ESC_CHECK_LAUNCH("esc_features_count.rd_class");
    }
    if (max_nodes > LDS_SUBGRAPH) {
      c.bucket = 3; c.ld = w.ld_huge;
      const size_t lds = class_lds_bytes(c.ld, n_cap, true);
      ESC_REQUIRE(lds <= 160 * 1024, "esc_features_count: class working set %zu B exceeds LDS", lds);
      allow_lds(feat_rd_class_kernel<true>, lds);
      esc::launch(ESC_K_FEATURES, feat_rd_class_kernel<true>, dim3((unsigned)w.n_slabs), dim3(256), lds, s, c);
      ESC_CHECK_LAUNCH("esc_features_count.rd_class_huge");
    }
  }
  EncodeArgs a = encode_args(w, node_ptr, edge_ptr, out_edge_ptr, G, max_nodes, h, use_rd, status);
  a.nnz_cnt = w.nnz_cnt;
  (void)hipMemsetAsync(w.nnz_cnt, 0, sizeof(int) * (size_t)cap_edges, s);
  const size_t lds = encode_lds_bytes(a.n_cap);
  allow_lds(feat_encode_kernel<false>, lds);
  esc::launch(ESC_K_FEATURES, feat_encode_kernel<false>, dim3((unsigned)cap_edges), dim3(64), lds, s, a);
  ESC_CHECK_LAUNCH("esc_features_count.encode");
  esc::launch(ESC_K_FEATURES, scan_counts_kernel, dim3(1), dim3(1024), 0, s, w.nnz_cnt, cap_edges, nnz_ptr);
  ESC_CHECK_LAUNCH("esc_features_count.scan_nnz");
  return ESC_OK;
}

int esc_features_fill(const int64_t* node_ptr, const int64_t* edge_ptr, int64_t G, int64_t total_nodes,
                      int64_t total_in_edges, int64_t sum_nodes_sq, int64_t max_nodes, int h, int use_rd, int self_loop,
                      const int64_t* out_edge_ptr, const int64_t* nnz_ptr, int64_t total_out_edges,
                      int64_t* out_src, int64_t* out_dst, int64_t* in_edge_of_out, int64_t* pos_enc,
                      int64_t* pos_index, int64_t* pos_batch, int32_t* status, void* work, void* stream) {
  int rc = check_common(node_ptr, edge_ptr, G, h, max_nodes);
  if (rc) return rc;
  (void)self_loop;
  ESC_REQUIRE(out_edge_ptr && nnz_ptr && status && work, "esc_features_fill: null pointer");
  ESC_REQUIRE(total_out_edges >= 0, "esc_features_fill: bad edge total");
  if (total_out_edges == 0) return ESC_OK;
  ESC_REQUIRE(out_src && out_dst && pos_enc && pos_index && pos_batch, "esc_features_fill: null output");
  hipStream_t s = (hipStream_t)stream;
  const WorkLayout w = carve(work, G, total_nodes, total_in_edges, sum_nodes_sq, max_nodes, use_rd);
  esc::launch(ESC_K_FEATURES, feat_edges_out_kernel, dim3((unsigned)G), dim3(256), 0, s, node_ptr, edge_ptr, w.w_src, w.w_dst,
                     w.w_in, out_edge_ptr, (int)G, out_src, out_dst, in_edge_of_out);
  ESC_CHECK_LAUNCH("esc_features_fill.edges");
  EncodeArgs a = encode_args(w, node_ptr, edge_ptr, out_edge_ptr, G, max_nodes, h, use_rd, status);
  a.nnz_ptr = nnz_ptr; a.pos_enc = pos_enc; a.pos_index = pos_index; a.pos_batch = pos_batch;
  const size_t lds = encode_lds_bytes(a.n_cap);
  allow_lds(feat_encode_kernel<true>, lds);
  esc::launch(ESC_K_FEATURES, feat_encode_kernel<true>, dim3((unsigned)total_out_edges), dim3(64), lds, s, a);
  ESC_CHECK_LAUNCH("esc_features_fill.encode");
  return ESC_OK;
}

}  // extern "C"
