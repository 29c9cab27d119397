// plan.hip — execution-plan builder for a FOREIGN batch (one that did not come from the device collate): the stable
// grouping of positions by key that turns edge_index / pos_index into the CSR / CSC views the kernels walk.
//
// The reference never builds these (PyG scatters by edge_index[1] in edge order, run_graphcount.py:161,169); the fast
// path gets them from esc_collate_fill.  For the drop-in DataLoader path and modules/gine_operations.py (one plan per
// distance class) they used to be derived with torch.sort / bincount / cumsum (rocPRIM radix sorts of int64 keys, ~10
// launches per call); here: a stable LSD radix sort of the POSITIONS by 8-bit digits of the key —
//   per pass   block histogram -> exclusive scan over (digit, block) -> one wave per block scatters its chunk in order
//              (rank inside a 64-element group by ballot matching, running per-digit counters in LDS)
// — 1 pass for keys < 256, 2 for < 65 536 (nodes, histogram bins), 3 for < 2^24; then the segment pointers from an
// integer histogram + scan.  Integer work, bit-exact by construction (stable = ascending position inside a key).
#include "common.h"

namespace esc {

constexpr int PLAN_CHUNK = 2048;      // positions per block and pass

__device__ __forceinline__ int digit_of(const int64_t* __restrict__ key, const int* __restrict__ idx, int64_t i, int shift) {
  const int64_t pos = idx ? idx[i] : i;
  return (int)((key[pos] >> shift) & 255);
}

__global__ __launch_bounds__(256) void plan_hist_kernel(const int64_t* __restrict__ key, const int* __restrict__ idx,
                                                        int64_t n, int shift, int nblk, int* __restrict__ hist) {
  __shared__ int cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * PLAN_CHUNK;
  for (int t = threadIdx.x; t < PLAN_CHUNK; t += 256) {
    const int64_t i = base + t;
    if (i < n) atomicAdd(&cnt[digit_of(key, idx, i, shift)], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];      // [digit][block]
}

// exclusive scan of n ints by ONE workgroup (n up to a few 10^5: 256 digits x blocks, or the key counts) -> out[n] (+ total
// at out[n] when with_total)
__global__ __launch_bounds__(1024) void plan_scan_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ out,
                                                         int with_total) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = (i < n) ? in[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const int carry = carry_s;
    if (i < n) out[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (with_total && threadIdx.x == 0) out[n] = carry_s;
}

// one wave per block walks its chunk IN ORDER, 64 positions at a time: lanes with equal digits are matched by 8 ballots,
// the rank inside the group is the number of matching lower lanes, the group's first lane of each digit advances the
// running counter — stable by construction
__global__ __launch_bounds__(64) void plan_scatter_kernel(const int64_t* __restrict__ key, const int* __restrict__ idx_in,
                                                          int64_t n, int shift, int nblk, const int* __restrict__ offs,
                                                          int* __restrict__ idx_out) {
  __shared__ int run[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) run[d] = offs[(size_t)d * nblk + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * PLAN_CHUNK;
  for (int g = 0; g < PLAN_CHUNK / 64; ++g) {
    const int64_t i = base + g * 64 + lane;
    const bool live = i < n;
    const int pos = live ? (idx_in ? idx_in[i] : (int)i) : 0;
    const int d = live ? (int)((key[pos] >> shift) & 255) : -1;
    unsigned long long same = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? m : ~m;
    }
    const unsigned long long lower = same & ((1ull << lane) - 1ull);
    const int rank = __popcll(lower);
    int dst = 0;
    if (live) dst = run[d] + rank;
    __syncthreads();                                   // every lane has read the counters of this group
    if (live && lower == 0ull) run[d] += __popcll(same);
    __syncthreads();
    if (live) idx_out[dst] = pos;
    if (base + (g + 1) * 64 >= n) break;               // wave-uniform
  }
}

__global__ __launch_bounds__(256) void plan_iota_kernel(int* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int)i;
}

__global__ __launch_bounds__(256) void plan_count_kernel(const int64_t* __restrict__ key, int64_t n, int64_t n_keys,
                                                         int* __restrict__ counts, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = key[i];
  if (k < 0 || k >= n_keys) { *bad = 1; return; }
  atomicAdd(&counts[k], 1);
}

// few keys (embedding tables, graph ids): thousands of positions hit the same counters — count a chunk in LDS first and
// add each non-zero bin once (integer adds: the result does not depend on the order)
constexpr int PLAN_SMALL_KEYS = 2048;
__global__ __launch_bounds__(256) void plan_count_small_kernel(const int64_t* __restrict__ key, int64_t n, int n_keys,
                                                               int* __restrict__ counts, int* __restrict__ bad) {
  __shared__ int cnt[PLAN_SMALL_KEYS];
  for (int k = threadIdx.x; k < n_keys; k += 256) cnt[k] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (PLAN_CHUNK * 4);
  for (int t = threadIdx.x; t < PLAN_CHUNK * 4; t += 256) {
    const int64_t i = base + t;
    if (i < n) {
      const int64_t k = key[i];
      if (k < 0 || k >= n_keys) *bad = 1; else atomicAdd(&cnt[(int)k], 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_keys; k += 256)
    if (cnt[k]) atomicAdd(&counts[k], cnt[k]);
}

// ---- sum-of-embeddings plan (AtomEncoder / BondEncoder: ogb_mol_gnn.py:264-282) ---------------------------------------
// index [n, k] (one id per feature column) -> keys into the k tables laid end to end: key = index + offset[col].  One launch
// writes the int64 keys the grouping reads, their int32 copy (the bag's index array), the all-ones weights and the row
// pointers 0, k, 2k, ...; ids outside their table raise the flag (and are clamped so that nothing downstream reads out of
// range).  A second launch turns the grouping permutation into the CSC view (c_row = entry / k, c_col = key).
struct EmbedDims { int k; int64_t dim[ESC_MAX_EMBED_COLS]; int64_t off[ESC_MAX_EMBED_COLS]; };

__global__ __launch_bounds__(256) void embed_keys_kernel(const int64_t* __restrict__ index, int64_t n, EmbedDims d,
                                                         int64_t* __restrict__ key, int* __restrict__ idx32,
                                                         int* __restrict__ ones, int* __restrict__ row_ptr,
                                                         int* __restrict__ bad) {
  const int64_t total = n * d.k;
  bool any_bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d.k);
    int64_t v = index[i];
    if (v < 0 || v >= d.dim[c]) { any_bad = true; v = v < 0 ? 0 : d.dim[c] - 1; }
    const int64_t kk = v + d.off[c];
    key[i] = kk;
    idx32[i] = (int)kk;
    ones[i] = 1;
    if (c == 0) row_ptr[i / d.k] = (int)i;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) row_ptr[n] = (int)total;
  if (any_bad) atomicOr(bad, 1);
}

// (perm and c_row may be the same buffer: entry i only reads perm[i] before it writes c_row[i])
__global__ __launch_bounds__(256) void embed_csc_kernel(const int* perm, const int* __restrict__ idx32, int64_t total,
                                                        int k, int* c_row, int* __restrict__ c_col) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = perm[i];
    c_row[i] = e / k;
    c_col[i] = idx32[e];
  }
}

}  // namespace esc

using namespace esc;

extern "C" {

int64_t esc_plan_csr_scratch(int64_t n, int64_t n_keys) {
  const int64_t nblk = cdiv(n > 0 ? n : 1, PLAN_CHUNK);
  return 2 * (n + 64) + 2 * (256 * nblk + 64) + (n_keys + 64) + 64;       // idx ping-pong, hist + offsets, counts, flag
}

int esc_plan_csr(const int64_t* key, int64_t n, int64_t n_keys, int32_t* ptr, int32_t* perm, int32_t* scratch,
                 int32_t* bad_flag, void* stream) {
  ESC_REQUIRE(ptr && scratch && bad_flag && (n == 0 || key), "esc_plan_csr: null pointer");
  // the radix passes sort on 24 key bits; segment pointers alone (perm == NULL: keys already grouped, e.g. row_ptr keyed on E) only
  // need the int32 counters to hold n_keys + 1 entries
  ESC_REQUIRE(n >= 0 && n < (1LL << 31) && n_keys > 0 && n_keys < (perm ? (1LL << 24) : (1LL << 31) - 64),
              "esc_plan_csr: bad sizes n=%ld n_keys=%ld", (long)n, (long)n_keys);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)cdiv(n > 0 ? n : 1, PLAN_CHUNK);
  int* idx_a = scratch;
  int* idx_b = idx_a + (n + 64);
  int* hist = idx_b + (n + 64);
  int* offs = hist + (256 * (int64_t)nblk + 64);
  int* counts = offs + (256 * (int64_t)nblk + 64);
  // segment pointers: integer histogram of the keys + exclusive scan (n_keys + 1 entries)
  if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(n_keys + 1), s) != hipSuccess || hipMemsetAsync(bad_flag, 0, sizeof(int), s) != hipSuccess) {
    set_error("esc_plan_csr: memset failed");
    return ESC_ELAUNCH;
  }
  if (n > 0 && n_keys <= PLAN_SMALL_KEYS)
    esc::launch(ESC_K_COLLATE, plan_count_small_kernel, dim3((unsigned)cdiv(n, PLAN_CHUNK * 4)), dim3(256), 0, s, key, n, (int)n_keys, counts, bad_flag);
  else if (n > 0)
    esc::launch(ESC_K_COLLATE, plan_count_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, key, n, n_keys, counts, bad_flag);
  esc::launch(ESC_K_COLLATE, plan_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)counts, n_keys, ptr, 1);
  ESC_CHECK_LAUNCH("esc_plan_csr.ptr");
  if (n == 0 || perm == nullptr) return ESC_OK;        // perm == NULL: segment pointers only (keys already grouped)
  int passes = 1;
  while (passes < 3 && (n_keys - 1) >> (8 * passes)) ++passes;
  const int* in = nullptr;                              // pass 0 reads the identity
  for (int p = 0; p < passes; ++p) {
    int* out = (p == passes - 1) ? perm : ((p & 1) ? idx_b : idx_a);
    esc::launch(ESC_K_COLLATE, plan_hist_kernel, dim3(nblk), dim3(256), 0, s, key, in, n, 8 * p, nblk, hist);
    esc::launch(ESC_K_COLLATE, plan_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)hist, (int64_t)256 * nblk, offs, 0);
    esc::launch(ESC_K_COLLATE, plan_scatter_kernel, dim3(nblk), dim3(64), 0, s, key, in, n, 8 * p, nblk, (const int*)offs, out);
    ESC_CHECK_LAUNCH("esc_plan_csr.pass");
    in = out;
  }
  return ESC_OK;
}

int64_t esc_embed_plan_scratch(int64_t n, int64_t k, int64_t rows) {       // int32 words: the int64 keys + the grouping's own scratch
  return 2 * (n * k + 8) + esc_plan_csr_scratch(n * k, rows > 0 ? rows : 1);
}

int esc_embed_plan(const int64_t* index, int64_t n, int64_t k, const int64_t* dims, int32_t* idx32, int32_t* ones,
                   int32_t* row_ptr, int32_t* col_ptr, int32_t* c_row, int32_t* c_col, int32_t* scratch, int32_t* bad_flag,
                   void* stream) {
  ESC_REQUIRE(dims && idx32 && ones && row_ptr && col_ptr && c_row && c_col && scratch && bad_flag && (n == 0 || index),
              "esc_embed_plan: null pointer");
  ESC_REQUIRE(n >= 0 && k > 0 && k <= ESC_MAX_EMBED_COLS && n * k < (1LL << 31) - 64, "esc_embed_plan: bad sizes n=%ld k=%ld", (long)n, (long)k);
  EmbedDims d;
  d.k = (int)k;
  int64_t rows = 0;
  for (int c = 0; c < ESC_MAX_EMBED_COLS; ++c) { d.dim[c] = 1; d.off[c] = 0; }
  for (int c = 0; c < (int)k; ++c) {
    ESC_REQUIRE(dims[c] > 0 && dims[c] < (1LL << 24), "esc_embed_plan: bad table size");
    d.dim[c] = dims[c]; d.off[c] = rows; rows += dims[c];
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = n * k;
  int64_t* key = reinterpret_cast<int64_t*>(scratch);                      // (the caller's int32 buffer is 8-byte aligned: see below)
  ESC_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "esc_embed_plan: scratch must be 8-byte aligned");
  int32_t* csr_scratch = scratch + 2 * (total + 8);
  if (hipMemsetAsync(bad_flag, 0, sizeof(int), s) != hipSuccess) { set_error("esc_embed_plan: memset failed"); return ESC_ELAUNCH; }
  const unsigned blocks = (unsigned)(cdiv(total > 0 ? total : 1, 256) < 2048 ? cdiv(total > 0 ? total : 1, 256) : 2048);
  esc::launch(ESC_K_COLLATE, embed_keys_kernel, dim3(blocks), dim3(256), 0, s, index, n, d, key, idx32, ones, row_ptr, bad_flag);
  ESC_CHECK_LAUNCH("esc_embed_plan.keys");
  // the grouping has its own flag word (it resets it): keys are in range by construction, the caller's flag keeps the id check
  int32_t* csr_flag = csr_scratch + esc_plan_csr_scratch(total, rows) - 8;
  const int rc = esc_plan_csr(key, total, rows, col_ptr, c_row /* the permutation, rewritten below */, csr_scratch, csr_flag, stream);
  if (rc != ESC_OK) return rc;
  if (total > 0) {
    // c_row currently holds the permutation; the CSC launch reads it through `perm` and overwrites c_row in place
    // element by element (entry i only depends on perm[i]) — a separate buffer is not needed
    esc::launch(ESC_K_COLLATE, embed_csc_kernel, dim3(blocks), dim3(256), 0, s, (const int*)c_row, (const int*)idx32, total, (int)k, c_row, c_col);
    ESC_CHECK_LAUNCH("esc_embed_plan.csc");
  }
  return ESC_OK;
}


}  // extern "C"
