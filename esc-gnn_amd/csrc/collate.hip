// collate.hip — device-side Batch.from_data_list: gather B graphs out of the HBM-resident dataset
// store and emit (i) the reference's batch tensors, bit-identical to /root/reference/batch.py:25-149
// (edge_index += running node count, pos_batch += running edge count :70-71, pos_enc/pos_index
// unshifted :72-73, batch = graph id :120-123), and (ii) the compact int32 execution plan (CSR by
// destination / source, bag rows, bag columns) by offsetting per-graph views that were sorted once
// when the store was built.  Pure HBM-bound gather/offset work: one pass over the batch's bytes.
#include "common.h"

namespace esc {

// per-column running counts over the batch's graphs: prefix[b][c] = #entries of column c in graphs
// 0..b-1 of the batch, total[c] = over all graphs.  One wave per column: lanes take 64 graphs at a
// time and a wave prefix scan replaces the serial dependent-load chain.
__global__ __launch_bounds__(256) void collate_col_count_kernel(const int* __restrict__ col_cnt_all, int n_cols,
                                                                const int64_t* __restrict__ graph_ids, int B,
                                                                int* __restrict__ prefix, int* __restrict__ total) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= n_cols) return;
  const int lane = lane_id();
  int run = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + lane;
    const int v = (b < B) ? col_cnt_all[(size_t)graph_ids[b] * n_cols + c] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (b < B) prefix[(size_t)b * n_cols + c] = run + incl - v;
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) total[c] = run;
}

// exclusive scan of n (<= a few thousand) ints by one workgroup -> out[n+1]
__global__ __launch_bounds__(1024) void small_scan_kernel(const int* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
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
  if (threadIdx.x == 0) out[n] = carry_s;
}

__global__ __launch_bounds__(256) void collate_fill_kernel(esc_collate_args a) {
  const int b = blockIdx.x;
  const int part = blockIdx.y, parts = gridDim.y;
  const int64_t g = a.graph_ids[b];
  const int64_t n0 = a.node_ptr[g], n_g = a.node_ptr[g + 1] - n0;
  const int64_t e0 = a.edge_ptr[g], e_g = a.edge_ptr[g + 1] - e0;
  const int64_t z0 = a.nnz_ptr[g], z_g = a.nnz_ptr[g + 1] - z0;
  const int64_t y0 = a.y_ptr[g], y_g = a.y_ptr[g + 1] - y0;
  const int64_t no = a.offsets[b], eo = a.offsets[(a.B + 1) + b], zo = a.offsets[2 * (a.B + 1) + b];
  const int64_t yo = a.offsets[3 * (a.B + 1) + b];
  const int64_t Nt = a.offsets[a.B], Et = a.offsets[(a.B + 1) + a.B], Zt = a.offsets[2 * (a.B + 1) + a.B];
  const int tid = part * blockDim.x + threadIdx.x;
  const int nthreads = parts * blockDim.x;

  // ---- nodes ----
  for (int64_t i = tid; i < n_g; i += nthreads) {
    a.batch[no + i] = b;
    a.in_ptr[no + i] = (int)(a.in_ptr_all[n0 + i] - e0 + eo);
    a.out_ptr[no + i] = (int)(a.out_ptr_all[n0 + i] - e0 + eo);
  }
  if (a.x_long) {                                      // categorical features: exact in fp32, handed out as int64
    for (int64_t i = tid; i < n_g * a.x_dim; i += nthreads) a.x_long[no * a.x_dim + i] = (int64_t)a.x_all[n0 * a.x_dim + i];
  } else {
    for (int64_t i = tid; i < n_g * a.x_dim; i += nthreads) a.x[no * a.x_dim + i] = a.x_all[n0 * a.x_dim + i];
  }
  if (a.graph_ptr && tid == 0) {
    a.graph_ptr[b] = (int)no;
    if (b == a.B - 1) a.graph_ptr[a.B] = (int)Nt;
  }
  for (int64_t i = tid; i < y_g * a.y_dim; i += nthreads) a.y[yo * a.y_dim + i] = a.y_all[y0 * a.y_dim + i];
  if (b == a.B - 1 && tid == 0) {
    a.in_ptr[Nt] = (int)Et; a.out_ptr[Nt] = (int)Et; a.row_ptr[Et] = (int)Zt;
  }
  // ---- edges ----
  for (int64_t k = tid; k < e_g; k += nthreads) {
    const int64_t s = a.esrc_all[e0 + k], d = a.edst_all[e0 + k];
    a.edge_index[eo + k] = s + no;
    a.edge_index[Et + eo + k] = d + no;
    const int64_t ki = a.in_edge_all[e0 + k];          // store-global edge id, dst-sorted order
    a.in_edge[eo + k] = (int)(ki - e0 + eo);
    a.in_src[eo + k] = (int)(a.esrc_all[ki] + no);
    const int64_t ko = a.out_edge_all[e0 + k];
    a.out_edge[eo + k] = (int)(ko - e0 + eo);
    a.out_dst[eo + k] = (int)(a.edst_all[ko] + no);
    a.row_ptr[eo + k] = (int)(a.row_ptr_all[e0 + k] - z0 + zo);
  }
  if (a.edge_attr) {                                   // attribute rows travel with their edges, word by word
    const uint32_t* __restrict__ src = static_cast<const uint32_t*>(a.edge_attr_all) + e0 * a.ea_words;
    uint32_t* __restrict__ dst = static_cast<uint32_t*>(a.edge_attr) + eo * a.ea_words;
    for (int64_t i = tid; i < e_g * a.ea_words; i += nthreads) dst[i] = src[i];
  }
  // ---- bag entries: reference tensors + compact row view + column (CSC) view ----
  for (int64_t j = tid; j < z_g; j += nthreads) {
    const int64_t v = a.pos_enc_all[z0 + j], c = a.pos_index_all[z0 + j];
    a.pos_enc[zo + j] = v;
    a.pos_index[zo + j] = c;
    a.pos_batch[zo + j] = a.pos_batch_all[z0 + j] + eo;
    a.bag_idx[zo + j] = (int)c;
    a.bag_val[zo + j] = (int)v;
    const int64_t e = a.c_perm_all[z0 + j];             // store-global entry id, (graph, column)-sorted order
    const int cc = (int)a.pos_index_all[e];
    const int dest = a.col_ptr[cc] + a.col_prefix[(size_t)b * a.n_cols + cc] + a.c_rank_all[z0 + j];
    a.col_row[dest] = (int)(a.pos_batch_all[e] + eo);
    a.col_val[dest] = (int)a.pos_enc_all[e];
    a.col_col[dest] = cc;
  }
}

}  // namespace esc

using namespace esc;

extern "C" {

int esc_collate_cols(const int32_t* col_cnt_all, int64_t n_cols, const int64_t* graph_ids, int64_t B,
                     int32_t* col_prefix, int32_t* col_total, int32_t* col_ptr, void* stream) {
  ESC_REQUIRE(col_cnt_all && graph_ids && col_prefix && col_total && col_ptr, "esc_collate_cols: null pointer");
  ESC_REQUIRE(n_cols > 0 && B > 0 && B < (1 << 24), "esc_collate_cols: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  esc::launch(ESC_K_COLLATE, collate_col_count_kernel, dim3((unsigned)cdiv(n_cols, 4)), dim3(256), 0, s, col_cnt_all, (int)n_cols, graph_ids, (int)B, col_prefix, col_total);
  ESC_CHECK_LAUNCH("esc_collate_cols.count");
  esc::launch(ESC_K_COLLATE, small_scan_kernel, dim3(1), dim3(1024), 0, s, col_total, (int)n_cols, col_ptr);
  ESC_CHECK_LAUNCH("esc_collate_cols.scan");
  return ESC_OK;
}

int esc_collate_fill(const esc_collate_args* args, void* stream) {
  ESC_REQUIRE(args, "esc_collate_fill: null args");
  const esc_collate_args& a = *args;
  ESC_REQUIRE(a.B > 0 && a.x_dim >= 0 && a.y_dim >= 0 && a.n_cols > 0, "esc_collate_fill: bad sizes");
  ESC_REQUIRE(a.graph_ids && a.offsets && a.node_ptr && a.edge_ptr && a.nnz_ptr && a.y_ptr, "esc_collate_fill: null index arrays");
  ESC_REQUIRE(a.batch && a.edge_index && a.in_ptr && a.out_ptr && a.row_ptr, "esc_collate_fill: null outputs");
  ESC_REQUIRE((a.x || a.x_long || a.x_dim == 0) && (!a.edge_attr || (a.edge_attr_all && a.ea_words > 0)), "esc_collate_fill: bad optional outputs");
  hipStream_t s = (hipStream_t)stream;
  esc::launch(ESC_K_COLLATE, collate_fill_kernel, dim3((unsigned)a.B, 8), dim3(256), 0, s, a);
  ESC_CHECK_LAUNCH("esc_collate_fill");
  return ESC_OK;
}

}  // extern "C"
