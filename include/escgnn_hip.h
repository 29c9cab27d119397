/* escgnn_hip.h — C ABI of libescgnn_hip.so: the MI355X (gfx950) hot path of ESC-GNN's
 * NestedGIN_eff: ego-net structural-encoding feature build, collate, ESC bag, GINE aggregate,
 * fp32-MFMA linear layers, BatchNorm/ReLU, L1 loss and Adam.
 *
 * Plain pointers + sizes only (no torch types).  All pointers are DEVICE pointers unless a
 * parameter is documented "host".  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Every function returns 0 on success or a negative ESC_E* code; esc_last_error()
 * returns a human-readable message for the calling thread.  Nothing here allocates device
 * memory or synchronises the stream unless documented: the caller owns every buffer
 * (SURVEY.md §8(b) "Ownership"), so the calls are hipGraph-capturable.
 *
 * The reference has no FFI for this path (it is pure Python on PyG); each entry point names
 * the reference call site (file:line under /root/reference) whose device work it replaces.
 * Index arrays are int32 ("compact") unless stated; user-visible int64 tensors are produced
 * only by esc_collate_* / esc_features_fill so they stay bit-identical to the reference's.
 */
#ifndef ESCGNN_HIP_H
#define ESCGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESC_OK 0
#define ESC_EINVAL (-1)   /* bad argument (null pointer, negative size, misaligned, unsupported width) */
#define ESC_ELAUNCH (-2)  /* HIP runtime error at launch */
#define ESC_ERANGE (-3)   /* input outside the encodable range (degree>=200, rd bin>=100, h>4, n too large) */

int esc_abi_version(void);                /* bumps when a signature changes */
const char* esc_last_error(void);         /* message of the last failing call on this thread */

/* ---- profiling hook: HIP-event timing of one kernel family on its own launch stream ------ */
enum { ESC_K_AGG_FWD = 0, ESC_K_AGG_BWD = 1, ESC_K_BAG_FWD = 2, ESC_K_BAG_BWD = 3,
       ESC_K_LINEAR = 4, ESC_K_COLLATE = 5, ESC_K_FEATURES = 6, ESC_K_NORM = 7,
       ESC_K_GEMM_EDGE = 8,   /* the 128-row GEMM tiles (edge-row Linear layers); counted under ESC_K_LINEAR unless enabled itself */
       ESC_K_COUNT = 9 };
int esc_prof_enable(int kind, int on);    /* on!=0: bracket every launch of `kind` with events */
int esc_prof_read(int kind, int64_t* launches, double* total_ms);
/* per-launch durations (ms) in launch order; returns the number written (<= cap) */
int64_t esc_prof_read_all(int kind, double* ms_out, int64_t cap);  /* host; syncs recorded events */
int esc_prof_reset(int kind);
/* In-kernel execution windows, for kernels that take a `span` argument (the scatter-add, esc_gine_aggregate_fwd_affine): after
 * esc_prof_span_arm(kind, n, stream) the first n profiled launches of `kind` also stamp the device's wall clock — the earliest first
 * instruction of a workgroup, the latest last instruction of a wave — into per-launch slots (plain stores, n <= 4096); esc_prof_span_read
 * returns max - min per launch in microseconds (after the caller has synchronised).  An event pair (above) = inter-kernel dispatch gap + kernel; rocprofv3 =
 * the dispatch packet's begin -> end; this = first wave in -> last wave out.  Re-arm after esc_prof_reset. */
int esc_prof_span_arm(int kind, int64_t launches, void* stream);
int64_t esc_prof_span_read(int kind, double* us_out, int64_t cap);

/* ---- a-6 ESC bag: z[k,:] = sum_j val_j * W[idx_j,:]  (run_graphcount.py:155) -------------
 * forward order = entry order inside the row with separate mul/add roundings, i.e. bitwise what
 * a sequential scatter_add of (W[idx]*val) produces. */
int esc_bag_fwd(const float* table, int64_t H, const int32_t* row_ptr, const int32_t* idx32,
                const int32_t* val32, int64_t E, float* out, int64_t ld_out, void* stream);
/* The same sums with the table height given (r03): a workgroup owns 128 consecutive edges x a 64-column slice and stages the
 * table rows ITS edges use in LDS once (north_star's "LDS-staged feature tiles"), instead of pulling every entry's row
 * through the L2; bit-identical results.  accumulate != 0 adds onto `out` (esc_bag_fwd_acc).  stats (may be NULL, not with
 * accumulate): float2[ceil(E / B)][H] (mean, M2) of the output columns per block of B = esc_bag_fwd_stats_block_rows(...)
 * rows — the partials esc_bn_stats_from_partials_rows merges, so the BatchNorm behind the bag needs no pass over `out`
 * (run_graphcount.py:155-156: z_embedding starts with BatchNorm).  Shapes the tiled kernel does not serve (table higher
 * than 4096 rows, H % 4 != 0, fewer than 512 edges) take the wave-per-row kernel; stats then must be NULL
 * (esc_bag_fwd_stats_block_rows returns 0). */
int esc_bag_fwd_rows(const float* table, int64_t rows, int64_t H, const int32_t* row_ptr, const int32_t* idx32,
                     const int32_t* val32, int64_t E, float* out, int64_t ld_out, int accumulate, float* stats, void* stream);
int64_t esc_bag_fwd_stats_block_rows(const float* table, int64_t rows, int64_t H, const float* out, int64_t ld_out, int64_t E);
/* dTable[c,:] = sum_{j: idx_j=c} val_j * dZ[row_j,:] — deterministic two-pass segmented sum over
 * the CSC view; `partials` = esc_bag_bwd_scratch(Z,H) floats of scratch.  Writes ALL n_cols rows
 * (zeros where a column has no entry). */
int64_t esc_bag_bwd_scratch(int64_t Z, int64_t H);
int esc_bag_bwd_table(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                      const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                      int64_t n_cols, float* dtable, float* partials, void* stream);
/* the same with the row count of dz given: when dz exceeds one XCD's L2 the 64-entry chunks of pass 1 are bucketed by
 * the row eighth they touch and each bucket is served by workgroups that share an XCD (L2-local schedule; identical
 * results).  The bucketing depends on the index arrays only: esc_bag_bwd_classify may run it ahead of time into the
 * same scratch (classified = 1 then skips it).  Scratch as esc_bag_bwd_scratch. */
int esc_bag_bwd_classify(const int32_t* c_row, int64_t Z, int64_t H, int64_t rows, float* partials, void* stream);
int esc_bag_bwd_table_rows(const float* dz, int64_t ld_dz, int64_t H, const int32_t* col_ptr,
                           const int32_t* c_row, const int32_t* c_val, const int32_t* c_col, int64_t Z,
                           int64_t n_cols, int64_t rows, int classified, float* dtable, float* partials,
                           void* stream);

/* ---- a-8 GINE aggregate (PyG GINEConv propagate; run_graphcount.py:161,169;
 * semantics GraphGPS/graphgps/layer/gine_conv_layer.py:56-84) -------------------------------
 * out[i,:] = (1+eps)*x[i,:] + sum_{k in in(i), ascending k} relu(x[src_k,:] + e[k,:]).
 * in_ptr/in_edge/in_src = CSR by destination from esc_csr_build(key=dst, other=src).
 * eps: device scalar.  Leading dimensions in floats.  e == NULL: message = relu(x[src]); eps == NULL: no
 * self term (plain neighbour sum) — the two forms GINEPLUS needs (modules/gine_operations.py:335-362). */
int esc_gine_aggregate_fwd(const float* x, int64_t ld_x, const float* e, int64_t ld_e,
                           const int32_t* in_ptr, const int32_t* in_edge, const int32_t* in_src,
                           const float* eps, int64_t N, int64_t C, float* out, int64_t ld_out,
                           void* stream);
/* backward, CSR by source (out_ptr/out_edge/out_dst): d_e[k,:] = [x[src_k]+e_k > 0] * g[dst_k,:];
 * dx[i,:] = (1+eps)*g[i,:] + sum_{k in out(i)} d_e[k,:]  (dx may be NULL);
 * deps_part (may be NULL): per-row partial dot products sum_c g[i,c]*x[i,c] whose total is d(eps): N * S floats with
 * S = esc_gine_aggregate_bwd_deps_slots(C) (1; 2 when rows of C floats are split over two waves, ESC_AGG_SPLIT_BWD=2). */
int esc_gine_aggregate_bwd_deps_slots(int64_t C);
int esc_gine_aggregate_bwd(const float* x, int64_t ld_x, const float* e, int64_t ld_e,
                           const float* g, int64_t ld_g, const int32_t* out_ptr,
                           const int32_t* out_edge, const int32_t* out_dst, const float* eps,
                           int64_t N, int64_t C, float* d_e, int64_t ld_de, float* dx,
                           int64_t ld_dx, int accumulate_dx /* dx += instead of = */, float* deps_part,
                           void* stream);

/* esc_gine_aggregate_fwd / _bwd with the layer input given as PRE-activation rows of a BatchNorm+ReLU whose output is never
 * written: x' = relu(x*x_scale + x_shift) is applied to every x row as it is read (the step engine's node chain saves one
 * elementwise launch per layer); dx is the gradient with respect to x'.  C >= 64, multiples of 4, 16-byte aligned. */
int esc_gine_aggregate_fwd_affine(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* e,
                                  int64_t ld_e, const int32_t* in_ptr, const int32_t* in_edge, const int32_t* in_src,
                                  const float* eps, int64_t N, int64_t C, float* out, int64_t ld_out, void* stream);
int esc_gine_aggregate_bwd_affine(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* e,
                                  int64_t ld_e, const float* g, int64_t ld_g, const int32_t* out_ptr, const int32_t* out_edge,
                                  const int32_t* out_dst, const float* eps, int64_t N, int64_t C, float* d_e, int64_t ld_de,
                                  float* dx, int64_t ld_dx, int accumulate_dx, float* deps_part, void* stream);

/* esc_gine_aggregate_bwd_affine that also leaves the column sums of the BatchNorm(+ReLU) backward IN FRONT of it (r03): dx
 * (required) is then the complete gradient of x' = relu(BN(x)); bn_mean / bn_invstd are that BatchNorm's statistics and
 * partial[slot][C] (float2; one slot per 4 consecutive source rows: esc_gine_aggregate_bwd_stats_slots(N) slots) receives
 * (sum g, sum g*xhat), g = dx * [x' > 0] — the input of esc_bn_bwd_coef_from_partials.  One launch less on the node chain
 * per GINE layer (the partial-sum pass of the previous layer's last BatchNorm, run_graphcount.py:80-87 backward). */
int64_t esc_gine_aggregate_bwd_stats_slots(int64_t N);
int esc_gine_aggregate_bwd_affine_stats(const float* x, int64_t ld_x, const float* x_scale, const float* x_shift, const float* bn_mean,
                                        const float* bn_invstd, const float* e, int64_t ld_e, const float* g, int64_t ld_g,
                                        const int32_t* out_ptr, const int32_t* out_edge, const int32_t* out_dst, const float* eps,
                                        int64_t N, int64_t C, float* d_e, int64_t ld_de, float* dx, int64_t ld_dx, int accumulate_dx,
                                        float* deps_part, float* partial, void* stream);

/* graph readout (a-10): global_add_pool / global_mean_pool (run_graphcount.py:179; zinc_models.py:602) over the
 * sorted node->graph vector given as segment pointers seg_ptr[G+1]; rows summed in node order (bit-identical to
 * a sequential index_add_), mean divides by max(count,1).  Backward broadcasts g[graph]/count to the nodes. */
int esc_segment_pool_fwd(const float* x, int64_t ld_x, const int32_t* seg_ptr, int64_t G, int64_t C,
                         int mean, float* out, int64_t ld_out, void* stream);
int esc_segment_pool_bwd(const float* g, int64_t ld_g, const int32_t* seg_ptr, int64_t G, int64_t C,
                         int mean, float* dx, int64_t ld_dx, void* stream);

/* deterministic sum of n floats (fp64 accumulation) -> out[0]; finishes deps from deps_part. */
int esc_reduce_sum(const float* v, int64_t n, float* out, void* stream);
/* the same sum for up to many independent vectors in one launch per ESC_MAX_SUM_JOBS jobs (the eps gradients of
 * all GINE layers at the end of a backward pass). */
typedef struct esc_sum_job { const float* v; int64_t n; float* out; } esc_sum_job;
#define ESC_MAX_SUM_JOBS 16
int esc_reduce_sum_jobs(const esc_sum_job* jobs, int count, void* stream);

/* ---- execution plan of a foreign batch (plan.hip): stable grouping of positions by key = the CSR / CSC views ----------
 * ptr[n_keys+1] = segment pointers, perm[n] = positions sorted by key, ties in ascending position (what a stable
 * torch.sort + bincount + cumsum gave; reference call sites that these views serve: run_graphcount.py:155,161,169).
 * perm may be NULL (keys already grouped: pointers only).  *bad_flag (device int) is set when a key is outside
 * [0, n_keys).  scratch: esc_plan_csr_scratch(n, n_keys) int32. */
int64_t esc_plan_csr_scratch(int64_t n, int64_t n_keys);
int esc_plan_csr(const int64_t* key, int64_t n, int64_t n_keys, int32_t* ptr, int32_t* perm, int32_t* scratch,
                 int32_t* bad_flag, void* stream);
/* Bag plan of a sum-of-embeddings lookup (AtomEncoder / BondEncoder, /root/reference/ogb_mol_gnn.py:264-282 and ogb's
 * BondEncoder): index int64[n, k] holds one id per feature column, the k tables (dims[c] rows each) lie end to end.
 * Outputs (int32): idx32[n*k] = index + table offset, ones[n*k], row_ptr[n+1] = 0, k, 2k, ... (CSR by output row) and
 * col_ptr[sum(dims)+1], c_row[n*k], c_col[n*k] (CSC by table row, stable).  bad_flag[0] != 0 afterwards: an id lay
 * outside its table (it was clamped).  scratch: esc_embed_plan_scratch(n, k, sum(dims)) int32, 8-byte aligned.
 * 9 launches instead of the ~25 torch index launches of the per-op mirror. */
#define ESC_MAX_EMBED_COLS 16
int64_t esc_embed_plan_scratch(int64_t n, int64_t k, int64_t rows);
int esc_embed_plan(const int64_t* index, int64_t n, int64_t k, const int64_t* dims /* host, k entries */, int32_t* idx32,
                   int32_t* ones, int32_t* row_ptr, int32_t* col_ptr, int32_t* c_row, int32_t* c_col, int32_t* scratch,
                   int32_t* bad_flag, void* stream);

/* ---- a-7/a-8/a-9/a-10 dense layers on the matrix cores (exact-fp32 MFMA) -----------------
 * torch.nn.Linear call sites run_graphcount.py:54-121,183-189 (+ GINEConv.lin).
 * Y[M,N] = act(X)[M,K] * W[N,K]^T + bias[N]   (bias may be NULL)
 * act(X) = X, or relu(X*in_scale[k] + in_shift[k]) when in_scale != NULL (fused BN+ReLU of the
 * producer layer).  col_stats (may be NULL; needs N > 32): float2[ceil(M/R)][N] — per R-row block the (mean, M2)
 * of every output column, R = esc_linear_stats_block_rows(...), written by the GEMM epilogue so that the BatchNorm
 * that follows needs no pass over Y (finish with esc_bn_stats_from_partials_rows, or fold it into the consumer). */
int esc_linear_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                   const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                   float* Y, int64_t ld_y, float* col_stats, void* stream);
/* Linear followed by training-mode BatchNorm statistics: the GEMM epilogue writes the col_stats partials, which are
 * merged — by a finalize launch, or with esc_tune_set(8, 1) and M <= 4096 by the last row-tile workgroup of every
 * column tile of the GEMM itself — into mean / invstd (saved for the backward), the running statistics (momentum update, unbiased variance) and, when
 * scale/shift are given, the consumer-side coefficients act(y) = relu(y*scale + shift).  Same results as
 * esc_linear_fwd + esc_bn_stats_from_partials up to fp64 rounding of the merge.  Needs N > 32 and M > 1. */
typedef struct esc_bn_fuse {
  float eps, momentum;
  float* mean;               /* [N] out */
  float* invstd;             /* [N] out */
  float* running_mean;       /* [N] in/out, may be NULL */
  float* running_var;        /* [N] in/out, may be NULL */
  const float* gamma;        /* [N] or NULL (= 1) */
  const float* beta;         /* [N] or NULL (= 0) */
  float* scale;              /* [N] out, may be NULL (together with shift) */
  float* shift;
} esc_bn_fuse;
/* The H -> 1 prediction head of a training step (run_graphcount.py:186-189 followed by F.l1_loss, :499): pred = relu(x*scale+shift) w^T + b
 * and, in the same launch, dpred[i] = sign(pred[i] - target[i]) * grad_scale / denom — exactly what esc_l1_loss leaves in its dpred —
 * so that the backward does not wait for the loss launch (whose value can then be computed on another stream). */
int esc_linear_fwd_l1(const float* X, int64_t ld_x, const float* w, const float* bias, const float* in_scale, const float* in_shift,
                      int64_t M, int64_t K, const float* target, int64_t denom, float grad_scale, float* pred, float* dpred, void* stream);
int esc_linear_fwd_l1_ok(const float* X, int64_t ld_x, const float* w, int64_t K, const float* in_scale, const float* in_shift);
/* Second half of a Linear whose reduction is cut in two: Y = Y0 + act(X) W^T + b, where Y0 [M, N] is what the first slice of the
 * input columns contributed (esc_linear_fwd over those columns, no bias).  The accumulators START from Y0, so the BatchNorm partials
 * in col_stats and the bias see the complete sums.  Used for the readout Linear over the layer concat (run_graphcount.py:183-185):
 * the slices of the layers that are already final are reduced on the idle second stream while the last layer runs.  The result differs
 * from the one-launch reduction in the order of the fp32 additions only (within the 1e-5 tolerance of north_star, tested).
 * esc_linear_fwd_from_ok: the shape is served (LDS-DMA tiles). */
int esc_linear_fwd_from(const float* Y0, int64_t ld_y0, const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                        const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K, float* Y, int64_t ld_y,
                        float* col_stats, void* stream);
int esc_linear_fwd_from_ok(const float* X, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K, int has_prologue);
int esc_linear_bn_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                      const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                      float* Y, int64_t ld_y, float* col_stats, const esc_bn_fuse* bn, void* stream);
/* ---- BatchNorm statistics consumed in place ("fold") ----------------------------------------------------------
 * The forward GEMM leaves one (mean, M2) partial per output column and ROW BLOCK in col_stats; the block height
 * depends on the kernel that served the shape: esc_linear_stats_block_rows() (32, 64 or 128).  Finish them either with
 * a finalize launch (esc_bn_stats_from_partials_rows) or — node-sized layers — inside the CONSUMER of the BatchNorm
 * output: esc_linear_fwd_fold / esc_affine_act_fold merge the partials in their prologue (every workgroup in the same
 * fixed order: bit-identical coefficients) and workgroup 0 stores mean / invstd / scale / shift and updates the running
 * statistics.  One launch less per BatchNorm on the latency-critical chain (~6 us each, 11 per training step). */
typedef struct esc_bn_fold {
  const float* partials;     /* float2[ceil(rows/block_rows)][C] */
  int64_t rows, block_rows, C;
  float eps, momentum;
  const float* gamma;        /* [C] or NULL (= 1) */
  const float* beta;         /* [C] or NULL (= 0) */
  float* mean;               /* [C] out */
  float* invstd;             /* [C] out */
  float* scale;              /* [C] out, may be NULL (together with shift) */
  float* shift;
  float* running_mean;       /* [C] in/out, may be NULL */
  float* running_var;        /* [C] in/out, may be NULL */
} esc_bn_fold;
int64_t esc_linear_stats_block_rows(const float* X, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int64_t N,
                                    int64_t K);
int esc_linear_fold_available(void);        /* 0 while esc_tune_set(11, 0) keeps every GEMM on the r01 tiles */
/* Y = relu(BN(X)) W^T + bias with the BatchNorm of X still in partial form; K = in_bn->C <= 1280, K % 32 == 0 and
 * 16-byte aligned rows (the H-wide layers), or N <= 4 (lin2).  ESC_EINVAL otherwise: finalize and call esc_linear_fwd. */
int esc_linear_fwd_fold(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                        const esc_bn_fold* in_bn, int64_t M, int64_t N, int64_t K, float* Y, int64_t ld_y,
                        float* col_stats, void* stream);
/* tile-shape / split knobs of the three GEMM forms (benchmark sweeps; defaults are the tuned ones):
 * 0 fwd tile for M>=8192, 1 fwd tile for small M, 2/3 same for dX, 4 dW tile, 5 dW target workgroups,
 * 6 dW minimum reduction rows per split (>=128), 7 node-sized fused-backward tile (0: 64x64xBK64, 1: 32x64xBK32 2-wave,
 * 2 (default): 64x64xBK32 — 37 KB of LDS, so that it fits next to three edge-stream workgroups on a CU).
 * tile ids: 0 128x128xBK32, 1 64x64xBK32, 2 128x32xBK32, 3 128x64xBK32, 4 64x64xBK64, 5 32x64xBK32 (2 waves),
 * 6 32x32xBK32 (1 wave), 7 64x32xBK32 (2 waves).
 * knob 8 (default 0): 1 = node-sized BatchNorm reductions (esc_linear_bn_fwd, esc_bn_bwd) are finished by the last
 * workgroup of the producing launch instead of a finalize launch (measured 1 % slower on the cfg1 step).
 * knob 9 (default 256, 1..512): workgroups per 256-column block of the BatchNorm-backward reduction kernel.
 * knob 10 (default 53248): dynamic-LDS floor in bytes of the GEMMs the step engine launches on its edge stream (caps them
 * at 3 workgroups per CU so that the node stream's kernels find a free wave slot; 0 = no cap).
 * knob 11 (default 15): bit mask of the LDS-DMA GEMM family (gemm_dma.h) — bit 0 forward, 1 gradients, 2 the tiny-dimension
 * kernels (linear_small.h), 3 the 64x32 narrow-output tile; 0 = every GEMM on the r01 register-staged tiles (bisecting).
 * knob 12 (default 0): node-sized BatchNorm backward with the finalize folded into the apply kernel (measured 7 % slower).
 * knob 13 (default 0): node-sized BatchNorm backward as ONE launch with a grid barrier (measured 9 % slower). */
int esc_tune_set(int knob, int value);
int esc_debug_gemm_occupancy(int tile_id);   /* resident workgroups/CU the runtime predicts (diagnostics) */
/* dX[M,K] = dY[M,N] * W[N,K]  (accumulate!=0: dX += ...) */
int esc_linear_bwd_input(const float* dY, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                         int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate,
                         void* stream);
/* dW[N,K] = dY[M,N]^T * act(X)[M,K], db[N] = colsum(dY) (db may be NULL).
 * `slabs` = float scratch of esc_linear_bwd_weight_scratch(M,N,K) floats (split-M partials,
 * summed in fixed order => bitwise reproducible). */
int64_t esc_linear_bwd_weight_scratch(int64_t M, int64_t N, int64_t K);
int esc_linear_bwd_weight(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x,
                          const float* in_scale, const float* in_shift, int64_t M, int64_t N,
                          int64_t K, float* dW, int64_t ld_dw, float* db, float* slabs,
                          void* stream);

/* both gradients of one Linear in ONE launch (dX tiles + split-M dW slabs share the dY stream), then the
 * ordered slab reduce.  dX may be NULL (weight gradient only).  Same scratch as esc_linear_bwd_weight. */
int esc_linear_bwd_both(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x,
                        const float* in_scale, const float* in_shift, const float* W, int64_t ld_w,
                        int64_t M, int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate,
                        float* dW, int64_t ld_dw, float* db, float* slabs, void* stream);

/* deferred form: the tiles run now, the ordered slab reduce is described in *job; esc_slab_reduce_jobs then
 * finishes up to ESC_MAX_REDUCE_JOBS weight gradients in ONE launch (they are only needed by the optimiser).
 * Each deferred call needs its own `slabs` region, untouched until the reduce. */
#define ESC_MAX_REDUCE_JOBS 48
typedef struct esc_reduce_job {
  const float* slabs; int64_t n; int32_t splits; int64_t cols; float* dw; int64_t ld_dw;
  const float* db_part; int64_t rows; float* db;
} esc_reduce_job;
int esc_linear_bwd_both_deferred(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x,
                                 const float* in_scale, const float* in_shift, const float* W, int64_t ld_w,
                                 int64_t M, int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate,
                                 float* dW, int64_t ld_dw, float* db, float* slabs, esc_reduce_job* job,
                                 void* stream);
int esc_slab_reduce_jobs(const esc_reduce_job* jobs /* host array */, int count, void* stream);

/* ---- Linear backward with the BatchNorm(+ReLU) backward in front of it folded in (r03) ---------------------------
 * Call sites: every `Linear -> BatchNorm1d -> ReLU` pair of the node-sized MLPs, run_graphcount.py:65-73 (x_embedding),
 * :78-87 / :98-107 (GINEConv.nn) and :183-186 (lin1 -> bn_lin1).  Their backward used to be
 *     bn_bwd partial -> finalize -> apply (writes dY) -> Linear backward (reads dY):
 * four dependent launches of ~5 us each on the latency-bound node chain.  Here the APPLY step happens while the Linear
 * backward stages its dY operand (and the PARTIAL sums of the next BatchNorm can ride in the dX epilogue), so only the
 * finalize stays on the chain.
 *   dOut       gradient of the BatchNorm(+ReLU) OUTPUT [M, N]
 *   bn         its input rows, batch statistics, forward coefficients (scale = gamma*invstd, shift = beta - mean*scale, as
 *              esc_bn_stats writes them) and coef = float2[N] (sum g, sum g*xhat) / M from esc_bn_bwd_coef[_from_partials];
 *              relu = the activation behind the BatchNorm: 0 none, 1 ReLU, 2 ELU (its derivative is recomputed from the
 *              pre-activation fmaf(x, scale, shift), the forward's own expression: [v > 0], resp. v > 0 ? 1 : exp(v))
 *   next       optional: dX is itself the gradient of a BatchNorm(+ReLU) output over K channels whose input rows are
 *              next->x; the dX tiles then also write partial[row_block][K] = (sum g, sum g*xhat) of their rows
 *              (row blocks of esc_linear_bwd_bn_block_rows(M, N, K) rows) for esc_bn_bwd_coef_from_partials.
 * Result: dX / dW / db equal esc_bn_bwd_apply followed by esc_linear_bwd_both[_deferred] (same arithmetic per element;
 * the operand is never written to memory).  job == NULL reduces the slabs at once.
 * esc_linear_bwd_both_bn_ok tells whether a shape is served (node-sized rows, 16-byte aligned operands, N <= 640 on
 * the MFMA tiles or K <= 16 on the narrow-input kernels); callers fall back to the unfused sequence otherwise.
 * bn == NULL (with next != NULL): dOut is a plain dY — only the next BatchNorm's column sums are folded in. */
typedef struct esc_bn_bwd_fused {
  const float* x; int64_t ld_x;
  const float* mean; const float* invstd; const float* scale; const float* shift;
  const float* coef;
  int32_t relu;
} esc_bn_bwd_fused;
typedef struct esc_bn_bwd_next {
  float* partial;
  const float* x; int64_t ld_x;
  const float* mean; const float* invstd; const float* scale; const float* shift;
  int32_t relu;
} esc_bn_bwd_next;
int esc_linear_bwd_both_bn_ok(const float* dOut, int64_t ld_dout, const esc_bn_bwd_fused* bn, const float* X, int64_t ld_x,
                              const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K, const float* dX, int64_t ld_dx,
                              const float* slabs, const esc_bn_bwd_next* next);
int64_t esc_linear_bwd_bn_block_rows(int64_t M, int64_t N, int64_t K);

/* Weight-gradient tiles on a stream of their own (thread-local setting; NULL = off, the default).
 * Inside a training step the dependent chain waits for a Linear backward's dX only (the reference's autograd has the same
 * dependency structure: torch.nn.Linear's grad_weight feeds nothing but the optimiser, run_graphcount.py:497-505).  While
 * `stream` is set, the node-sized (64-row tile) launches of esc_linear_bwd_both_deferred / esc_linear_bwd_both_bn whose slab
 * reduce is DEFERRED (job != NULL) enqueue their dW tiles on `stream`, ordered behind everything queued so far on the launch
 * stream; the dX tiles stay where they were.  The caller (a) keeps dY / X / the BatchNorm operands of such a call unchanged
 * until `stream` has drained, and (b) orders esc_slab_reduce_jobs behind `stream`.  Results are bit-identical to the single
 * launch (same tiles, same slabs, same ordered reduce).  The step engine turns it on with ESC_WGRAD_STREAM=1. */
int esc_linear_bwd_set_wgrad_stream(void* stream);
int esc_linear_bwd_both_bn(const float* dOut, int64_t ld_dout, const esc_bn_bwd_fused* bn, const float* X, int64_t ld_x,
                           const float* in_scale, const float* in_shift, const float* W, int64_t ld_w, int64_t M,
                           int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw,
                           float* db, float* slabs, esc_reduce_job* job, const esc_bn_bwd_next* next, void* stream);
/* coef[c] = (sum g, sum g*xhat) / M, dgamma, dbeta of a BatchNorm(+ReLU) backward — the first two of esc_bn_bwd's three
 * steps (relu in {0, 1, 2}; Y as in esc_bn_bwd) ... */
int esc_bn_bwd_coef(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY, int64_t ld_dy, int64_t M,
                    int64_t C, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                    float* coef, float* dgamma, float* dbeta, float* scratch, void* stream);
/* ... or only the second, from `slots` partial rows float2[slots][C] a producer left (esc_linear_bwd_both_bn's `next`,
 * esc_gine_aggregate_bwd_stats): summed in slot order in fp64 */
int esc_bn_bwd_coef_from_partials(const float* partial, int64_t slots, int64_t M, int64_t C, float* coef, float* dgamma,
                                  float* dbeta, void* stream);

/* ---- BatchNorm1d (training statistics) + ReLU, torch.nn.BatchNorm1d call sites
 * run_graphcount.py:55-60,66-72,80-87,115 --------------------------------------------------
 * `relu` arguments below select the fused activation: 0 none, 1 ReLU, 2 ELU(alpha=1) (zinc_models.py:513-522).
 * stats: mean[C], invstd[C] of X[M,C] (biased variance, eps), optional running-stat update
 * (momentum, unbiased variance) exactly as torch does; then Y = relu?(gamma*(X-mean)*invstd+beta). */
int64_t esc_bn_scratch(int64_t C);      /* floats of scratch the three calls below need */
/* scale/shift (may be NULL together): the consumer-side fused form act(x) = relu(x*scale + shift),
 * scale = gamma*invstd, shift = beta - mean*scale — fed to esc_linear_* as in_scale/in_shift. */
int esc_bn_stats(const float* X, int64_t ld_x, int64_t M, int64_t C, float eps, float momentum,
                 float* mean, float* invstd, float* running_mean, float* running_var,
                 const float* gamma, const float* beta, float* scale, float* shift,
                 float* scratch, void* stream);
/* same outputs as esc_bn_stats, from the per-32-row (mean, M2) partials a forward GEMM left in col_stats */
int esc_bn_stats_from_partials(const float* partials, int64_t M, int64_t C, float eps, float momentum,
                               float* mean, float* invstd, float* running_mean, float* running_var,
                               const float* gamma, const float* beta, float* scale, float* shift,
                               void* stream);
int esc_bn_stats_from_partials_rows(const float* partials, int64_t M, int64_t C, int64_t block_rows, float eps,
                                    float momentum, float* mean, float* invstd, float* running_mean,
                                    float* running_var, const float* gamma, const float* beta, float* scale,
                                    float* shift, void* stream);
/* Y = act(BN(X)) with the BatchNorm still in partial form (bn->C == C <= 1024, C % 4 == 0, 16-byte aligned rows) */
int esc_affine_act_fold(const float* X, int64_t ld_x, int64_t M, int64_t C, const esc_bn_fold* bn, int relu, float* Y,
                        int64_t ld_y, void* stream);
int esc_bn_apply(const float* X, int64_t ld_x, int64_t M, int64_t C, const float* mean,
                 const float* invstd, const float* gamma, const float* beta, int relu, float* Y,
                 int64_t ld_y, void* stream);
/* backward of Y = relu?(BN(X)): dX (may alias dY), dgamma[C], dbeta[C].  Y is the forward output used
 * for the relu mask; NULL => the mask is recomputed from X, gamma, beta (output never materialised). */
int esc_bn_bwd(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
               int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
               const float* gamma, const float* beta, int relu, float* dX, int64_t ld_dx,
               float* dgamma, float* dbeta, float* scratch, void* stream);
/* BatchNorm(+ReLU) backward with a dropout backward folded in (OGB layer updates, /root/reference/ogb_mol_gnn.py:638-645,
 * :744-755): no Y variant (the ReLU mask comes from the pre-BatchNorm rows), relu in {0, 1}, mask = the keep bytes [M][C] of
 * esc_dropout_fwd / esc_affine_act_dropout_fwd, p its rate.
 *   mask_on_output == 0: dY is the gradient of dropout(act(bn(X))) — g = dY * keep / (1-p) before everything else;
 *   mask_on_output == 1: X itself was dropout(input) — the result dX is multiplied by keep / (1-p).
 * Same three launches and the same arithmetic as esc_dropout_bwd followed by esc_bn_bwd (or the reverse).  Needs
 * esc_bn_bwd_dropout_ok(...) (widths and leading dimensions multiples of 4) and 16-byte aligned operands. */
int esc_bn_bwd_dropout_ok(int64_t C, int64_t ld_x, int64_t ld_dy, int64_t ld_dx);
int esc_bn_bwd_dropout(const float* X, int64_t ld_x, const float* dY, int64_t ld_dy, int64_t M, int64_t C, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, int relu, const uint8_t* mask, float p,
                       int mask_on_output, float* dX, int64_t ld_dx, float* dgamma, float* dbeta, float* scratch, void* stream);
/* The two halves of esc_bn_bwd for a BatchNorm whose statistics span several ranks (SyncBN):
 * _sums writes sums[c] = (sum g, sum g*xhat) over the LOCAL rows (float2[C]; g = dY * act'), with mean / invstd the
 * GLOBAL statistics, plus the local dgamma / dbeta; the caller all-reduces `sums`, divides by the global row count
 * and hands the result to _apply as `coef`. */
int esc_bn_bwd_sums(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
                    int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, int relu, float* sums, float* dgamma,
                    float* dbeta, float* scratch, void* stream);
int esc_bn_bwd_apply(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
                     int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, const float* coef, float* dX,
                     int64_t ld_dx, void* stream);

/* Y = relu?(X*scale + shift) — materialises a BatchNorm(+ReLU) output from its fused coefficients */
int esc_affine_act(const float* X, int64_t ld_x, int64_t M, int64_t C, const float* scale,
                   const float* shift, int relu, float* Y, int64_t ld_y, void* stream);
/* inference-mode coefficients from running statistics: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale */
int esc_bn_eval_coef(const float* running_mean, const float* running_var, const float* gamma,
                     const float* beta, float eps, int64_t C, float* scale, float* shift, void* stream);

/* ---- a-11 loss + optimiser (run_graphcount.py:478,500-505) -------------------------------- */
/* loss[0] = sum_i |pred_i - y_i| / denom ; dpred_i = sign(pred_i - y_i) * grad_scale / denom
 * (dpred may be NULL).  denom = M for the reference's L1Loss(mean); = global node count under
 * graph-sharded data parallelism. */
int esc_l1_loss(const float* pred, const float* y, int64_t M, int64_t denom, float grad_scale,
                float* loss, float* dpred, void* stream);
/* BCEWithLogitsLoss()(pred[labeled], y[labeled]), labeled = (y == y) — the OGB criterion with NaN
 * targets ignored (run_ogb_mol.py:65-72).  denom <= 0: mean over the labeled entries of this call;
 * > 0: explicit divisor (global labeled count under graph sharding).  dpred (may be NULL) receives
 * d loss / d pred, 0 at unlabeled entries. */
int esc_bce_logits_loss(const float* pred, const float* y, int64_t M, int64_t denom, float* loss,
                        float* dpred, void* stream);
/* torch.optim.Adam (no amsgrad, no weight decay) over one flat buffer, torch's operation order;
 * `step` is the 1-based step number. */
int esc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step, void* stream);
/* the same update on grad[i] / grad_denom[0] (device scalar): the data-parallel gradient bucket is all-reduced
 * as SUMS over the global batch together with the global target count, and the division rides on this launch. */
int esc_adam_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                         double lr, double beta1, double beta2, double eps, int64_t step,
                         const float* grad_denom, void* stream);

/* ---- whole-step engine: NestedGIN_eff forward + L1 + backward in ONE host call ---------------------
 * (run_graphcount.py:134-194 forward, :500-503 loss/backward).  The host passes pointer tables of the
 * model's parameters / gradient slots / BatchNorm buffers and of the batch + plan; every kernel above
 * is enqueued from C++ (no per-op host round trip) with BatchNorm+ReLU fused into the consumer GEMMs
 * (z_emb and the hidden MLP activations are never materialised) and layer outputs written straight
 * into the [N,(L+1)H] concatenation buffer.  Gradients are WRITTEN (not accumulated) into the d* slots. */
#define ESC_MAX_LAYERS 16
typedef struct esc_linear_t { const float* w; const float* b; float* dw; float* db; int64_t in_dim, out_dim; } esc_linear_t;
typedef struct esc_bn_t { const float* gamma; const float* beta; float* dgamma; float* dbeta;
                          float* running_mean; float* running_var; float eps, momentum; } esc_bn_t;
typedef struct esc_mlp_t { esc_linear_t lin0; esc_bn_t bn0; esc_linear_t lin1; esc_bn_t bn1; } esc_mlp_t;
typedef struct esc_conv_t { const float* eps; float* deps; esc_mlp_t nn; esc_linear_t lin; } esc_conv_t;
typedef struct esc_nested_gin_t {
  int64_t num_layers, hidden, in_dim, z_rows;
  const float* z_table; float* dz_table;                 /* z_initial.weight */
  esc_bn_t zbn0; esc_linear_t zlin; esc_bn_t zbn1;       /* z_embedding.{1,3,5} */
  esc_mlp_t xemb;                                        /* x_embedding */
  esc_conv_t conv[ESC_MAX_LAYERS];                       /* conv1, convs.* */
  esc_linear_t lin1; esc_bn_t bn_lin1; esc_linear_t lin2;
} esc_nested_gin_t;
typedef struct esc_batch_t {
  int64_t N, E, Z;
  const float* x; const float* y;                        /* x [N,in_dim]; y [N] (train only) */
  const int32_t *in_ptr, *in_edge, *in_src, *out_ptr, *out_edge, *out_dst;
  const int32_t *row_ptr, *bag_idx, *bag_val, *col_ptr, *col_row, *col_val, *col_col;
} esc_batch_t;
/* (bit 3: do NOT apply the knob-10 occupancy cap to the forward's edge GEMMs; bit 4: apply it also to the z_embedding GEMM
 * of the backward tail; bit 5: edge terms two layers ahead of the node chain instead of one; bit 2: edge stream at the
 * highest instead of the lowest priority — both for experiments.)
 * bit 1 (default on): the edge-sized conv.lin GEMMs of all layers run on a second HIP stream, ordered against the
 * node chain by one event per dependency; bit 0 (default off): the x_embedding branch on a further stream.  Default 2. */
int esc_engine_set_side_stream(int on);
/* all three engines: smallest batch (in edges) whose edge pipeline runs on the second stream (default 12 000; 0 = always).
 * Smaller batches — ZINC at bs 128, the 16-graph per-rank slices of a strong-scaling run of the counting model — are launch-latency
 * bound on both pipelines and the cross-stream events cost more than the overlap returns (counting model, bs 16: 0.68 ms on one
 * stream, 0.84 ms on two). */
int esc_engine_set_two_stream_min_edges(int64_t edges);
/* 1 (default): write relu(BN(.)) of the two EDGE-sized z_embedding activations once instead of re-applying the
 * affine+ReLU prologue in every consumer GEMM; 0: fully fused (less memory, slower on MI355X r01). */
int esc_engine_set_materialise_edge_act(int on);
/* 1 (default): BatchNorm statistics come from the producing GEMM's epilogue (col_stats) and are merged by that
 * launch's last workgroups (esc_linear_bn_fwd); 3: same epilogue, separate finalize launch; 0: a pass over Y */
int esc_engine_set_gemm_stats(int on);
/* ---- SyncBN inside the step engine: BatchNorm statistics over ALL ranks of a graph-sharded step ----------
 * The library links no communication stack: the host hands the engine ONE function that sums a device buffer over the
 * ranks, in place, ordered on the given HIP stream (RCCL via torch.distributed in esc_gnn_amd/engine.py; gloo in the
 * two-rank tests).  buf_node / buf_edge are the caller-owned exchange buffers of the two pipelines (cap floats each,
 * >= world * 3 * hidden).  Per BatchNorm: forward = local statistics -> esc_bn_sync_pack -> all-reduce (an all-gather of
 * (count, mean, M2) slots) -> esc_bn_sync_finalize (Chan merge in rank order); backward = esc_bn_bwd_sums -> all-reduce
 * of 2C sums -> esc_bn_sync_coef -> esc_bn_bwd_apply.  fn == NULL or world <= 1 switches it off. */
typedef int (*esc_allreduce_fn)(float* buf, int64_t n, void* stream, void* user);
int esc_engine_set_collective(esc_allreduce_fn fn, void* user, int rank, int world, float* buf_node, float* buf_edge,
                              int64_t cap);
int esc_bn_sync_pack(const float* mean, const float* invstd, int64_t n_local, float eps, int64_t C, int rank, int world,
                     float* buf, void* stream);
int esc_bn_sync_finalize(const float* buf, int world, int64_t C, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, const float* gamma, const float* beta, float* scale,
                         float* shift, float* n_total, void* stream);
int esc_bn_sync_coef(float* coef, int64_t C, const float* n_total, void* stream);
/* diagnostics (ESC_PHASE_TIMING=1 in the environment): mean ms between event marks of the two pipelines over the recorded
 * steps; call after a device synchronise.  Returns the number of steps averaged. */
int esc_engine_phase_times(double* out6, int skip_first);
int64_t esc_engine_workspace_floats(const esc_nested_gin_t* m, int64_t N, int64_t E, int64_t Z);
/* loss[0] = sum|pred-y| / loss_denom (loss_denom <= 0: N).  pred (may be NULL): float[N]. */
int esc_engine_train_step(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                          int64_t loss_denom, float* loss, float* pred, void* stream);
/* eval-mode forward (running statistics, no gradient state kept): pred float[N] */
/* The same step in two halves: _begin enqueues everything except the final join with the edge stream and the
 * edge-side weight-gradient reductions; _end (same thread) enqueues those.  Whatever the caller enqueues on `stream`
 * in between (the next batch's collate) runs while the edge pipeline is still finishing.  The gradients and the loss
 * are complete only after _end; a following esc_engine_train_step* call closes an open step by itself. */
int esc_engine_train_step_begin(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace,
                                int64_t loss_denom, float* loss, float* pred, void* stream);
int esc_engine_train_step_end(void);
/* The step as two autograd halves: forward in training mode (batch statistics; the activations stay in `workspace`,
 * which must not be touched until the backward), and the backward from d(loss)/d(pred) of any loss (float[N]).
 * Parameter gradients land where the model descriptor points, like esc_engine_train_step. */
int esc_engine_forward_train(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, float* pred,
                             void* stream);
int esc_engine_backward(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, const float* dpred,
                        void* stream);
int esc_engine_predict(const esc_nested_gin_t* m, const esc_batch_t* b, float* workspace, float* pred,
                       void* stream);

/* ---- whole-step engine, ZINC variant: NestedGIN_eff of zinc_models.py:504-611 (BASELINE config 4) -------------
 * x = node_type_embedding(x) (:581), z = z_embedding(ESC bag) with ELU (:513-522,:589-590), edge term input
 * [z_emb | edge_type_embedding(edge_attr)] (:591, edge_dim = hidden + 32), L GINEConv layers with ELU MLPs (:593-598),
 * readout global_add_pool(cat(xs)) -> lin1 -> BatchNorm -> ELU -> lin2 (:601-609), L1 loss over the graphs (run_zinc.py).
 * Same conventions as esc_engine_*: gradients are WRITTEN into the d* slots; the batch needs >= 2 graphs
 * (the reference skips bn_lin1 for a single graph, :603-604 — that case stays on the per-op path).
 * Activations are materialised (the GEMM prologue of the counting engine is ReLU-only); the edge pipeline (bag, z_embedding,
 * edge terms and their backward) runs on the engine's second stream like the counting model's when the batch has >= 12 000
 * edges (below that one stream is faster). */
typedef struct esc_embed_t { const float* w; float* dw; int64_t rows, dim; } esc_embed_t;
typedef struct esc_zinc_gin_t {
  int64_t num_layers, hidden, z_rows;
  const float* z_table; float* dz_table;                 /* z_initial.weight */
  esc_bn_t zbn0; esc_linear_t zlin; esc_bn_t zbn1;       /* z_embedding.{1,3,5} */
  esc_embed_t node_emb, edge_emb;                        /* node_type_embedding, edge_type_embedding */
  esc_conv_t conv[ESC_MAX_LAYERS];                       /* conv1, convs.*; conv.lin.in_dim = hidden + edge_emb.dim */
  esc_linear_t lin1; esc_bn_t bn_lin1; esc_linear_t lin2;
} esc_zinc_gin_t;
typedef struct esc_mol_batch_t {
  int64_t N, E, Z, G;
  const int64_t* node_type; const int64_t* edge_type;    /* [N], [E]: data.x, data.edge_attr flattened */
  const float* y;                                        /* [G] (train_step only) */
  const int32_t* graph_ptr;                              /* [G+1]: node range of every graph (batch is sorted) */
  const int32_t *in_ptr, *in_edge, *in_src, *out_ptr, *out_edge, *out_dst;
  const int32_t *row_ptr, *bag_idx, *bag_val, *col_ptr, *col_row, *col_val, *col_col;
} esc_mol_batch_t;
int64_t esc_zinc_workspace_floats(const esc_zinc_gin_t* m, int64_t N, int64_t E, int64_t Z, int64_t G);
/* loss[0] = sum|pred-y| / loss_denom (loss_denom <= 0: G).  pred (may be NULL): float[G]. */
int esc_zinc_train_step(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, int64_t loss_denom,
                        float* loss, float* pred, void* stream);
int esc_zinc_forward_train(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, float* pred, void* stream);
int esc_zinc_backward(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, const float* dpred, void* stream);
int esc_zinc_predict(const esc_zinc_gin_t* m, const esc_mol_batch_t* b, float* workspace, float* pred, void* stream);
/* lookups in small embedding tables (rows <= 4096): out[i,:] = table[idx[i],:] (out-of-range index: zero row, *bad_flag
 * = 1 when given); dtable[r,:] = sum over i with idx[i] == r of g[i,:] in a fixed order (one workgroup per table row) */
int esc_embed_fwd(const float* table, int64_t rows, int64_t C, const int64_t* idx, int64_t M, float* out, int64_t ld_out,
                  int32_t* bad_flag, void* stream);
int esc_embed_bwd(const float* g, int64_t ld_g, const int64_t* idx, int64_t M, int64_t rows, int64_t C, float* dtable,
                  void* stream);
/* out[i,:] = x[i,:] (0 if x == NULL) + rows[graph(i),:] for the n_rows = seg_ptr[G] rows: the virtual-node broadcast
 * h + vn[batch] (ogb_mol_gnn.py:739) */
int esc_segment_broadcast_add(const float* x, int64_t ld_x, const float* rows, int64_t ld_rows, const int32_t* seg_ptr,
                              int64_t G, int64_t n_rows, int64_t C, float* out, int64_t ld_out, void* stream);
/* y = F.dropout(x, p, training=True) (+ res): keep with probability 1-p and scale by 1/(1-p); the keep mask (one byte
 * per element, [M*C]) is written for the backward.  The stream of random numbers is a counter-based hash of (seed,
 * element index) — NOT torch's generator: masks differ from the reference's, their distribution does not.  p == 0 is
 * y = x + res with no mask.  _bwd: dx = dy * mask / (1-p) (+ add). */
/* y = dropout_p(act(x * scale[c] + shift[c])) (+ res): BatchNorm in coefficient form -> activation (0 none, 1 ReLU) ->
 * dropout -> residual add of one layer update (/root/reference/ogb_mol_gnn.py:744-755) in one pass; element for element
 * what esc_affine_act followed by esc_dropout_fwd produce (same masks). */
int esc_affine_act_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int64_t C, const float* scale, const float* shift, int act,
                               float p, uint64_t seed, const float* res, int64_t ld_res, float* y, int64_t ld_y, uint8_t* mask,
                               void* stream);
int esc_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int64_t C, float p, uint64_t seed, const float* res, int64_t ld_res,
                    float* y, int64_t ld_y, uint8_t* mask, void* stream);
int esc_dropout_bwd(const float* dy, int64_t ld_dy, int64_t M, int64_t C, float p, const uint8_t* mask, const float* add,
                    int64_t ld_add, float* dx, int64_t ld_dx, void* stream);
/* the tables of sum-of-embeddings encoders (AtomEncoder ogb_mol_gnn.py:264-282, ogb's BondEncoder) gathered into one
 * [sum rows, C] buffer, and the gradient of that buffer scattered back to the tables' gradient slots */
#define ESC_MAX_TABLES 64
typedef struct esc_table_list { int32_t count; int32_t rows[ESC_MAX_TABLES]; const float* w[ESC_MAX_TABLES]; float* dw[ESC_MAX_TABLES]; } esc_table_list;
int esc_table_pack(const esc_table_list* tables, int64_t C, float* cat, void* stream);
int esc_table_unpack_grad(const esc_table_list* tables, int64_t C, const float* dcat, void* stream);
/* esc_bag_fwd that ADDS the bag sums onto what `out` already holds (edge term = Linear(z) + BondEncoder(edge_attr)) */
int esc_bag_fwd_acc(const float* table, int64_t H, const int32_t* row_ptr, const int32_t* idx32,
                    const int32_t* val32, int64_t E, float* out, int64_t ld_out, void* stream);

/* ---- whole-step engine, OGB molecule variant: GNN(gnn_type='gin_eff') of ogb_mol_gnn.py (BASELINE config 5) ----------
 * AtomEncoder (:264-282) -> per layer h + vn[batch] (:739), GINConv_eff (:346-358: edge term = BondEncoder(edge_attr) +
 * edge_encoder_pos(z_emb), mlp = Linear(H,2H) BN ReLU Linear(2H,H)), batch_norms[l] (+ReLU except last), dropout,
 * residual (:744-752), virtual-node update add_pool(h)+vn -> MLP -> dropout (:757-783); JK = last; sum / mean graph
 * pooling + graph_pred_linear (:66-261); BCE-with-logits over the labeled targets (run_ogb_mol.py:65-72).
 * z_embedding = Dropout BN ReLU Linear Dropout BN ReLU (:638-645).  Dropout uses esc_dropout_fwd's own random stream.
 * Two streams (edge pipeline on the engine's second stream, batches of >= 12 000 edges); gradients are WRITTEN into the d* slots; needs >= 2 graphs. */
typedef struct esc_ogb_layer_t {
  const float* eps; float* deps;
  esc_linear_t pos;                                        /* convs[l].edge_encoder_pos */
  esc_linear_t lin0; esc_bn_t bn0; esc_linear_t lin1;      /* convs[l].mlp.{0,1,3} */
  esc_bn_t bn;                                             /* batch_norms[l] */
  esc_linear_t vlin0; esc_bn_t vbn0; esc_linear_t vlin1; esc_bn_t vbn1;   /* mlp_virtualnode_list[l].{0,1,3,4}, l < L-1 */
  int64_t bond_row0;                                       /* first packed-table row of convs[l].edge_encoder */
} esc_ogb_layer_t;
typedef struct esc_ogb_gnn_t {
  int64_t num_layers, hidden, z_rows, num_tasks;
  int32_t residual, mean_pool;
  float drop_ratio; int32_t pad_;
  const float* z_table; float* dz_table;
  esc_bn_t zbn0; esc_linear_t zlin; esc_bn_t zbn1;
  esc_table_list tables;                                   /* atom tables first, then each layer's bond tables */
  int64_t atom_rows, bond_rows;                            /* rows of all atom tables / of one layer's bond tables */
  const float* vn_w; float* vn_dw;                         /* virtualnode_embedding.weight [1,H] */
  esc_ogb_layer_t layer[ESC_MAX_LAYERS];
  esc_linear_t head;                                       /* graph_pred_linear */
} esc_ogb_gnn_t;
/* a sum-of-embeddings lookup as a bag over the packed tables: CSR by output row + CSC by table row (weights all 1) */
typedef struct esc_bag_plan_t { int64_t n_entries; const int32_t *row_ptr, *idx, *ones, *col_ptr, *c_row, *c_col; } esc_bag_plan_t;
typedef struct esc_ogb_batch_t {
  int64_t N, E, Z, G;
  esc_bag_plan_t atoms, bonds;                             /* 9 entries per node / 3 per edge, indices into the packed tables
                                                              (bonds: relative to a layer's bond_row0) */
  const float* y;                                          /* [G, num_tasks], NaN = unlabeled (train_step only) */
  const int32_t* graph_ptr;                                /* [G+1] */
  const int64_t* zero_idx;                                 /* [G] zeros: virtualnode_embedding(0) for every graph (:701) */
  const int32_t *in_ptr, *in_edge, *in_src, *out_ptr, *out_edge, *out_dst;
  const int32_t *row_ptr, *bag_idx, *bag_val, *col_ptr, *col_row, *col_val, *col_col;
  uint64_t seed;                                           /* dropout stream of this step */
} esc_ogb_batch_t;
int64_t esc_ogb_workspace_floats(const esc_ogb_gnn_t* m, int64_t N, int64_t E, int64_t Z, int64_t G, int64_t atom_entries,
                                 int64_t bond_entries);
/* loss[0] = BCEWithLogits over the labeled entries (loss_denom <= 0: their count in this batch).  logits (may be NULL):
 * float[G, num_tasks]. */
int esc_ogb_train_step(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, int64_t loss_denom, float* loss,
                       float* logits, void* stream);
int esc_ogb_forward_train(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, float* logits, void* stream);
int esc_ogb_backward(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, const float* dlogits, void* stream);
int esc_ogb_predict(const esc_ogb_gnn_t* m, const esc_ogb_batch_t* b, float* workspace, float* logits, void* stream);

/* ---- a-5 collate (batch.py:25-149): gather B graphs out of the HBM-resident dataset store ------
 * The store keeps the reference's InMemoryDataset layout (per-key concatenation + slice pointers,
 * GraphCountDataset.py:119-120) plus views sorted ONCE at build time (suffix _all):
 *   in_ptr_all[nodes+1] / in_edge_all[edges]   edges grouped by (global) destination, stable
 *   out_ptr_all / out_edge_all                 edges grouped by source
 *   row_ptr_all[edges+1]                       first bag entry of each edge
 *   c_perm_all[nnz] / c_rank_all[nnz]          entries grouped by (graph, histogram bin) + rank inside the group
 *   col_cnt_all[graphs][n_cols]                entries per bin per graph
 * offsets = int64[4][B+1]: exclusive prefix sums of the selected graphs' node / edge / nnz / y-row
 * counts (computed by the host from its copy of the slice pointers).
 * Outputs: the reference's batch tensors (x, y, edge_index[2,E], batch, pos_enc, pos_index, pos_batch:
 * int64 index tensors bit-identical to Batch.from_data_list) and the compact int32 plan consumed by
 * esc_bag_* / esc_gine_aggregate_*. */
typedef struct esc_collate_args {
  int64_t B, x_dim, y_dim, n_cols;
  const int64_t* graph_ids;     /* [B] device */
  const int64_t* offsets;       /* [4][B+1] device */
  /* store */
  const int64_t *node_ptr, *edge_ptr, *nnz_ptr, *y_ptr;
  const float *x_all, *y_all;
  const int64_t *esrc_all, *edst_all, *pos_enc_all, *pos_index_all, *pos_batch_all;
  const int64_t *in_ptr_all, *in_edge_all, *out_ptr_all, *out_edge_all, *row_ptr_all, *c_perm_all;
  const int32_t* c_rank_all;
  /* from esc_collate_cols */
  const int32_t *col_ptr, *col_prefix;
  /* reference-visible outputs */
  float *x, *y;
  int64_t *edge_index, *batch, *pos_enc, *pos_index, *pos_batch;
  /* plan outputs */
  int32_t *in_ptr, *in_edge, *in_src, *out_ptr, *out_edge, *out_dst;
  int32_t *row_ptr, *bag_idx, *bag_val, *col_row, *col_val, *col_col;
  /* optional (ABI 3; NULL / 0 = absent) */
  const void* edge_attr_all;    /* per-edge attribute rows of the store, ea_words 4-byte words each (edge_attr of ZINC / OGB) */
  void* edge_attr;              /* [E][ea_words] gathered like edge_index (Batch.from_data_list, batch.py:112-113: no offset) */
  int64_t ea_words;
  int64_t* x_long;              /* categorical node features: x written as int64 here instead of float into `x` */
  int32_t* graph_ptr;           /* [B+1] first node of every graph of the batch (= offsets row 0 as int32) */
} esc_collate_args;
/* column bookkeeping of the batch: col_prefix[B][n_cols], col_total[n_cols], col_ptr[n_cols+1] */
int esc_collate_cols(const int32_t* col_cnt_all, int64_t n_cols, const int64_t* graph_ids, int64_t B,
                     int32_t* col_prefix, int32_t* col_total, int32_t* col_ptr, void* stream);
int esc_collate_fill(const esc_collate_args* args /* host struct of device pointers */, void* stream);

/* ---- a-1..a-4 feature build (utils_edge_efficient.py:20-152,201-294) -----------------------
 * G graphs per call.  node_ptr[G+1] / edge_ptr[G+1]: int64 prefix sums of node and input-edge
 * counts (the InMemoryDataset `slices` layout); src/dst: concatenated int64 edge lists with
 * graph-LOCAL node ids.  If self_loop, every (a,a) is dropped and (i,i), i<n appended at the END
 * (:33-36).  max_nodes = largest graph (<= 4096; sizes the LDS working sets), sum_nodes_sq = sum over
 * graphs of n_g^2 (sizes the per-root hop tables).  Ego-nets of any size are encoded: the rd pseudo-inverse of
 * one of more than 96 nodes runs on a global-memory slab instead of LDS (slower, same result).
 * Pass 1 (_count): out_edge_ptr[G+1] (edges after normalisation) and nnz_ptr[cap+1], cap =
 * total_in_edges (+ total_nodes if self_loop): exclusive scan of per-output-edge nonzero counts
 * (entries past the real edge total repeat the grand total).  The host reads out_edge_ptr[G] and
 * nnz_ptr[that] to size the outputs.  Pass 2 (_fill): edge_index' (out_src/out_dst, int64, local
 * ids), in_edge_of_out (global input edge id or -1 for an appended loop; may be NULL) and the
 * sparse encoding pos_enc / pos_index / pos_batch (int64; pos_batch = graph-local edge id, :143).
 * status[G]: 0, or ESC_ERANGE where the reference's one_hot would raise (degree >= 200, rd bin
 * outside [0,100), edge code >= 1300) or a node id is out of range.
 * work: esc_features_scratch_bytes(...) bytes, must be the SAME buffer for both passes (the count pass leaves
 * the hop tables and the per-edge rd rows in it).
 * max_nodes_per_hop (k_hop_subgraph :235-237, python random.sample on the host's Mersenne twister, one draw per
 * root and level in edge order) has no entry point: it is a sequential host-RNG walk, never set by the
 * reference's run scripts, and is permanently outside this path (the python mirror raises). */
int64_t esc_features_scratch_bytes(int64_t G, int64_t total_nodes, int64_t total_in_edges, int64_t sum_nodes_sq,
                                   int64_t max_nodes, int use_rd);
int esc_features_count(const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* src,
                       const int64_t* dst, int64_t G, int64_t total_nodes, int64_t total_in_edges,
                       int64_t sum_nodes_sq, int64_t max_nodes, int h, int use_rd, int self_loop,
                       int64_t* out_edge_ptr, int64_t* nnz_ptr, int32_t* status, void* work, void* stream);
int esc_features_fill(const int64_t* node_ptr, const int64_t* edge_ptr, int64_t G, int64_t total_nodes,
                      int64_t total_in_edges, int64_t sum_nodes_sq, int64_t max_nodes, int h, int use_rd,
                      int self_loop, const int64_t* out_edge_ptr, const int64_t* nnz_ptr, int64_t total_out_edges,
                      int64_t* out_src, int64_t* out_dst, int64_t* in_edge_of_out, int64_t* pos_enc,
                      int64_t* pos_index, int64_t* pos_batch, int32_t* status, void* work, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESCGNN_HIP_H */
