// linear_mfma.hip — the dense layers of NestedGIN_eff on the gfx950 matrix cores in exact fp32.
//
//   forward      Y[M,N]  = act(X)[M,K] * W[N,K]^T + bias           (torch.nn.Linear)
//   input grad   dX[M,K] = dY[M,N] * W[N,K]
//   weight grad  dW[N,K] = dY[M,N]^T * act(X)[M,K],  db[N] = colsum(dY)
//
// Call sites replaced: every torch.nn.Linear of /root/reference/run_graphcount.py:54-121,183-189
// and GINEConv.lin (edge_dim -> in_channels), the only GEMM-shaped work on the path.
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate) is a k-ordered fp32 fma chain — no TF32-like
// truncation exists on gfx950 — so results stay within fp32 rounding of the CPU oracle (1e-5 bar).
// Peak 157 TFLOP/s; these shapes (M = E or N_nodes, N,K <= 1280) are short-K, so the kernel is a
// classic LDS-tiled, register-prefetched (global -> VGPR -> LDS, one barrier per 32-deep K step)
// design with 4 waves per workgroup, each owning (BM/WM) x (BN/WN) of the tile as 32x32 MFMA blocks.
//
// Operand forms.  "k-contiguous": the reduction index is the fastest-moving index in memory
// (X[M,K], W[N,K] in forward).  LDS image [row][BK+4]; a lane fetches 4 consecutive k of its row
// with one ds_read_b128 (conflict-free with the +4 pad) and feeds 4 MFMAs — lane half h owns
// k = 8c+4h+t, so the k order inside an 8-chunk is permuted identically for A and B.
// "reduction-major": the reduction index is the row index in memory (W[N,K] for dX, dY and X for
// dW).  LDS image [k][cols+4]; a lane reads single floats (ds_read_b32, consecutive lanes ->
// consecutive banks).
#include "common.h"
#include <array>
#include <cstdlib>
#include <map>
#include "gemm_dma.h"
#include "linear_small.h"
#include <type_traits>

namespace esc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KPAD = 4;

template <int ROWS, int BK, int NTHR = 256>
struct KContigTile {            // [ROWS][BK+KPAD]
  static constexpr int LD = BK + KPAD;
  static constexpr int FLOATS = ROWS * LD;
  static constexpr int QPR = BK / 4;                           // float4 per row
  static constexpr int PER_THREAD = ROWS * QPR / NTHR;         // float4 per thread
  static_assert(ROWS * QPR % NTHR == 0 && NTHR % QPR == 0, "tile must split evenly over the workgroup");
};
template <int COLS, int BK, int NTHR = 256>
struct RedMajorTile {           // [BK][COLS+KPAD]
  static constexpr int LD = COLS + KPAD;
  static constexpr int FLOATS = BK * LD;
  static constexpr int PER_THREAD = BK * (COLS / 4) / NTHR;
  static_assert(BK * (COLS / 4) % NTHR == 0 && NTHR % (COLS / 4) == 0, "tile must split evenly over the workgroup");
};

// ---- global -> register staging ------------------------------------------------------------------
// k-contiguous: rows r0.. of `src` (ld), reduction range [k0, k0+BK); element (r, k) valid iff
// r < rows && k < kdim.  Optional per-k affine+relu (fused BatchNorm+ReLU of the producer).
template <int ROWS, int BK, int NTHR, bool PRO>
__device__ __forceinline__ bool load_kcontig(const float* __restrict__ src, int64_t ld, int r0, int rows,
                                             int k0, int kdim, bool vec_ok,
                                             const float* __restrict__ sc, const float* __restrict__ sh,
                                             float4 (&reg)[KContigTile<ROWS, BK, NTHR>::PER_THREAD]) {
  using T = KContigTile<ROWS, BK, NTHR>;
  const int tid = threadIdx.x;
  const int kq = tid % T::QPR;
  const int k = k0 + kq * 4;
  // Fast path (block-uniform condition): whole 16-B quads inside K.  Loads are UNCONDITIONAL — an
  // out-of-range row is clamped to the last valid row and zeroed by a select — so hipcc emits straight
  // global_load_dwordx4 streams instead of a branch + vmcnt(0) per load.
  if (vec_ok && k0 + BK <= kdim) {
    // RAW loads only: nothing here may depend on the loaded values, or hipcc waits for them on the spot
    // and the prefetch collapses.  The affine+ReLU prologue and the row mask run in finish_kcontig(),
    // right before the LDS store one K-step later.
#pragma unroll
    for (int p = 0; p < T::PER_THREAD; ++p) {
      const int r = r0 + tid / T::QPR + p * (NTHR / T::QPR);
      const int rc = min(r, rows - 1);
      reg[p] = *reinterpret_cast<const float4*>(src + (size_t)rc * ld + k);
    }
    return true;
  }
#pragma unroll
  for (int p = 0; p < T::PER_THREAD; ++p) {
    const int r = r0 + tid / T::QPR + p * (NTHR / T::QPR);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const float* q = src + (size_t)r * ld + k;
      if (k + 0 < kdim) v.x = q[0];
      if (k + 1 < kdim) v.y = q[1];
      if (k + 2 < kdim) v.z = q[2];
      if (k + 3 < kdim) v.w = q[3];
      if constexpr (PRO) {
        if (k + 0 < kdim) v.x = fmaxf(fmaf(v.x, sc[k + 0], sh[k + 0]), 0.f);
        if (k + 1 < kdim) v.y = fmaxf(fmaf(v.y, sc[k + 1], sh[k + 1]), 0.f);
        if (k + 2 < kdim) v.z = fmaxf(fmaf(v.z, sc[k + 2], sh[k + 2]), 0.f);
        if (k + 3 < kdim) v.w = fmaxf(fmaf(v.w, sc[k + 3], sh[k + 3]), 0.f);
      }
    }
    reg[p] = v;
  }
  return false;
}
// prologue + row mask of a RAW k-contiguous tile (see load_kcontig)
template <int ROWS, int BK, int NTHR, bool PRO>
__device__ __forceinline__ void finish_kcontig(int r0, int rows, float4 s4, float4 h4,
                                               float4 (&reg)[KContigTile<ROWS, BK, NTHR>::PER_THREAD]) {
  using T = KContigTile<ROWS, BK, NTHR>;
  if constexpr (!PRO) {
    if (r0 + ROWS <= rows) return;        // interior tile (block-uniform): nothing to mask, nothing to transform
  }
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < T::PER_THREAD; ++p) {
    const int r = r0 + tid / T::QPR + p * (NTHR / T::QPR);
    float4 v = reg[p];
    if constexpr (PRO) {
      v.x = fmaxf(fmaf(v.x, s4.x, h4.x), 0.f); v.y = fmaxf(fmaf(v.y, s4.y, h4.y), 0.f);
      v.z = fmaxf(fmaf(v.z, s4.z, h4.z), 0.f); v.w = fmaxf(fmaf(v.w, s4.w, h4.w), 0.f);
    }
    const bool ok = r < rows;
    reg[p] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
}
template <int ROWS, int BK, int NTHR>
__device__ __forceinline__ void store_kcontig(float* __restrict__ lds, const float4 (&reg)[KContigTile<ROWS, BK, NTHR>::PER_THREAD]) {
  using T = KContigTile<ROWS, BK, NTHR>;
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < T::PER_THREAD; ++p) {
    const int r = tid / T::QPR + p * (NTHR / T::QPR);
    *reinterpret_cast<float4*>(lds + r * T::LD + (tid % T::QPR) * 4) = reg[p];
  }
}

// reduction-major: rows (reduction) [k0, k0+BK) of `src`, columns c0..c0+COLS; valid iff k < kdim && c < cols.
// Optional per-COLUMN affine+relu (for act(X) in the weight gradient).
template <int COLS, int BK, int NTHR, bool PRO>
__device__ __forceinline__ bool load_redmajor(const float* __restrict__ src, int64_t ld, int k0, int kdim,
                                              int c0, int cols, bool vec_ok,
                                              const float* __restrict__ sc, const float* __restrict__ sh,
                                              float4 (&reg)[RedMajorTile<COLS, BK, NTHR>::PER_THREAD]) {
  const int tid = threadIdx.x;
  constexpr int QPR = COLS / 4;  // float4 per row
  if (vec_ok && c0 + COLS <= cols) {   // block-uniform fast path: RAW unconditional loads, clamped reduction row
#pragma unroll
    for (int p = 0; p < RedMajorTile<COLS, BK, NTHR>::PER_THREAD; ++p) {
      const int f = tid + p * NTHR;
      const int kk = f / QPR, cq = f % QPR;
      const int kc = min(k0 + kk, kdim - 1);
      reg[p] = *reinterpret_cast<const float4*>(src + (size_t)kc * ld + c0 + cq * 4);
    }
    return true;
  }
#pragma unroll
  for (int p = 0; p < RedMajorTile<COLS, BK, NTHR>::PER_THREAD; ++p) {
    const int f = tid + p * NTHR;
    const int kk = f / QPR, cq = f % QPR;
    const int k = k0 + kk, c = c0 + cq * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kdim) {
      const float* q = src + (size_t)k * ld + c;
      if (c + 0 < cols) v.x = q[0];
      if (c + 1 < cols) v.y = q[1];
      if (c + 2 < cols) v.z = q[2];
      if (c + 3 < cols) v.w = q[3];
      if constexpr (PRO) {
        if (c + 0 < cols) v.x = fmaxf(fmaf(v.x, sc[c + 0], sh[c + 0]), 0.f);
        if (c + 1 < cols) v.y = fmaxf(fmaf(v.y, sc[c + 1], sh[c + 1]), 0.f);
        if (c + 2 < cols) v.z = fmaxf(fmaf(v.z, sc[c + 2], sh[c + 2]), 0.f);
        if (c + 3 < cols) v.w = fmaxf(fmaf(v.w, sc[c + 3], sh[c + 3]), 0.f);
      }
    }
    reg[p] = v;
  }
  return false;
}
template <int COLS, int BK, int NTHR, bool PRO>
__device__ __forceinline__ void finish_redmajor(int k0, int kdim, float4 s4, float4 h4,
                                                float4 (&reg)[RedMajorTile<COLS, BK, NTHR>::PER_THREAD]) {
  const int tid = threadIdx.x;
  constexpr int QPR = COLS / 4;
  static_assert(NTHR % QPR == 0, "a thread keeps the same column quad for every pass");
  if constexpr (!PRO) {
    if (k0 + BK <= kdim) return;          // full K-step (block-uniform): nothing to mask
  }
#pragma unroll
  for (int p = 0; p < RedMajorTile<COLS, BK, NTHR>::PER_THREAD; ++p) {
    const int f = tid + p * NTHR;
    const int kk = f / QPR;
    float4 v = reg[p];
    if constexpr (PRO) {
      v.x = fmaxf(fmaf(v.x, s4.x, h4.x), 0.f); v.y = fmaxf(fmaf(v.y, s4.y, h4.y), 0.f);
      v.z = fmaxf(fmaf(v.z, s4.z, h4.z), 0.f); v.w = fmaxf(fmaf(v.w, s4.w, h4.w), 0.f);
    }
    const bool ok = k0 + kk < kdim;
    reg[p] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
}
template <int COLS, int BK, int NTHR>
__device__ __forceinline__ void store_redmajor(float* __restrict__ lds, const float4 (&reg)[RedMajorTile<COLS, BK, NTHR>::PER_THREAD]) {
  const int tid = threadIdx.x;
  constexpr int QPR = COLS / 4;
#pragma unroll
  for (int p = 0; p < RedMajorTile<COLS, BK, NTHR>::PER_THREAD; ++p) {
    const int f = tid + p * NTHR;
    *reinterpret_cast<float4*>(lds + (f / QPR) * RedMajorTile<COLS, BK, NTHR>::LD + (f % QPR) * 4) = reg[p];
  }
}

// ---- fragment reads: 4 consecutive MFMA k-steps of one 32-row block -----------------------------
// returns f[t] = operand value for MFMA t of 8-chunk `c8` (k = 8*c8 + 4*h + t)
template <int ROWS, int BK>
__device__ __forceinline__ float4 frag_kcontig(const float* __restrict__ lds, int row0, int c8) {
  const int l = lane_id();
  return *reinterpret_cast<const float4*>(lds + (row0 + (l & 31)) * KContigTile<ROWS, BK>::LD + c8 * 8 + (l >> 5) * 4);
}
template <int COLS, int BK>
__device__ __forceinline__ float4 frag_redmajor(const float* __restrict__ lds, int col0, int c8) {
  const int l = lane_id();
  constexpr int LD = RedMajorTile<COLS, BK>::LD;
  const float* p = lds + (c8 * 8 + (l >> 5) * 4) * LD + col0 + (l & 31);
  return make_float4(p[0], p[LD], p[2 * LD], p[3 * LD]);
}

// =================================================================================================
// Generic tile kernel.  C[BM x BN] (+)= A_op[BM x R] * B_op[R x BN] over reduction range
// [red0, red1) (blockIdx.z selects the split for the weight gradient).
//   A_KC : A operand k-contiguous (rows = output rows)   else reduction-major (cols = output rows)
//   B_KC : B operand k-contiguous (rows = output cols)   else reduction-major (cols = output cols)
// =================================================================================================
// BatchNorm finalize fused behind the statistics epilogue: the last row-tile workgroup of every column tile
// (grid_last_block on tickets[bx]) merges that tile's partials and writes what bn_finalize_kernel would
struct BnFuse {
  unsigned* tickets;        // nullptr = off; one counter per column tile
  int row_tiles;            // workgroups sharing a counter
  float eps, momentum;
  float* mean; float* invstd; float* running_mean; float* running_var;
  const float* gamma; const float* beta; float* scale; float* shift;
};

struct GemmArgs {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  const float* bias;        // per output column (forward) or nullptr
  const float* pro_scale;   // prologue affine (applies to A if A_KC: per k; to B if !B_KC && !A_KC: per col)
  const float* pro_shift;
  float* db_part;           // weight grad: per-split column sums of A' (= dY)   [splits][rowsC]
  float2* col_stats;        // forward: per 32-row block (mean, M2) of the outputs, [ceil(rowsC/32)][colsC]
  int rowsC, colsC, red;    // output rows, output cols, reduction length
  int red_per_split;
  int accumulate;
  int a_vec, b_vec, c_slab; // alignment flags; c_slab: C is a [splits][rowsC][colsC] slab buffer
  BnFuse fin;
};

// Merge the per-32-row (mean, M2) partials of columns [n0, n0+BN) — every group but possibly the last holds exactly
// 32 rows, so the merge is division-free: with d_p = mean_p - pivot,  mean = pivot + S1/G,
// M2 = sum M2_p + 32 (S2 - S1^2/G)  (fp64, shifted by the first group's mean: no cancellation), then one Chan merge
// with the ragged last group.  NTHR/BN threads share a column (contiguous slot ranges, summed in fixed order).
template <int BN, int NTHR>
__device__ __forceinline__ void bn_finalize_cols(const GemmArgs& g, int n0, float* lds) {
  static_assert(NTHR % BN == 0, "threads must tile the column block");
  constexpr int TPC = NTHR / BN;
  const int tid = threadIdx.x, cl = tid % BN, part = tid / BN;
  const int col = n0 + cl;
  const int M = g.rowsC, C = g.colsC;
  const int full = M / 32;
  double S1 = 0.0, S2 = 0.0, SM = 0.0, pivot = 0.0;
  if (col < C && full > 0) {
    pivot = (double)g.col_stats[col].x;
    const int per = (full + TPC - 1) / TPC;
    const int p0 = part * per, p1 = min(full, p0 + per);
    // the partials were written through to memory by other workgroups: every load is a long-latency miss, so
    // keep 16 of them in flight per thread
    for (int p = p0; p < p1; p += 16) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = g.col_stats[(size_t)min(p + u, p1 - 1) * C + col];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (p + u < p1) {
          const double d = (double)v[u].x - pivot;
          S1 += d;
          S2 += d * d;
          SM += (double)v[u].y;
        }
      }
    }
  }
  double* sh = reinterpret_cast<double*>(lds);      // [3][NTHR]; the staging buffers are dead by now
  sh[tid] = S1; sh[NTHR + tid] = S2; sh[2 * NTHR + tid] = SM;
  __syncthreads();
  if (part != 0 || col >= C) return;
#pragma unroll
  for (int q = 1; q < TPC; ++q) { S1 += sh[q * BN + cl]; S2 += sh[NTHR + q * BN + cl]; SM += sh[2 * NTHR + q * BN + cl]; }
  double n = 0.0, mu = 0.0, m2 = 0.0;
  if (full > 0) {
    n = 32.0 * full;
    mu = pivot + S1 / full;
    m2 = SM + 32.0 * (S2 - S1 * S1 / full);
    if (m2 < 0.0) m2 = 0.0;
  }
  if (M > 32 * full) {
    const float2 v = g.col_stats[(size_t)full * C + col];
    chan_merge(n, mu, m2, (double)(M - 32 * full), (double)v.x, (double)v.y);
  }
  const BnFuse& f = g.fin;
  const float is = (float)(1.0 / sqrt(m2 / (double)M + (double)f.eps));
  f.mean[col] = (float)mu;
  f.invstd[col] = is;
  if (f.scale) {
    const float sc = (f.gamma ? f.gamma[col] : 1.f) * is;
    f.scale[col] = sc;
    f.shift[col] = (f.beta ? f.beta[col] : 0.f) - (float)mu * sc;
  }
  if (f.running_mean) f.running_mean[col] = (1.f - f.momentum) * f.running_mean[col] + f.momentum * (float)mu;
  if (f.running_var) f.running_var[col] = (1.f - f.momentum) * f.running_var[col] + f.momentum * (float)(m2 / (double)(M - 1));
}

template <int BM, int BN, int WM, int WN, int BK, bool A_KC, bool B_KC, bool PRO, bool DB, int KW = 1>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, float* __restrict__ lds, int bx, int by, int bz) {
  // KW > 1: KW wave groups share ONE output tile and split every K-step between them (wave group wk owns
  // 8-chunks [wk*BK/8/KW, (wk+1)*BK/8/KW)); their accumulators are summed through LDS in group order at
  // the end.  Node-sized layers only have ~600 32x32 output blocks, i.e. 0.6 waves per SIMD — splitting K
  // in the workgroup is what puts >2 waves on every SIMD so MFMA, LDS and barrier phases overlap.
  constexpr int NTHR = WM * WN * KW * 64;
  static_assert(!DB || KW == 1, "bias-gradient column sums assume one wave group");
  static_assert((BK / 8) % KW == 0, "K-step must split evenly over the wave groups");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 32, NT = TN / 32;
  static_assert(MT >= 1 && NT >= 1, "wave tile must hold at least one 32x32 block");
  using ATile = typename std::conditional<A_KC, KContigTile<BM, BK, NTHR>, RedMajorTile<BM, BK, NTHR>>::type;
  using BTile = typename std::conditional<B_KC, KContigTile<BN, BK, NTHR>, RedMajorTile<BN, BK, NTHR>>::type;
  constexpr int STAGE = ATile::FLOATS + BTile::FLOATS;   // one K-step of A then B (LDS holds 2 stages)

  const int m0 = by * BM;
  const int n0 = bx * BN;
  const int split = bz;
  const int red0 = split * g.red_per_split;
  const int red1 = min(g.red, red0 + g.red_per_split);
  const int wave = threadIdx.x >> 6;
  const int wk = wave / (WM * WN);
  const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int l = lane_id();

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // two register sets: tile kt+1 waits in one while tile kt+2 is being fetched into the other
  float4 ra0[ATile::PER_THREAD], rb0[BTile::PER_THREAD];
  float4 ra1[ATile::PER_THREAD], rb1[BTile::PER_THREAD];
  float dbsum = 0.f;

  // Prologue coefficients travel with the tile they belong to (loaded as RAW values next to it): fetching
  // them at finish time would sit behind the NEXT tile's loads in the in-order vmcnt queue and drain the
  // prefetch.  k-contiguous A: one (scale, shift) quad per K-step; reduction-major B: the thread's column
  // quad never changes, so it is loaded once.
  const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ps0 = one4, ph0 = zero4, ps1 = one4, ph1 = zero4, pcs = one4, pch = zero4;
  bool pro_vec = false;
  if constexpr (PRO) {
    pro_vec = A_KC ? (g.a_vec != 0) : (g.b_vec != 0);   // host sets *_vec only if the coefficient vectors are 16-B aligned
    if constexpr (!A_KC) {
      const int c = n0 + (threadIdx.x % (BN / 4)) * 4;
      if (pro_vec && c + 3 < g.colsC) {
        pcs = *reinterpret_cast<const float4*>(g.pro_scale + c);
        pch = *reinterpret_cast<const float4*>(g.pro_shift + c);
      }
    }
  }
  // returns bit0: A tile is RAW (needs finish), bit1: B tile is RAW
  auto gload = [&](int k0, float4 (&ra)[ATile::PER_THREAD], float4 (&rb)[BTile::PER_THREAD], float4& ps, float4& ph) -> int {
    bool rawa, rawb;
    if constexpr (PRO && A_KC) {
      const int k = k0 + (threadIdx.x % (BK / 4)) * 4;
      if (pro_vec && k0 + BK <= red1) {
        ps = *reinterpret_cast<const float4*>(g.pro_scale + k);
        ph = *reinterpret_cast<const float4*>(g.pro_shift + k);
      }
    }
    if constexpr (A_KC) rawa = load_kcontig<BM, BK, NTHR, PRO>(g.A, g.lda, m0, g.rowsC, k0, red1, g.a_vec, g.pro_scale, g.pro_shift, ra);
    else                rawa = load_redmajor<BM, BK, NTHR, false>(g.A, g.lda, k0, red1, m0, g.rowsC, g.a_vec, nullptr, nullptr, ra);
    if constexpr (B_KC) rawb = load_kcontig<BN, BK, NTHR, false>(g.B, g.ldb, n0, g.colsC, k0, red1, g.b_vec, nullptr, nullptr, rb);
    else                rawb = load_redmajor<BN, BK, NTHR, PRO && !A_KC>(g.B, g.ldb, k0, red1, n0, g.colsC, g.b_vec, g.pro_scale, g.pro_shift, rb);
    return (rawa ? 1 : 0) | (rawb ? 2 : 0);
  };
  auto lstore = [&](int buf, int k0, int raw, float4 (&ra)[ATile::PER_THREAD], float4 (&rb)[BTile::PER_THREAD], float4 ps, float4 ph) {
    if (raw & 1) {
      if constexpr (A_KC) finish_kcontig<BM, BK, NTHR, PRO>(m0, g.rowsC, ps, ph, ra);
      else                finish_redmajor<BM, BK, NTHR, false>(k0, red1, zero4, zero4, ra);
    }
    if (raw & 2) {
      if constexpr (B_KC) finish_kcontig<BN, BK, NTHR, false>(n0, g.colsC, zero4, zero4, rb);
      else                finish_redmajor<BN, BK, NTHR, PRO && !A_KC>(k0, red1, pcs, pch, rb);
    }
    float* a_w = lds + buf * STAGE;
    float* b_w = a_w + ATile::FLOATS;
    if constexpr (A_KC) store_kcontig<BM, BK, NTHR>(a_w, ra); else store_redmajor<BM, BK, NTHR>(a_w, ra);
    if constexpr (B_KC) store_kcontig<BN, BK, NTHR>(b_w, rb); else store_redmajor<BN, BK, NTHR>(b_w, rb);
  };
  auto compute = [&](int cur) {
    const float* a_l = lds + cur * STAGE;
    const float* b_l = a_l + ATile::FLOATS;
    if constexpr (DB) {   // column sums of the reduction-major A' tile (bias gradient), block column 0 only
      if (bx == 0 && threadIdx.x < BM) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) dbsum += a_l[kk * ATile::LD + threadIdx.x];
      }
    }
#pragma unroll
    for (int cc = 0; cc < BK / 8 / KW; ++cc) {
      const int c8 = wk * (BK / 8 / KW) + cc;
      float4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (A_KC) af[i] = frag_kcontig<BM, BK>(a_l, wm * TM + i * 32, c8);
        else                af[i] = frag_redmajor<BM, BK>(a_l, wm * TM + i * 32, c8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (B_KC) bf[j] = frag_kcontig<BN, BK>(b_l, wn * TN + j * 32, c8);
        else                bf[j] = frag_redmajor<BN, BK>(b_l, wn * TN + j * 32, c8);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nk = (red1 > red0) ? (red1 - red0 + BK - 1) / BK : 0;
  int raw0 = 0, raw1 = 0;
  if (nk > 0) raw0 = gload(red0, ra0, rb0, ps0, ph0);
  if (nk > 1) raw1 = gload(red0 + BK, ra1, rb1, ps1, ph1);
  if (nk > 0) lstore(0, red0, raw0, ra0, rb0, ps0, ph0);
  __syncthreads();
  // invariant at the top of iteration kt: LDS[kt&1] = tile kt; register set (kt+1)&1 = tile kt+1 (in flight)
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) raw0 = gload(red0 + (kt + 2) * BK, ra0, rb0, ps0, ph0);
    compute(0);
    if (kt + 1 < nk) lstore(1, red0 + (kt + 1) * BK, raw1, ra1, rb1, ps1, ph1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) raw1 = gload(red0 + (kt + 3) * BK, ra1, rb1, ps1, ph1);
    compute(1);
    if (kt + 2 < nk) lstore(0, red0 + (kt + 2) * BK, raw0, ra0, rb0, ps0, ph0);
    __syncthreads();
  }

  if constexpr (KW > 1) {      // sum the KW partial accumulators in group order (deterministic) through LDS
    constexpr int TILE_F = MT * NT * 16 * 64;                  // floats one wave holds
    __syncthreads();                                           // staging buffers are dead from here on
    float* red = lds + (size_t)(wave % (WM * WN)) * (KW - 1) * TILE_F;
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(size_t)(wk - 1) * TILE_F + ((i * NT + j) * 16 + r) * 64 + l] = acc[i][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int q = 0; q < KW - 1; ++q)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(size_t)q * TILE_F + ((i * NT + j) * 16 + r) * 64 + l];
  }
  // ---- optional BatchNorm statistics of the OUTPUT (bias included), one (mean, M2) pair per column and per
  // 32-row block, merged later by Chan's formula (esc_bn_stats_from_partials): the following BatchNorm needs no
  // extra pass over Y.  Lane halves hold rows 4h..4h+3 (+8k): one cross-half shuffle completes a column.
  if (g.col_stats != nullptr) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + (l & 31);
        const int row0 = m0 + wm * TM + i * 32;
        const float bv = (g.bias && col < g.colsC) ? g.bias[col] : 0.f;
        const int nvalid = min(32, g.rowsC - row0);
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          if (row < g.rowsC) s1 += acc[i][j][r] + bv;
        }
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = nvalid > 0 ? s1 / (float)nvalid : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          if (row < g.rowsC) { const float d = acc[i][j][r] + bv - mean; m2 = fmaf(d, d, m2); }
        }
        m2 += __shfl_xor(m2, 32, 64);
        if (l < 32 && col < g.colsC && nvalid > 0) {
          float2* dst = g.col_stats + (size_t)(row0 / 32) * g.colsC + col;
          if (g.fin.tickets != nullptr) store_agent(dst, make_float2(mean, m2));   // read by another workgroup
          else *dst = make_float2(mean, m2);
        }
      }
  }
  // ---- epilogue: C/D map of the 32x32 block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cbase = g.C + (g.c_slab ? (size_t)split * g.rowsC * g.ldc : 0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * TN + j * 32 + (l & 31);
      if (col >= g.colsC) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (row < g.rowsC) {
          float* p = Cbase + (size_t)row * g.ldc + col;
          float v = acc[i][j][r] + bv;
          if (g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  if constexpr (DB) {
    if (bx == 0 && threadIdx.x < BM && m0 + (int)threadIdx.x < g.rowsC)
      g.db_part[(size_t)split * g.rowsC + m0 + threadIdx.x] = dbsum;
  }
  if constexpr (KW == 1 && !DB) {
    if (g.fin.tickets != nullptr) {      // uniform over the grid
      if (grid_last_block(g.fin.tickets + bx, (unsigned)g.fin.row_tiles)) bn_finalize_cols<BN, NTHR>(g, n0, lds);
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK, bool A_KC, bool B_KC, bool PRO, bool DB, int KW = 1>
__global__ __launch_bounds__(WM * WN * KW * 64) void gemm_tile_kernel(GemmArgs g) {
  ESC_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  gemm_tile_body<BM, BN, WM, WN, BK, A_KC, B_KC, PRO, DB, KW>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Backward of one Linear in ONE launch: the first gx workgroups compute dX = dY*W tiles, the rest the
// split-M dW = dY^T*act(X) slabs.  Both stream the same dY; fusing them removes a launch boundary and lets
// the two under-filled grids of the node-sized layers (152 + 304 workgroups) share the chip.
struct DualArgs { GemmArgs dx; GemmArgs dw; int dx_nx, dx_ny, dw_nx, dw_ny, dw_nz; };
template <int BM, int BN, int WM, int WN, int BK, bool PRO>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bwd_dual_kernel(DualArgs a) {
  ESC_PRIO();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x;
  const int n_dx = a.dx_nx * a.dx_ny;
  if (b < n_dx) {
    gemm_tile_body<BM, BN, WM, WN, BK, true, false, false, false>(a.dx, lds, b % a.dx_nx, b / a.dx_nx, 0);
  } else {
    const int r = b - n_dx;
    const int per = a.dw_nx * a.dw_ny;
    gemm_tile_body<BM, BN, WM, WN, BK, false, false, PRO, true>(a.dw, lds, (r % per) % a.dw_nx, (r % per) / a.dw_nx, r / per);
  }
}

// sum `splits` slabs in fixed order: out[i] = sum_s slab[s][i] for the N*K weight-gradient elements and,
// in the same launch, the N bias-gradient partials that follow them in every slab group
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t n, int splits,
                                                          int cols, float* __restrict__ out, int64_t ld_out,
                                                          const float* __restrict__ db_part, int rows,
                                                          float* __restrict__ db) {
  ESC_PRIO();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float s = 0.f;
#pragma unroll 4
    for (int q = 0; q < splits; ++q) s += slabs[(size_t)q * n + i];
    out[(i / cols) * ld_out + (i % cols)] = s;
  } else if (db != nullptr && i - n < rows) {
    const int64_t r = i - n;
    float s = 0.f;
    for (int q = 0; q < splits; ++q) s += db_part[(size_t)q * rows + r];
    db[r] = s;
  }
}

// every weight gradient of a training step reduced in ONE launch (the slabs are only needed by the optimiser):
// block b belongs to the job whose [block_start, block_start+blocks) range contains it
struct ReduceJobs {
  esc_reduce_job job[ESC_MAX_REDUCE_JOBS];
  int block_start[ESC_MAX_REDUCE_JOBS + 1];
  unsigned char vec[ESC_MAX_REDUCE_JOBS];     // 1: four consecutive gradient elements per thread (float4 slab reads)
  int count;
};
// One workgroup owns 64 consecutive UNITS of a job (a unit = four consecutive gradient elements when the job allows float4
// reads, one element otherwise; bias-gradient rows are further units); its four waves each add a quarter of the slabs
// in split order — batches of 8 reads in flight, a short last batch padded by clamping the slab index and masking the
// term, so that no wave ever walks a tail of dependent single loads — and wave 0 adds the four shares in wave order: a
// fixed association, bitwise reproducible.  (One thread per unit walking all 60-75 slabs left the launch latency-bound:
// 17 us for one edge-sized gradient, 20 us for the three 10-wide ones.)
template <int VEC>
__device__ __forceinline__ void slab_sum(const float* __restrict__ base, int64_t stride, int k0, int k1, int64_t off, bool live,
                                         float (&s)[VEC]) {
#pragma unroll
  for (int t = 0; t < VEC; ++t) s[t] = 0.f;
  if (!live) return;
  for (int k = k0; k < k1; k += 8) {
    float v[8][VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = min(k + u, k1 - 1);
      const float* p = base + (size_t)kk * stride + off;
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
      } else {
        v[u][0] = *p;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < k1) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) s[t] += v[u][t];
      }
  }
}

__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(ReduceJobs t) {
  ESC_PRIO();
  __shared__ float part[3][64][4];
  int j = 0;
  while (j + 1 < t.count && (int)blockIdx.x >= t.block_start[j + 1]) ++j;
  const esc_reduce_job& q = t.job[j];
  const int blk = (int)blockIdx.x - t.block_start[j];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = (q.splits + 3) / 4;
  const int k0 = min(q.splits, w * per), k1 = min(q.splits, k0 + per);
  const bool vec = t.vec[j] != 0;
  const int64_t units = vec ? q.n / 4 : q.n;
  const int64_t nblk = (units + 63) / 64;                   // blocks that cover the weight gradient; bias rows follow
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const bool bias = blk >= nblk;
  const int64_t i = bias ? (int64_t)(blk - nblk) * 64 + lane : (int64_t)blk * 64 + lane;
  const bool live = bias ? (q.db != nullptr && i < q.rows) : i < units;
  if (bias) {
    float o[1];
    slab_sum<1>(q.db_part, q.rows, k0, k1, i, live, o);
    s[0] = o[0];
  } else if (vec) {
    slab_sum<4>(q.slabs, q.n, k0, k1, 4 * i, live, s);
  } else {
    float o[1];
    slab_sum<1>(q.slabs, q.n, k0, k1, i, live, o);
    s[0] = o[0];
  }
  if (w > 0) { part[w - 1][lane][0] = s[0]; part[w - 1][lane][1] = s[1]; part[w - 1][lane][2] = s[2]; part[w - 1][lane][3] = s[3]; }
  __syncthreads();
  if (w != 0 || !live) return;
#pragma unroll
  for (int p = 0; p < 3; ++p) { s[0] += part[p][lane][0]; s[1] += part[p][lane][1]; s[2] += part[p][lane][2]; s[3] += part[p][lane][3]; }
  if (bias) {
    q.db[i] = s[0];
  } else if (vec) {
    const int64_t e = 4 * i;
    *reinterpret_cast<float4*>(q.dw + (e / q.cols) * q.ld_dw + (e % q.cols)) = make_float4(s[0], s[1], s[2], s[3]);
  } else {
    q.dw[(i / q.cols) * q.ld_dw + (i % q.cols)] = s[0];
  }
}

template <int BM, int BN, int WM, int WN, int BK, bool A_KC, bool B_KC, bool PRO, bool DB, int KW = 1>
static void launch_tile(const GemmArgs& g, int splits, hipStream_t s) {
  constexpr int NTHR = WM * WN * KW * 64;
  using ATile = typename std::conditional<A_KC, KContigTile<BM, BK, NTHR>, RedMajorTile<BM, BK, NTHR>>::type;
  using BTile = typename std::conditional<B_KC, KContigTile<BN, BK, NTHR>, RedMajorTile<BN, BK, NTHR>>::type;
  constexpr size_t lds_stage = 2 * (ATile::FLOATS + BTile::FLOATS) * sizeof(float);
  constexpr size_t lds_red = (size_t)WM * WN * (KW - 1) * (BM / WM / 32) * (BN / WN / 32) * 16 * 64 * sizeof(float);
  constexpr size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  auto kern = gemm_tile_kernel<BM, BN, WM, WN, BK, A_KC, B_KC, PRO, DB, KW>;
  if (lds > 64 * 1024) {
    static bool raised = false;          // per instantiation
    if (!raised) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); raised = true; }
  }
  dim3 grid((unsigned)cdiv(g.colsC, BN), (unsigned)cdiv(g.rowsC, BM), (unsigned)splits);
  const size_t floor_ = (size_t)gemm_lds_floor();
  const size_t use = lds > floor_ ? lds : floor_;
  if (use > 64 * 1024 && use > lds) {
    static size_t raised_to = 0;     // per instantiation
    if (use > raised_to) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)use); raised_to = use; }
  }
  esc::launch(ESC_K_LINEAR, kern, grid, dim3(NTHR), use, s, g);
}

// tile shapes, selectable per call site (esc_tune_set) — ids are stable
//   0: 128x128 BK32   1: 64x64 BK32   2: 128x32 BK32 (narrow outputs)   3: 128x64 BK32   4: 64x64 BK64
//   5: 32x64 BK32, 2 waves   6: 32x32 BK32, 1 wave   7: 64x32 BK32, 2 waves   (smaller workgroups: measured slower)
//   8: 32x32 tile, 4 wave groups splitting BK64   9: same with BK128   10: 64x32, 2 groups, BK64   (in-workgroup split-K)
#define ESC_TILE_DISPATCH(ID, AKC, BKC, PRO, DB)                                                     \
  switch (ID) {                                                                                      \
    case 0: launch_tile<128, 128, 2, 2, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                 \
    case 2: launch_tile<128, 32, 4, 1, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                  \
    case 3: launch_tile<128, 64, 2, 2, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                  \
    case 4: launch_tile<64, 64, 2, 2, 64, AKC, BKC, PRO, DB>(g, splits, s); break;                   \
    case 5: launch_tile<32, 64, 1, 2, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                   \
    case 6: launch_tile<32, 32, 1, 1, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                   \
    case 7: launch_tile<64, 32, 2, 1, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                   \
    case 8: if constexpr (!(DB)) { launch_tile<32, 32, 1, 1, 64, AKC, BKC, PRO, false, 4>(g, splits, s); break; } \
    case 9: if constexpr (!(DB)) { launch_tile<32, 32, 1, 1, 128, AKC, BKC, PRO, false, 4>(g, splits, s); break; } \
    case 10: if constexpr (!(DB)) { launch_tile<64, 32, 2, 1, 64, AKC, BKC, PRO, false, 2>(g, splits, s); break; } \
    default: launch_tile<64, 64, 2, 2, 32, AKC, BKC, PRO, DB>(g, splits, s); break;                  \
  }

static void tile_dims(int id, int* bm, int* bn, int* bk) {
  switch (id) {
    case 0: *bm = 128; *bn = 128; *bk = 32; break;
    case 2: *bm = 128; *bn = 32; *bk = 32; break;
    case 3: *bm = 128; *bn = 64; *bk = 32; break;
    case 4: *bm = 64; *bn = 64; *bk = 64; break;
    case 5: *bm = 32; *bn = 64; *bk = 32; break;
    case 6: *bm = 32; *bn = 32; *bk = 32; break;
    case 7: *bm = 64; *bn = 32; *bk = 32; break;
    default: *bm = 64; *bn = 64; *bk = 32; break;
  }
}

static inline bool vec_ok(const void* p, int64_t ld) { return aligned16(p) && (ld % 4 == 0); }

// =================================================================================================
// Narrow-output linears (N <= 4 output features, K <= 256): lin2 (H -> 1).  On the 128x32 MFMA tile this is 19
// workgroups that pad N to 32 and crawl (17 us forward, 19 us backward in two launches vs 5 / 8 us here); they are
// really bandwidth problems — X is read once — so: one wave per row batch, a lane owns one k-quad, the N weight
// rows sit in registers, dot products by wave reduction.  Backward: dX rows and the workgroup's share of dW / db in
// one pass; the shares are summed by the ordinary slab-reduce job (deterministic, one share per 128 rows).
// =================================================================================================
constexpr int NARROW_N = 4, NARROW_K = 256;                      // measured: at N = 10 the wave reductions cost more than the padded MFMA tile
constexpr int NARROW_ROWS = 32;                                  // rows per workgroup of linear_narrow_bwd: 2 400 rows = 75 workgroups (128 left 19 on 256 CUs)

// L1: the H -> 1 prediction head of a training step — the wave that has a node's prediction also leaves d|pred - y| / d pred for it
// (the same expression as l1_loss_kernel), so the backward does not wait for the loss launch
struct NarrowL1 { const float* target; float gs; float* dpred; };
template <int NMAX, bool PRO, bool FOLD = false, bool L1 = false>
__global__ __launch_bounds__(256) void linear_narrow_fwd(const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ sc, const float* __restrict__ sh,
                                                         int M, int N, int K, float* __restrict__ Y, int64_t ldy,
                                                         BnFoldDev fold, NarrowL1 l1) {
  ESC_PRIO();
  const int lane = lane_id();
  const int k = lane * 4;
  const bool valid = k < K;
  float4 w[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    w[n] = (valid && n < N) ? *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), ph = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO && FOLD) {           // the BatchNorm in front of X is still in partial form: merge it here (common.h)
    if (valid) {
      const bool writer = blockIdx.x == 0 && threadIdx.x < 64;
      bn_fold_column(fold, k + 0, writer, ps.x, ph.x); bn_fold_column(fold, k + 1, writer, ps.y, ph.y);
      bn_fold_column(fold, k + 2, writer, ps.z, ph.z); bn_fold_column(fold, k + 3, writer, ps.w, ph.w);
    }
  } else if constexpr (PRO) {
    if (valid) { ps = *reinterpret_cast<const float4*>(sc + k); ph = *reinterpret_cast<const float4*>(sh + k); }
  }
  const float bv = (bias != nullptr && lane < N) ? bias[lane] : 0.f;
  const int stride = gridDim.x * 4;
  for (int r0 = blockIdx.x * 4 + (threadIdx.x >> 6); r0 < M; r0 += 4 * stride) {
    float4 x[4];                                   // 4 rows in flight per wave
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * stride;
      x[u] = (valid && r < M) ? *reinterpret_cast<const float4*>(X + (size_t)r * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * stride;
      if (r >= M) break;                           // wave-uniform
      float4 v = x[u];
      if constexpr (PRO) {
        v = valid ? make_float4(fmaxf(fmaf(v.x, ps.x, ph.x), 0.f), fmaxf(fmaf(v.y, ps.y, ph.y), 0.f),
                                fmaxf(fmaf(v.z, ps.z, ph.z), 0.f), fmaxf(fmaf(v.w, ps.w, ph.w), 0.f))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float mine = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {                               // wave-uniform
          const float p = wave_sum(v.x * w[n].x + v.y * w[n].y + v.z * w[n].z + v.w * w[n].w);
          mine = (lane == n) ? p : mine;
        }
      }
      if (lane < N) Y[(size_t)r * ldy + lane] = mine + bv;
      if constexpr (L1) {
        if (lane == 0) {
          const float d = (mine + bv) - l1.target[r];
          l1.dpred[r] = d > 0.f ? l1.gs : (d < 0.f ? -l1.gs : 0.f);
        }
      }
    }
  }
}

template <int NMAX, bool PRO>
__global__ __launch_bounds__(256) void linear_narrow_bwd(const float* __restrict__ dY, int64_t lddy,
                                                         const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ sc, const float* __restrict__ sh,
                                                         int M, int N, int K, float* __restrict__ dX, int64_t lddx,
                                                         int accumulate, float* __restrict__ slab,
                                                         float* __restrict__ db_part) {
  ESC_PRIO();
  __shared__ float4 red[3][64];
  __shared__ float redb[3][NMAX];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int k = lane * 4;
  const bool valid = k < K;
  float4 w[NMAX], acc[NMAX];
  float dbacc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    w[n] = (valid && n < N) ? *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    dbacc[n] = 0.f;
  }
  float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), ph = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO) {
    if (valid) { ps = *reinterpret_cast<const float4*>(sc + k); ph = *reinterpret_cast<const float4*>(sh + k); }
  }
  const int row_end = min(M, (int)(blockIdx.x + 1) * NARROW_ROWS);
  for (int r0 = blockIdx.x * NARROW_ROWS + wave; r0 < row_end; r0 += 16) {     // 4 rows in flight per wave
    float4 x[4], old[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * 4;
      const bool live = valid && r < row_end;
      x[u] = live ? *reinterpret_cast<const float4*>(X + (size_t)r * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      old[u] = (live && dX != nullptr && accumulate) ? *reinterpret_cast<const float4*>(dX + (size_t)r * lddx + k)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * 4;
      if (r >= row_end) break;                     // wave-uniform
      float4 v = x[u];
      if constexpr (PRO) {
        v = valid ? make_float4(fmaxf(fmaf(v.x, ps.x, ph.x), 0.f), fmaxf(fmaf(v.y, ps.y, ph.y), 0.f),
                                fmaxf(fmaf(v.z, ps.z, ph.z), 0.f), fmaxf(fmaf(v.w, ps.w, ph.w), 0.f))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float4 d = old[u];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {
          const float g = dY[(size_t)r * lddy + n];              // same address in every lane: one broadcast load
          d.x = fmaf(g, w[n].x, d.x); d.y = fmaf(g, w[n].y, d.y); d.z = fmaf(g, w[n].z, d.z); d.w = fmaf(g, w[n].w, d.w);
          acc[n].x = fmaf(g, v.x, acc[n].x); acc[n].y = fmaf(g, v.y, acc[n].y);
          acc[n].z = fmaf(g, v.z, acc[n].z); acc[n].w = fmaf(g, v.w, acc[n].w);
          dbacc[n] += g;
        }
      }
      if (valid && dX != nullptr) *reinterpret_cast<float4*>(dX + (size_t)r * lddx + k) = d;
    }
  }
  // the four waves' shares, added in wave order, become this workgroup's slab
  float* out = slab + (size_t)blockIdx.x * N * K;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    if (n >= N) break;
    if (wave > 0) red[wave - 1][lane] = acc[n];
    __syncthreads();
    if (wave == 0 && valid) {
      float4 t = acc[n];
#pragma unroll
      for (int q = 0; q < 3; ++q) { const float4 o = red[q][lane]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      *reinterpret_cast<float4*>(out + (size_t)n * K + k) = t;
    }
    __syncthreads();
  }
  if (wave > 0 && lane == 0) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n) redb[wave - 1][n] = dbacc[n];
  }
  __syncthreads();
  if (wave == 0 && lane == 0) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) db_part[(size_t)blockIdx.x * N + n] = ((dbacc[n] + redb[0][n]) + redb[1][n]) + redb[2][n];
  }
}

// dX = dY[M,N] * W[N,K] for a short reduction (N <= 16: conv1.lin's input gradient, 15 200 x 256 from 10 features):
// an outer-product-like, purely bandwidth-bound pass — the MFMA tile pads N to a 32-deep K-step (29 us vs 8 us).
template <int NMAX>
__global__ __launch_bounds__(256) void linear_narrow_dx(const float* __restrict__ dY, int64_t lddy,
                                                        const float* __restrict__ W, int64_t ldw, int M, int N, int K,
                                                        float* __restrict__ dX, int64_t lddx, int accumulate) {
  ESC_PRIO();
  const int lane = lane_id();
  const int k = lane * 4;
  const bool valid = k < K;                                 // (every lane stays: lanes < N carry the dY values of a row)
  float4 w[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    w[n] = (valid && n < N) ? *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int stride = gridDim.x * 4;
  for (int r0 = blockIdx.x * 4 + (threadIdx.x >> 6); r0 < M; r0 += 4 * stride) {
    float4 d[4];
    float gy[4];                                            // lane n < N holds dY[r, n] of each of the wave's four rows: ONE load per row
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * stride;
      d[u] = (valid && accumulate && r < M) ? *reinterpret_cast<const float4*>(dX + (size_t)r * lddx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      gy[u] = (r < M && lane < N) ? dY[(size_t)r * lddy + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * stride;
      if (r >= M) break;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N) {
          const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gy[u]), n));
          d[u].x = fmaf(g, w[n].x, d[u].x); d[u].y = fmaf(g, w[n].y, d[u].y);
          d[u].z = fmaf(g, w[n].z, d[u].z); d[u].w = fmaf(g, w[n].w, d[u].w);
        }
      }
      if (valid) *reinterpret_cast<float4*>(dX + (size_t)r * lddx + k) = d[u];
    }
  }
}

static inline bool narrow_ok(int64_t N, int64_t K, const float* X, int64_t ld_x, const float* W, int64_t ld_w,
                             const float* sc, const float* sh) {
  return N <= NARROW_N && K <= NARROW_K && K % 4 == 0 && vec_ok(X, ld_x) && vec_ok(W, ld_w) &&
         (sc == nullptr || (aligned16(sc) && aligned16(sh)));
}

// tuning knobs (esc_tune_set): defaults chosen from scratch/gemm_bench.py sweeps on MI355X
enum { KNOB_FWD_BIG = 0, KNOB_FWD_SMALL = 1, KNOB_DX_BIG = 2, KNOB_DX_SMALL = 3, KNOB_DW_TILE = 4,
       KNOB_DW_BLOCKS = 5, KNOB_DW_MIN_ROWS = 6, KNOB_DUAL_SMALL = 7, KNOB_COUNT = 8 };
static int g_knob[KNOB_COUNT] = {1, 4, 1, 4, 4, 512, 128, 2};


// ---- dispatch to the LDS-DMA family (gemm_dma.h): the H-wide layers ---------------------------------------------
// knob 11 (default 1): 0 keeps every GEMM on the r01 register-staged tiles above (A/B runs in one process)
static int g_use_dma = 15;        // bit 0: forward, bit 1: gradients, bit 2: the tiny-dimension kernels (linear_small.h), bit 3: 64x32 narrow-output tile
static inline bool dma_ok(const void* p, int64_t rows, int64_t ld) {
  return aligned16(p) && ld % 4 == 0 && rows * ld * 4 < (1LL << 31);
}
// a 128-wide tile dimension over `dim` columns: acceptable when the padding to a multiple of 128 wastes <= 10 % of the MFMA
// work (256, 600 -> yes; 300 -> 384 is 28 % waste -> 64-wide tiles: 320)
static inline bool tile128_ok(int64_t dim) { return cdiv(dim, 128) * 128 * 10 <= dim * 11; }
// forward GEMM on 128-row tiles (4 compute + 4 loader waves, one workgroup per CU): edge-sized inputs.  Mid-sized launches that
// would still fill the chip with them (ogbg-mol node rows: 6 500 x 600 = 51 x 5 tiles) are faster ALONE on the big tile (32.8 ->
// 27.7 us) but slower inside the two-stream step (4.87 vs 4.77 ms: a one-workgroup-per-CU tile on the node stream shuts the
// edge stream's GEMMs out) — ESC_BIG_MIN_WGS=<workgroups> enables the rule for experiments.  The BatchNorm partials of the
// epilogue are per row tile, so esc_linear_stats_block_rows answers with the same predicate.
static inline bool dma_big(int64_t M, int64_t N) {
  if (N < 128) return false;
  if (M >= 8192) return true;
  static const int64_t min_wgs = getenv("ESC_BIG_MIN_WGS") ? atoll(getenv("ESC_BIG_MIN_WGS")) : (1LL << 62);
  return cdiv(M, 128) * cdiv(N, tile128_ok(N) ? 128 : 64) >= min_wgs;
}
// 300 / 600-wide layers (ogbg-mol emb_dim 300, its 2H hidden layer): a 128-row x 160-column tile (r03) pads them by 6.7 % at 2.2x the
// arithmetic intensity of the 64x64 tile they take (128-wide tiles would pad 300 by 28 %).  Built (reduction-major tiles with
// 640-byte rows: one DMA piece per row, 40 of 64 lanes active), correct (tests/test_hip_ops.py) and MEASURED SLOWER
// (profiles/r03_kernel_roofline_tile160.txt): one workgroup per CU and 157 x 2 = 314 tiles for 256 CUs leave the second round of
// workgroups on 58 CUs — 20000x300x300 forward 60.1 us against 51.5 us on the 128x64 tile, dX+dW 117 against 108 us, the
// config-5 step 4.34 against 4.27 ms.  OFF by default; ESC_TILE160=1 enables it for experiments.
static inline bool tile160_ok(int64_t dim) { return cdiv(dim, 160) * 160 * 10 <= dim * 11; }
static inline bool use160(int64_t rows, int64_t cols) {
  static const int on = getenv("ESC_TILE160") ? atoi(getenv("ESC_TILE160")) : 0;
  static const int64_t min_wgs = getenv("ESC_TILE160_MIN_WGS") ? atoll(getenv("ESC_TILE160_MIN_WGS")) : 150;
  return on && tile160_ok(cols) && cols % 128 != 0 && cdiv(rows, 128) * cdiv(cols, 160) >= min_wgs;
}
static inline hipError_t dma_check(hipError_t e, const char* what) {
  if (e != hipSuccess) set_error("%s: %s", what, hipGetErrorString(e));
  return e;
}
// split-M plan of the weight gradient for a BMxBN output tile: ~one workgroup per CU, splits >= 128 rows deep
static void dma_wgrad_plan(int64_t M, int64_t N, int64_t K, int bm, int bn, int* splits, int* per) {
  const int64_t tiles = cdiv(N, bm) * cdiv(K, bn);
  int64_t sp = cdiv(256, tiles);
  const int64_t max_sp = cdiv(M, 128);
  if (sp > max_sp) sp = max_sp;
  if (sp < 1) sp = 1;
  int64_t pr = cdiv(cdiv(M, sp), 32) * 32;
  if (pr < 128) pr = 128;
  *per = (int)pr;
  *splits = (int)cdiv(M, pr);
}
static bool dma_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias, const float* in_scale,
                    const float* in_shift, int64_t M, int64_t N, int64_t K, float* Y, int64_t ld_y, float* col_stats,
                    hipStream_t s, int* rc, const float* c_init = nullptr, int64_t ld_init = 0) {
  if (!(g_use_dma & 1) || K % 4 != 0 || K < 32 || !dma_ok(X, M, ld_x) || !dma_ok(W, N, ld_w) || (in_scale && cdiv(K, 32) * 32 > 1280)) return false;
  if (N <= 32 && (!(g_use_dma & 8) || col_stats != nullptr)) return false;
  dma::GArgs g{};
  g.A = X; g.lda = (int)ld_x; g.B = W; g.ldb = (int)ld_w; g.C = Y; g.ldc = (int)ld_y; g.bias = bias;
  g.pro_scale = in_scale; g.pro_shift = in_shift; g.col_stats = reinterpret_cast<float2*>(col_stats);
  g.M = (int)M; g.N = (int)N; g.R = (int)K; g.red_per_split = (int)K; g.accumulate = 0;
  g.c_init = c_init; g.ld_init = (int)ld_init;
  hipError_t e;
  if (c_init != nullptr) {         // esc_linear_fwd_from: the instantiations whose accumulators start from a partial result
    if (N <= 32 || !tile128_ok(N) && dma_big(M, N)) return false;
    if (dma_big(M, N)) e = in_scale ? dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, false, 1, true, false, true>(g, 0, s)
                                    : dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, false, 0, true, false, true>(g, 0, s);
    else               e = in_scale ? dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, false, 1, true, false, true>(g, 0, s)
                                    : dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, false, 0, true, false, true>(g, 0, s);
    *rc = dma_check(e, "esc_linear_fwd_from") == hipSuccess ? ESC_OK : ESC_ELAUNCH;
    return true;
  }
  if (N <= 32) {                   // narrow outputs (GINEConv.lin 256 -> 10): a 64x32 tile, bandwidth-bound on X
    e = in_scale ? dma::launch_gemm<64, 32, 32, 2, 1, 3, 2, false, false, 1, false, false>(g, 0, s)
                 : dma::launch_gemm<64, 32, 32, 2, 1, 3, 2, false, false, 0, false, false>(g, 0, s);
  } else if (use160(M, N)) {              // 300 / 600-wide outputs with enough row tiles: the 128x160 tile
    e = in_scale ? dma::launch_gemm<128, 160, 32, 4, 1, 3, 4, false, false, 1, true, false>(g, 0, s)
                 : dma::launch_gemm<128, 160, 32, 4, 1, 3, 4, false, false, 0, true, false>(g, 0, s);
  } else if (dma_big(M, N)) {             // edge-sized (or enough 128-row tiles to fill the chip): 128x128 tile, 4 compute + 4 loader waves
    if (in_scale)               e = dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, false, 1, true, false>(g, 0, s);
    else if (tile128_ok(N))     e = dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, false, 0, true, false>(g, 0, s);
    else                        e = dma::launch_gemm<128, 64, 32, 2, 2, 3, 2, false, false, 0, true, false>(g, 0, s);   // N = 300: 5 x 64 instead of 3 x 128
  } else {                         // node-sized: 64x64 tile, 4 compute + 2 loader waves
    e = in_scale ? dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, false, 1, true, false>(g, 0, s)
                 : dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, false, 0, true, false>(g, 0, s);
  }
  *rc = dma_check(e, "esc_linear_fwd") == hipSuccess ? ESC_OK : ESC_ELAUNCH;
  return true;
}
static bool dma_bwd_ok(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* W, int64_t ld_w,
                       int64_t M, int64_t N, int64_t K, const float* dX, int64_t ld_dx, const float* slabs, bool need_dx,
                       bool need_dw) {
  if (!(g_use_dma & 2) || N <= 32 || K <= 32 || N % 4 != 0 || K % 4 != 0 || !dma_ok(dY, M, ld_dy)) return false;
  if (need_dx && (!dma_ok(W, N, ld_w) || !dma_ok(dX, M, ld_dx))) return false;      // (N % 4 == 0 checked above: partial last K-step)
  if (need_dw && (!dma_ok(X, M, ld_x) || !aligned16(slabs))) return false;
  return true;
}
static void dma_fill_dx(dma::GArgs& g, const float* dY, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M, int64_t N,
                        int64_t K, float* dX, int64_t ld_dx, int accumulate) {
  g.A = dY; g.lda = (int)ld_dy; g.B = W; g.ldb = (int)ld_w; g.C = dX; g.ldc = (int)ld_dx;
  g.M = (int)M; g.N = (int)K; g.R = (int)N; g.red_per_split = (int)N; g.accumulate = accumulate;
}
static void dma_fill_dw(dma::GArgs& g, const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                        const float* in_shift, int64_t M, int64_t N, int64_t K, float* slabs, int splits, int per) {
  g.A = dY; g.lda = (int)ld_dy; g.B = X; g.ldb = (int)ld_x; g.C = slabs; g.ldc = (int)K;
  g.pro_scale = in_scale; g.pro_shift = in_shift; g.db_part = slabs + (size_t)splits * N * K;
  g.M = (int)N; g.N = (int)K; g.R = (int)M; g.red_per_split = per; g.accumulate = 0;
}
}  // namespace esc

using namespace esc;

extern "C" {

// resident workgroups per CU the runtime predicts for the forward kernel of a tile id (diagnostics)
int esc_debug_gemm_occupancy(int tile_id) {
  int n = -1;
#define ESC_OCC(BM_, BN_, WM_, WN_, BK_)                                                                  \
  {                                                                                                       \
    auto kern = gemm_tile_kernel<BM_, BN_, WM_, WN_, BK_, true, true, false, false>;                      \
    const size_t lds = 2 * (KContigTile<BM_, BK_>::FLOATS + KContigTile<BN_, BK_>::FLOATS) * sizeof(float); \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, lds);                               \
  }
  switch (tile_id) {
    case 0: ESC_OCC(128, 128, 2, 2, 32) break;
    case 2: ESC_OCC(128, 32, 4, 1, 32) break;
    case 3: ESC_OCC(128, 64, 2, 2, 32) break;
    case 4: ESC_OCC(64, 64, 2, 2, 64) break;
    default: ESC_OCC(64, 64, 2, 2, 32) break;
  }
#undef ESC_OCC
  return n;
}

int esc_tune_set(int knob, int value) {
  if (knob == 8) { set_last_block_finalize(value); return ESC_OK; }
  if (knob == 9) { set_norm_rowblock_cap(value); return ESC_OK; }
  if (knob == 12) { set_bn_bwd_fold(value); return ESC_OK; }
  if (knob == 13) { set_bn_bwd_one_launch(value); return ESC_OK; }
  if (knob == 10) { set_edge_lds_floor(value); return ESC_OK; }
  if (knob == 11) { g_use_dma = value; return ESC_OK; }
  ESC_REQUIRE(knob >= 0 && knob < KNOB_COUNT, "esc_tune_set: unknown knob %d", knob);
  g_knob[knob] = value;
  return ESC_OK;
}

#define ESC_TRY_(x) do { int rc__ = (x); if (rc__ != ESC_OK) return rc__; } while (0)
constexpr int64_t FUSE_FINALIZE_MAX_ROWS = 4096;

static int linear_fwd_impl(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                           const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                           float* Y, int64_t ld_y, float* col_stats, const esc_bn_fuse* bn, void* stream) {
  ESC_REQUIRE(X && W && Y, "esc_linear_fwd: null pointer");
  ESC_REQUIRE(M >= 0 && N > 0 && K > 0 && ld_x >= K && ld_w >= K && ld_y >= N, "esc_linear_fwd: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  ESC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "esc_linear_fwd: in_scale/in_shift must come together");
  ESC_REQUIRE(col_stats == nullptr || N > 32, "esc_linear_fwd: col_stats needs N > 32");
  ESC_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "esc_linear_fwd: dimension too large");
  ESC_REQUIRE(bn == nullptr || (col_stats && bn->mean && bn->invstd && M > 1), "esc_linear_bn_fwd: needs col_stats, mean, invstd and M > 1");
  ESC_REQUIRE(bn == nullptr || ((bn->scale == nullptr) == (bn->shift == nullptr)), "esc_linear_bn_fwd: scale/shift must come together");
  if (M == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (col_stats == nullptr && narrow_ok(N, K, X, ld_x, W, ld_w, in_scale, in_shift)) {
    const unsigned blocks = (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048);      // 4 rows per wave and pass
#define ESC_NARROW_FWD(NM) \
    if (in_scale) esc::launch(ESC_K_LINEAR, linear_narrow_fwd<NM, true>, dim3(blocks), dim3(256), 0, s, X, ld_x, W, ld_w, bias, in_scale, in_shift, (int)M, (int)N, (int)K, Y, ld_y, BnFoldDev{}, NarrowL1{}); \
    else          esc::launch(ESC_K_LINEAR, linear_narrow_fwd<NM, false>, dim3(blocks), dim3(256), 0, s, X, ld_x, W, ld_w, bias, in_scale, in_shift, (int)M, (int)N, (int)K, Y, ld_y, BnFoldDev{}, NarrowL1{})
    if (N == 1) { ESC_NARROW_FWD(1); } else { ESC_NARROW_FWD(4); }
#undef ESC_NARROW_FWD
    ESC_CHECK_LAUNCH("esc_linear_fwd.narrow");
    return ESC_OK;
  }
  if ((g_use_dma & 4) && K <= small::SMALL_MAX && in_scale == nullptr && N > 32) {     // in_dim-wide inputs: see linear_small.h
    esc::launch(ESC_K_LINEAR, small::smallk_fwd<small::SMALL_MAX>, dim3((unsigned)cdiv(M, small::ROWS_FWD), (unsigned)cdiv(N, 256)),
                dim3(256), 0, s, X, ld_x, W, ld_w, bias, (int)M, (int)N, (int)K, Y, ld_y, reinterpret_cast<float2*>(col_stats));
    ESC_CHECK_LAUNCH("esc_linear_fwd.smallk");
    if (bn == nullptr) return ESC_OK;
    return esc_bn_stats_from_partials_rows(col_stats, M, N, small::ROWS_FWD, bn->eps, bn->momentum, bn->mean, bn->invstd,
                                           bn->running_mean, bn->running_var, bn->gamma, bn->beta, bn->scale, bn->shift, stream);
  }
  if (bn == nullptr || M > FUSE_FINALIZE_MAX_ROWS || !last_block_finalize()) {
    int rc = ESC_OK;
    if (dma_fwd(X, ld_x, W, ld_w, bias, in_scale, in_shift, M, N, K, Y, ld_y, col_stats, s, &rc)) {
      if (rc != ESC_OK || bn == nullptr) return rc;
      return esc_bn_stats_from_partials_rows(col_stats, M, N, (dma_big(M, N) || use160(M, N)) ? 128 : 64, bn->eps, bn->momentum, bn->mean,
                                             bn->invstd, bn->running_mean, bn->running_var, bn->gamma, bn->beta, bn->scale,
                                             bn->shift, stream);
    }
  }
  GemmArgs g{};
  g.A = X; g.lda = ld_x; g.B = W; g.ldb = ld_w; g.C = Y; g.ldc = ld_y; g.bias = bias;
  g.pro_scale = in_scale; g.pro_shift = in_shift; g.db_part = nullptr;
  g.col_stats = reinterpret_cast<float2*>(col_stats);
  g.rowsC = (int)M; g.colsC = (int)N; g.red = (int)K; g.red_per_split = (int)K; g.accumulate = 0;
  g.a_vec = vec_ok(X, ld_x) && (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
  g.b_vec = vec_ok(W, ld_w); g.c_slab = 0;
  const int splits = 1;
  int id = (N <= 32) ? 2 : (M >= 8192 ? g_knob[KNOB_FWD_BIG] : g_knob[KNOB_FWD_SMALL]);
  // node-sized rows with a long reduction (lin1: K = (L+1)*H): 152 workgroups would each walk 20 K-steps alone;
  // the 32x32 tile whose 4 wave groups split every K-step puts 4x the waves on the chip (42 -> 32 us)
  if (N > 32 && M < 8192 && K >= 1024 && id == 4) id = 8;
  if (bn && (M > FUSE_FINALIZE_MAX_ROWS || !last_block_finalize())) {     // edge-sized: hundreds of partials per column — a wide finalize launch is faster
    ESC_TRY_(linear_fwd_impl(X, ld_x, W, ld_w, bias, in_scale, in_shift, M, N, K, Y, ld_y, col_stats, nullptr, stream));
    return esc_bn_stats_from_partials(col_stats, M, N, bn->eps, bn->momentum, bn->mean, bn->invstd, bn->running_mean,
                                      bn->running_var, bn->gamma, bn->beta, bn->scale, bn->shift, stream);
  }
  if (bn) {
    if (id >= 8) id = 4;                       // the fused finalize lives in the one-wave-group tiles only
    int bm, bn_cols, bk;
    tile_dims(id, &bm, &bn_cols, &bk);
    const int col_tiles = (int)cdiv(N, bn_cols);
    g.fin.tickets = tickets(col_tiles);
    ESC_REQUIRE(g.fin.tickets != nullptr, "esc_linear_bn_fwd: no ticket counters");
    g.fin.row_tiles = (int)cdiv(M, bm);
    g.fin.eps = bn->eps; g.fin.momentum = bn->momentum; g.fin.mean = bn->mean; g.fin.invstd = bn->invstd;
    g.fin.running_mean = bn->running_mean; g.fin.running_var = bn->running_var; g.fin.gamma = bn->gamma;
    g.fin.beta = bn->beta; g.fin.scale = bn->scale; g.fin.shift = bn->shift;
  }
  if (in_scale) { ESC_TILE_DISPATCH(id, true, true, true, false) }
  else          { ESC_TILE_DISPATCH(id, true, true, false, false) }
  ESC_CHECK_LAUNCH("esc_linear_fwd");
  return ESC_OK;
}

int esc_linear_fold_available(void) { return (g_use_dma & 1) != 0; }

int64_t esc_linear_stats_block_rows(const float* X, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int64_t N,
                                    int64_t K) {
  if ((g_use_dma & 4) && K <= small::SMALL_MAX && N > 32) return small::ROWS_FWD;
  if ((g_use_dma & 1) && K % 4 == 0 && K >= 32 && N > 32 && dma_ok(X, M, ld_x) && dma_ok(W, N, ld_w)) return (dma_big(M, N) || use160(M, N)) ? 128 : 64;
  return 32;
}

int esc_linear_fwd_fold(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                        const esc_bn_fold* in_bn, int64_t M, int64_t N, int64_t K, float* Y, int64_t ld_y,
                        float* col_stats, void* stream) {
  ESC_REQUIRE(X && W && Y && in_bn && in_bn->partials && in_bn->mean && in_bn->invstd, "esc_linear_fwd_fold: null pointer");
  ESC_REQUIRE(M > 0 && N > 0 && K > 0 && ld_x >= K && ld_w >= K && ld_y >= N && M < (1LL << 31), "esc_linear_fwd_fold: bad sizes");
  ESC_REQUIRE(in_bn->C == K && in_bn->rows > 1 && in_bn->block_rows > 0, "esc_linear_fwd_fold: the folded BatchNorm must have K channels");
  ESC_REQUIRE((in_bn->scale == nullptr) == (in_bn->shift == nullptr), "esc_linear_fwd_fold: scale/shift must come together");
  hipStream_t s = (hipStream_t)stream;
  BnFoldDev f{reinterpret_cast<const float2*>(in_bn->partials), (int)cdiv(in_bn->rows, in_bn->block_rows), (int)in_bn->block_rows,
              (int)in_bn->rows, (int)in_bn->C, in_bn->eps, in_bn->momentum, in_bn->gamma, in_bn->beta, in_bn->mean, in_bn->invstd,
              in_bn->scale, in_bn->shift, in_bn->running_mean, in_bn->running_var};
  if (col_stats == nullptr && N <= NARROW_N && K <= NARROW_K && K % 4 == 0 && vec_ok(X, ld_x) && vec_ok(W, ld_w)) {
    const unsigned blocks = (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048);
    if (N == 1) esc::launch(ESC_K_LINEAR, linear_narrow_fwd<1, true, true>, dim3(blocks), dim3(256), 0, s, X, ld_x, W, ld_w, bias, (const float*)nullptr, (const float*)nullptr, (int)M, (int)N, (int)K, Y, ld_y, f, NarrowL1{});
    else        esc::launch(ESC_K_LINEAR, linear_narrow_fwd<4, true, true>, dim3(blocks), dim3(256), 0, s, X, ld_x, W, ld_w, bias, (const float*)nullptr, (const float*)nullptr, (int)M, (int)N, (int)K, Y, ld_y, f, NarrowL1{});
    ESC_CHECK_LAUNCH("esc_linear_fwd_fold.narrow");
    return ESC_OK;
  }
  ESC_REQUIRE((g_use_dma & 1) && K % 4 == 0 && K >= 32 && cdiv(K, 32) * 32 <= 1280 && N > 32 && dma_ok(X, M, ld_x) && dma_ok(W, N, ld_w),
              "esc_linear_fwd_fold: shape not served by the folding kernels (M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
  dma::GArgs g{};
  g.A = X; g.lda = (int)ld_x; g.B = W; g.ldb = (int)ld_w; g.C = Y; g.ldc = (int)ld_y; g.bias = bias;
  g.col_stats = reinterpret_cast<float2*>(col_stats); g.fold = f;
  g.M = (int)M; g.N = (int)N; g.R = (int)K; g.red_per_split = (int)K; g.accumulate = 0;
  const hipError_t e = dma_big(M, N) ? dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, false, 3, true, false>(g, 0, s)
                                               : dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, false, 3, true, false>(g, 0, s);
  return dma_check(e, "esc_linear_fwd_fold") == hipSuccess ? ESC_OK : ESC_ELAUNCH;
}

int esc_linear_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                   const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                   float* Y, int64_t ld_y, float* col_stats, void* stream) {
  return linear_fwd_impl(X, ld_x, W, ld_w, bias, in_scale, in_shift, M, N, K, Y, ld_y, col_stats, nullptr, stream);
}

/* Y = Y0 + act(X) W^T + b with the BatchNorm partials of the RESULT in col_stats: the second half of a Linear whose reduction was cut
 * in two (see include/escgnn_hip.h).  LDS-DMA tiles only. */
int esc_linear_fwd_from(const float* Y0, int64_t ld_y0, const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                        const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K, float* Y, int64_t ld_y,
                        float* col_stats, void* stream) {
  ESC_REQUIRE(Y0 && X && W && Y, "esc_linear_fwd_from: null pointer");
  ESC_REQUIRE(M > 0 && N > 32 && K >= 32 && ld_y0 >= N && ld_x >= K && ld_w >= K && ld_y >= N && M < (1LL << 31), "esc_linear_fwd_from: bad sizes");
  ESC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "esc_linear_fwd_from: in_scale/in_shift must come together");
  int rc = ESC_OK;
  ESC_REQUIRE(dma_fwd(X, ld_x, W, ld_w, bias, in_scale, in_shift, M, N, K, Y, ld_y, col_stats, (hipStream_t)stream, &rc, Y0, ld_y0),
              "esc_linear_fwd_from: shape not served by the LDS-DMA tiles (M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
  return rc;
}
int esc_linear_fwd_from_ok(const float* X, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K, int has_prologue) {
  return ((g_use_dma & 1) && K % 4 == 0 && K >= 32 && N > 32 && dma_ok(X, M, ld_x) && dma_ok(W, N, ld_w) && !(has_prologue && cdiv(K, 32) * 32 > 1280) &&
          !(dma_big(M, N) && !tile128_ok(N))) ? 1 : 0;
}

/* pred = act(X) w^T + b for ONE output column (the H -> 1 head) and, in the same launch, dpred = d(sum |pred - target| * grad_scale /
 * denom) / d pred — what esc_l1_loss would leave in its dpred (same expression, bit for bit); see include/escgnn_hip.h */
int esc_linear_fwd_l1(const float* X, int64_t ld_x, const float* w, const float* bias, const float* in_scale, const float* in_shift,
                      int64_t M, int64_t K, const float* target, int64_t denom, float grad_scale, float* pred, float* dpred, void* stream) {
  ESC_REQUIRE(X && w && target && pred && dpred, "esc_linear_fwd_l1: null pointer");
  ESC_REQUIRE(M > 0 && K > 0 && ld_x >= K && denom > 0 && M < (1LL << 31), "esc_linear_fwd_l1: bad sizes");
  ESC_REQUIRE(in_scale != nullptr && in_shift != nullptr, "esc_linear_fwd_l1: the head reads pre-BatchNorm rows (in_scale / in_shift)");
  ESC_REQUIRE(narrow_ok(1, K, X, ld_x, w, K, in_scale, in_shift), "esc_linear_fwd_l1: shape / alignment not served (K=%ld)", (long)K);
  const unsigned blocks = (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048);
  const NarrowL1 l1{target, (float)((double)grad_scale / (double)denom), dpred};
  esc::launch(ESC_K_LINEAR, linear_narrow_fwd<1, true, false, true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ld_x, w, K, bias, in_scale,
              in_shift, (int)M, 1, (int)K, pred, (int64_t)1, BnFoldDev{}, l1);
  ESC_CHECK_LAUNCH("esc_linear_fwd_l1");
  return ESC_OK;
}
int esc_linear_fwd_l1_ok(const float* X, int64_t ld_x, const float* w, int64_t K, const float* in_scale, const float* in_shift) {
  return (in_scale && in_shift && narrow_ok(1, K, X, ld_x, w, K, in_scale, in_shift)) ? 1 : 0;
}

int esc_linear_bn_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                      const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                      float* Y, int64_t ld_y, float* col_stats, const esc_bn_fuse* bn, void* stream) {
  ESC_REQUIRE(bn != nullptr, "esc_linear_bn_fwd: null bn");
  return linear_fwd_impl(X, ld_x, W, ld_w, bias, in_scale, in_shift, M, N, K, Y, ld_y, col_stats, bn, stream);
}

int esc_linear_bwd_input(const float* dY, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                         int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate,
                         void* stream) {
  ESC_REQUIRE(dY && W && dX, "esc_linear_bwd_input: null pointer");
  ESC_REQUIRE(M >= 0 && N > 0 && K > 0 && ld_dy >= N && ld_w >= K && ld_dx >= K, "esc_linear_bwd_input: bad sizes");
  ESC_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "esc_linear_bwd_input: dimension too large");
  if (M == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N <= 16 && K <= NARROW_K && K % 4 == 0 && vec_ok(W, ld_w) && vec_ok(dX, ld_dx)) {
    const unsigned blocks = (unsigned)(cdiv(M, 16) < 4096 ? cdiv(M, 16) : 4096);
    if (N <= 4) esc::launch(ESC_K_LINEAR, linear_narrow_dx<4>, dim3(blocks), dim3(256), 0, s, dY, ld_dy, W, ld_w, (int)M, (int)N, (int)K, dX, ld_dx, accumulate);
    else        esc::launch(ESC_K_LINEAR, linear_narrow_dx<16>, dim3(blocks), dim3(256), 0, s, dY, ld_dy, W, ld_w, (int)M, (int)N, (int)K, dX, ld_dx, accumulate);
    ESC_CHECK_LAUNCH("esc_linear_bwd_input.narrow");
    return ESC_OK;
  }
  if ((g_use_dma & 4) && K <= small::SMALL_MAX && N % 4 == 0 && N <= 1024 && aligned16(dY) && ld_dy % 4 == 0) {
    const size_t lds = (size_t)(32 + small::SMALL_MAX) * (N + 4) * sizeof(float);
    static size_t raised_to = 64 * 1024;
    if (dma_check(dma::raise_lds(small::smalln_dx<small::SMALL_MAX>, lds, raised_to), "esc_linear_bwd_input") != hipSuccess) return ESC_ELAUNCH;
    esc::launch(ESC_K_LINEAR, small::smalln_dx<small::SMALL_MAX>, dim3((unsigned)cdiv(M, 32)), dim3(256), lds, s, dY, ld_dy, W, ld_w,
                (int)M, (int)N, (int)K, dX, ld_dx, accumulate, BnbDev{});
    ESC_CHECK_LAUNCH("esc_linear_bwd_input.smalln");
    return ESC_OK;
  }
  if (dma_bwd_ok(dY, ld_dy, nullptr, 0, W, ld_w, M, N, K, dX, ld_dx, nullptr, true, false)) {
    dma::GArgs d{};
    dma_fill_dx(d, dY, ld_dy, W, ld_w, M, N, K, dX, ld_dx, accumulate);
    const hipError_t e = use160(M, K) ? dma::launch_gemm<128, 160, 32, 4, 1, 3, 4, false, true, 0, false, false>(d, 0, s)
                         : (M >= 8192 && tile128_ok(K)) ? dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, false, true, 0, false, false>(d, 0, s)
                                   : dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, false, true, 0, false, false>(d, 0, s);
    return dma_check(e, "esc_linear_bwd_input") == hipSuccess ? ESC_OK : ESC_ELAUNCH;
  }
  GemmArgs g{};
  g.A = dY; g.lda = ld_dy; g.B = W; g.ldb = ld_w; g.C = dX; g.ldc = ld_dx; g.bias = nullptr;
  g.rowsC = (int)M; g.colsC = (int)K; g.red = (int)N; g.red_per_split = (int)N; g.accumulate = accumulate;
  g.a_vec = vec_ok(dY, ld_dy); g.b_vec = vec_ok(W, ld_w); g.c_slab = 0;
  const int splits = 1;
  const int id = (K <= 32) ? 2 : (M >= 8192 ? g_knob[KNOB_DX_BIG] : g_knob[KNOB_DX_SMALL]);
  ESC_TILE_DISPATCH(id, true, false, false, false)
  ESC_CHECK_LAUNCH("esc_linear_bwd_input");
  return ESC_OK;
}

static void wgrad_plan_tile(int64_t M, int64_t N, int64_t K, int bm, int bn, int bk, int* splits, int* per_split);

static void wgrad_plan(int64_t M, int64_t N, int64_t K, int* splits, int* per_split) {
  int bm, bn, bk;
  tile_dims(g_knob[KNOB_DW_TILE], &bm, &bn, &bk);
  wgrad_plan_tile(M, N, K, bm, bn, bk, splits, per_split);
}

static void wgrad_plan_tile(int64_t M, int64_t N, int64_t K, int bm, int bn, int bk, int* splits, int* per_split) {
  // enough splits along M for ~KNOB_DW_BLOCKS workgroups, each >= 128 rows deep (fixed by the shape
  // only, so scratch sizing and the launch agree)
  const int64_t tiles = cdiv(N, bm) * cdiv(K, bn);
  int64_t want = cdiv(g_knob[KNOB_DW_BLOCKS], tiles);
  int64_t max_splits = cdiv(M, g_knob[KNOB_DW_MIN_ROWS] < 128 ? 128 : g_knob[KNOB_DW_MIN_ROWS]);
  int64_t sp = want < 1 ? 1 : (want > max_splits ? max_splits : want);
  if (sp < 1) sp = 1;
  int64_t per = cdiv(cdiv(M, sp), bk) * bk;
  sp = cdiv(M, per);
  if (sp < 1) sp = 1;
  *splits = (int)sp;
  *per_split = (int)per;
}

int64_t esc_linear_bwd_weight_scratch(int64_t M, int64_t N, int64_t K) {
  // upper bound over every tunable plan: at most ceil(M/128) splits; the tiny-dimension kernels (linear_small.h) cut the
  // rows finer, their slabs are a few KB each
  if ((K <= small::SMALL_MAX) != (N <= small::SMALL_MAX)) return (cdiv(M, small::ROWS_WGRAD) + 1) * (N * K + N);
  return (cdiv(M, 128) + 1) * (N * K + N);
}

static int weight_impl(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                       const float* in_shift, int64_t M, int64_t N, int64_t K, float* dW, int64_t ld_dw, float* db,
                       float* slabs, esc_reduce_job* defer, void* stream);

int esc_linear_bwd_weight(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x,
                          const float* in_scale, const float* in_shift, int64_t M, int64_t N,
                          int64_t K, float* dW, int64_t ld_dw, float* db, float* slabs,
                          void* stream) {
  return weight_impl(dY, ld_dy, X, ld_x, in_scale, in_shift, M, N, K, dW, ld_dw, db, slabs, nullptr, stream);
}

static void fill_job(esc_reduce_job* j, const float* slabs, int64_t n, int splits, int64_t cols, float* dW, int64_t ld_dw,
                     const float* db_part, int64_t rows, float* db) {
  j->slabs = slabs; j->n = n; j->splits = splits; j->cols = cols; j->dw = dW; j->ld_dw = ld_dw;
  j->db_part = db_part; j->rows = rows; j->db = db;
}

static int weight_impl(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                       const float* in_shift, int64_t M, int64_t N, int64_t K, float* dW, int64_t ld_dw, float* db,
                       float* slabs, esc_reduce_job* defer, void* stream) {
  ESC_REQUIRE(dY && X && dW && slabs, "esc_linear_bwd_weight: null pointer");
  ESC_REQUIRE(M > 0 && N > 0 && K > 0 && ld_dy >= N && ld_x >= K && ld_dw >= K, "esc_linear_bwd_weight: bad sizes");
  ESC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "esc_linear_bwd_weight: in_scale/in_shift must come together");
  ESC_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "esc_linear_bwd_weight: dimension too large");
  hipStream_t s = (hipStream_t)stream;
  int splits, per;
  if ((g_use_dma & 4) && (K <= small::SMALL_MAX) != (N <= small::SMALL_MAX)) {        // one tiny feature dimension: linear_small.h
    splits = (int)cdiv(M, small::ROWS_WGRAD);
    float* db_part = slabs + (size_t)splits * N * K;
    const bool small_k = K <= small::SMALL_MAX;
    const dim3 grid((unsigned)splits, (unsigned)cdiv(small_k ? N : K, 256));
#define ESC_WGRAD_SMALL(SK, PR) esc::launch(ESC_K_LINEAR, small::wgrad_small<small::SMALL_MAX, SK, PR>, grid, dim3(256), 0, s, dY, ld_dy, X, ld_x, in_scale, in_shift, (int)M, (int)N, (int)K, slabs, db_part, BnbDev{})
    if (small_k) { if (in_scale) ESC_WGRAD_SMALL(true, true); else ESC_WGRAD_SMALL(true, false); }
    else         { if (in_scale) ESC_WGRAD_SMALL(false, true); else ESC_WGRAD_SMALL(false, false); }
#undef ESC_WGRAD_SMALL
    ESC_CHECK_LAUNCH("esc_linear_bwd_weight.small");
    const int64_t n = N * K;
    if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, db_part, N, db); return ESC_OK; }
    esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
                splits, (int)K, dW, ld_dw, db_part, (int)N, db);
    ESC_CHECK_LAUNCH("esc_linear_bwd_weight.reduce");
    return ESC_OK;
  }
  if (dma_bwd_ok(dY, ld_dy, X, ld_x, nullptr, 0, M, N, K, nullptr, 0, slabs, false, true)) {
    const bool big = M >= 8192 && tile128_ok(N) && tile128_ok(K);
    dma_wgrad_plan(M, N, K, big ? 128 : 64, big ? 128 : 64, &splits, &per);
    dma::GArgs d{};
    dma_fill_dw(d, dY, ld_dy, X, ld_x, in_scale, in_shift, M, N, K, slabs, splits, per);
    hipError_t e;
    if (big) e = in_scale ? dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, true, true, 2, false, true>(d, 0, s)
                          : dma::launch_gemm<128, 128, 32, 2, 2, 3, 4, true, true, 0, false, true>(d, 0, s);
    else     e = in_scale ? dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, true, true, 2, false, true>(d, 0, s)
                          : dma::launch_gemm<64, 64, 32, 2, 2, 3, 2, true, true, 0, false, true>(d, 0, s);
    if (dma_check(e, "esc_linear_bwd_weight") != hipSuccess) return ESC_ELAUNCH;
    const int64_t n = N * K;
    if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, d.db_part, N, db); return ESC_OK; }
    esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
                splits, (int)K, dW, ld_dw, d.db_part, (int)N, db);
    ESC_CHECK_LAUNCH("esc_linear_bwd_weight.reduce");
    return ESC_OK;
  }
  wgrad_plan(M, N, K, &splits, &per);
  GemmArgs g{};
  g.A = dY; g.lda = ld_dy; g.B = X; g.ldb = ld_x; g.C = slabs; g.ldc = K; g.bias = nullptr;
  g.pro_scale = in_scale; g.pro_shift = in_shift;
  g.db_part = slabs + (size_t)splits * N * K;
  g.rowsC = (int)N; g.colsC = (int)K; g.red = (int)M; g.red_per_split = per; g.accumulate = 0;
  g.a_vec = vec_ok(dY, ld_dy);
  g.b_vec = vec_ok(X, ld_x) && (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
  g.c_slab = 1;
  const int id = g_knob[KNOB_DW_TILE];
  if (in_scale) { ESC_TILE_DISPATCH(id, false, false, true, true) }
  else          { ESC_TILE_DISPATCH(id, false, false, false, true) }
  ESC_CHECK_LAUNCH("esc_linear_bwd_weight.tiles");
  const int64_t n = N * K;
  if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, g.db_part, N, db); return ESC_OK; }
  esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
              splits, (int)K, dW, ld_dw, g.db_part, (int)N, db);
  ESC_CHECK_LAUNCH("esc_linear_bwd_weight.reduce");
  return ESC_OK;
}

}  // extern "C"

// ---- weight-gradient tiles on a stream of their own (esc_linear_bwd_set_wgrad_stream) -------------------------------------
// Inside a step the node chain waits for a Linear backward's dX only; its dW slabs are needed by the optimiser.  While a side
// stream is set (thread-local), the node-sized dual launches with a DEFERRED slab reduce put their dW tiles there, ordered
// behind everything queued on the launch stream so far (one event from a small per-device pool per launch).
struct WgradSide {
  hipStream_t stream = nullptr;
  std::map<int, std::array<hipEvent_t, 16>> events;     // per device
  int next = 0;
};
static thread_local WgradSide g_wside;
static hipStream_t wgrad_stream_after(hipStream_t s) {
  WgradSide& w = g_wside;
  if (w.stream == nullptr || w.stream == s) return s;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return s;
  auto it = w.events.find(dev);
  if (it == w.events.end()) it = w.events.emplace(dev, std::array<hipEvent_t, 16>{}).first;
  hipEvent_t& e = it->second[w.next];
  w.next = (w.next + 1) & 15;
  if (e == nullptr && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return s; }
  if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(w.stream, e, 0) != hipSuccess) return s;
  return w.stream;
}

template <int BM, int BN, int WM, int WN, int BK, bool PRO>
static void launch_dual(const DualArgs& a, hipStream_t s) {
  constexpr int NTHR = WM * WN * 64;
  constexpr size_t lds = 2 * (size_t)(KContigTile<BM, BK, NTHR>::FLOATS + RedMajorTile<BN, BK, NTHR>::FLOATS) * sizeof(float);
  constexpr size_t lds2 = 2 * (size_t)(RedMajorTile<BM, BK, NTHR>::FLOATS + RedMajorTile<BN, BK, NTHR>::FLOATS) * sizeof(float);
  constexpr size_t need = lds > lds2 ? lds : lds2;
  auto kern = gemm_bwd_dual_kernel<BM, BN, WM, WN, BK, PRO>;
  if (need > 64 * 1024) {
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need); raised = true; }
  }
  const unsigned blocks = (unsigned)(a.dx_nx * a.dx_ny + a.dw_nx * a.dw_ny * a.dw_nz);
  const size_t floor_ = (size_t)gemm_lds_floor();        // occupancy cap requested by the caller (see common.h)
  const size_t use = need > floor_ ? need : floor_;
  if (use > 64 * 1024 && use > need) {
    static size_t raised_to = 0;
    if (use > raised_to) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)use); raised_to = use; }
  }
  esc::launch(ESC_K_LINEAR, kern, dim3(blocks), dim3(NTHR), use, s, a);
}

extern "C" {

static int both_impl(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                     const float* in_shift, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K,
                     float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw, float* db,
                     float* slabs, esc_reduce_job* defer, void* stream);

int esc_linear_bwd_both(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                        const float* in_shift, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K,
                        float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw, float* db,
                        float* slabs, void* stream) {
  return both_impl(dY, ld_dy, X, ld_x, in_scale, in_shift, W, ld_w, M, N, K, dX, ld_dx, accumulate, dW, ld_dw, db, slabs,
                   nullptr, stream);
}

/* same, but the ordered slab reduce is NOT launched: its description is returned in *job for esc_slab_reduce_jobs.
 * `slabs` must then stay untouched until that call. */
int esc_linear_bwd_both_deferred(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                                 const float* in_shift, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K,
                                 float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw, float* db,
                                 float* slabs, esc_reduce_job* job, void* stream) {
  ESC_REQUIRE(job, "esc_linear_bwd_both_deferred: null job");
  return both_impl(dY, ld_dy, X, ld_x, in_scale, in_shift, W, ld_w, M, N, K, dX, ld_dx, accumulate, dW, ld_dw, db, slabs,
                   job, stream);
}

int esc_slab_reduce_jobs(const esc_reduce_job* jobs, int count, void* stream) {
  ESC_REQUIRE(jobs && count > 0 && count <= ESC_MAX_REDUCE_JOBS, "esc_slab_reduce_jobs: 1..%d jobs", ESC_MAX_REDUCE_JOBS);
  ReduceJobs t{};
  t.count = count;
  int blocks = 0;
  for (int j = 0; j < count; ++j) {
    ESC_REQUIRE(jobs[j].slabs && jobs[j].dw && jobs[j].n > 0 && jobs[j].splits > 0, "esc_slab_reduce_jobs: bad job %d", j);
    t.job[j] = jobs[j];
    t.block_start[j] = blocks;
    const esc_reduce_job& q = jobs[j];
    const bool vec = q.n % 4 == 0 && q.cols % 4 == 0 && q.ld_dw % 4 == 0 && aligned16(q.slabs) && aligned16(q.dw);
    t.vec[j] = vec ? 1 : 0;
    blocks += (int)(cdiv(vec ? q.n / 4 : q.n, 64) + (q.db ? cdiv(q.rows, 64) : 0));
  }
  t.block_start[count] = blocks;
  esc::launch(ESC_K_LINEAR, slab_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
  ESC_CHECK_LAUNCH("esc_slab_reduce_jobs");
  return ESC_OK;
}

static int both_impl(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x, const float* in_scale,
                     const float* in_shift, const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K,
                     float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw, float* db,
                     float* slabs, esc_reduce_job* defer, void* stream) {
  ESC_REQUIRE(dY && X && W && dW && slabs, "esc_linear_bwd_both: null pointer");
  if (M > 0 && M < (1LL << 31) && ld_dy >= N && ld_dw >= K && (dX == nullptr || (vec_ok(dX, ld_dx) && ld_dx >= K)) &&
      aligned16(slabs) && narrow_ok(N, K, X, ld_x, W, ld_w, in_scale, in_shift) &&
      (in_scale == nullptr) == (in_shift == nullptr)) {
    // dX rows and the dW / db shares in one pass over X and dY (see linear_narrow_bwd)
    hipStream_t s = (hipStream_t)stream;
    const int splits = (int)cdiv(M, NARROW_ROWS);
    float* db_part = slabs + (size_t)splits * N * K;
#define ESC_NARROW_BWD(NM) \
    if (in_scale) esc::launch(ESC_K_LINEAR, linear_narrow_bwd<NM, true>, dim3(splits), dim3(256), 0, s, dY, ld_dy, X, ld_x, W, ld_w, in_scale, in_shift, (int)M, (int)N, (int)K, dX, ld_dx, accumulate, slabs, db_part); \
    else          esc::launch(ESC_K_LINEAR, linear_narrow_bwd<NM, false>, dim3(splits), dim3(256), 0, s, dY, ld_dy, X, ld_x, W, ld_w, in_scale, in_shift, (int)M, (int)N, (int)K, dX, ld_dx, accumulate, slabs, db_part)
    if (N == 1) { ESC_NARROW_BWD(1); } else { ESC_NARROW_BWD(4); }
#undef ESC_NARROW_BWD
    ESC_CHECK_LAUNCH("esc_linear_bwd_both.narrow");
    const int64_t n = N * K;
    if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, db_part, N, db); return ESC_OK; }
    esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
                splits, (int)K, dW, ld_dw, db_part, (int)N, db);
    ESC_CHECK_LAUNCH("esc_linear_bwd_both.narrow_reduce");
    return ESC_OK;
  }
  if (dX != nullptr && M > 0 && M < (1LL << 31) && ld_dw >= K && (in_scale == nullptr) == (in_shift == nullptr) &&
      dma_bwd_ok(dY, ld_dy, X, ld_x, W, ld_w, M, N, K, dX, ld_dx, slabs, true, true)) {
    // dX tiles + split-M dW slabs of the LDS-DMA family in ONE launch
    hipStream_t s = (hipStream_t)stream;
    const bool big = M >= 8192 && tile128_ok(N) && tile128_ok(K);
    const bool t160 = use160(M, K);            // dX tiles 128 rows x 160 of the K columns; the dW job rides on the same tile over [N, K]
    int splits, per;
    dma_wgrad_plan(M, N, K, (big || t160) ? 128 : 64, t160 ? 160 : (big ? 128 : 64), &splits, &per);
    dma::DualArgs a{};
    dma_fill_dx(a.dx, dY, ld_dy, W, ld_w, M, N, K, dX, ld_dx, accumulate);
    dma_fill_dw(a.dw, dY, ld_dy, X, ld_x, in_scale, in_shift, M, N, K, slabs, splits, per);
    hipError_t e;
    if (t160) e = in_scale ? dma::launch_dual<128, 160, 32, 4, 1, 3, 4, true>(a, 0, s)
                           : dma::launch_dual<128, 160, 32, 4, 1, 3, 4, false>(a, 0, s);
    else if (big) e = in_scale ? dma::launch_dual<128, 128, 32, 2, 2, 3, 4, true>(a, 0, s)
                          : dma::launch_dual<128, 128, 32, 2, 2, 3, 4, false>(a, 0, s);
    else {
      hipStream_t sw = defer ? wgrad_stream_after(s) : s;
      e = in_scale ? dma::launch_dual<64, 64, 32, 2, 2, 3, 2, true>(a, 0, s, ESC_K_LINEAR, sw) : dma::launch_dual<64, 64, 32, 2, 2, 3, 2, false>(a, 0, s, ESC_K_LINEAR, sw);
    }
    if (dma_check(e, "esc_linear_bwd_both") != hipSuccess) return ESC_ELAUNCH;
    const int64_t n = N * K;
    if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, a.dw.db_part, N, db); return ESC_OK; }
    esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
                splits, (int)K, dW, ld_dw, a.dw.db_part, (int)N, db);
    ESC_CHECK_LAUNCH("esc_linear_bwd_both.reduce");
    return ESC_OK;
  }
  if (dX == nullptr || N <= 32 || K <= 32) {             // other narrow shapes keep their dedicated tiles
    int rc = weight_impl(dY, ld_dy, X, ld_x, in_scale, in_shift, M, N, K, dW, ld_dw, db, slabs, defer, stream);
    if (rc || dX == nullptr) return rc;
    return esc_linear_bwd_input(dY, ld_dy, W, ld_w, M, N, K, dX, ld_dx, accumulate, stream);
  }
  ESC_REQUIRE(M > 0 && ld_dy >= N && ld_x >= K && ld_w >= K && ld_dx >= K && ld_dw >= K, "esc_linear_bwd_both: bad sizes");
  ESC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "esc_linear_bwd_both: in_scale/in_shift must come together");
  ESC_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "esc_linear_bwd_both: dimension too large");
  hipStream_t s = (hipStream_t)stream;
  // edge-sized: 64x64xBK32 (4 workgroups/CU); node-sized: KNOB_DUAL_SMALL picks 64x64xBK64 (0) or the 2-wave
  // 32x64xBK32 tile (1) that doubles the workgroup count of these under-filled grids
  const int small_tile = g_knob[KNOB_DUAL_SMALL];
  const int bm = (M >= 8192 || small_tile != 1) ? 64 : 32, bn = 64;
  const int bk = (M >= 8192 || small_tile != 0) ? 32 : 64;         // knob 7: 0 64x64xBK64, 1 32x64xBK32, 2 64x64xBK32
  int splits, per;
  wgrad_plan_tile(M, N, K, bm, bn, bk, &splits, &per);
  DualArgs a{};
  GemmArgs& g = a.dx;
  g.A = dY; g.lda = ld_dy; g.B = W; g.ldb = ld_w; g.C = dX; g.ldc = ld_dx;
  g.rowsC = (int)M; g.colsC = (int)K; g.red = (int)N; g.red_per_split = (int)N; g.accumulate = accumulate;
  g.a_vec = vec_ok(dY, ld_dy); g.b_vec = vec_ok(W, ld_w); g.c_slab = 0;
  GemmArgs& w = a.dw;
  w.A = dY; w.lda = ld_dy; w.B = X; w.ldb = ld_x; w.C = slabs; w.ldc = K;
  w.pro_scale = in_scale; w.pro_shift = in_shift; w.db_part = slabs + (size_t)splits * N * K;
  w.rowsC = (int)N; w.colsC = (int)K; w.red = (int)M; w.red_per_split = per; w.accumulate = 0;
  w.a_vec = vec_ok(dY, ld_dy);
  w.b_vec = vec_ok(X, ld_x) && (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
  w.c_slab = 1;
  a.dx_nx = (int)cdiv(K, bn); a.dx_ny = (int)cdiv(M, bm);
  a.dw_nx = (int)cdiv(K, bn); a.dw_ny = (int)cdiv(N, bm); a.dw_nz = splits;
  if (bm == 32)      { if (in_scale) launch_dual<32, 64, 1, 2, 32, true>(a, s); else launch_dual<32, 64, 1, 2, 32, false>(a, s); }
  else if (bk == 32) { if (in_scale) launch_dual<64, 64, 2, 2, 32, true>(a, s); else launch_dual<64, 64, 2, 2, 32, false>(a, s); }
  else               { if (in_scale) launch_dual<64, 64, 2, 2, 64, true>(a, s); else launch_dual<64, 64, 2, 2, 64, false>(a, s); }
  ESC_CHECK_LAUNCH("esc_linear_bwd_both.tiles");
  const int64_t n = N * K;
  if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, w.db_part, N, db); return ESC_OK; }
  esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
              splits, (int)K, dW, ld_dw, w.db_part, (int)N, db);
  ESC_CHECK_LAUNCH("esc_linear_bwd_both.reduce");
  return ESC_OK;
}


// ---- Linear backward with the BatchNorm(+ReLU) backward of its dY folded in (include/escgnn_hip.h, r03) ----------------
static inline BnbDev bnb_dev(const esc_bn_bwd_fused* b) {
  return BnbDev{b->x, (int)b->ld_x, b->mean, b->invstd, b->scale, b->shift, reinterpret_cast<const float2*>(b->coef), b->relu};
}
static inline bool bnb_operands_ok(const esc_bn_bwd_fused* b, int64_t M, int64_t N) {
  return b && b->x && b->mean && b->invstd && b->scale && b->shift && b->coef && b->relu >= 0 && b->relu <= 2 && b->ld_x >= N &&
         dma_ok(b->x, M, b->ld_x) && aligned16(b->mean) && aligned16(b->invstd) && aligned16(b->scale) && aligned16(b->shift) &&
         aligned16(b->coef);
}
static inline bool bnb_big_shape(int64_t M, int64_t N, int64_t K) { return M >= 8192 && tile128_ok(N) && tile128_ok(K); }
static inline bool in_range_big(int64_t N) { return N <= 640; }
static inline bool bnb_small_shape(int64_t N, int64_t K) { return (g_use_dma & 4) && K <= small::SMALL_MAX && N > small::SMALL_MAX && N % 4 == 0 && N <= 1024; }

int esc_linear_bwd_set_wgrad_stream(void* stream) {
  g_wside.stream = (hipStream_t)stream;
  return ESC_OK;
}

int esc_linear_bwd_both_bn_ok(const float* dOut, int64_t ld_dout, const esc_bn_bwd_fused* bn, const float* X, int64_t ld_x,
                              const float* W, int64_t ld_w, int64_t M, int64_t N, int64_t K, const float* dX, int64_t ld_dx,
                              const float* slabs, const esc_bn_bwd_next* next) {
  if (!dOut || !X || !W || !slabs || M <= 1 || N <= 0 || K <= 0 || (bn == nullptr && next == nullptr)) return 0;
  // edge-sized rows ride on the 128x128 tile (4 compute + 4 loader waves, one workgroup per CU, three ring stages even with the
  // third operand image): only the square-ish H-wide layers it serves; everything else is node-sized
  if (M >= 8192 && !(bnb_big_shape(M, N, K) && bn != nullptr && in_range_big(N))) return 0;
  if (bn != nullptr && !bnb_operands_ok(bn, M, N)) return 0;
  if (bn != nullptr && bnb_small_shape(N, K))
    return next == nullptr && aligned16(dOut) && ld_dout % 4 == 0 && ld_dout >= N && ld_x >= K && ld_w >= K && (dX == nullptr || ld_dx >= K);
  if (dX == nullptr || N > dma::Cfg<64, 64, 32, 2, 2, 2, 2, false, true, 4, false, false>::BNB_MAXK) return 0;
  if (!dma_bwd_ok(dOut, ld_dout, X, ld_x, W, ld_w, M, N, K, dX, ld_dx, slabs, true, true)) return 0;
  if (next) {
    if (!next->partial || !next->x || !next->mean || !next->invstd || !next->scale || !next->shift || next->ld_x < K || next->ld_x % 4 != 0 ||
        !aligned16(next->x) || !aligned16(next->mean) || !aligned16(next->invstd) || !aligned16(next->scale) || !aligned16(next->shift) ||
        !aligned16(next->partial) || K % 4 != 0 || ld_dx % 4 != 0 || !aligned16(dX) || next->relu < 0 || next->relu > 2)
      return 0;
  }
  return 1;
}

int64_t esc_linear_bwd_bn_block_rows(int64_t M, int64_t N, int64_t K) { return bnb_big_shape(M, N, K) ? 128 : 64; }

int esc_linear_bwd_both_bn(const float* dOut, int64_t ld_dout, const esc_bn_bwd_fused* bn, const float* X, int64_t ld_x,
                           const float* in_scale, const float* in_shift, const float* W, int64_t ld_w, int64_t M,
                           int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate, float* dW, int64_t ld_dw,
                           float* db, float* slabs, esc_reduce_job* defer, const esc_bn_bwd_next* next, void* stream) {
  ESC_REQUIRE(dOut && X && W && dW && slabs, "esc_linear_bwd_both_bn: null pointer");
  ESC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "esc_linear_bwd_both_bn: in_scale/in_shift must come together");
  ESC_REQUIRE(ld_dw >= K, "esc_linear_bwd_both_bn: bad sizes");
  ESC_REQUIRE(esc_linear_bwd_both_bn_ok(dOut, ld_dout, bn, X, ld_x, W, ld_w, M, N, K, dX, ld_dx, slabs, next),
              "esc_linear_bwd_both_bn: shape / alignment not served (M=%ld N=%ld K=%ld) - use esc_bn_bwd + esc_linear_bwd_both", (long)M, (long)N, (long)K);
  hipStream_t s = (hipStream_t)stream;
  const BnbDev bd = bn ? bnb_dev(bn) : BnbDev{};
  const int64_t n = N * K;
  if (bn && bnb_small_shape(N, K)) {                       // in_dim-wide Linear (x_embedding.0, conv1.nn.0): see linear_small.h
    const int splits = (int)cdiv(M, small::ROWS_WGRAD);
    float* db_part = slabs + (size_t)splits * N * K;
    const dim3 grid((unsigned)splits, (unsigned)cdiv(N, 256));
    hipStream_t sw = defer ? wgrad_stream_after(s) : s;
#define ESC_WG(PR, ACT) esc::launch(ESC_K_LINEAR, small::wgrad_small<small::SMALL_MAX, true, PR, ACT>, grid, dim3(256), 0, sw, dOut, ld_dout, X, ld_x, in_scale, in_shift, (int)M, (int)N, (int)K, slabs, db_part, bd)
    if (bd.relu == 2) { if (in_scale) ESC_WG(true, 2); else ESC_WG(false, 2); }
    else              { if (in_scale) ESC_WG(true, 1); else ESC_WG(false, 1); }
#undef ESC_WG
    ESC_CHECK_LAUNCH("esc_linear_bwd_both_bn.small");
    if (dX != nullptr) {
      const size_t lds = (size_t)(32 + small::SMALL_MAX) * (N + 4) * sizeof(float);
      static size_t raised_to = 64 * 1024;
      auto kern = bd.relu == 2 ? small::smalln_dx<small::SMALL_MAX, 2> : small::smalln_dx<small::SMALL_MAX, 1>;
      if (dma_check(dma::raise_lds(small::smalln_dx<small::SMALL_MAX, 1>, lds, raised_to), "esc_linear_bwd_both_bn") != hipSuccess) return ESC_ELAUNCH;
      static size_t raised_elu = 64 * 1024;
      if (bd.relu == 2 && dma_check(dma::raise_lds(small::smalln_dx<small::SMALL_MAX, 2>, lds, raised_elu), "esc_linear_bwd_both_bn") != hipSuccess) return ESC_ELAUNCH;
      esc::launch(ESC_K_LINEAR, kern, dim3((unsigned)cdiv(M, 32)), dim3(256), lds, s, dOut, ld_dout, W, ld_w, (int)M, (int)N, (int)K, dX, ld_dx, accumulate, bd);
      ESC_CHECK_LAUNCH("esc_linear_bwd_both_bn.smalln");
    }
    if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, db_part, N, db); return ESC_OK; }
    esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
                splits, (int)K, dW, ld_dw, db_part, (int)N, db);
    ESC_CHECK_LAUNCH("esc_linear_bwd_both_bn.reduce");
    return ESC_OK;
  }
  const bool big = bnb_big_shape(M, N, K);
  int splits, per;
  dma_wgrad_plan(M, N, K, big ? 128 : 64, big ? 128 : 64, &splits, &per);
  dma::DualArgs a{};
  dma_fill_dx(a.dx, dOut, ld_dout, W, ld_w, M, N, K, dX, ld_dx, accumulate);
  dma_fill_dw(a.dw, dOut, ld_dout, X, ld_x, in_scale, in_shift, M, N, K, slabs, splits, per);
  a.dx.bnb = bd; a.dw.bnb = bd;
  if (next)
    a.dx.bst = BnStatDev{reinterpret_cast<float2*>(next->partial), next->x, (int)next->ld_x, next->mean, next->invstd, next->scale, next->shift, next->relu};
  hipError_t e;
  // (the third operand image of the fused apply costs a ring stage: two stages keep the workgroup at 54 KB, which fits beside
  // an edge-stream GEMM on a CU; three stages — 78 KB — measured slower inside the two-stream step, and neutral (0.988 vs 0.985 ms)
  // once the backward's node workgroups keep off the edge GEMMs' CUs altogether)
  if (big) {
    if (next) e = in_scale ? dma::launch_dual<128, 128, 32, 2, 2, 3, 4, true, true, true>(a, 0, s) : dma::launch_dual<128, 128, 32, 2, 2, 3, 4, false, true, true>(a, 0, s);
    else      e = in_scale ? dma::launch_dual<128, 128, 32, 2, 2, 3, 4, true, true, false>(a, 0, s) : dma::launch_dual<128, 128, 32, 2, 2, 3, 4, false, true, false>(a, 0, s);
  } else {
    hipStream_t sw = defer ? wgrad_stream_after(s) : s;         // node-sized: the dW tiles may ride on the side stream
    constexpr int KL = ESC_K_LINEAR;
    if (bn == nullptr) e = in_scale ? dma::launch_dual<64, 64, 32, 2, 2, 3, 2, true, false, true>(a, 0, s, KL, sw) : dma::launch_dual<64, 64, 32, 2, 2, 3, 2, false, false, true>(a, 0, s, KL, sw);
    else if (next) e = in_scale ? dma::launch_dual<64, 64, 32, 2, 2, 2, 2, true, true, true>(a, 0, s, KL, sw) : dma::launch_dual<64, 64, 32, 2, 2, 2, 2, false, true, true>(a, 0, s, KL, sw);
    else      e = in_scale ? dma::launch_dual<64, 64, 32, 2, 2, 2, 2, true, true, false>(a, 0, s, KL, sw) : dma::launch_dual<64, 64, 32, 2, 2, 2, 2, false, true, false>(a, 0, s, KL, sw);
  }
  if (dma_check(e, "esc_linear_bwd_both_bn") != hipSuccess) return ESC_ELAUNCH;
  if (defer) { fill_job(defer, slabs, n, splits, K, dW, ld_dw, a.dw.db_part, N, db); return ESC_OK; }
  esc::launch(ESC_K_LINEAR, slab_reduce_kernel, dim3((unsigned)cdiv(n + (db ? N : 0), 256)), dim3(256), 0, s, slabs, n,
              splits, (int)K, dW, ld_dw, a.dw.db_part, (int)N, db);
  ESC_CHECK_LAUNCH("esc_linear_bwd_both_bn.reduce");
  return ESC_OK;
}

}  // extern "C"
