// linear_small.h — Linear layers with a tiny feature dimension (the dataset's in_dim = 10 of run_graphcount.py:65,78
// and GINEConv.lin 256 -> 10, :77): x_embedding.0, conv1.nn.0 and conv1.lin, forward and both gradients.
//
// A padded MFMA tile spends 14-28 us of the latency-critical node chain on each of them (measured, r01 kernels);
// they are bandwidth problems of a few MB:
//   smallk_fwd   Y[M,N]  = X[M,K<=16] W[N,K]^T + b   (+ the BatchNorm partials of the GEMM epilogue contract)
//   wgrad_small  dW[N,K] = dY[M,N]^T act(X)[M,K], db = colsum(dY), one of N, K <= 16: split-M slabs, summed in order later
//   smalln_dx    dX[M,K<=16] = dY[M,N] W[N,K]
// (the N <= 16 forward rides on a 64x32 tile of gemm_dma.h, the K-wide dX of an N <= 16 layer is linear_narrow_dx).
#pragma once
#include "common.h"

namespace esc {
namespace small {

constexpr int SMALL_MAX = 16;
constexpr int ROWS_FWD = 32;          // rows per workgroup = rows per BatchNorm partial (the GEMM epilogue's contract)
constexpr int ROWS_WGRAD = 32;        // reduction rows per slab (a slab is N*K <= 16*1280 floats: hundreds of them are cheap to sum)

// thread = output column n; the K weights of its column live in registers; X rows are workgroup-uniform scalars
template <int KMAX>
__global__ __launch_bounds__(256) void smallk_fwd(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                  int64_t ldw, const float* __restrict__ bias, int M, int N, int K,
                                                  float* __restrict__ Y, int64_t ldy, float2* __restrict__ col_stats) {
  ESC_PRIO();
  __shared__ float xs[ROWS_FWD * KMAX];
  const int n = blockIdx.y * 256 + threadIdx.x;
  const int r0 = blockIdx.x * ROWS_FWD;
  const int rows = min(ROWS_FWD, M - r0);
  for (int i = threadIdx.x; i < ROWS_FWD * KMAX; i += 256) {
    const int r = i / KMAX, k = i % KMAX;
    xs[i] = (r < rows && k < K) ? X[(size_t)(r0 + r) * ldx + k] : 0.f;
  }
  float w[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) w[k] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.f;
  const float b = (bias != nullptr && n < N) ? bias[n] : 0.f;
  __syncthreads();
  if (n >= N) return;
  float y[ROWS_FWD];
  float s1 = 0.f;
#pragma unroll
  for (int r = 0; r < ROWS_FWD; ++r) {
    float a = b;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) a = fmaf(xs[r * KMAX + k], w[k], a);      // k >= K: both factors are 0
    y[r] = a;
    if (r < rows) { Y[(size_t)(r0 + r) * ldy + n] = a; s1 += a; }
  }
  if (col_stats != nullptr) {
    const float mean = s1 / (float)rows;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS_FWD; ++r)
      if (r < rows) { const float d = y[r] - mean; m2 = fmaf(d, d, m2); }
    col_stats[(size_t)blockIdx.x * N + n] = make_float2(mean, m2);
  }
}

// SMALL_K: thread = n (a column of dY), accumulators over k; X rows are uniform.  act(X) = relu(X*scale+shift) per k.
// !SMALL_K (small N): thread = k (a column of X), accumulators over n; dY rows are uniform.  act per thread column.
// BNB (SMALL_K only): dY is the gradient of a BatchNorm(+ReLU) OUTPUT and its backward is applied as the column is read
// (common.h BnbDev) — the elementwise launch in front of this one is gone.
template <int SMAX, bool SMALL_K, bool PRO, int BNB = 0>     // BNB: 0 plain dY, 1 fused BatchNorm backward (no activation / ReLU), 2 ... with ELU
__global__ __launch_bounds__(256) void wgrad_small(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X,
                                                   int64_t ldx, const float* __restrict__ sc, const float* __restrict__ sh,
                                                   int M, int N, int K, float* __restrict__ slabs, float* __restrict__ db_part,
                                                   BnbDev bnb) {
  static_assert(BNB == 0 || SMALL_K, "the fused BatchNorm backward transforms the wide dY operand");
  ESC_PRIO();
  __shared__ float us[ROWS_WGRAD * SMAX];            // the uniform operand's rows
  const int t = blockIdx.y * 256 + threadIdx.x;      // wide index: n (SMALL_K) or k
  const int split = blockIdx.x;
  const int r0 = split * ROWS_WGRAD;
  const int rows = min(ROWS_WGRAD, M - r0);
  const int S = SMALL_K ? K : N;                      // small extent
  const int Wd = SMALL_K ? N : K;                     // wide extent
  const float* U = SMALL_K ? X : dY;
  const int64_t ldu = SMALL_K ? ldx : lddy;
  for (int i = threadIdx.x; i < rows * S; i += 256) {
    float v = U[(size_t)(r0 + i / S) * ldu + (i % S)];
    if constexpr (PRO && SMALL_K) v = fmaxf(fmaf(v, sc[i % S], sh[i % S]), 0.f);
    us[(i / S) * SMAX + (i % S)] = v;
  }
  for (int i = threadIdx.x; i < ROWS_WGRAD * SMAX; i += 256)
    if (i / SMAX >= rows || i % SMAX >= S) us[i] = 0.f;
  __syncthreads();
  float acc[SMAX];
#pragma unroll
  for (int q = 0; q < SMAX; ++q) acc[q] = 0.f;
  float colsum = 0.f;
  if (t < Wd) {
    const float* V = SMALL_K ? dY : X;
    const int64_t ldv = SMALL_K ? lddy : ldx;
    float psc = 1.f, psh = 0.f;
    if constexpr (PRO && !SMALL_K) { psc = sc[t]; psh = sh[t]; }
    float b_mu = 0.f, b_a = 0.f, b_ms = 0.f, b_mh = 1.f, b_k1 = 0.f, b_k2 = 0.f;
    if constexpr (BNB != 0) {
      const float2 kk = bnb.coef[t];
      b_mu = bnb.mean[t]; b_a = bnb.scale[t]; b_k1 = kk.x; b_k2 = bnb.invstd[t] * kk.y;
      if (bnb.relu) { b_ms = b_a; b_mh = bnb.shift[t]; }
    }
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      float v = V[(size_t)(r0 + r) * ldv + t];
      if constexpr (BNB == 1) v = bnb_apply(v, bnb.x[(size_t)(r0 + r) * bnb.ldx + t], b_mu, b_a, b_ms, b_mh, b_k1, b_k2);
      if constexpr (BNB == 2) v = bnb_apply_elu(v, bnb.x[(size_t)(r0 + r) * bnb.ldx + t], b_mu, b_a, b_ms, b_mh, b_k1, b_k2);
      if constexpr (PRO && !SMALL_K) v = fmaxf(fmaf(v, psc, psh), 0.f);
      if constexpr (SMALL_K) colsum += v;
#pragma unroll
      for (int q = 0; q < SMAX; ++q) acc[q] = fmaf(us[r * SMAX + q], v, acc[q]);
    }
    float* out = slabs + (size_t)split * N * K;
#pragma unroll
    for (int q = 0; q < SMAX; ++q)
      if (q < S) out[SMALL_K ? (size_t)t * K + q : (size_t)q * K + t] = acc[q];
    if constexpr (SMALL_K) db_part[(size_t)split * N + t] = colsum;
  }
  if constexpr (!SMALL_K) {                          // db[n] = sum over rows of the uniform dY rows
    if (blockIdx.y == 0 && (int)threadIdx.x < N) {
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += us[r * SMAX + threadIdx.x];
      db_part[(size_t)split * N + threadIdx.x] = s;
    }
  }
}

// dX[M,K<=16] = dY[M,N] W[N,K]: 32 rows per workgroup staged in LDS, thread (row, k-slot) walks the N-long dot product
template <int KMAX, int BNB = 0>
__global__ __launch_bounds__(256) void smalln_dx(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W,
                                                 int64_t ldw, int M, int N, int K, float* __restrict__ dX, int64_t lddx,
                                                 int accumulate, BnbDev bnb) {
  ESC_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldn = N + 4;                             // padded rows: 8 rows / 8 k-slots per wave hit different banks
  float* ys = sm;                                    // [32][N+4]
  float* wt = sm + 32 * ldn;                         // [KMAX][N+4]  (W transposed)
  const int r0 = blockIdx.x * 32;
  const int rows = min(32, M - r0);
  for (int i = threadIdx.x; i < rows * (N / 4); i += 256) {
    const int r = i / (N / 4), c = (i % (N / 4)) * 4;
    float4 v = *reinterpret_cast<const float4*>(dY + (size_t)(r0 + r) * lddy + c);
    if constexpr (BNB != 0) {     // the BatchNorm backward of the staged gradient rows (per-channel coefficients: L1/L2 hits)
      const float4 x = *reinterpret_cast<const float4*>(bnb.x + (size_t)(r0 + r) * bnb.ldx + c);
      const float4 mu = *reinterpret_cast<const float4*>(bnb.mean + c), a = *reinterpret_cast<const float4*>(bnb.scale + c);
      const float4 is = *reinterpret_cast<const float4*>(bnb.invstd + c);
      const float4 k01 = *reinterpret_cast<const float4*>(bnb.coef + c), k23 = *reinterpret_cast<const float4*>(bnb.coef + c + 2);
      float4 ms = make_float4(0.f, 0.f, 0.f, 0.f), mh = make_float4(1.f, 1.f, 1.f, 1.f);
      if (bnb.relu) { ms = a; mh = *reinterpret_cast<const float4*>(bnb.shift + c); }
      if constexpr (BNB == 2) {
        v.x = bnb_apply_elu(v.x, x.x, mu.x, a.x, ms.x, mh.x, k01.x, is.x * k01.y); v.y = bnb_apply_elu(v.y, x.y, mu.y, a.y, ms.y, mh.y, k01.z, is.y * k01.w);
        v.z = bnb_apply_elu(v.z, x.z, mu.z, a.z, ms.z, mh.z, k23.x, is.z * k23.y); v.w = bnb_apply_elu(v.w, x.w, mu.w, a.w, ms.w, mh.w, k23.z, is.w * k23.w);
      } else {
        v.x = bnb_apply(v.x, x.x, mu.x, a.x, ms.x, mh.x, k01.x, is.x * k01.y); v.y = bnb_apply(v.y, x.y, mu.y, a.y, ms.y, mh.y, k01.z, is.y * k01.w);
        v.z = bnb_apply(v.z, x.z, mu.z, a.z, ms.z, mh.z, k23.x, is.z * k23.y); v.w = bnb_apply(v.w, x.w, mu.w, a.w, ms.w, mh.w, k23.z, is.w * k23.w);
      }
    }
    *reinterpret_cast<float4*>(ys + r * ldn + c) = v;
  }
  for (int i = threadIdx.x; i < N * K; i += 256) wt[(i % K) * ldn + (i / K)] = W[(size_t)(i / K) * ldw + (i % K)];
  __syncthreads();
  const int r = threadIdx.x >> 3, ks = threadIdx.x & 7;
  if (r >= rows) return;
#pragma unroll
  for (int kk = 0; kk < KMAX / 8; ++kk) {
    const int k = ks + kk * 8;
    if (k >= K) break;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < N; c += 4) {
      const float4 y = *reinterpret_cast<const float4*>(ys + r * ldn + c);
      const float4 w = *reinterpret_cast<const float4*>(wt + k * ldn + c);
      a0 = fmaf(y.x, w.x, a0); a1 = fmaf(y.y, w.y, a1); a2 = fmaf(y.z, w.z, a2); a3 = fmaf(y.w, w.w, a3);
    }
    float v = (a0 + a1) + (a2 + a3);
    float* dst = dX + (size_t)(r0 + r) * lddx + k;
    if (accumulate) v += *dst;
    *dst = v;
  }
}

}  // namespace small
}  // namespace esc
