// norm.hip — BatchNorm1d with batch statistics (+ fused ReLU) forward/backward for [rows, C] activations.
//
// Replaces torch.nn.BatchNorm1d(+ReLU) at /root/reference/run_graphcount.py:55-60 (z_embedding, over
// all E edge rows of the batch), :66-72 (x_embedding), :80-87,:100-107 (GINEConv.nn), :115 (bn_lin1).
// HBM-bound elementwise/reduction work: every pass reads whole 256 B..1 KiB row segments per wave
// (lane = column), rows are strided over waves.  Statistics are accumulated per wave around a shift
// (first row of the wave's stripe) and merged with Chan's formula in fp64, so fp32 cancellation in
// E[x^2]-E[x]^2 never shows up at the 1e-5 parity bar.
#include "common.h"

#include <cstdlib>

namespace esc {

constexpr int NORM_ROWBLOCKS = 512;          // scratch sizing: most row blocks (= workgroups per column block) ever used

// fused activation after the affine: 0 none, 1 ReLU, 2 ELU(alpha=1) (zinc_models.py:513-522 uses ELU)
__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v > 0.f ? v : expm1f(v);
  return v;
}
// d act / d v expressed through the forward OUTPUT y (relu: [y>0]; elu: y>0 ? 1 : y+1)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  if (act == 1) return y > 0.f ? 1.f : 0.f;
  if (act == 2) return y > 0.f ? 1.f : y + 1.f;
  return 1.f;
}
// The pre-activation exactly as the forward pass formed it (consumer-side fused BatchNorm: fmaf(x, scale, shift) with
// scale = gamma*invstd, shift = beta - mean*scale, the same float expressions as bn_finalize / bn_fold_column): a mask
// recomputed as ((x-mean)*invstd)*gamma+beta rounds differently, and an element within an ulp of the kink then gets a
// gradient although the forward clipped it (or vice versa) — measured: a handful per 6*10^5 activations, each worth
// ~1e-3 of relative gradient error at node size.
__device__ __forceinline__ float pre_act_fwd(float x, float mu, float is, float ga, float be) {
  const float sc = ga * is;
  return fmaf(x, sc, be - mu * sc);
}
// ... or through the pre-activation v when the output was never materialised
__device__ __forceinline__ float act_grad_from_pre(float v, int act) {
  if (act == 1) return v > 0.f ? 1.f : 0.f;
  if (act == 2) return v > 0.f ? 1.f : expf(v);
  return 1.f;
}          // grid.y; x4 waves => 256 row slots

// slot p owns rows p, p+P, p+2P, ...   partial[(p*C + c)] = {mean, M2}
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ X, int64_t ld, int M, int C,
                                                         float2* __restrict__ partial) {
  ESC_PRIO();
  const int c = blockIdx.x * 64 + lane_id();
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  if (c >= C) return;
  float shift = 0.f, s1 = 0.f, s2 = 0.f;
  int n = 0;
  if (slot < M) shift = X[(size_t)slot * ld + c];
#pragma unroll 4
  for (int r = slot; r < M; r += P) {
    const float d = X[(size_t)r * ld + c] - shift;
    s1 += d;
    s2 = fmaf(d, d, s2);
    ++n;
  }
  float2 out = make_float2(0.f, 0.f);
  if (n > 0) {
    const double m = (double)s1 / n;
    out.x = (float)((double)shift + m);
    out.y = (float)fmax((double)s2 - (double)s1 * m, 0.0);
  }
  partial[(size_t)slot * C + c] = out;
}

// float4 variants (C % 4 == 0, 16-B aligned rows): a wave covers 256 columns with one 1-KiB load per row, so a
// few rows per wave already put tens of KiB in flight per CU — the scalar form above tops out near 1 TB/s on the
// edge-sized tensors.
__global__ __launch_bounds__(256) void bn_partial_kernel_v4(const float* __restrict__ X, int64_t ld, int M, int C,
                                                            float2* __restrict__ partial) {
  ESC_PRIO();
  const int c = (blockIdx.x * 64 + lane_id()) * 4;
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  if (c >= C) return;
  float4 shift = make_float4(0.f, 0.f, 0.f, 0.f), s1 = shift, s2 = shift;
  int n = 0;
  if (slot < M) shift = *reinterpret_cast<const float4*>(X + (size_t)slot * ld + c);
#pragma unroll 8
  for (int r = slot; r < M; r += P) {
    const float4 v = *reinterpret_cast<const float4*>(X + (size_t)r * ld + c);
    const float dx = v.x - shift.x, dy = v.y - shift.y, dz = v.z - shift.z, dw = v.w - shift.w;
    s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
    s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
    ++n;
  }
  float2 o[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
  if (n > 0) {
    const float sh[4] = {shift.x, shift.y, shift.z, shift.w}, a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double m = (double)a1[t] / n;
      o[t] = make_float2((float)((double)sh[t] + m), (float)fmax((double)a2[t] - (double)a1[t] * m, 0.0));
    }
  }
  float2* dst = partial + (size_t)slot * C + c;
  *reinterpret_cast<float4*>(dst) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
  *reinterpret_cast<float4*>(dst + 2) = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
}

// ACT / HAS_Y are compile-time so that every load in the row loop is unconditional (a run-time select around a
// load makes hipcc branch and wait per element, which is what kept the scalar kernel at ~1 TB/s).
// The four waves of a workgroup add their sums through LDS (fixed order) into ONE slot per workgroup, and the
// workgroup that finishes last (grid_last_block) folds the slots into dgamma / dbeta / coef: no finalize launch.
// DROP: dY is the gradient of dropout(act(bn(X))): g = dY * keep / (1-p) first (dmask: one byte per element, [M][C]) —
// the dropout backward of an OGB layer update (ogb_mol_gnn.py:750-755) without its own pass over the rows
template <int ACT, bool HAS_Y, bool DROP>
__device__ __forceinline__ void bn_bwd_partial_body(const float* __restrict__ X, int64_t ldx,
                                                                const float* __restrict__ Y, int64_t ldy,
                                                                const float* __restrict__ dY, int64_t ldg, int M,
                                                                int C, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float2* partial, unsigned* tickets,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float2* __restrict__ coef,
                                                                const unsigned char* __restrict__ dmask, float dscale) {
  ESC_PRIO();
  constexpr int relu = ACT;
  __shared__ float4 sh[3][2][64];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int slot = blockIdx.y * 4 + wave;
  const int P = gridDim.y * 4;
  const bool active = c < C;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (active) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 be = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = slot; r < M; r += P) {
      float4 g = *reinterpret_cast<const float4*>(dY + (size_t)r * ldg + c);
      if constexpr (DROP) {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(dmask + (size_t)r * C + c);
        g.x = mk.x ? g.x * dscale : 0.f; g.y = mk.y ? g.y * dscale : 0.f; g.z = mk.z ? g.z * dscale : 0.f; g.w = mk.w ? g.w * dscale : 0.f;
      }
      const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
      const float4 xh = make_float4((x.x - mu.x) * is.x, (x.y - mu.y) * is.y, (x.z - mu.z) * is.z, (x.w - mu.w) * is.w);
      if constexpr (ACT != 0) {
        if constexpr (HAS_Y) {
          const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
          g.x *= act_grad_from_out(y.x, relu); g.y *= act_grad_from_out(y.y, relu);
          g.z *= act_grad_from_out(y.z, relu); g.w *= act_grad_from_out(y.w, relu);
        } else {
          g.x *= act_grad_from_pre(pre_act_fwd(x.x, mu.x, is.x, ga.x, be.x), relu); g.y *= act_grad_from_pre(pre_act_fwd(x.y, mu.y, is.y, ga.y, be.y), relu);
          g.z *= act_grad_from_pre(pre_act_fwd(x.z, mu.z, is.z, ga.z, be.z), relu); g.w *= act_grad_from_pre(pre_act_fwd(x.w, mu.w, is.w, ga.w, be.w), relu);
        }
      }
      s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
      s2.x = fmaf(g.x, xh.x, s2.x); s2.y = fmaf(g.y, xh.y, s2.y); s2.z = fmaf(g.z, xh.z, s2.z); s2.w = fmaf(g.w, xh.w, s2.w);
    }
  }
  if (wave > 0) { sh[wave - 1][0][lane] = s1; sh[wave - 1][1][lane] = s2; }
  __syncthreads();
  if (wave == 0 && active) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float4 a = sh[w][0][lane], b = sh[w][1][lane];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float2* dst = partial + (size_t)blockIdx.y * C + c;
    if (tickets != nullptr) {
      store_agent(dst, make_float2(s1.x, s2.x));
      store_agent(dst + 1, make_float2(s1.y, s2.y));
      store_agent(dst + 2, make_float2(s1.z, s2.z));
      store_agent(dst + 3, make_float2(s1.w, s2.w));
    } else {
      *reinterpret_cast<float4*>(dst) = make_float4(s1.x, s2.x, s1.y, s2.y);
      *reinterpret_cast<float4*>(dst + 2) = make_float4(s1.z, s2.z, s1.w, s2.w);
    }
  }
  if (tickets == nullptr) return;          // large M: a wide finalize launch follows
  if (!grid_last_block(tickets + blockIdx.x, gridDim.y)) return;
  const int col = blockIdx.x * 256 + threadIdx.x;      // the last workgroup: one thread per column, slots in order
  if (col >= C) return;
  double t1 = 0.0, t2 = 0.0;
  const int P1 = (int)gridDim.y;
  for (int p = 0; p < P1; p += 16) {       // written through to memory by other workgroups: 16 misses in flight
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)min(p + u, P1 - 1) * C + col];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (p + u < P1) { t1 += (double)v[u].x; t2 += (double)v[u].y; }
    }
  }
  if (dgamma) dgamma[col] = (float)t2;
  if (dbeta) dbeta[col] = (float)t1;
  coef[col] = make_float2((float)(t1 / M), (float)(t2 / M));
}

template <int ACT, bool HAS_Y>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel_v4(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y, int64_t ldy,
                                                                const float* __restrict__ dY, int64_t ldg, int M, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float2* partial, unsigned* tickets, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float2* __restrict__ coef) {
  bn_bwd_partial_body<ACT, HAS_Y, false>(X, ldx, Y, ldy, dY, ldg, M, C, mean, invstd, gamma, beta, partial, tickets, dgamma, dbeta, coef, nullptr, 0.f);
}
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_partial_drop_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldg,
                                                                  int M, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float2* partial,
                                                                  const unsigned char* __restrict__ dmask, float dscale) {
  bn_bwd_partial_body<ACT, false, true>(X, ldx, nullptr, 0, dY, ldg, M, C, mean, invstd, gamma, beta, partial, nullptr, nullptr, nullptr, nullptr, dmask, dscale);
}

// one wave per column: lanes own slots lane, lane+64, ... then a shuffle-tree merge (fixed order)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float2* __restrict__ partial, int M, int C, int P,
                                                          int block_rows, float eps, float momentum,
                                                          float* __restrict__ mean,
                                                          float* __restrict__ invstd,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
  ESC_PRIO();
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= C) return;
  const int lane = lane_id();
  double n = 0.0, mu = 0.0, m2 = 0.0;
  // block_rows == 0: slot p owns rows p, p+P, ... (bn_partial_kernel); else slot p owns rows [p*block_rows, ...)
  for (int p = lane; p < P && (block_rows > 0 || p < M); p += 64) {
    const float2 v = partial[(size_t)p * C + c];
    const int np = block_rows > 0 ? min(block_rows, M - p * block_rows) : (M - p + P - 1) / P;
    chan_merge(n, mu, m2, (double)np, (double)v.x, (double)v.y);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double nb = __shfl_xor(n, o, 64), mub = __shfl_xor(mu, o, 64), m2b = __shfl_xor(m2, o, 64);
    // merge the higher lane into the lower one in both partners identically (commutative up to rounding:
    // order the pair by lane parity so both compute the same expression)
    if ((lane & o) == 0) chan_merge(n, mu, m2, nb, mub, m2b);
    else { double n2 = nb, mu2 = mub, m22 = m2b; chan_merge(n2, mu2, m22, n, mu, m2); n = n2; mu = mu2; m2 = m22; }
  }
  if (lane == 0) {
    const double var = m2 / (double)M;
    mean[c] = (float)mu;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    invstd[c] = is;
    if (scale) {      // consumer-side fused form: act(x) = relu(x*scale + shift)
      const float sc = (gamma ? gamma[c] : 1.f) * is;
      scale[c] = sc;
      shift[c] = (beta ? beta[c] : 0.f) - (float)mu * sc;
    }
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2 / (double)(M - 1));
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int C,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu,
                                                       float* __restrict__ Y, int64_t ldy) {
  ESC_PRIO();
  const int cv = C / VEC;
  const int64_t total = M * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    float xv[VEC], yv[VEC];
    if constexpr (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4*>(X + r * ldx + c);
      xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
    } else {
      xv[0] = X[r * ldx + c];
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      float v = (xv[t] - mean[c + t]) * invstd[c + t];
      v = fmaf(v, gamma ? gamma[c + t] : 1.f, beta ? beta[c + t] : 0.f);
      yv[t] = act_fwd(v, relu);
    }
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(Y + r * ldy + c) = make_float4(yv[0], yv[1], yv[2], yv[3]);
    } else {
      Y[r * ldy + c] = yv[0];
    }
  }
}

// backward partials: slot sums of g and g*xhat, g = dY * [Y > 0] (relu) ; partial[(p*C+c)] = {s1, s2}
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ Y, int64_t ldy,
                                                             const float* __restrict__ dY, int64_t ldg, int M,
                                                             int C, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int relu,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float2* __restrict__ partial) {
  ESC_PRIO();
  const int c = blockIdx.x * 64 + lane_id();
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  if (c >= C) return;
  const float mu = mean[c], is = invstd[c];
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int r = slot; r < M; r += P) {
    float g = dY[(size_t)r * ldg + c];
    const float xv = X[(size_t)r * ldx + c];
    const float xh = (xv - mu) * is;
    if (relu) {   // activation derivative from the forward output (if kept) or from the recomputed pre-activation
      g *= Y ? act_grad_from_out(Y[(size_t)r * ldy + c], relu) : act_grad_from_pre(pre_act_fwd(xv, mu, is, ga, be), relu);
    }
    s1 += g;
    s2 = fmaf(g, xh, s2);
  }
  partial[(size_t)slot * C + c] = make_float2(s1, s2);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float2* __restrict__ partial, int M, int C,
                                                              int P, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta,
                                                              float2* __restrict__ coef) {
  ESC_PRIO();
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // one wave per column
  if (c >= C) return;
  const int lane = lane_id();
  double s1 = 0.0, s2 = 0.0;
  for (int p = lane; p < P; p += 64) {
    const float2 v = partial[(size_t)p * C + c];
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    coef[c] = make_float2((float)(s1 / M), (float)(s2 / M));
  }
}

template <int VEC, int ACT, bool HAS_Y>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ X, int64_t ldx,
                                                           const float* __restrict__ Y, int64_t ldy,
                                                           const float* __restrict__ dY, int64_t ldg, int64_t M,
                                                           int C, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float2* __restrict__ coef,
                                                           float* __restrict__ dX, int64_t ldd) {
  ESC_PRIO();
  constexpr int relu = ACT;
  const int cv = C / VEC;
  const int64_t total = M * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    float xv[VEC], yv[VEC], gv[VEC], ov[VEC];
    if constexpr (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4*>(X + r * ldx + c);
      const float4 g = *reinterpret_cast<const float4*>(dY + r * ldg + c);
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
      gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
      if constexpr (ACT != 0 && HAS_Y) {
        const float4 y = *reinterpret_cast<const float4*>(Y + r * ldy + c);
        yv[0] = y.x; yv[1] = y.y; yv[2] = y.z; yv[3] = y.w;
      }
    } else {
      xv[0] = X[r * ldx + c];
      gv[0] = dY[r * ldg + c];
      if constexpr (ACT != 0 && HAS_Y) yv[0] = Y[r * ldy + c];
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      const float is = invstd[c + t];
      const float xh = (xv[t] - mean[c + t]) * is;
      float g = gv[t];
      if constexpr (ACT != 0) {
        if constexpr (HAS_Y) g *= act_grad_from_out(yv[t], relu);
        else g *= act_grad_from_pre(pre_act_fwd(xv[t], mean[c + t], is, gamma ? gamma[c + t] : 1.f, beta ? beta[c + t] : 0.f), relu);
      }
      const float2 k = coef[c + t];
      ov[t] = (gamma ? gamma[c + t] : 1.f) * is * (g - k.x - xh * k.y);
    }
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(dX + r * ldd + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    } else {
      dX[r * ldd + c] = ov[0];
    }
  }
}

// Row-strided float4 form of the two elementwise passes above: a lane owns one column quad for the whole launch, so
// the per-column coefficients are loaded ONCE as float4 (the flat-index kernels re-read ~20 scalars per element) and a
// wave keeps 4 rows x 2-3 arrays in flight.  Edge-sized BatchNorm backward: 17 -> 11 us.
// FOLD: there was no finalize launch — `coef` is the partial-sum array [fold_slots][C] of bn_bwd_partial_kernel_v4 and
// every wave adds the (few) slots of its four columns itself, in slot order with fp64 accumulators like the finalize
// kernel; the first row block also writes dgamma / dbeta.  Node-sized BatchNorms only (<= 32 slots): a dependent launch
// costs more than 64 extra loads per lane.
// DROP_IN: as in the partial kernel (g = dY * keep / (1-p)); DROP_OUT: X itself was dropout(input), so the result is
// multiplied by ITS keep mask / (1-p) on the way out (z_embedding's Dropout -> BatchNorm order, ogb_mol_gnn.py:638-645)
template <int ACT, bool HAS_Y, bool FOLD, bool DROP_IN, bool DROP_OUT>
__device__ __forceinline__ void bn_bwd_apply_rows_body(const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ dY, int64_t ldg, int M, int C,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float2* __restrict__ coef,
                                                         float* __restrict__ dX, int64_t ldd, int fold_slots,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         const unsigned char* __restrict__ dmask, float dscale) {
  ESC_PRIO();
  constexpr int relu = ACT;
  const int c = (blockIdx.x * 64 + lane_id()) * 4;
  if (c >= C) return;
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
  const float4 ga = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 be = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 k01, k23;
  if constexpr (FOLD) {
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int p = 0; p < fold_slots; ++p) {
      const float4 u = *reinterpret_cast<const float4*>(coef + (size_t)p * C + c), v = *reinterpret_cast<const float4*>(coef + (size_t)p * C + c + 2);
      s1[0] += (double)u.x; s2[0] += (double)u.y; s1[1] += (double)u.z; s2[1] += (double)u.w;
      s1[2] += (double)v.x; s2[2] += (double)v.y; s1[3] += (double)v.z; s2[3] += (double)v.w;
    }
    k01 = make_float4((float)(s1[0] / M), (float)(s2[0] / M), (float)(s1[1] / M), (float)(s2[1] / M));
    k23 = make_float4((float)(s1[2] / M), (float)(s2[2] / M), (float)(s1[3] / M), (float)(s2[3] / M));
    if (slot == 0) {
      if (dgamma) *reinterpret_cast<float4*>(dgamma + c) = make_float4((float)s2[0], (float)s2[1], (float)s2[2], (float)s2[3]);
      if (dbeta) *reinterpret_cast<float4*>(dbeta + c) = make_float4((float)s1[0], (float)s1[1], (float)s1[2], (float)s1[3]);
    }
  } else {
    k01 = *reinterpret_cast<const float4*>(coef + c); k23 = *reinterpret_cast<const float4*>(coef + c + 2);
  }
  const float4 a = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
#pragma unroll 4
  for (int r = slot; r < M; r += P) {
    const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
    float4 g = *reinterpret_cast<const float4*>(dY + (size_t)r * ldg + c);
    uchar4 mk = make_uchar4(1, 1, 1, 1);
    if constexpr (DROP_IN || DROP_OUT) mk = *reinterpret_cast<const uchar4*>(dmask + (size_t)r * C + c);
    if constexpr (DROP_IN) {
      g.x = mk.x ? g.x * dscale : 0.f; g.y = mk.y ? g.y * dscale : 0.f; g.z = mk.z ? g.z * dscale : 0.f; g.w = mk.w ? g.w * dscale : 0.f;
    }
    const float4 xh = make_float4((x.x - mu.x) * is.x, (x.y - mu.y) * is.y, (x.z - mu.z) * is.z, (x.w - mu.w) * is.w);
    if constexpr (ACT != 0) {
      if constexpr (HAS_Y) {
        const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
        g.x *= act_grad_from_out(y.x, relu); g.y *= act_grad_from_out(y.y, relu);
        g.z *= act_grad_from_out(y.z, relu); g.w *= act_grad_from_out(y.w, relu);
      } else {
        g.x *= act_grad_from_pre(pre_act_fwd(x.x, mu.x, is.x, ga.x, be.x), relu); g.y *= act_grad_from_pre(pre_act_fwd(x.y, mu.y, is.y, ga.y, be.y), relu);
        g.z *= act_grad_from_pre(pre_act_fwd(x.z, mu.z, is.z, ga.z, be.z), relu); g.w *= act_grad_from_pre(pre_act_fwd(x.w, mu.w, is.w, ga.w, be.w), relu);
      }
    }
    // same expression as bn_bwd_apply_kernel: gamma * invstd * (g - k.x - xhat * k.y)
    float4 o = make_float4(a.x * (g.x - k01.x - xh.x * k01.y), a.y * (g.y - k01.z - xh.y * k01.w),
                           a.z * (g.z - k23.x - xh.z * k23.y), a.w * (g.w - k23.z - xh.w * k23.w));
    if constexpr (DROP_OUT) {
      o.x = mk.x ? o.x * dscale : 0.f; o.y = mk.y ? o.y * dscale : 0.f; o.z = mk.z ? o.z * dscale : 0.f; o.w = mk.w ? o.w * dscale : 0.f;
    }
    *reinterpret_cast<float4*>(dX + (size_t)r * ldd + c) = o;
  }
}

template <int ACT, bool HAS_Y, bool FOLD = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ dY, int64_t ldg, int M, int C,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float2* __restrict__ coef, float* __restrict__ dX, int64_t ldd,
                                                         int fold_slots, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  bn_bwd_apply_rows_body<ACT, HAS_Y, FOLD, false, false>(X, ldx, Y, ldy, dY, ldg, M, C, mean, invstd, gamma, beta, coef, dX, ldd, fold_slots, dgamma, dbeta, nullptr, 0.f);
}
template <int ACT, bool DROP_IN, bool DROP_OUT>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_drop(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldg,
                                                              int M, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float2* __restrict__ coef, float* __restrict__ dX, int64_t ldd,
                                                              const unsigned char* __restrict__ dmask, float dscale) {
  bn_bwd_apply_rows_body<ACT, false, false, DROP_IN, DROP_OUT>(X, ldx, nullptr, 0, dY, ldg, M, C, mean, invstd, gamma, beta, coef, dX, ldd, 0, nullptr, nullptr, dmask, dscale);
}

__global__ __launch_bounds__(256) void affine_act_rows(const float* __restrict__ X, int64_t ldx, int M, int C,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu,
                                                       float* __restrict__ Y, int64_t ldy) {
  ESC_PRIO();
  const int c = (blockIdx.x * 64 + lane_id()) * 4;
  if (c >= C) return;
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
#pragma unroll 4
  for (int r = slot; r < M; r += P) {
    const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
    *reinterpret_cast<float4*>(Y + (size_t)r * ldy + c) =
        make_float4(act_fwd(fmaf(x.x, sc.x, sh.x), relu), act_fwd(fmaf(x.y, sc.y, sh.y), relu),
                    act_fwd(fmaf(x.z, sc.z, sh.z), relu), act_fwd(fmaf(x.w, sc.w, sh.w), relu));
  }
}

// affine_act_rows with the BatchNorm still in partial form: every workgroup merges the partials of its column quad
// (bn_fold_column, common.h), the first row block stores the coefficients for the backward pass
__global__ __launch_bounds__(256) void affine_act_fold_rows(const float* __restrict__ X, int64_t ldx, int M, int C,
                                                            BnFoldDev f, int relu, float* __restrict__ Y, int64_t ldy) {
  ESC_PRIO();
  __shared__ float coef[2][256];
  // one column per thread for the merge (all its partial loads in flight at once), a column quad per lane afterwards
  const int cm = blockIdx.x * 256 + threadIdx.x;
  if (cm < C) bn_fold_column(f, cm, blockIdx.y == 0, coef[0][threadIdx.x], coef[1][threadIdx.x]);
  __syncthreads();
  const int lane = lane_id();
  const int c = (blockIdx.x * 64 + lane) * 4;
  if (c >= C) return;
  const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int P = gridDim.y * 4;
  const float4 sc = *reinterpret_cast<const float4*>(&coef[0][lane * 4]), sh = *reinterpret_cast<const float4*>(&coef[1][lane * 4]);
#pragma unroll 4
  for (int r = slot; r < M; r += P) {
    const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
    *reinterpret_cast<float4*>(Y + (size_t)r * ldy + c) =
        make_float4(act_fwd(fmaf(x.x, sc.x, sh.x), relu), act_fwd(fmaf(x.y, sc.y, sh.y), relu),
                    act_fwd(fmaf(x.z, sc.z, sh.z), relu), act_fwd(fmaf(x.w, sc.w, sh.w), relu));
  }
}

// y = relu?(x*scale + shift): materialises a consumer-side-fused BatchNorm(+ReLU) output
template <int VEC>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int C,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu,
                                                         float* __restrict__ Y, int64_t ldy) {
  ESC_PRIO();
  const int cv = C / VEC;
  const int64_t total = M * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * VEC;
    if constexpr (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4*>(X + r * ldx + c);
      const float4 s4 = *reinterpret_cast<const float4*>(scale + c);
      const float4 h4 = *reinterpret_cast<const float4*>(shift + c);
      float4 o = make_float4(fmaf(q.x, s4.x, h4.x), fmaf(q.y, s4.y, h4.y), fmaf(q.z, s4.z, h4.z), fmaf(q.w, s4.w, h4.w));
      o.x = act_fwd(o.x, relu); o.y = act_fwd(o.y, relu); o.z = act_fwd(o.z, relu); o.w = act_fwd(o.w, relu);
      *reinterpret_cast<float4*>(Y + r * ldy + c) = o;
    } else {
      Y[r * ldy + c] = act_fwd(fmaf(X[r * ldx + c], scale[c], shift[c]), relu);
    }
  }
}

// inference-mode coefficients from running statistics
__global__ __launch_bounds__(256) void bn_eval_coef_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int C,
                                                           float* __restrict__ scale, float* __restrict__ shift) {
  ESC_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = (gamma ? gamma[c] : 1.f) * (1.0f / sqrtf(rv[c] + eps));
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

// ---- SyncBN inside the step engine (graph-sharded data parallelism) -----------------------------------------
// Every rank packs its (count, mean, M2) per channel into ITS slot of a zeroed [world][3][C] buffer; one SUM all-reduce
// is then an all-gather; the slots are Chan-merged in rank order (fp64) on every rank: identical global statistics.
__global__ __launch_bounds__(256) void bn_sync_pack_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float n_local, float eps, int C, int rank, int world,
                                                           float* __restrict__ buf) {
  ESC_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = invstd[c];
  const float m2 = n_local * fmaxf(1.f / (is * is) - eps, 0.f);        // biased variance * n
  for (int r = 0; r < world; ++r) {
    float* b = buf + (size_t)r * 3 * C;
    const bool me = r == rank;
    b[c] = me ? n_local : 0.f;
    b[C + c] = me ? mean[c] : 0.f;
    b[2 * C + c] = me ? m2 : 0.f;
  }
}
__global__ __launch_bounds__(256) void bn_sync_finalize_kernel(const float* __restrict__ buf, int world, int C, float eps,
                                                               float momentum, float* __restrict__ mean,
                                                               float* __restrict__ invstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ n_total) {
  ESC_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, mu = 0.0, m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const float* b = buf + (size_t)r * 3 * C;
    chan_merge(n, mu, m2, (double)b[c], (double)b[C + c], (double)b[2 * C + c]);
  }
  const float is = (float)(1.0 / sqrt(m2 / n + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  if (scale) {
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - (float)mu * sc;
  }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2 / (n - 1.0));
  if (c == 0 && n_total) *n_total = (float)n;
}
// backward: the all-reduced column sums (sum g, sum g*xhat) become the means over ALL ranks' rows
__global__ __launch_bounds__(256) void bn_sync_coef_kernel(float2* __restrict__ coef, int C, const float* __restrict__ n_total) {
  ESC_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = 1.f / *n_total;
  const float2 v = coef[c];
  coef[c] = make_float2(v.x * inv, v.y * inv);
}

// >= 4 rows per wave slot (scalar kernels keep the old 64-block cap: their finalize cost grows with the slot count)
// forward statistics: 64 (their finalize merges 4 slots per block with Chan's formula, its cost grows with the count);
// ---- node-sized BatchNorm backward in ONE launch ---------------------------------------------------------------------------
// partial sums -> grid barrier -> every workgroup adds the slots of its 256 columns -> dX from the rows it still holds in
// registers.  Replaces partial + finalize + apply (three dependent launches at the ~4.5 us floor each, 13 times per training
// step of the counting model).  The barrier is an agent-scope counter pair from the ticket pool: workgroups arrive, spin
// (bounded: a workgroup that gives up raises *err and the results are invalid, but nothing hangs) and the last one to LEAVE
// re-zeroes the pair.  All gridDim.y <= 256 workgroups of a column block are small (256 threads, 3 KB LDS), so they are
// co-resident unless other kernels occupy every CU — those finish without waiting for this one, so the spin always ends.
// Rows per wave are fixed (4, in registers): the host picks gridDim.y = ceil(M / 16).
template <int ACT, bool HAS_Y>
__global__ __launch_bounds__(256) void bn_bwd_node_kernel(const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ Y, int64_t ldy,
                                                          const float* __restrict__ dY, int64_t ldg, int M, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float2* slots, unsigned* bar, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dX, int64_t ldd,
                                                          int* err) {
  ESC_PRIO();
  constexpr int relu = ACT;
  constexpr int RPW = 4;
  __shared__ float4 sh[3][2][64];
  __shared__ float2 coef_s[256];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool active = c < C;
  const int P = gridDim.y * 4;
  const int slot0 = blockIdx.y * 4 + wave;
  float4 gq[RPW], xq[RPW];
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, ga = make_float4(1.f, 1.f, 1.f, 1.f), be = mu;
  float4 s1 = mu, s2 = mu;
  if (active) {
    mu = *reinterpret_cast<const float4*>(mean + c); is = *reinterpret_cast<const float4*>(invstd + c);
    if (gamma) ga = *reinterpret_cast<const float4*>(gamma + c);
    if (beta) be = *reinterpret_cast<const float4*>(beta + c);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = slot0 + j * P;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), xh = g;
      if (r < M) {
        g = *reinterpret_cast<const float4*>(dY + (size_t)r * ldg + c);
        const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
        xh = make_float4((x.x - mu.x) * is.x, (x.y - mu.y) * is.y, (x.z - mu.z) * is.z, (x.w - mu.w) * is.w);
        if constexpr (ACT != 0) {
          if constexpr (HAS_Y) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ldy + c);
            g.x *= act_grad_from_out(y.x, relu); g.y *= act_grad_from_out(y.y, relu);
            g.z *= act_grad_from_out(y.z, relu); g.w *= act_grad_from_out(y.w, relu);
          } else {
            g.x *= act_grad_from_pre(pre_act_fwd(x.x, mu.x, is.x, ga.x, be.x), relu); g.y *= act_grad_from_pre(pre_act_fwd(x.y, mu.y, is.y, ga.y, be.y), relu);
            g.z *= act_grad_from_pre(pre_act_fwd(x.z, mu.z, is.z, ga.z, be.z), relu); g.w *= act_grad_from_pre(pre_act_fwd(x.w, mu.w, is.w, ga.w, be.w), relu);
          }
        }
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, xh.x, s2.x); s2.y = fmaf(g.y, xh.y, s2.y); s2.z = fmaf(g.z, xh.z, s2.z); s2.w = fmaf(g.w, xh.w, s2.w);
      }
      gq[j] = g; xq[j] = xh;
    }
  }
  if (wave > 0) { sh[wave - 1][0][lane] = s1; sh[wave - 1][1][lane] = s2; }
  __syncthreads();
  if (wave == 0 && active) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float4 a = sh[w][0][lane], b = sh[w][1][lane];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float2* dst = slots + (size_t)blockIdx.y * C + c;
    store_agent(dst, make_float2(s1.x, s2.x));
    store_agent(dst + 1, make_float2(s1.y, s2.y));
    store_agent(dst + 2, make_float2(s1.z, s2.z));
    store_agent(dst + 3, make_float2(s1.w, s2.w));
  }
  // ---- grid barrier over the gridDim.y workgroups of this column block
  unsigned* arrive = bar + 2 * blockIdx.x;
  unsigned* leave = arrive + 1;
  __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): the write-through slot stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.y) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { if (err) *err = 1; break; }      // ~seconds: never in a healthy run
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // the slots are read from memory, not from stale lines
  // ---- every workgroup adds the slots of its 256 columns (slot order, fp64 like the finalize kernel)
  {
    const int col = blockIdx.x * 256 + threadIdx.x;
    double t1 = 0.0, t2 = 0.0;
    if (col < C) {
      const int nb = gridDim.y;
      int p = 0;
      for (; p + 8 <= nb; p += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slots[(size_t)(p + u) * C + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) { t1 += (double)v[u].x; t2 += (double)v[u].y; }
      }
      for (; p < nb; ++p) { const float2 v = slots[(size_t)p * C + col]; t1 += (double)v.x; t2 += (double)v.y; }
      if (blockIdx.y == 0) {
        if (dgamma) dgamma[col] = (float)t2;
        if (dbeta) dbeta[col] = (float)t1;
      }
    }
    coef_s[threadIdx.x] = make_float2((float)(t1 / M), (float)(t2 / M));
  }
  __syncthreads();
  if (active) {
    const float2 k0 = coef_s[lane * 4], k1 = coef_s[lane * 4 + 1], k2 = coef_s[lane * 4 + 2], k3 = coef_s[lane * 4 + 3];
    const float4 a = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = slot0 + j * P;
      if (r < M)
        *reinterpret_cast<float4*>(dX + (size_t)r * ldd + c) =
            make_float4(a.x * (gq[j].x - k0.x - xq[j].x * k0.y), a.y * (gq[j].y - k1.x - xq[j].y * k1.y),
                        a.z * (gq[j].z - k2.x - xq[j].z * k2.y), a.w * (gq[j].w - k3.x - xq[j].w * k3.y));
    }
  }
  // ---- the last workgroup to leave re-zeroes the counter pair for the next launch that draws it from the pool
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.y - 1) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// backward sums: norm_rowblock_cap() = 256 (one slot per block, plain sums) — swept on MI355X: 35 -> 28 us edge-sized
static inline int rowblocks(int64_t M, bool wide, bool backward = false) {
  // (forward statistics of edge-sized inputs on 256 instead of 64 row blocks: measured no gain inside the step, r03)
  const int64_t cap = (wide && backward) ? norm_rowblock_cap() : 64;
  const int64_t want = cdiv(M, 16);
  return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

}  // namespace esc

using namespace esc;

extern "C" {

int64_t esc_bn_scratch(int64_t C) { return (int64_t)NORM_ROWBLOCKS * 4 * C * 2 + 2 * C; }

int esc_bn_stats(const float* X, int64_t ld_x, int64_t M, int64_t C, float eps, float momentum,
                 float* mean, float* invstd, float* running_mean, float* running_var,
                 const float* gamma, const float* beta, float* scale, float* shift, float* scratch,
                 void* stream) {
  ESC_REQUIRE((scale == nullptr) == (shift == nullptr), "esc_bn_stats: scale/shift must come together");
  ESC_REQUIRE(X && mean && invstd && scratch, "esc_bn_stats: null pointer");
  ESC_REQUIRE(M > 1 && C > 0 && ld_x >= C && M < (1LL << 31), "esc_bn_stats: need more than 1 row per channel (M=%ld, C=%ld)", (long)M, (long)C);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = (C % 4 == 0) && (ld_x % 4 == 0) && aligned16(X) && aligned16(scratch);
  const int rb = rowblocks(M, wide);
  if (wide) esc::launch(ESC_K_NORM, bn_partial_kernel_v4, dim3((unsigned)cdiv(C, 256), rb), dim3(256), 0, s, X, ld_x, (int)M, (int)C, (float2*)scratch);
  else      esc::launch(ESC_K_NORM, bn_partial_kernel, dim3((unsigned)cdiv(C, 64), rb), dim3(256), 0, s, X, ld_x, (int)M, (int)C, (float2*)scratch);
  ESC_CHECK_LAUNCH("esc_bn_stats.partial");
  esc::launch(ESC_K_NORM, bn_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, s, (const float2*)scratch, (int)M, (int)C, rb * 4, 0, eps, momentum, mean, invstd, running_mean, running_var, gamma, beta, scale, shift);
  ESC_CHECK_LAUNCH("esc_bn_stats.finalize");
  return ESC_OK;
}

int esc_bn_stats_from_partials(const float* partials, int64_t M, int64_t C, float eps, float momentum, float* mean,
                                float* invstd, float* running_mean, float* running_var, const float* gamma,
                                const float* beta, float* scale, float* shift, void* stream) {
  ESC_REQUIRE(partials && mean && invstd, "esc_bn_stats_from_partials: null pointer");
  ESC_REQUIRE(M > 1 && C > 0 && M < (1LL << 31), "esc_bn_stats_from_partials: need more than 1 row per channel");
  ESC_REQUIRE((scale == nullptr) == (shift == nullptr), "esc_bn_stats_from_partials: scale/shift must come together");
  esc::launch(ESC_K_NORM, bn_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream,
              (const float2*)partials, (int)M, (int)C, (int)cdiv(M, 32), 32, eps, momentum, mean, invstd, running_mean,
              running_var, gamma, beta, scale, shift);
  ESC_CHECK_LAUNCH("esc_bn_stats_from_partials");
  return ESC_OK;
}

int esc_bn_stats_from_partials_rows(const float* partials, int64_t M, int64_t C, int64_t block_rows, float eps,
                                    float momentum, float* mean, float* invstd, float* running_mean,
                                    float* running_var, const float* gamma, const float* beta, float* scale,
                                    float* shift, void* stream) {
  ESC_REQUIRE(partials && mean && invstd, "esc_bn_stats_from_partials_rows: null pointer");
  ESC_REQUIRE(M > 1 && C > 0 && M < (1LL << 31) && block_rows > 0, "esc_bn_stats_from_partials_rows: need more than 1 row per channel");
  ESC_REQUIRE((scale == nullptr) == (shift == nullptr), "esc_bn_stats_from_partials_rows: scale/shift must come together");
  esc::launch(ESC_K_NORM, bn_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream,
              (const float2*)partials, (int)M, (int)C, (int)cdiv(M, block_rows), (int)block_rows, eps, momentum, mean, invstd,
              running_mean, running_var, gamma, beta, scale, shift);
  ESC_CHECK_LAUNCH("esc_bn_stats_from_partials_rows");
  return ESC_OK;
}

int esc_affine_act_fold(const float* X, int64_t ld_x, int64_t M, int64_t C, const esc_bn_fold* bn, int relu, float* Y,
                        int64_t ld_y, void* stream) {
  ESC_REQUIRE(X && Y && bn && bn->partials && bn->mean && bn->invstd, "esc_affine_act_fold: null pointer");
  ESC_REQUIRE(M > 1 && M < (1LL << 31) && C > 0 && C % 4 == 0 && bn->C == C && bn->rows > 1 && bn->block_rows > 0 && ld_x >= C && ld_y >= C &&
              ld_x % 4 == 0 && ld_y % 4 == 0 && aligned16(X) && aligned16(Y), "esc_affine_act_fold: bad sizes / alignment");
  ESC_REQUIRE((bn->scale == nullptr) == (bn->shift == nullptr), "esc_affine_act_fold: scale/shift must come together");
  BnFoldDev f{reinterpret_cast<const float2*>(bn->partials), (int)cdiv(bn->rows, bn->block_rows), (int)bn->block_rows, (int)bn->rows,
              (int)bn->C, bn->eps, bn->momentum, bn->gamma, bn->beta, bn->mean, bn->invstd, bn->scale, bn->shift,
              bn->running_mean, bn->running_var};
  // 32 rows per workgroup: every workgroup re-reads the partials (77 KB for 2 400 rows of 256 columns), so fewer and
  // fatter workgroups than the plain affine pass
  const unsigned rb = (unsigned)(cdiv(M, 32) < 1024 ? cdiv(M, 32) : 1024);
  esc::launch(ESC_K_NORM, affine_act_fold_rows, dim3((unsigned)cdiv(C, 256), rb), dim3(256), 0, (hipStream_t)stream, X, ld_x,
              (int)M, (int)C, f, relu, Y, ld_y);
  ESC_CHECK_LAUNCH("esc_affine_act_fold");
  return ESC_OK;
}

int esc_bn_apply(const float* X, int64_t ld_x, int64_t M, int64_t C, const float* mean,
                 const float* invstd, const float* gamma, const float* beta, int relu, float* Y,
                 int64_t ld_y, void* stream) {
  ESC_REQUIRE(X && Y && mean && invstd, "esc_bn_apply: null pointer");
  ESC_REQUIRE(M >= 0 && C > 0 && ld_x >= C && ld_y >= C, "esc_bn_apply: bad sizes");
  if (M == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_y % 4 == 0) && aligned16(X) && aligned16(Y);
  const int64_t work = M * (vec ? C / 4 : C);
  const unsigned blocks = (unsigned)(cdiv(work, 256) < 4096 ? cdiv(work, 256) : 4096);
  if (vec) esc::launch(ESC_K_NORM, bn_apply_kernel<4>, dim3(blocks), dim3(256), 0, s, X, ld_x, M, (int)C, mean, invstd, gamma, beta, relu, Y, ld_y);
  else     esc::launch(ESC_K_NORM, bn_apply_kernel<1>, dim3(blocks), dim3(256), 0, s, X, ld_x, M, (int)C, mean, invstd, gamma, beta, relu, Y, ld_y);
  ESC_CHECK_LAUNCH("esc_bn_apply");
  return ESC_OK;
}

int esc_affine_act(const float* X, int64_t ld_x, int64_t M, int64_t C, const float* scale, const float* shift,
                   int relu, float* Y, int64_t ld_y, void* stream) {
  ESC_REQUIRE(X && Y && scale && shift, "esc_affine_act: null pointer");
  ESC_REQUIRE(M >= 0 && C > 0 && ld_x >= C && ld_y >= C, "esc_affine_act: bad sizes");
  if (M == 0) return ESC_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_y % 4 == 0) && aligned16(X) && aligned16(Y) &&
                   aligned16(scale) && aligned16(shift);
  const int64_t work = M * (vec ? C / 4 : C);
  const unsigned blocks = (unsigned)(cdiv(work, 256) < 4096 ? cdiv(work, 256) : 4096);
  if (vec && M < (1LL << 31)) {
    const unsigned rb = (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048);        // 4 rows per wave and pass
    esc::launch(ESC_K_NORM, affine_act_rows, dim3((unsigned)cdiv(C, 256), rb), dim3(256), 0, s, X, ld_x, (int)M, (int)C, scale, shift, relu, Y, ld_y);
  }
  else if (vec) esc::launch(ESC_K_NORM, affine_act_kernel<4>, dim3(blocks), dim3(256), 0, s, X, ld_x, M, (int)C, scale, shift, relu, Y, ld_y);
  else          esc::launch(ESC_K_NORM, affine_act_kernel<1>, dim3(blocks), dim3(256), 0, s, X, ld_x, M, (int)C, scale, shift, relu, Y, ld_y);
  ESC_CHECK_LAUNCH("esc_affine_act");
  return ESC_OK;
}

int esc_bn_sync_pack(const float* mean, const float* invstd, int64_t n_local, float eps, int64_t C, int rank, int world,
                     float* buf, void* stream) {
  ESC_REQUIRE(mean && invstd && buf && C > 0 && world > 0 && rank >= 0 && rank < world && n_local > 0, "esc_bn_sync_pack: bad argument");
  esc::launch(ESC_K_NORM, bn_sync_pack_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, invstd,
              (float)n_local, eps, (int)C, rank, world, buf);
  ESC_CHECK_LAUNCH("esc_bn_sync_pack");
  return ESC_OK;
}

int esc_bn_sync_finalize(const float* buf, int world, int64_t C, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, const float* gamma, const float* beta, float* scale,
                         float* shift, float* n_total, void* stream) {
  ESC_REQUIRE(buf && mean && invstd && C > 0 && world > 0, "esc_bn_sync_finalize: bad argument");
  ESC_REQUIRE((scale == nullptr) == (shift == nullptr), "esc_bn_sync_finalize: scale/shift must come together");
  esc::launch(ESC_K_NORM, bn_sync_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, buf, world,
              (int)C, eps, momentum, mean, invstd, running_mean, running_var, gamma, beta, scale, shift, n_total);
  ESC_CHECK_LAUNCH("esc_bn_sync_finalize");
  return ESC_OK;
}

int esc_bn_sync_coef(float* coef, int64_t C, const float* n_total, void* stream) {
  ESC_REQUIRE(coef && n_total && C > 0, "esc_bn_sync_coef: bad argument");
  esc::launch(ESC_K_NORM, bn_sync_coef_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
              reinterpret_cast<float2*>(coef), (int)C, n_total);
  ESC_CHECK_LAUNCH("esc_bn_sync_coef");
  return ESC_OK;
}

int esc_bn_eval_coef(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                     float eps, int64_t C, float* scale, float* shift, void* stream) {
  ESC_REQUIRE(running_mean && running_var && scale && shift && C > 0, "esc_bn_eval_coef: bad argument");
  esc::launch(ESC_K_NORM, bn_eval_coef_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
              running_mean, running_var, gamma, beta, eps, (int)C, scale, shift);
  ESC_CHECK_LAUNCH("esc_bn_eval_coef");
  return ESC_OK;
}

// column sums of the BatchNorm backward: coef[c] = (sum g, sum g*xhat) / divisor, dgamma = sum g*xhat, dbeta = sum g
// (g = dY * act'(.)).  divisor = M for the local BatchNorm; 1 when the sums are still to be all-reduced (SyncBN).
static int bn_bwd_reduce(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY, int64_t ld_dy,
                         int64_t M, int64_t C, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, int relu, int64_t divisor, bool allow_fuse, float* dgamma, float* dbeta,
                         float2* partial, float2* coef, hipStream_t s) {
  const bool wide = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_dy % 4 == 0) && (!Y || ld_y % 4 == 0) && aligned16(X) &&
                    aligned16(dY) && (!Y || aligned16(Y)) && aligned16(mean) && aligned16(invstd) &&
                    (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) && aligned16(partial);
  // node-sized inputs: few fat workgroups (>= 32 rows each) whose last one folds the <= 64 slots itself (knob 8);
  // edge-sized: many workgroups + a wide finalize launch (one workgroup cannot pull hundreds of slots quickly)
  const bool fuse = allow_fuse && wide && M <= 4096 && divisor == M && last_block_finalize();
  const int rb = fuse ? (int)(cdiv(M, 32) < 64 ? cdiv(M, 32) : 64) : rowblocks(M, wide, true);
  if (wide) {
    const dim3 grid((unsigned)cdiv(C, 256), rb);
    unsigned* tk = fuse ? tickets((int)grid.x) : nullptr;
    ESC_REQUIRE(!fuse || tk != nullptr, "esc_bn_bwd: no ticket counters");
#define ESC_BWD_PARTIAL(A, H) esc::launch(ESC_K_NORM, bn_bwd_partial_kernel_v4<A, H>, grid, dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, tk, dgamma, dbeta, coef)
    if (relu == 0)      ESC_BWD_PARTIAL(0, false);
    else if (relu == 1) { if (Y) ESC_BWD_PARTIAL(1, true); else ESC_BWD_PARTIAL(1, false); }
    else                { if (Y) ESC_BWD_PARTIAL(2, true); else ESC_BWD_PARTIAL(2, false); }
#undef ESC_BWD_PARTIAL
    if (!fuse) {
      ESC_CHECK_LAUNCH("esc_bn_bwd.partial");
      esc::launch(ESC_K_NORM, bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, s, partial, (int)divisor, (int)C, rb, dgamma, dbeta, coef);
    }
  }
  else {
    esc::launch(ESC_K_NORM, bn_bwd_partial_kernel, dim3((unsigned)cdiv(C, 64), rb), dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, relu, gamma, beta, partial);
    ESC_CHECK_LAUNCH("esc_bn_bwd.partial");
    esc::launch(ESC_K_NORM, bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, s, partial, (int)divisor, (int)C, rb * 4, dgamma, dbeta, coef);
  }
  ESC_CHECK_LAUNCH("esc_bn_bwd.finalize");
  return ESC_OK;
}

// dX = gamma * invstd * (g - coef.x - xhat * coef.y)
static int bn_bwd_apply_impl(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY, int64_t ld_dy,
                             int64_t M, int64_t C, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, int relu, const float2* coef, float* dX, int64_t ld_dx, hipStream_t s) {
  const bool vec = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_dy % 4 == 0) && (ld_dx % 4 == 0) && (!Y || ld_y % 4 == 0) &&
                   aligned16(X) && aligned16(dY) && aligned16(dX) && (!Y || aligned16(Y)) &&
                   (!gamma || aligned16(gamma));
  const int64_t work = M * (vec ? C / 4 : C);
  const unsigned blocks = (unsigned)(cdiv(work, 256) < 4096 ? cdiv(work, 256) : 4096);
#define ESC_BWD_APPLY(V, A, H) esc::launch(ESC_K_NORM, bn_bwd_apply_kernel<V, A, H>, dim3(blocks), dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, M, (int)C, mean, invstd, gamma, beta, coef, dX, ld_dx)
#define ESC_BWD_APPLY_V(V)                                                                 \
  if (relu == 0)      ESC_BWD_APPLY(V, 0, false);                                           \
  else if (relu == 1) { if (Y) ESC_BWD_APPLY(V, 1, true); else ESC_BWD_APPLY(V, 1, false); } \
  else                { if (Y) ESC_BWD_APPLY(V, 2, true); else ESC_BWD_APPLY(V, 2, false); }
  const bool rows_form = vec && aligned16(mean) && aligned16(invstd) && (!beta || aligned16(beta)) && aligned16(coef);
  if (rows_form) {
    const dim3 grid((unsigned)cdiv(C, 256), (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048));
#define ESC_BWD_ROWS(A, H) esc::launch(ESC_K_NORM, bn_bwd_apply_rows<A, H, false>, grid, dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, coef, dX, ld_dx, 0, (float*)nullptr, (float*)nullptr)
    if (relu == 0)      ESC_BWD_ROWS(0, false);
    else if (relu == 1) { if (Y) ESC_BWD_ROWS(1, true); else ESC_BWD_ROWS(1, false); }
    else                { if (Y) ESC_BWD_ROWS(2, true); else ESC_BWD_ROWS(2, false); }
#undef ESC_BWD_ROWS
  }
  else if (vec) { ESC_BWD_APPLY_V(4) } else { ESC_BWD_APPLY_V(1) }
#undef ESC_BWD_APPLY_V
#undef ESC_BWD_APPLY
  ESC_CHECK_LAUNCH("esc_bn_bwd.apply");
  return ESC_OK;
}

int esc_bn_bwd(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
               int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
               const float* gamma, const float* beta, int relu, float* dX, int64_t ld_dx, float* dgamma,
               float* dbeta, float* scratch, void* stream) {
  ESC_REQUIRE(X && dY && dX && mean && invstd && scratch, "esc_bn_bwd: null pointer");
  ESC_REQUIRE(M > 0 && C > 0 && ld_x >= C && ld_dy >= C && ld_dx >= C && (!Y || ld_y >= C) && M < (1LL << 31), "esc_bn_bwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  float2* partial = (float2*)scratch;
  float2* coef = partial + (size_t)NORM_ROWBLOCKS * 4 * C;
  // node-sized: 32 fat row blocks leave 32 partial slots and the apply kernel adds them itself — no finalize launch
  const bool all16 = aligned16(X) && aligned16(dY) && aligned16(dX) && (!Y || aligned16(Y)) && aligned16(mean) && aligned16(invstd) &&
                     (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) && aligned16(partial) &&
                     (!dgamma || aligned16(dgamma)) && (!dbeta || aligned16(dbeta));
  if (bn_bwd_one_launch() && M >= 64 && M <= 4096 && C % 4 == 0 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0 && (!Y || ld_y % 4 == 0) && all16) {
    const dim3 grid((unsigned)cdiv(C, 256), (unsigned)cdiv(M, 16));
    unsigned* bar = tickets(2 * (int)grid.x);
    ESC_REQUIRE(bar != nullptr, "esc_bn_bwd: no barrier counters");
    int* noerr = nullptr;
#define ESC_BWD_NODE(A, H) esc::launch(ESC_K_NORM, bn_bwd_node_kernel<A, H>, grid, dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, bar, dgamma, dbeta, dX, ld_dx, noerr)
    if (relu == 0)      ESC_BWD_NODE(0, false);
    else if (relu == 1) { if (Y) ESC_BWD_NODE(1, true); else ESC_BWD_NODE(1, false); }
    else                { if (Y) ESC_BWD_NODE(2, true); else ESC_BWD_NODE(2, false); }
#undef ESC_BWD_NODE
    ESC_CHECK_LAUNCH("esc_bn_bwd.node");
    return ESC_OK;
  }
  if (bn_bwd_fold() && M >= 64 && M <= 4096 && C % 4 == 0 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0 && (!Y || ld_y % 4 == 0) && all16 &&
      !last_block_finalize()) {
    const int rb = (int)(cdiv(M, 32) < 32 ? cdiv(M, 32) : 32);
    const dim3 grid((unsigned)cdiv(C, 256), rb);
    float* nof = nullptr; float2* noc = nullptr; unsigned* tk = nullptr;
#define ESC_BWD_PARTIAL(A, H) esc::launch(ESC_K_NORM, bn_bwd_partial_kernel_v4<A, H>, grid, dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, tk, nof, nof, noc)
    if (relu == 0)      ESC_BWD_PARTIAL(0, false);
    else if (relu == 1) { if (Y) ESC_BWD_PARTIAL(1, true); else ESC_BWD_PARTIAL(1, false); }
    else                { if (Y) ESC_BWD_PARTIAL(2, true); else ESC_BWD_PARTIAL(2, false); }
#undef ESC_BWD_PARTIAL
    ESC_CHECK_LAUNCH("esc_bn_bwd.partial");
    const dim3 agrid((unsigned)cdiv(C, 256), (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048));
    const float2* cpart = partial;
#define ESC_BWD_ROWS(A, H) esc::launch(ESC_K_NORM, bn_bwd_apply_rows<A, H, true>, agrid, dim3(256), 0, s, X, ld_x, Y, ld_y, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, cpart, dX, ld_dx, rb, dgamma, dbeta)
    if (relu == 0)      ESC_BWD_ROWS(0, false);
    else if (relu == 1) { if (Y) ESC_BWD_ROWS(1, true); else ESC_BWD_ROWS(1, false); }
    else                { if (Y) ESC_BWD_ROWS(2, true); else ESC_BWD_ROWS(2, false); }
#undef ESC_BWD_ROWS
    ESC_CHECK_LAUNCH("esc_bn_bwd.apply_fold");
    return ESC_OK;
  }
  int rc = bn_bwd_reduce(X, ld_x, Y, ld_y, dY, ld_dy, M, C, mean, invstd, gamma, beta, relu, M, true, dgamma, dbeta, partial, coef, s);
  if (rc != ESC_OK) return rc;
  return bn_bwd_apply_impl(X, ld_x, Y, ld_y, dY, ld_dy, M, C, mean, invstd, gamma, beta, relu, coef, dX, ld_dx, s);
}

int esc_bn_bwd_dropout_ok(int64_t C, int64_t ld_x, int64_t ld_dy, int64_t ld_dx) {
  return C % 4 == 0 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_dx % 4 == 0;
}

int esc_bn_bwd_dropout(const float* X, int64_t ld_x, const float* dY, int64_t ld_dy, int64_t M, int64_t C, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, int relu, const uint8_t* mask, float p,
                       int mask_on_output, float* dX, int64_t ld_dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
  ESC_REQUIRE(X && dY && dX && mean && invstd && scratch && mask, "esc_bn_bwd_dropout: null pointer");
  ESC_REQUIRE(M > 0 && C > 0 && ld_x >= C && ld_dy >= C && ld_dx >= C && M < (1LL << 31) && p > 0.f && p < 1.f && (relu == 0 || relu == 1),
              "esc_bn_bwd_dropout: bad arguments");
  float2* partial = (float2*)scratch;
  float2* coef = partial + (size_t)NORM_ROWBLOCKS * 4 * C;
  ESC_REQUIRE(esc_bn_bwd_dropout_ok(C, ld_x, ld_dy, ld_dx) && aligned16(X) && aligned16(dY) && aligned16(dX) && aligned16(mean) &&
              aligned16(invstd) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) && aligned16(partial) && aligned16(coef) &&
              (reinterpret_cast<uintptr_t>(mask) & 3) == 0, "esc_bn_bwd_dropout: operands must be 16-byte aligned with widths a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const float dscale = (float)(1.0 / (1.0 - (double)p));
  const int rb = rowblocks(M, true, true);
  const dim3 grid((unsigned)cdiv(C, 256), rb);
  const unsigned char* mk = (const unsigned char*)mask;
  if (mask_on_output) {        // the sums are those of the plain BatchNorm backward
    const float* noy = nullptr; unsigned* tk = nullptr; float* nof = nullptr; float2* noc = nullptr;
    if (relu) esc::launch(ESC_K_NORM, bn_bwd_partial_kernel_v4<1, false>, grid, dim3(256), 0, s, X, ld_x, noy, (int64_t)0, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, tk, nof, nof, noc);
    else      esc::launch(ESC_K_NORM, bn_bwd_partial_kernel_v4<0, false>, grid, dim3(256), 0, s, X, ld_x, noy, (int64_t)0, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, tk, nof, nof, noc);
  } else {
    if (relu) esc::launch(ESC_K_NORM, bn_bwd_partial_drop_kernel<1>, grid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, mk, dscale);
    else      esc::launch(ESC_K_NORM, bn_bwd_partial_drop_kernel<0>, grid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, partial, mk, dscale);
  }
  ESC_CHECK_LAUNCH("esc_bn_bwd_dropout.partial");
  esc::launch(ESC_K_NORM, bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, s, (const float2*)partial, (int)M, (int)C, rb, dgamma, dbeta, coef);
  ESC_CHECK_LAUNCH("esc_bn_bwd_dropout.finalize");
  const dim3 agrid((unsigned)cdiv(C, 256), (unsigned)(cdiv(M, 16) < 2048 ? cdiv(M, 16) : 2048));
  const float2* kc = coef;
  if (mask_on_output) {
    if (relu) esc::launch(ESC_K_NORM, bn_bwd_apply_rows_drop<1, false, true>, agrid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, kc, dX, ld_dx, mk, dscale);
    else      esc::launch(ESC_K_NORM, bn_bwd_apply_rows_drop<0, false, true>, agrid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, kc, dX, ld_dx, mk, dscale);
  } else {
    if (relu) esc::launch(ESC_K_NORM, bn_bwd_apply_rows_drop<1, true, false>, agrid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, kc, dX, ld_dx, mk, dscale);
    else      esc::launch(ESC_K_NORM, bn_bwd_apply_rows_drop<0, true, false>, agrid, dim3(256), 0, s, X, ld_x, dY, ld_dy, (int)M, (int)C, mean, invstd, gamma, beta, kc, dX, ld_dx, mk, dscale);
  }
  ESC_CHECK_LAUNCH("esc_bn_bwd_dropout.apply");
  return ESC_OK;
}

int esc_bn_bwd_sums(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
                    int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, int relu, float* sums, float* dgamma,
                    float* dbeta, float* scratch, void* stream) {
  ESC_REQUIRE(X && dY && sums && mean && invstd && scratch, "esc_bn_bwd_sums: null pointer");
  ESC_REQUIRE(M > 0 && C > 0 && ld_x >= C && ld_dy >= C && (!Y || ld_y >= C) && M < (1LL << 31), "esc_bn_bwd_sums: bad sizes");
  return bn_bwd_reduce(X, ld_x, Y, ld_y, dY, ld_dy, M, C, mean, invstd, gamma, beta, relu, 1, false, dgamma, dbeta,
                       (float2*)scratch, (float2*)sums, (hipStream_t)stream);
}

int esc_bn_bwd_coef(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY, int64_t ld_dy, int64_t M,
                    int64_t C, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                    float* coef, float* dgamma, float* dbeta, float* scratch, void* stream) {
  ESC_REQUIRE(X && dY && coef && mean && invstd && scratch, "esc_bn_bwd_coef: null pointer");
  ESC_REQUIRE(M > 0 && C > 0 && ld_x >= C && ld_dy >= C && (!Y || ld_y >= C) && M < (1LL << 31), "esc_bn_bwd_coef: bad sizes");
  return bn_bwd_reduce(X, ld_x, Y, ld_y, dY, ld_dy, M, C, mean, invstd, gamma, beta, relu, M, true, dgamma, dbeta,
                       (float2*)scratch, (float2*)coef, (hipStream_t)stream);
}

int esc_bn_bwd_coef_from_partials(const float* partial, int64_t slots, int64_t M, int64_t C, float* coef, float* dgamma,
                                  float* dbeta, void* stream) {
  ESC_REQUIRE(partial && coef && slots > 0 && M > 0 && C > 0 && M < (1LL << 31) && slots < (1LL << 31), "esc_bn_bwd_coef_from_partials: bad arguments");
  esc::launch(ESC_K_NORM, bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream,
              reinterpret_cast<const float2*>(partial), (int)M, (int)C, (int)slots, dgamma, dbeta, reinterpret_cast<float2*>(coef));
  ESC_CHECK_LAUNCH("esc_bn_bwd_coef_from_partials");
  return ESC_OK;
}

int esc_bn_bwd_apply(const float* X, int64_t ld_x, const float* Y, int64_t ld_y, const float* dY,
                     int64_t ld_dy, int64_t M, int64_t C, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, const float* coef, float* dX,
                     int64_t ld_dx, void* stream) {
  ESC_REQUIRE(X && dY && dX && mean && invstd && coef, "esc_bn_bwd_apply: null pointer");
  ESC_REQUIRE(M > 0 && C > 0 && ld_x >= C && ld_dy >= C && ld_dx >= C && (!Y || ld_y >= C) && M < (1LL << 31), "esc_bn_bwd_apply: bad sizes");
  return bn_bwd_apply_impl(X, ld_x, Y, ld_y, dY, ld_dy, M, C, mean, invstd, gamma, beta, relu, (const float2*)coef, dX,
                           ld_dx, (hipStream_t)stream);
}

}  // extern "C"
