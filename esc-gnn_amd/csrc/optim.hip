// optim.hip — L1 regression loss and the Adam update of the training step.
//
// Replaces torch.nn.L1Loss()(model(data), y) + backward seed (/root/reference/run_graphcount.py:500-503)
// and torch.optim.Adam(...).step() (:478,:505).  One flat parameter buffer => one launch per step.
#include "common.h"

namespace esc {

// single workgroup: M is the node count of a batch (thousands).  fp64 accumulation, fixed order.
__global__ __launch_bounds__(1024) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                       int64_t M, double denom, float grad_scale,
                                                       float* __restrict__ loss, float* __restrict__ dpred) {
  ESC_PRIO();
  __shared__ double sh[16];
  double acc = 0.0;
  const float gs = (float)((double)grad_scale / denom);
  for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
    const float d = pred[i] - y[i];
    acc += (double)fabsf(d);
    if (dpred) dpred[i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    loss[0] = (float)(t / denom);
  }
}

// BCEWithLogitsLoss()(pred[is_labeled], y[is_labeled]) with is_labeled = (y == y): the OGB training criterion
// (/root/reference/run_ogb_mol.py:65-72).  Single workgroup, fp64 accumulation, fixed order.
__global__ __launch_bounds__(1024) void bce_logits_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                          int64_t M, double denom_in, float* __restrict__ loss,
                                                          float* __restrict__ dpred) {
  ESC_PRIO();
  __shared__ double sh[16];
  __shared__ double shc[16];
  double acc = 0.0, cnt = 0.0;
  for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
    const float t = y[i], x = pred[i];
    if (t == t) {
      acc += (double)(fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
      cnt += 1.0;
    }
  }
  acc = wave_sum(acc);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = acc; shc[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  double t = 0.0, c = 0.0;
  for (int w = 0; w < 16; ++w) { t += sh[w]; c += shc[w]; }
  const double denom = denom_in > 0.0 ? denom_in : c;
  if (threadIdx.x == 0) loss[0] = denom > 0.0 ? (float)(t / denom) : 0.f;
  if (dpred) {
    const float inv = denom > 0.0 ? (float)(1.0 / denom) : 0.f;
    for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
      const float tt = y[i], x = pred[i];
      const float sg = 1.f / (1.f + expf(-x));
      dpred[i] = (tt == tt) ? (sg - tt) * inv : 0.f;
    }
  }
}

// torch.optim.Adam single-tensor arithmetic, in torch's operation order:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p += -(lr/bc1) * m/denom
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float one_minus_b1, float b2, float one_minus_b2,
                                                   float bc2_sqrt, float eps, float neg_step,
                                                   const float* __restrict__ grad_denom) {
  ESC_PRIO();
  // grad_denom (device scalar, may be null): the all-reduced bucket holds SUMS over the global batch; dividing
  // here saves a pass over the bucket (same rounding as grad.div_(total) followed by the plain update)
  const float den = grad_denom ? grad_denom[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = grad_denom ? g[i] / den : g[i];
    const float mi = m[i] + one_minus_b1 * (gi - m[i]);
    const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] + neg_step * (mi / denom);
  }
}

}  // namespace esc

using namespace esc;

extern "C" {

int esc_l1_loss(const float* pred, const float* y, int64_t M, int64_t denom, float grad_scale, float* loss,
                float* dpred, void* stream) {
  ESC_REQUIRE(pred && y && loss, "esc_l1_loss: null pointer");
  ESC_REQUIRE(M > 0 && denom > 0, "esc_l1_loss: empty batch");
  esc::launch(-1, l1_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, y, M, (double)denom, grad_scale, loss, dpred);
  ESC_CHECK_LAUNCH("esc_l1_loss");
  return ESC_OK;
}

int esc_bce_logits_loss(const float* pred, const float* y, int64_t M, int64_t denom, float* loss, float* dpred,
                        void* stream) {
  ESC_REQUIRE(pred && y && loss, "esc_bce_logits_loss: null pointer");
  ESC_REQUIRE(M > 0, "esc_bce_logits_loss: empty batch");
  esc::launch(-1, bce_logits_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, y, M, (double)denom, loss, dpred);
  ESC_CHECK_LAUNCH("esc_bce_logits_loss");
  return ESC_OK;
}

int esc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
  return esc_adam_step_scaled(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, nullptr, stream);
}

int esc_adam_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                         double lr, double beta1, double beta2, double eps, int64_t step,
                         const float* grad_denom, void* stream) {
  ESC_REQUIRE(param && grad && exp_avg && exp_avg_sq, "esc_adam_step: null pointer");
  ESC_REQUIRE(n >= 0 && step >= 1, "esc_adam_step: bad n/step");
  if (n == 0) return ESC_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const unsigned blocks = (unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048);
  esc::launch(-1, adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                     (float)(-lr / bc1), grad_denom);
  ESC_CHECK_LAUNCH("esc_adam_step");
  return ESC_OK;
}

}  // extern "C"
