// ns_loss.hip -- BCEWithLogitsKgeLoss over the [n, 1 + K] score block of a negative-sampling slot
// (kge/util/loss.py:136-189 on the scores TrainingJobNegativeSampling._process_subbatch assembles,
// train_negative_sampling.py:120-151: column 0 = the positive, columns 1.. = its negatives), forward and gradient
// in ONE pass over the block:
//   kind 0  "bce"                   sum_j l(x_j, y_j)                                      (reduction "sum")
//   kind 1  "bce_mean"              ( l(x_0, 1) + sum_{j>=1} l(x_j, 0) / K ) / 2           (loss.py:160-168)
//   kind 2  "bce_self_adversarial"  ( l(x_0, 1) + sum_{j>=1} w_j l(x_j, 0) ) / 2,  w = softmax_j(T x_j), detached
//                                                                                          (loss.py:169-186)
// with x = score + offset (train.loss_arg, loss.py:153-154) and l = torch's BCEWithLogitsLoss element:
// (1 - y) x + m + log(exp(-m) + exp(-x - m)), m = max(-x, 0).  The reference spends ~15 launches per slot on this, and
// the self-adversarial form two torch.nonzero calls = two device -> host waits per slot and step.
// One wave per row; HBM-bound on a 2 MB block (n = 512, K = 1000): read once, gradient written once.
#include "common.hpp"

namespace kge {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float bce_elem(float x, float y) {
  const float m = __builtin_fmaxf(-x, 0.0f);
  return (1.0f - y) * x + m + __logf(__expf(-m) + __expf(-x - m));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int KIND>
__global__ __launch_bounds__(256) void ns_bce_kernel(const float* __restrict__ sc, long long ld, long long n, long long c,
                                                     float offset, float temp, float* __restrict__ loss_rows,
                                                     float* __restrict__ grad, long long ldg) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* x = sc + row * ld;
  float* g = grad ? grad + row * ldg : nullptr;
  const float K = (float)(c - 1);
  float wmax = -__builtin_inff(), wsum = 0.0f;
  if (KIND == 2) {  // softmax statistics of the negatives' T (x + offset)
    for (long long j = 1 + lane; j < c; j += 64) wmax = __builtin_fmaxf(wmax, temp * (x[j] + offset));
    wmax = wave_max(wmax);
    for (long long j = 1 + lane; j < c; j += 64) wsum += __expf(temp * (x[j] + offset) - wmax);
    wsum = wave_sum(wsum);
  }
  float acc = 0.0f;
  for (long long j = lane; j < c; j += 64) {
    const float v = x[j] + offset;
    const float y = j == 0 ? 1.0f : 0.0f;
    const float l = bce_elem(v, y);
    const float d = sigmoidf(v) - y;  // d l / d x
    float w;
    if (KIND == 0) w = 1.0f;
    else if (KIND == 1) w = j == 0 ? 0.5f : 0.5f / K;
    else w = j == 0 ? 0.5f : 0.5f * __expf(temp * v - wmax) / wsum;
    acc += w * l;
    if (g) g[j] = w * d;
  }
  acc = wave_sum(acc);
  if (lane == 0) loss_rows[row] = acc;
}

int run_ns_bce(int kind, const float* scores, long long ld, long long n, long long c, float offset, float temp,
               float* loss_rows, float* grad, long long ldg, hipStream_t st) {
  if (n == 0) return KGE_OK;
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
  if (kind == 0) hipLaunchKernelGGL(ns_bce_kernel<0>, grid, block, 0, st, scores, ld, n, c, offset, temp, loss_rows, grad, ldg);
  else if (kind == 1) hipLaunchKernelGGL(ns_bce_kernel<1>, grid, block, 0, st, scores, ld, n, c, offset, temp, loss_rows, grad, ldg);
  else hipLaunchKernelGGL(ns_bce_kernel<2>, grid, block, 0, st, scores, ld, n, c, offset, temp, loss_rows, grad, ldg);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
