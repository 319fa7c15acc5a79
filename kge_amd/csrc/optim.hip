// optim.hip -- dense Adagrad step over a parameter table in ONE pass.
//
// After the scoring and the loss are fused, the optimizer is the largest remaining item of a 1vsAll
// training step: torch.optim.Adagrad's multi-tensor path is five element-wise kernels (addcmul,
// sqrt, add eps, mul -clr, addcdiv: kge/util/optimizer.py:15-20 -> torch.optim.Adagrad), i.e. the
// tables, gradients and accumulators cross HBM ~14 times per element, and a mixed-precision model
// re-casts both tables to bf16 afterwards.  Here: read param, grad, sum once, write param and sum
// once, and (optionally) the bf16 copy of the new param in the same pass: 20-22 bytes per element.
//
// Arithmetic = torch's multi-tensor sequence with every operation rounded on its own:
//   g   = grad + weight_decay * param              (only if weight_decay != 0)
//   sum = sum + g * g
//   p   = p + (minus_clr * g) / (sqrt(sum) + eps)
// (-ffp-contract=off; torch's kernels may contract the first two lines into fma: results agree to
// an ulp or two, tests/test_gpu_optim.py states the tolerance.)
#include "common.hpp"

namespace kge {

__global__ __launch_bounds__(256) void adagrad_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                      float* __restrict__ sum, long long count, float minus_clr,
                                                      float weight_decay, float eps,
                                                      unsigned short* __restrict__ copy16) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  if (i + 4 <= count) {
    f32x4 p = *reinterpret_cast<const f32x4*>(param + i);
    f32x4 g = *reinterpret_cast<const f32x4*>(grad + i);
    f32x4 s = *reinterpret_cast<const f32x4*>(sum + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e];
      if (weight_decay != 0.0f) ge = ge + weight_decay * p[e];
      s[e] = s[e] + ge * ge;
      p[e] = p[e] + (minus_clr * ge) / (__builtin_sqrtf(s[e]) + eps);
    }
    *reinterpret_cast<f32x4*>(param + i) = p;
    *reinterpret_cast<f32x4*>(sum + i) = s;
    if (copy16 != nullptr) {
      u32x2 c = {bf16_pack(p[0], p[1]), bf16_pack(p[2], p[3])};
      *reinterpret_cast<u32x2*>(copy16 + i) = c;
    }
  } else {
    for (long long j = i; j < count; ++j) {
      float ge = grad[j];
      if (weight_decay != 0.0f) ge = ge + weight_decay * param[j];
      const float s = sum[j] + ge * ge;
      const float p = param[j] + (minus_clr * ge) / (__builtin_sqrtf(s) + eps);
      sum[j] = s;
      param[j] = p;
      if (copy16 != nullptr) copy16[j] = (unsigned short)(bf16_pack(p, 0.0f) & 0xffffu);
    }
  }
}

// Several tables in ONE launch (kge_adagrad_step_multi): a training step updates the entity and the relation table,
// and the relation table's launch is all latency (237 x 512 elements at the FB15k-237 shape: ~2.5 us of a graph
// replay for 0.1 us of traffic).  Workgroup b serves the segment whose block range holds it; arithmetic as above.
struct AdagradSegs {
  float* param[KGE_ADAGRAD_MAX_SEGS];
  const float* grad[KGE_ADAGRAD_MAX_SEGS];
  float* sum[KGE_ADAGRAD_MAX_SEGS];
  unsigned short* copy16[KGE_ADAGRAD_MAX_SEGS];
  long long count[KGE_ADAGRAD_MAX_SEGS];
  unsigned int first_block[KGE_ADAGRAD_MAX_SEGS + 1];
  float minus_clr[KGE_ADAGRAD_MAX_SEGS], weight_decay[KGE_ADAGRAD_MAX_SEGS], eps[KGE_ADAGRAD_MAX_SEGS];
  int num;
};

__global__ __launch_bounds__(256) void adagrad_multi_kernel(AdagradSegs a) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < KGE_ADAGRAD_MAX_SEGS; ++j)
    if (j < a.num && blockIdx.x >= a.first_block[j]) k = j;
  float* __restrict__ param = a.param[k];
  const float* __restrict__ grad = a.grad[k];
  float* __restrict__ sum = a.sum[k];
  unsigned short* __restrict__ copy16 = a.copy16[k];
  const long long count = a.count[k];
  const float minus_clr = a.minus_clr[k], weight_decay = a.weight_decay[k], eps = a.eps[k];
  const long long i = ((long long)(blockIdx.x - a.first_block[k]) * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  if (i + 4 <= count) {
    f32x4 p = *reinterpret_cast<const f32x4*>(param + i);
    f32x4 g = *reinterpret_cast<const f32x4*>(grad + i);
    f32x4 s = *reinterpret_cast<const f32x4*>(sum + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e];
      if (weight_decay != 0.0f) ge = ge + weight_decay * p[e];
      s[e] = s[e] + ge * ge;
      p[e] = p[e] + (minus_clr * ge) / (__builtin_sqrtf(s[e]) + eps);
    }
    *reinterpret_cast<f32x4*>(param + i) = p;
    *reinterpret_cast<f32x4*>(sum + i) = s;
    if (copy16 != nullptr) {
      u32x2 c = {bf16_pack(p[0], p[1]), bf16_pack(p[2], p[3])};
      *reinterpret_cast<u32x2*>(copy16 + i) = c;
    }
  } else {
    for (long long j = i; j < count; ++j) {
      float ge = grad[j];
      if (weight_decay != 0.0f) ge = ge + weight_decay * param[j];
      const float s = sum[j] + ge * ge;
      const float p = param[j] + (minus_clr * ge) / (__builtin_sqrtf(s) + eps);
      sum[j] = s;
      param[j] = p;
      if (copy16 != nullptr) copy16[j] = (unsigned short)(bf16_pack(p, 0.0f) & 0xffffu);
    }
  }
}

int run_adagrad_multi(const kge_adagrad_seg* segs, int num, hipStream_t st) {
  AdagradSegs a{};
  long long blocks = 0;
  int k = 0;
  for (int j = 0; j < num; ++j) {
    if (segs[j].count == 0) continue;
    a.param[k] = segs[j].param;
    a.grad[k] = segs[j].grad;
    a.sum[k] = segs[j].state_sum;
    a.copy16[k] = (unsigned short*)segs[j].bf16_copy;
    a.count[k] = segs[j].count;
    a.minus_clr[k] = segs[j].minus_clr;
    a.weight_decay[k] = segs[j].weight_decay;
    a.eps[k] = segs[j].eps;
    a.first_block[k] = (unsigned int)blocks;
    blocks += (segs[j].count + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
    ++k;
  }
  if (k == 0) return KGE_OK;
  a.first_block[k] = (unsigned int)blocks;
  a.num = k;
  hipLaunchKernelGGL(adagrad_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_adagrad(float* param, const float* grad, float* sum, long long count, float minus_clr, float weight_decay,
                float eps, unsigned short* copy16, hipStream_t st) {
  if (count == 0) return KGE_OK;
  const long long blocks = (count + 1023) / 1024;
  if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adagrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, sum, count, minus_clr,
                     weight_decay, eps, copy16);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- Adam (torch.optim.Adam, single-tensor sequence, amsgrad / maximize off), one pass ------------
//   g          = grad + weight_decay * param            (only if weight_decay != 0)
//   exp_avg    = exp_avg + (g - exp_avg) * (1 - beta1)   (lerp)
//   exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) * g * g
//   param      = param + (-step_size) * exp_avg / (sqrt(exp_avg_sq) / bias_corr2_sqrt + eps)
// step_size = lr / (1 - beta1^t), bias_corr2_sqrt = sqrt(1 - beta2^t): computed by the caller.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m1, float* __restrict__ m2, long long count,
                                                   float step_size, float bc2_sqrt, float omb1, float beta2, float omb2,
                                                   float weight_decay, float eps, unsigned short* __restrict__ copy16) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= count) return;
  auto one = [&](float p, float g, float& a, float& b) {
    if (weight_decay != 0.0f) g = g + weight_decay * p;
    a = a + (g - a) * omb1;
    b = b * beta2 + (omb2 * g) * g;
    const float denom = __builtin_sqrtf(b) / bc2_sqrt + eps;
    return p + (-step_size) * (a / denom);
  };
  if (i0 + 4 <= count) {
    f32x4 p = *reinterpret_cast<const f32x4*>(param + i0);
    const f32x4 g = *reinterpret_cast<const f32x4*>(grad + i0);
    f32x4 a = *reinterpret_cast<const f32x4*>(m1 + i0);
    f32x4 b = *reinterpret_cast<const f32x4*>(m2 + i0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ae = a[e], be = b[e];
      p[e] = one(p[e], g[e], ae, be);
      a[e] = ae;
      b[e] = be;
    }
    *reinterpret_cast<f32x4*>(param + i0) = p;
    *reinterpret_cast<f32x4*>(m1 + i0) = a;
    *reinterpret_cast<f32x4*>(m2 + i0) = b;
    if (copy16 != nullptr) {
      u32x2 c = {bf16_pack(p[0], p[1]), bf16_pack(p[2], p[3])};
      *reinterpret_cast<u32x2*>(copy16 + i0) = c;
    }
  } else {
    for (long long j = i0; j < count; ++j) {
      float a = m1[j], b = m2[j];
      const float p = one(param[j], grad[j], a, b);
      m1[j] = a;
      m2[j] = b;
      param[j] = p;
      if (copy16 != nullptr) copy16[j] = (unsigned short)(bf16_pack(p, 0.0f) & 0xffffu);
    }
  }
}

int run_adam(float* param, const float* grad, float* m1, float* m2, long long count, float step_size, float bc2_sqrt,
             float omb1, float beta2, float omb2, float weight_decay, float eps, unsigned short* copy16, hipStream_t st) {
  if (count == 0) return KGE_OK;
  const long long blocks = (count + 1023) / 1024;
  if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, m1, m2, count, step_size,
                     bc2_sqrt, omb1, beta2, omb2, weight_decay, eps, copy16);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- row-sparse Adagrad (lookup_embedder.yaml:78-81 `sparse: True` -> torch.optim.Adagrad's sparse
// path): only the `nrows` listed rows of the table have a gradient (coalesced: unique row ids, one
// [dim] value row each).  One wave per listed row:
//   sum[row] += g * g;   param[row] += minus_clr * g / (sqrt(sum[row]) + eps)
// The untouched rows are neither read nor written: a 4.6 M-row table with 1 M touched rows moves a
// fifth of the dense sweep's bytes, and there is no dense gradient to zero.
__global__ __launch_bounds__(256) void adagrad_rows_kernel(float* __restrict__ param, long long param_ld,
                                                           const float* __restrict__ grows, long long g_ld,
                                                           float* __restrict__ sum, long long sum_ld,
                                                           const long long* __restrict__ rows, long long nrows, int dim,
                                                           float minus_clr, float eps,
                                                           unsigned short* __restrict__ copy16, long long c_ld) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int lane = threadIdx.x & 63;
  const long long row = rows[r];
  float* p = param + row * param_ld;
  float* s = sum + row * sum_ld;
  const float* g = grows + r * g_ld;
  for (int c = lane; c < dim; c += 64) {
    const float ge = g[c];
    const float se = s[c] + ge * ge;
    const float pe = p[c] + (minus_clr * ge) / (__builtin_sqrtf(se) + eps);
    s[c] = se;
    p[c] = pe;
    if (copy16 != nullptr) copy16[row * c_ld + c] = (unsigned short)(bf16_pack(pe, 0.0f) & 0xffffu);
  }
}

int run_adagrad_rows(float* param, long long param_ld, const float* grows, long long g_ld, float* sum, long long sum_ld,
                     const long long* rows, long long nrows, int dim, float minus_clr, float eps,
                     unsigned short* copy16, long long c_ld, hipStream_t st) {
  if (nrows == 0 || dim == 0) return KGE_OK;
  const long long blocks = (nrows + 3) / 4;
  if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adagrad_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, param_ld, grows, g_ld, sum,
                     sum_ld, rows, nrows, dim, minus_clr, eps, copy16, c_ld);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
