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

int run_adagrad(float* param, const float* grad, float* sum, long long count, float minus_clr, float weight_decay,
                float eps, unsigned short* copy16, hipStream_t st) {
  if (count == 0) return KGE_OK;
  const long long blocks = (count + 1023) / 1024;
  if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adagrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, sum, count, minus_clr,
                     weight_decay, eps, copy16);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
