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

// ---- Adagrad with the embedder's penalty term folded in (round 6) -------------------------------------------------
// LookupEmbedder.penalty (kge/model/embedder/lookup_embedder.py:122-177), unweighted: weight / p * sum |x|^p over the
// whole table (lp), or over |z| = sqrt(re^2 + im^2 + 1e-14) of the complex coordinates (n3 with space complex, p = 3;
// im = col + row_dim / 2).  TrainingJob.run_epoch back-propagates the term between the batch and optimizer.step()
// (kge/job/train.py:417-436): its gradient, weight * sign(x) |x|^(p-1) (n3: weight * |z| * re, weight * |z| * im), is a
// function of the element (pair) alone, so it is added to the gradient in the registers of the Adagrad pass -- no
// autograd graph over [E, d], no extra sweep -- and the term's VALUE (of the pre-step parameters, what the reference's
// trace shows) is summed in the same pass: per-workgroup partial in double, one atomic add per workgroup.
struct AdagradPen {
  int kind[KGE_ADAGRAD_MAX_SEGS], p[KGE_ADAGRAD_MAX_SEGS];
  float weight[KGE_ADAGRAD_MAX_SEGS];
  long long half_dim[KGE_ADAGRAD_MAX_SEGS];  // kind 2: row_dim / 2
  double* value[KGE_ADAGRAD_MAX_SEGS];
};

__device__ __forceinline__ float pen_grad_lp(float x, int p, float w, float& v) {
  const float a = __builtin_fabsf(x);
  if (p == 2) { v += x * x; return w * x; }
  if (p == 1) { v += a; return x > 0.0f ? w : (x < 0.0f ? -w : 0.0f); }
  v += a * a * a;  // p == 3
  return w * (x * a);
}

__global__ __launch_bounds__(256) void adagrad_multi_pen_kernel(AdagradSegs a, AdagradPen q) {
  __shared__ double part[4];
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
  const int kind = q.kind[k], pp = q.p[k];
  const float w = q.weight[k];
  float v = 0.0f;  // this thread's part of sum |x|^p
  auto update = [&](long long i, f32x4 p, f32x4 pg) {  // pg: the penalty's gradient of these four elements
    f32x4 g = *reinterpret_cast<const f32x4*>(grad + i);
    f32x4 s = *reinterpret_cast<const f32x4*>(sum + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e] + pg[e];
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
  };
  const long long t = (long long)(blockIdx.x - a.first_block[k]) * 256 + threadIdx.x;
  if (kind == 2) {  // one (re, im) pair of quads per thread; count % row_dim == 0, half_dim % 4 == 0 (checked by the caller)
    const long long h = q.half_dim[k];
    const long long e0 = t * 4;  // index among the count / 2 real parts
    if (e0 < count / 2) {
      const long long i_re = (e0 / h) * (2 * h) + (e0 % h), i_im = i_re + h;
      const f32x4 re = *reinterpret_cast<const f32x4*>(param + i_re);
      const f32x4 im = *reinterpret_cast<const f32x4*>(param + i_im);
      f32x4 g_re, g_im;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = __builtin_sqrtf(re[e] * re[e] + im[e] * im[e] + 1e-14f);
        v += z * z * z;
        g_re[e] = w * (z * re[e]);
        g_im[e] = w * (z * im[e]);
      }
      update(i_re, re, g_re);
      update(i_im, im, g_im);
    }
  } else {
    const long long i = t * 4;
    if (i + 4 <= count) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(param + i);
      f32x4 pg = {0.0f, 0.0f, 0.0f, 0.0f};
      if (kind == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pg[e] = pen_grad_lp(p[e], pp, w, v);
      }
      update(i, p, pg);
    } else {
      for (long long j = i; j < count; ++j) {
        const float x = param[j];
        float ge = grad[j];
        if (kind == 1) ge = ge + pen_grad_lp(x, pp, w, v);
        if (weight_decay != 0.0f) ge = ge + weight_decay * x;
        const float s = sum[j] + ge * ge;
        const float p = x + (minus_clr * ge) / (__builtin_sqrtf(s) + eps);
        sum[j] = s;
        param[j] = p;
        if (copy16 != nullptr) copy16[j] = (unsigned short)(bf16_pack(p, 0.0f) & 0xffffu);
      }
    }
  }
  if (kind != 0) {  // (uniform per workgroup)
    double d = (double)v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(q.value[k], part[0] + part[1] + part[2] + part[3]);
  }
}

int run_adagrad_multi_pen(const kge_adagrad_seg* segs, const kge_penalty_seg* pens, int num, hipStream_t st) {
  AdagradSegs a{};
  AdagradPen q{};
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
    q.kind[k] = pens[j].kind;
    q.p[k] = pens[j].p;
    q.weight[k] = pens[j].weight;
    q.half_dim[k] = pens[j].row_dim / 2;
    q.value[k] = pens[j].value;
    a.first_block[k] = (unsigned int)blocks;
    const long long quads = pens[j].kind == 2 ? segs[j].count / 2 : segs[j].count;
    blocks += (quads + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
    ++k;
  }
  if (k == 0) return KGE_OK;
  a.first_block[k] = (unsigned int)blocks;
  a.num = k;
  hipLaunchKernelGGL(adagrad_multi_pen_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, q);
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
