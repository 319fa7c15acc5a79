// bwd_gemm.hip -- backward of the ComplEx / DistMult sp_ / _po scores on f32 tables as two plain
// GEMMs + two elementwise kernels.
//
// With q_i = s_i (x) r_i (the query vector, common.hpp build_q4) the forward is S = Q * T^T with
// T the target rows themselves, so (autograd of complex.py:30-39 / distmult.py:15-21, triggered
// at kge/job/train_1vsAll.py:70,81):
//     dT = G^T * Q      [m, d]      (g_tgt)
//     dQ = G * T        [n, d]      then the chain rule of q = s (x) r to the gathered entity row
//                                   (g_a) and relation row (g_p) of query i.
// The two contractions are hand-written on the f32 matrix cores (bwd_gemm32.hip: gemm32_kernel, split-K for the
// long reduction of dQ); here: the query build, the target-row gather for listed subsets and the chain rule.  No
// scratch memory of our own: Q lives in g_p and dQ in g_a until the chain rule overwrites both in place, gathered
// target rows live in g_tgt until dT overwrites them, and when the targets are all entities g_tgt holds the split-K
// partials of the dQ product.  Tolerance-level parity with the reference's autograd (summation order unspecified
// on both sides).  No BLAS library is linked.
#include "common.hpp"


namespace kge {

// bwd_gemm16.hip: the hand-written bf16 contractions
int run_gemm16_dq(int d, long long rows, long long m, const unsigned short* T, long long ldt,
                  const unsigned short* G16, long long mp, float* C, float* scratch, size_t scratch_bytes,
                  hipStream_t st);
bool run_gemm16_dt(int d, long long rows, long long m, const unsigned short* Q16, const unsigned short* G16,
                   long long mp, float* dT, hipStream_t st);
// bwd_gemm32.hip: C[M, N] = A * B on the f32 matrix cores (float32 operands, or bf16 widened)
bool run_gemm32(bool a_kcont, int in16, long long M, long long N, long long K, const void* A, long long lda,
                const void* B, long long ldb, float* C, long long ldc, float* scratch, size_t scratch_bytes,
                hipStream_t st);

// Q[i, :] = q(a_i, r_i), f32, ld = d.  One thread per (row, coordinate of the first half).
template <int SCORER>
__global__ __launch_bounds__(256) void bwdg_build_q_kernel(Operand A, Operand R, int dir, int d,
                                                           long long n, float* __restrict__ Q) {
  const int h = SCORER == KGE_COMPLEX ? d / 2 : d;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = t / h;
  const int c = (int)(t % h);
  if (i >= n) return;
  const float* a = (const float*)A.base + index_at(A.idx, i) * A.ld;
  const float* r = (const float*)R.base + index_at(R.idx, i) * R.ld;
  float* q = Q + i * d;
  if (SCORER == KGE_DISTMULT) {
    q[c] = a[c] * r[c];
  } else if (dir == KGE_SP_) {
    q[c] = a[c] * r[c] - a[h + c] * r[h + c];
    q[h + c] = a[h + c] * r[c] + a[c] * r[h + c];
  } else {
    q[c] = r[c] * a[c] + r[h + c] * a[h + c];
    q[h + c] = r[c] * a[h + c] - r[h + c] * a[c];
  }
}

// in: g_a = dQ.  out: g_a = d(score)/d(a row), g_p = d(score)/d(r row).  In place: a thread reads
// and writes only its own coordinate (pair).
template <int SCORER>
__global__ __launch_bounds__(256) void bwdg_chain_kernel(Operand A, Operand R, int dir, int d,
                                                         long long n, float* __restrict__ g_a,
                                                         float* __restrict__ g_p) {
  const int h = SCORER == KGE_COMPLEX ? d / 2 : d;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = t / h;
  const int c = (int)(t % h);
  if (i >= n) return;
  const float* a = (const float*)A.base + index_at(A.idx, i) * A.ld;
  const float* r = (const float*)R.base + index_at(R.idx, i) * R.ld;
  float* ga = g_a + i * d;
  float* gp = g_p + i * d;
  if (SCORER == KGE_DISTMULT) {
    const float dq = ga[c];
    ga[c] = dq * r[c];
    gp[c] = dq * a[c];
    return;
  }
  const float dre = ga[c], dim_ = ga[h + c];
  const float are = a[c], aim = a[h + c], rre = r[c], rim = r[h + c];
  if (dir == KGE_SP_) {  // q_re = a_re r_re - a_im r_im, q_im = a_im r_re + a_re r_im
    ga[c] = dre * rre + dim_ * rim;
    ga[h + c] = dim_ * rre - dre * rim;
    gp[c] = dre * are + dim_ * aim;
    gp[h + c] = dim_ * are - dre * aim;
  } else {  // q_re = r_re a_re + r_im a_im, q_im = r_re a_im - r_im a_re
    ga[c] = dre * rre - dim_ * rim;
    ga[h + c] = dre * rim + dim_ * rre;
    gp[c] = dre * are + dim_ * aim;
    gp[h + c] = dre * aim - dim_ * are;
  }
}

// out[j, :] = table row TG.idx[j] (listed target subset), ld = d
__global__ __launch_bounds__(256) void bwdg_gather_kernel(Operand TG, int d, long long m,
                                                          float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long j = t / d;
  const int c = (int)(t % d);
  if (j >= m) return;
  out[j * d + c] = ((const float*)TG.base + index_at(TG.idx, j) * TG.ld)[c];
}

// `zero` (may be NULL): zero_cnt floats cleared on the way -- the relation-gradient accumulator the chain launch behind
// the products adds into, when no query-build launch of the backward is there to do it (KGE_FLAG_CE_KEEP_QUERIES)
__global__ __launch_bounds__(256) void bwdg_reduce_kernel(const float* __restrict__ part, long long cnt, int P,
                                                          float* __restrict__ out, float* __restrict__ zero,
                                                          long long zero_cnt) {
  if (zero != nullptr) {
    const long long nthreads = (long long)gridDim.x * 256;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < zero_cnt; k += nthreads) zero[k] = 0.0f;
  }
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= cnt) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(part + i);
  for (int p = 1; p < P; ++p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + (long long)p * cnt + i);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  *reinterpret_cast<f32x4*>(out + i) = acc;
}

template <int SCORER>
static int bwdg_run(int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                    long long n, long long m, const float* gout, long long ldg, float* g_a,
                    float* g_p, float* g_tgt, hipStream_t st) {
  const int half = SCORER == KGE_COMPLEX ? d / 2 : d;
  const unsigned qblocks = (unsigned)((n * half + 255) / 256);
  // target rows as a dense [m, d] matrix: the table itself, or gathered into g_tgt for now
  const float* T = (const float*)TG.base;
  long long ldt = TG.ld;
  if (TG.idx.ptr != nullptr) {
    hipLaunchKernelGGL(bwdg_gather_kernel, dim3((unsigned)((m * d + 255) / 256)), dim3(256), 0, st, TG, d, m,
                       g_tgt);
    T = g_tgt;
    ldt = d;
  }
  // dQ = G * T  ([n, d]; K = m, split over the K ranges): all entities -> g_tgt (written by the second product
  // only) holds the partials
  float* ws = TG.idx.ptr == nullptr ? g_tgt : nullptr;
  const size_t ws_bytes = TG.idx.ptr == nullptr ? (size_t)m * d * sizeof(float) : 0;
  if (!run_gemm32(true, 0, n, d, m, gout, ldg, T, ldt, g_a, d, ws, ws_bytes, st)) return KGE_ERR_UNSUPPORTED;
  // Q -> g_p, then dT = G^T * Q  ([m, d]; K = n)
  hipLaunchKernelGGL((bwdg_build_q_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, g_p);
  if (!run_gemm32(false, 0, m, d, n, gout, ldg, g_p, d, g_tgt, d, nullptr, 0, st)) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((bwdg_chain_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, g_a, g_p);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- bf16 tables (mixed-precision training): the same two products on the bf16 matrix cores.
// G is rounded to bf16 (what autocast does to the incoming gradient of a bf16 matmul), Q is the
// bf16 query matrix of the forward; f32 accumulation, f32 gradients.  Scratch (caller-provided):
// G16 [n, m] + Q16 [n, d] bf16.

__global__ __launch_bounds__(256) void bwdg_cast_kernel(const float* __restrict__ g, long long ldg, long long n,
                                                        long long m, unsigned short* __restrict__ out) {
  // two columns per thread; row pitch of `out` = m rounded up to 8 elements (16-byte rows)
  const long long mp2 = ((m + 7) & ~7LL) >> 1;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = t / mp2;
  const long long j = (t % mp2) * 2;
  if (i >= n) return;
  const float a = j < m ? g[i * ldg + j] : 0.0f, b = (j + 1 < m) ? g[i * ldg + j + 1] : 0.0f;
  reinterpret_cast<unsigned int*>(out)[i * mp2 + (j >> 1)] = bf16_pack(a, b);
}

__device__ __forceinline__ float bwdg_w(const unsigned short* p, long long k) {
  return __uint_as_float((unsigned int)p[k] << 16);
}

template <int SCORER>
__global__ __launch_bounds__(256) void bwdg_build_q16_kernel(Operand A, Operand R, int dir, int d, long long n,
                                                             unsigned short* __restrict__ Q, Operand A2, Operand R2,
                                                             long long n2, float* __restrict__ zero,
                                                             long long zero_cnt) {
  // `zero` (may be NULL): zero_cnt floats cleared on the way -- the relation-gradient accumulator the chain launch
  // behind the two products adds into (kge_ce_sp_po_bwd_accum); every thread of the grid takes its share
  if (zero != nullptr) {
    const long long nthreads = (long long)gridDim.x * gridDim.y * 256;
    for (long long k = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; k < zero_cnt; k += nthreads)
      zero[k] = 0.0f;
  }
  if (blockIdx.y == 1) {  // two-sided launch: the n2 _po queries (entity rows A2, relation rows R2), rows [n, n + n2)
    A = A2;
    R = R2;
    dir = KGE_PO_;
    Q += n * d;
    n = n2;
  }
  const int h = SCORER == KGE_COMPLEX ? d / 2 : d;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = t / h;
  const int c = (int)(t % h);
  if (i >= n) return;
  const unsigned short* a = (const unsigned short*)A.base + index_at(A.idx, i) * A.ld;
  const unsigned short* r = (const unsigned short*)R.base + index_at(R.idx, i) * R.ld;
  unsigned short* q = Q + i * d;
  float q0, q1 = 0.0f;
  if (SCORER == KGE_DISTMULT) {
    q0 = bwdg_w(a, c) * bwdg_w(r, c);
  } else if (dir == KGE_SP_) {
    q0 = bwdg_w(a, c) * bwdg_w(r, c) - bwdg_w(a, h + c) * bwdg_w(r, h + c);
    q1 = bwdg_w(a, h + c) * bwdg_w(r, c) + bwdg_w(a, c) * bwdg_w(r, h + c);
  } else {
    q0 = bwdg_w(r, c) * bwdg_w(a, c) + bwdg_w(r, h + c) * bwdg_w(a, h + c);
    q1 = bwdg_w(r, c) * bwdg_w(a, h + c) - bwdg_w(r, h + c) * bwdg_w(a, c);
  }
  const unsigned int pk = bf16_pack(q0, q1);  // RNE, the forward's rounding of q
  q[c] = (unsigned short)(pk & 0xffffu);
  if (SCORER == KGE_COMPLEX) q[h + c] = (unsigned short)(pk >> 16);
}

// acc_ent == NULL: row gradients written to g_a (in place over dQ) / g_p.  Otherwise they are added
// straight into table gradients (float atomics; the scatter-add of the gathered rows, which the
// reference gets from autograd's index backward): acc_ent[A.idx[i], :] += ..., acc_rel[R.idx[i], :] += ...
template <int SCORER>
__global__ __launch_bounds__(256) void bwdg_chain16_kernel(Operand A, Operand R, int dir, int d, long long n,
                                                           float* __restrict__ g_a, float* __restrict__ g_p,
                                                           Operand A2, Operand R2, long long n2,
                                                           float* __restrict__ acc_ent, long long acc_ent_ld,
                                                           float* __restrict__ acc_rel, long long acc_rel_ld) {
  if (blockIdx.y == 1) {  // two-sided launch: the n2 _po queries, rows [n, n + n2)
    A = A2;
    R = R2;
    dir = KGE_PO_;
    g_a += n * d;
    if (g_p != nullptr) g_p += n * d;
    n = n2;
  }
  const int h = SCORER == KGE_COMPLEX ? d / 2 : d;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = t / h;
  const int c = (int)(t % h);
  if (i >= n) return;
  const long long ia = index_at(A.idx, i), ir = index_at(R.idx, i);
  const unsigned short* a = (const unsigned short*)A.base + ia * A.ld;
  const unsigned short* r = (const unsigned short*)R.base + ir * R.ld;
  float* ga = g_a + i * d;
  float va0, va1 = 0.0f, vp0, vp1 = 0.0f;
  if (SCORER == KGE_DISTMULT) {
    const float dq = ga[c];
    va0 = dq * bwdg_w(r, c);
    vp0 = dq * bwdg_w(a, c);
  } else {
    const float dre = ga[c], dim_ = ga[h + c];
    const float are = bwdg_w(a, c), aim = bwdg_w(a, h + c), rre = bwdg_w(r, c), rim = bwdg_w(r, h + c);
    if (dir == KGE_SP_) {
      va0 = dre * rre + dim_ * rim;
      va1 = dim_ * rre - dre * rim;
      vp0 = dre * are + dim_ * aim;
      vp1 = dim_ * are - dre * aim;
    } else {
      va0 = dre * rre - dim_ * rim;
      va1 = dre * rim + dim_ * rre;
      vp0 = dre * are + dim_ * aim;
      vp1 = dre * aim - dim_ * are;
    }
  }
  if (acc_ent != nullptr) {
    unsafeAtomicAdd(acc_ent + ia * acc_ent_ld + c, va0);
    unsafeAtomicAdd(acc_rel + ir * acc_rel_ld + c, vp0);
    if (SCORER == KGE_COMPLEX) {
      unsafeAtomicAdd(acc_ent + ia * acc_ent_ld + h + c, va1);
      unsafeAtomicAdd(acc_rel + ir * acc_rel_ld + h + c, vp1);
    }
    return;
  }
  float* gp = g_p + i * d;
  ga[c] = va0;
  gp[c] = vp0;
  if (SCORER == KGE_COMPLEX) {
    ga[h + c] = va1;
    gp[h + c] = vp1;
  }
}

__global__ __launch_bounds__(256) void bwdg_gather16_kernel(Operand TG, int d, long long m,
                                                            unsigned short* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long j = t / d;
  const int c = (int)(t % d);
  if (j >= m) return;
  out[j * d + c] = ((const unsigned short*)TG.base + index_at(TG.idx, j) * TG.ld)[c];
}

long long pairs_bwd_workspace_bytes(int dtype, int scorer, int d, long long n, long long m) {
  if (dtype != KGE_BF16 || (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT)) return 0;
  const long long mp = (m + 7) & ~7LL;
  return ((n * mp * 2 + 255) & ~255LL) + ((n * (long long)d * 2 + 255) & ~255LL);
}

// dQ = G16 * T and dT = G16^T * Q16 on the bf16 matrix cores: the hand-written kernel of bwd_gemm16.hip,
// the f32 matrix cores on the widened operands (gemm32_kernel) where that declines (d not a multiple of 256,
// KGE_BWD_GEMM_LIB=1).  `lws`: scratch for the
// split-K partials of dQ (g_tgt before dT overwrites it) or NULL.
static bool bwdg_dq16(int d, long long rows, long long m, const unsigned short* T, long long ldt,
                      const unsigned short* G16, long long mp, float* g_a, float* lws, size_t lws_bytes,
                      hipStream_t st, float* zero = nullptr, long long zero_cnt = 0) {
  const int sp = run_gemm16_dq(d, rows, m, T, ldt, G16, mp, g_a, lws, lws_bytes, st);
  if (sp > 1) {
    const long long cnt = rows * d;  // d % 256 == 0: cnt % 4 == 0
    hipLaunchKernelGGL(bwdg_reduce_kernel, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, st, lws, cnt, sp,
                       g_a, zero, zero_cnt);
  } else if (zero != nullptr && zero_cnt > 0) {  // (no split-K sum to ride in: a fill of its own)
    if (!fill_words_async(zero, 0, (size_t)zero_cnt * sizeof(float), st)) return false;
  }
  if (sp >= 1) return true;
  return run_gemm32(true, 1, rows, d, m, G16, mp, T, ldt, g_a, d, lws, lws_bytes, st);
}

static bool bwdg_dt16(int d, long long rows, long long m, const unsigned short* Q16, const unsigned short* G16,
                      long long mp, float* g_tgt, hipStream_t st) {
  if (run_gemm16_dt(d, rows, m, Q16, G16, mp, g_tgt, st)) return true;
  return run_gemm32(false, 1, m, d, rows, G16, mp, Q16, d, g_tgt, d, nullptr, 0, st);
}

// tests / tools: one of the two products on its own (which = 0: dQ [rows, d] from T [m, d]; 1: dT [m, d] from
// Q16 [rows, d]); lib != 0 forces gemm32_kernel on the widened operands (the cross-check)
int run_debug_gemm16(int which, int lib, int d, long long rows, long long m, const unsigned short* X, long long ldx,
                     const unsigned short* G16, long long mp, float* out, float* scratch, long long scratch_bytes,
                     hipStream_t st) {
  bool ok;
  if (which == 0)
    ok = lib ? run_gemm32(true, 1, rows, d, m, G16, mp, X, ldx, out, d, scratch, (size_t)scratch_bytes, st)
             : bwdg_dq16(d, rows, m, X, ldx, G16, mp, out, scratch, (size_t)scratch_bytes, st);
  else
    ok = lib ? run_gemm32(false, 1, m, d, rows, G16, mp, X, d, out, d, nullptr, 0, st)
             : bwdg_dt16(d, rows, m, X, G16, mp, out, st);
  if (!ok) return KGE_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Both gradient products from G16 = d loss / d score in bf16 ([n, mp], row pitch mp elements):
// the tail of the mixed-precision backward, shared with the fused 1vsAll loss (ce_loss.hip), whose
// scoring kernel writes G16 itself.
template <int SCORER>
static int bwdg_products16(int dir, const Operand& A, const Operand& R, const Operand& TG, int d, long long n,
                           long long m, const unsigned short* G16, long long mp, unsigned short* Q16,
                           float* g_a, float* g_p, float* g_tgt, hipStream_t st, bool q16_ready = false) {
  const int half = SCORER == KGE_COMPLEX ? d / 2 : d;
  const unsigned qblocks = (unsigned)((n * half + 255) / 256);
  if (!q16_ready)
    hipLaunchKernelGGL((bwdg_build_q16_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, Q16, A, R, n,
                       (float*)nullptr, 0LL);
  const unsigned short* T = (const unsigned short*)TG.base;
  long long ldt = TG.ld;
  if (TG.idx.ptr != nullptr) {  // gathered target rows live in g_tgt until dT overwrites it
    hipLaunchKernelGGL(bwdg_gather16_kernel, dim3((unsigned)((m * d + 255) / 256)), dim3(256), 0, st, TG, d, m,
                       (unsigned short*)g_tgt);
    T = (const unsigned short*)g_tgt;
    ldt = d;
  }
  // dQ^T[d, n] = T^T[d, m] * G^T[m, n];  dT^T[d, m] = Q^T[d, n] * G[n, m]  (column-major views)
  // all entities: g_tgt (written by the second product only) is the library's split-K workspace
  void* lws = TG.idx.ptr == nullptr ? (void*)g_tgt : nullptr;
  const size_t lws_bytes = TG.idx.ptr == nullptr ? (size_t)m * d * sizeof(float) : 0;
  if (!bwdg_dq16(d, n, m, T, ldt, G16, mp, g_a, (float*)lws, lws_bytes, st)) return KGE_ERR_UNSUPPORTED;
  if (!bwdg_dt16(d, n, m, Q16, G16, mp, g_tgt, st)) return KGE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((bwdg_chain16_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, g_a, g_p, A, R, n,
                     (float*)nullptr, 0LL, (float*)nullptr, 0LL);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Two-sided variant (kge_ce_sp_po_bwd, kge_kl2_bwd_accum): rows [0, n) of G16 / Q16 / g_a / g_p belong to the n sp_
// queries (entity rows A1, relation rows R), rows [n, n + n2) to the n2 _po queries (A2, R2); both products run ONCE
// over the n + n2 rows (dT sums the two sides inside the product instead of in a separate accumulation pass).
template <int SCORER>
static int bwdg_products16_two(const Operand& A1, const Operand& A2, const Operand& R, const Operand& R2,
                               const Operand& TG, int d, long long n, long long n2, long long m,
                               const unsigned short* G16, long long mp, unsigned short* Q16, float* g_a, float* g_p,
                               float* g_tgt, float* acc_rel, long long acc_rel_rows, long long acc_rel_ld,
                               float* dq_scratch, long long dq_scratch_bytes, hipStream_t st, bool q16_ready,
                               bool clear_acc_rel = false) {
  // clear_acc_rel (with q16_ready): Q16 was left by the FORWARD's build launch (KGE_FLAG_CE_KEEP_QUERIES) and nothing
  // has cleared acc_rel yet -- the split-K sum's launch does it on its way
  if (TG.idx.ptr != nullptr) return KGE_ERR_UNSUPPORTED;  // all entities only
  const int half = SCORER == KGE_COMPLEX ? d / 2 : d;
  const long long nrows = n + n2, nmax = n > n2 ? n : n2;
  const dim3 qgrid((unsigned)((nmax * half + 255) / 256), 2);  // y = side
  // (q16_ready: Q16 was written and acc_rel cleared by the launch that built the query fragments of the gradient
  // pass -- run_query_build_q16, ce_loss.hip)
  if (!q16_ready)
    hipLaunchKernelGGL((bwdg_build_q16_kernel<SCORER>), qgrid, dim3(256), 0, st, A1, R, KGE_SP_, d, n, Q16, A2, R2, n2,
                       acc_rel, acc_rel != nullptr ? acc_rel_rows * acc_rel_ld : 0LL);
  const unsigned short* T = (const unsigned short*)TG.base;
  // split-K scratch of dQ: the caller's (sized for as many splits as the product wants: 16 at the FB15k-237 shape),
  // else g_tgt, which dT overwrites afterwards (14 fit).  [Summing the partials inside the chain launch instead of
  // in bwdg_reduce_kernel was tried in round 5: 13.3 us scalar / 17+ us with 16-byte loads against 6.1 + 5.8 us.]
  float* lws = dq_scratch != nullptr ? dq_scratch : g_tgt;
  const size_t lws_bytes = dq_scratch != nullptr ? (size_t)dq_scratch_bytes : (size_t)m * d * sizeof(float);
  const bool clr = clear_acc_rel && acc_rel != nullptr;
  // [Tried in round 6 and reverted: dQ + its split-K sum on a second stream beside dT (fork behind the launch that wrote
  // G16, join in front of the chain rule; the events become edges of the captured step).  The replayed step went from
  // 0.1409 to 0.1432-0.1448 ms, the eager step from 0.234 to 0.255+: both products fill the chip (one 144 KB workgroup
  // per CU), so only the 7 us sum could overlap, and the two cross-stream edges of the graph cost more than that.]
  if (!bwdg_dq16(d, nrows, m, T, TG.ld, G16, mp, g_a, lws, lws_bytes, st, clr ? acc_rel : nullptr,
                 clr ? acc_rel_rows * acc_rel_ld : 0LL))
    return KGE_ERR_UNSUPPORTED;
  if (!bwdg_dt16(d, nrows, m, Q16, G16, mp, g_tgt, st)) return KGE_ERR_UNSUPPORTED;
  // acc_rel != NULL: the row gradients go straight into the table gradients -- the entity rows
  // on top of dT in g_tgt [m, d] (all entities: row ids are table rows), the relation rows into acc_rel
  hipLaunchKernelGGL((bwdg_chain16_kernel<SCORER>), qgrid, dim3(256), 0, st, A1, R, KGE_SP_, d, n, g_a, g_p, A2, R2, n2,
                     acc_rel != nullptr ? g_tgt : (float*)nullptr, (long long)d, acc_rel, acc_rel_ld);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_pairs_bwd_products16_two(int scorer, const Operand& A1, const Operand& A2, const Operand& R,
                                 const Operand& R2, const Operand& TG, int d, long long n, long long n2, long long m,
                                 const unsigned short* G16, long long mp, unsigned short* Q16, float* g_a, float* g_p,
                                 float* g_tgt, float* acc_rel, long long acc_rel_rows, long long acc_rel_ld,
                                 float* dq_scratch, long long dq_scratch_bytes, hipStream_t st, bool q16_ready,
                                 bool clear_acc_rel) {
  if (n + n2 == 0 || m == 0) return KGE_OK;  // (ce_loss.hip handles empty batches itself: acc_rel is cleared there)
  if (n + n2 >= (1LL << 31) || m >= (1LL << 31) || mp >= (1LL << 31) || TG.ld >= (1LL << 31))
    return KGE_ERR_UNSUPPORTED;
  int rc = KGE_ERR_UNSUPPORTED;
  if (scorer == KGE_COMPLEX)
    rc = bwdg_products16_two<KGE_COMPLEX>(A1, A2, R, R2, TG, d, n, n2, m, G16, mp, Q16, g_a, g_p, g_tgt, acc_rel,
                                          acc_rel_rows, acc_rel_ld, dq_scratch, dq_scratch_bytes, st, q16_ready, clear_acc_rel);
  else if (scorer == KGE_DISTMULT)
    rc = bwdg_products16_two<KGE_DISTMULT>(A1, A2, R, R2, TG, d, n, n2, m, G16, mp, Q16, g_a, g_p, g_tgt, acc_rel,
                                           acc_rel_rows, acc_rel_ld, dq_scratch, dq_scratch_bytes, st, q16_ready, clear_acc_rel);
  return rc;
}

template <int SCORER>
static int bwdg_run16(int dir, const Operand& A, const Operand& R, const Operand& TG, int d, long long n,
                      long long m, const float* gout, long long ldg, float* g_a, float* g_p, float* g_tgt,
                      void* wsp, long long ws_bytes, hipStream_t st) {
  const long long mp = (m + 7) & ~7LL;  // row pitch of G16 (elements)
  const long long g16_bytes = (n * mp * 2 + 255) & ~255LL;
  if (wsp == nullptr || ((uintptr_t)wsp & 255) || ws_bytes < g16_bytes + n * (long long)d * 2) return KGE_ERR_WORKSPACE;
  unsigned short* G16 = (unsigned short*)wsp;
  unsigned short* Q16 = (unsigned short*)((char*)wsp + g16_bytes);
  hipLaunchKernelGGL(bwdg_cast_kernel, dim3((unsigned)((n * (mp / 2) + 255) / 256)), dim3(256), 0, st, gout, ldg, n,
                     m, G16);
  return bwdg_products16<SCORER>(dir, A, R, TG, d, n, m, G16, mp, Q16, g_a, g_p, g_tgt, st);
}

// ce_loss.hip: G16 (pitch mp) was written by the scoring kernel; Q16 = n * d * 2 bytes of scratch
int run_pairs_bwd_products16(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                             long long n, long long m, const unsigned short* G16, long long mp,
                             unsigned short* Q16, float* g_a, float* g_p, float* g_tgt, hipStream_t st, bool q16_ready) {
  if (n == 0 || m == 0) return KGE_OK;
  if (n >= (1LL << 31) || m >= (1LL << 31) || mp >= (1LL << 31) || TG.ld >= (1LL << 31)) return KGE_ERR_UNSUPPORTED;
  int rc = KGE_ERR_UNSUPPORTED;
  if (scorer == KGE_COMPLEX)
    rc = bwdg_products16<KGE_COMPLEX>(dir, A, R, TG, d, n, m, G16, mp, Q16, g_a, g_p, g_tgt, st, q16_ready);
  else if (scorer == KGE_DISTMULT)
    rc = bwdg_products16<KGE_DISTMULT>(dir, A, R, TG, d, n, m, G16, mp, Q16, g_a, g_p, g_tgt, st, q16_ready);
  return rc;
}

int run_pairs_bwd_gemm16(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                         int dr, long long n, long long m, const float* gout, long long ldg, float* g_a,
                         float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st) {
  if (n == 0 || m == 0) return KGE_OK;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return KGE_ERR_UNSUPPORTED;
  if (dr != d || !g_a || !g_p || !g_tgt || (d % 2)) return KGE_ERR_UNSUPPORTED;
  if (n >= (1LL << 31) || m >= (1LL << 31) || TG.ld >= (1LL << 31)) return KGE_ERR_UNSUPPORTED;
  const int rc = scorer == KGE_COMPLEX
                     ? bwdg_run16<KGE_COMPLEX>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, ws, ws_bytes, st)
                     : bwdg_run16<KGE_DISTMULT>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, ws, ws_bytes, st);
  return rc;
}

// KGE_ERR_UNSUPPORTED: the caller uses the self-contained kernels of bwd.hip
int run_pairs_bwd_gemm(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG,
                       int d, int dr, long long n, long long m, const float* gout, long long ldg,
                       float* g_a, float* g_p, float* g_tgt, hipStream_t st) {
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return KGE_ERR_UNSUPPORTED;
  if (dr != d || !g_a || !g_p || !g_tgt) return KGE_ERR_UNSUPPORTED;
  if (n >= (1LL << 31) || m >= (1LL << 31) || ldg >= (1LL << 31) || TG.ld >= (1LL << 31))
    return KGE_ERR_UNSUPPORTED;
  // (Until round 6 this declined under stream capture -- a leftover of the rounds that called a BLAS library, which
  // may allocate.  Nothing below allocates; with the check a CAPTURED float32 step fell through to the VALU kernels of
  // bwd.hip and replayed slower than the eager step ran: 1.22 against 0.86 ms, profiles/r5_train_step_kernels.txt.)
  if (scorer == KGE_COMPLEX) return bwdg_run<KGE_COMPLEX>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, st);
  return bwdg_run<KGE_DISTMULT>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, st);
}

}  // namespace kge
