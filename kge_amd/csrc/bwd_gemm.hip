// bwd_gemm.hip -- backward of the ComplEx / DistMult sp_ / _po scores on f32 tables as two plain
// GEMMs + two elementwise kernels.
//
// With q_i = s_i (x) r_i (the query vector, common.hpp build_q4) the forward is S = Q * T^T with
// T the target rows themselves, so (autograd of complex.py:30-39 / distmult.py:15-21, triggered
// at kge/job/train_1vsAll.py:70,81):
//     dT = G^T * Q      [m, d]      (g_tgt)
//     dQ = G * T        [n, d]      then the chain rule of q = s (x) r to the gathered entity row
//                                   (g_a) and relation row (g_p) of query i.
// The two GEMMs are plain library GEMMs (hipBLASLt, f32 MFMA inside; rocBLAS sgemm if hipBLASLt
// has no plan -- its 128x128 macro-tile without split-K takes 0.8 ms on the 512 x 512 x 14,541
// dQ product that hipBLASLt does in 0.1 ms); what is hand-written here is the query build, the
// target-row gather for listed subsets and the chain rule.  No scratch memory of our own: Q
// lives in g_p and dQ in g_a until the chain rule overwrites both in place, gathered target
// rows live in g_tgt until dT overwrites them, and when the targets are all entities g_tgt
// doubles as the library's split-K workspace for the dQ product.  Tolerance-level parity with
// the reference's autograd (summation order unspecified on both sides).
#include "common.hpp"

#include <mutex>
#include <vector>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>

namespace kge {

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

static rocblas_handle bwdg_handle() {
  static std::mutex mu;
  static rocblas_handle handles[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!handles[dev]) {
    rocblas_handle h = nullptr;
    if (rocblas_create_handle(&h) != rocblas_status_success) return nullptr;
    rocblas_set_pointer_mode(h, rocblas_pointer_mode_host);
    handles[dev] = h;
  }
  return handles[dev];
}

// ---- column-major f32 GEMM C[m,n] = op(A) * op(B) through hipBLASLt, plans cached per shape
struct LtPlan {
  int ta, tb;
  long long m, n, k, lda, ldb, ldc;
  size_t ws_avail;
  bool ok;
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t la, lb, lc;
  hipblasLtMatmulHeuristicResult_t res;
};

static bool lt_gemm(int ta, int tb, long long m, long long n, long long k, const float* A, long long lda,
                    const float* B, long long ldb, float* C, long long ldc, void* ws, size_t ws_bytes,
                    hipStream_t st) {
  static std::mutex mu;
  static hipblasLtHandle_t handles[64] = {};
  static std::vector<LtPlan> plans[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (!handles[dev] && hipblasLtCreate(&handles[dev]) != HIPBLAS_STATUS_SUCCESS) {
    handles[dev] = nullptr;
    return false;
  }
  if (ws_bytes > (64u << 20)) ws_bytes = 64u << 20;
  ws_bytes &= ~(size_t)255;
  LtPlan* pl = nullptr;
  for (auto& q : plans[dev])
    if (q.ta == ta && q.tb == tb && q.m == m && q.n == n && q.k == k && q.lda == lda && q.ldb == ldb &&
        q.ldc == ldc && q.ws_avail == ws_bytes) {
      pl = &q;
      break;
    }
  if (!pl) {
    LtPlan q{ta, tb, m, n, k, lda, ldb, ldc, ws_bytes, false, nullptr, nullptr, nullptr, nullptr, {}};
    const hipblasOperation_t opa = ta ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = tb ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    hipblasLtMatmulPreference_t pref = nullptr;
    int found = 0;
    bool good = hipblasLtMatmulDescCreate(&q.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulDescSetAttribute(q.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)) ==
                       HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulDescSetAttribute(q.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)) ==
                       HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatrixLayoutCreate(&q.la, HIP_R_32F, ta ? k : m, ta ? m : k, lda) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatrixLayoutCreate(&q.lb, HIP_R_32F, tb ? n : k, tb ? k : n, ldb) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatrixLayoutCreate(&q.lc, HIP_R_32F, m, n, ldc) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes,
                                                         sizeof(ws_bytes)) == HIPBLAS_STATUS_SUCCESS;
    good = good && hipblasLtMatmulAlgoGetHeuristic(handles[dev], q.desc, q.la, q.lb, q.lc, q.lc, pref, 1, &q.res,
                                                   &found) == HIPBLAS_STATUS_SUCCESS;
    if (pref) hipblasLtMatmulPreferenceDestroy(pref);
    q.ok = good && found > 0 && q.res.workspaceSize <= ws_bytes;
    plans[dev].push_back(q);
    pl = &plans[dev].back();
  }
  if (!pl->ok) return false;
  const float one = 1.0f, zero = 0.0f;
  return hipblasLtMatmul(handles[dev], pl->desc, &one, A, pl->la, B, pl->lb, &zero, C, pl->lc, C, pl->lc,
                         &pl->res.algo, ws, pl->res.workspaceSize, st) == HIPBLAS_STATUS_SUCCESS;
}

template <int SCORER>
static int bwdg_run(int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                    long long n, long long m, const float* gout, long long ldg, float* g_a,
                    float* g_p, float* g_tgt, hipStream_t st) {
  const int half = SCORER == KGE_COMPLEX ? d / 2 : d;
  const unsigned qblocks = (unsigned)((n * half + 255) / 256);
  const float one = 1.0f, zero = 0.0f;
  // target rows as a dense [m, d] matrix: the table itself, or gathered into g_tgt for now
  const float* T = (const float*)TG.base;
  long long ldt = TG.ld;
  if (TG.idx.ptr != nullptr) {
    hipLaunchKernelGGL(bwdg_gather_kernel, dim3((unsigned)((m * d + 255) / 256)), dim3(256), 0, st, TG, d, m,
                       g_tgt);
    T = g_tgt;
    ldt = d;
  }
  rocblas_handle h = nullptr;  // fallback only
  auto fallback = [&]() {
    if (h) return true;
    h = bwdg_handle();
    return h != nullptr && rocblas_set_stream(h, st) == rocblas_status_success;
  };
  // dQ = G * T  (row-major [n, d]) == column-major dQ^T[d, n] = T^T[d, m] * G^T[m, n]; all
  // entities: g_tgt (written by the second product only) is the library's workspace here
  void* ws = TG.idx.ptr == nullptr ? (void*)g_tgt : nullptr;
  const size_t ws_bytes = TG.idx.ptr == nullptr ? (size_t)m * d * sizeof(float) : 0;
  if (!lt_gemm(0, 0, d, n, m, T, ldt, gout, ldg, g_a, d, ws, ws_bytes, st)) {
    if (!fallback()) return KGE_ERR_UNSUPPORTED;
    if (rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, d, (int)n, (int)m, &one, T, (int)ldt,
                      gout, (int)ldg, &zero, g_a, d) != rocblas_status_success)
      return KGE_ERR_LAUNCH;
  }
  // Q -> g_p, then dT = G^T * Q (row-major [m, d]) == column-major dT^T[d, m] = Q^T[d, n] * G[n, m]
  hipLaunchKernelGGL((bwdg_build_q_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, g_p);
  if (!lt_gemm(0, 1, d, m, n, g_p, d, gout, ldg, g_tgt, d, nullptr, 0, st)) {
    if (!fallback()) return KGE_ERR_UNSUPPORTED;
    if (rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, d, (int)m, (int)n, &one, g_p, d,
                      gout, (int)ldg, &zero, g_tgt, d) != rocblas_status_success)
      return KGE_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((bwdg_chain_kernel<SCORER>), dim3(qblocks), dim3(256), 0, st, A, R, dir, d, n, g_a, g_p);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// KGE_ERR_UNSUPPORTED: the caller uses the self-contained kernels of bwd.hip
int run_pairs_bwd_gemm(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG,
                       int d, int dr, long long n, long long m, const float* gout, long long ldg,
                       float* g_a, float* g_p, float* g_tgt, hipStream_t st) {
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return KGE_ERR_UNSUPPORTED;
  if (dr != d || !g_a || !g_p || !g_tgt) return KGE_ERR_UNSUPPORTED;
  if (n >= (1LL << 31) || m >= (1LL << 31) || ldg >= (1LL << 31) || TG.ld >= (1LL << 31))
    return KGE_ERR_UNSUPPORTED;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;  // rocBLAS may allocate: not under capture
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
    return KGE_ERR_UNSUPPORTED;
  if (scorer == KGE_COMPLEX) return bwdg_run<KGE_COMPLEX>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, st);
  return bwdg_run<KGE_DISTMULT>(dir, A, R, TG, d, n, m, gout, ldg, g_a, g_p, g_tgt, st);
}

}  // namespace kge
