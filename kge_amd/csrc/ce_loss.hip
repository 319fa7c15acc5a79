// ce_loss.hip -- 1vsAll training loss fused with the sp_/_po scoring.
//
// Reference: TrainingJob1vsAll scores a batch against all entities and feeds the [n, E] matrix to
// KLDivWithSoftmaxKgeLoss, which for index labels is CrossEntropyLoss(reduction="sum")
// (kge/job/train_1vsAll.py:64-65, 75-76; kge/util/loss.py:192-207).  Unfused, the score matrix is
// written once (29.8 MB at the FB15k-237 shape, the dominant HBM term of the scoring kernel), read
// twice by softmax + nll forward, and a same-size gradient matrix is written and read again by the
// backward.  Here the scoring kernel keeps the tiles in registers:
//
//   kge_ce_fwd  pairs_bf16_v3_kernel<.., V3_LSE>: per row and column group (max, sum exp) by
//               online softmax over the tiles + the label's score; ce_combine_kernel merges the
//               column groups:  lse_i = logsumexp_j score(i, j),  loss_i = lse_i - score(i, label_i).
//               HBM traffic: the tables once + n * ncg * 8 bytes.
//   kge_ce_bwd  pairs_bf16_v3_kernel<.., V3_DS> recomputes the tiles and writes
//               G16 = g_i * (softmax(i, j) - [j == label_i]) in bf16 (half the bytes of the f32
//               gradient matrix, no cast pass), then the two gradient GEMMs of the mixed-precision
//               backward (bwd_gemm.hip) consume it.
//
// Scores inside both kernels are bit-identical to kge_score_sp / kge_score_po on the same bf16 tables
// (same MFMA chain); exp/log are the hardware v_exp_f32 / libm logf: results match
// CrossEntropyLoss on those scores to float rounding (tests/test_gpu_ce.py states the tolerance).
#include "common.hpp"
#include <cstdlib>

namespace kge {

bool pairs_bf16_v3_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R, const Operand& TG);
int pairs_bf16_v3_column_groups(long long n, long long m);
long long pairs_bf16_v3_workspace_bytes(int d, long long n);
int run_pairs_bf16_v4_lse(int scorer, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG,
                          int dir, int d, long long n, long long m, hipStream_t st, void* ws, long long ws_bytes,
                          const CeArgs& ce, unsigned long long* dbg);
int run_pairs_bf16_v3_ce(int scorer, int epi, const Operand& A, const Operand& R, const Operand& TG, int dir,
                         int d, long long n, long long m, hipStream_t st, void* ws, long long ws_bytes,
                         const CeArgs& ce, unsigned long long* dbg);
int run_pairs_bwd_products16(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                             long long n, long long m, const unsigned short* G16, long long mp,
                             unsigned short* Q16, float* g_a, float* g_p, float* g_tgt, hipStream_t st,
                             bool q16_ready = false);

int run_pairs_bwd_products16_two(int scorer, const Operand& A1, const Operand& A2, const Operand& R,
                                 const Operand& R2, const Operand& TG, int d, long long n, long long n2, long long m,
                                 const unsigned short* G16, long long mp, unsigned short* Q16, float* g_a, float* g_p,
                                 float* g_tgt, float* acc_rel, long long acc_rel_rows, long long acc_rel_ld,
                                 float* dq_scratch, long long dq_scratch_bytes, hipStream_t st, bool q16_ready = false,
                                 bool clear_acc_rel = false);
long long gemm16_dq_scratch_bytes(int d, long long rows, long long m);
// ce_pairs_v8.hip: the loss passes on the persistent two-consumer-wave structure, from prepared query fragments
bool pairs_bf16_v8_ce_takes(int d, long long n, long long m);
int pairs_bf16_v8_ce_column_groups(int d, long long n, long long m, bool two_sided);
int run_pairs_bf16_v8_ce(int epi, const Operand& TG, int d, long long n, long long m, bool two_sided, const void* qf,
                         const CeArgs& ce, hipStream_t st);
int run_query_build(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                    long long n, void* qf, hipStream_t st);
int run_query_build_q16(int scorer, const Operand& A, const Operand* A2, const Operand& R, int dir, int d, long long n,
                        void* qf, unsigned short* q16, float* zero, long long zero_cnt, hipStream_t st);

// merge the column groups of a row: M = max_c m_c, L = sum_c l_c exp(m_c - M).  One wave per row,
// lanes over the column groups, xor-butterfly reductions (fixed order: deterministic).
//   * A label outside [0, m) was found by no lane of the scoring kernel -- its true_score slot is stale scratch --:
//     the row's loss reads NaN (another shard owns the label: the sharded loss takes the owner's value).  The labels
//     are checked HERE; until round 5 a fill launch wrote NaN over the slots before every scoring launch.
//   * sum.out != NULL: scale * sum_i loss_rows[i] as well, in the same launch: every workgroup leaves the sum of its
//     rows in sum.blk, the workgroup that arrives last (a wrapping counter in the workspace's control block,
//     zero between calls) adds them up in a fixed order -- the same bits whatever the arrival order.
struct CeSum {
  float* out;              // [1], or NULL: no sum
  float* blk;              // [gridDim.x] scratch
  unsigned int* counter;   // zero before the launch, zero after it
  const float* scale_dev;  // [1] or NULL
  float scale;
};

__global__ __launch_bounds__(1024) void ce_combine_kernel(const float* __restrict__ part, int ncg, long long n,
                                                         const float* __restrict__ true_score,
                                                         float* __restrict__ loss_rows, float* __restrict__ lse,
                                                         Index label, Index label2, long long side2_off, long long m,
                                                         CeSum sum) {
  // one wave per row, blockDim.x / 64 rows per workgroup: 4 (256 threads), or 16 with the sum -- a quarter of the
  // arrivals (fence + counter) of the 4-row launch, which cost as much as the launch they replaced
  __shared__ float sh_row[16];
  __shared__ int sh_last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long long i = (long long)blockIdx.x * nw + w;
  float li = 0.0f;
  if (i < n) {
    const float* p = part + i * ncg * 2;
    float M = -__builtin_inff();
    for (int c = lane; c < ncg; c += 64) M = fmaxf(M, p[2 * c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
    float L = 0.0f;
    for (int c = lane; c < ncg; c += 64) L += p[2 * c + 1] * expf(p[2 * c] - M);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) L += __shfl_xor(L, off, 64);
    if (lane == 0) {
      const float z = M + logf(L);
      const Index& lix = i < side2_off ? label : label2;
      const long long lab = lix.ptr != nullptr ? index_at(lix, i < side2_off ? i : i - side2_off) : -1;
      li = lab >= 0 && lab < m ? z - true_score[i] : __builtin_nanf("");
      lse[i] = z;
      loss_rows[i] = li;
    }
  }
  if (sum.out == nullptr) return;  // (uniform over the launch)
  if (lane == 0) sh_row[w] = li;
  __syncthreads();
  if (threadIdx.x == 0) {
    float sb = sh_row[0];
    for (int k = 1; k < nw; ++k) sb += sh_row[k];
    __hip_atomic_store(sum.blk + blockIdx.x, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();  // the partial is visible device-wide before this workgroup counts as arrived
    // atomicInc wraps at gridDim.x - 1: a counter that was ZERO before the launch is zero again once the last workgroup
    // has arrived.  A counter that was NOT zero (a workspace whose control block was never cleared) has period
    // gridDim.x too, but some workgroup then sees "last" before the others have written their partials: the sum would
    // be wrong on EVERY call, not only the first.  The control block must be zeroed by the caller before first use
    // (include/kge_amd.h: kge_ce_sp_po_fwd_sum; engine._ce_workspace allocates it with torch.zeros) -- ADVICE r5.
    sh_last = atomicInc(sum.counter, gridDim.x - 1) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (sh_last == 0 || w != 0) return;
  __threadfence();
  float acc = 0.0f;
  for (unsigned int b = lane; b < gridDim.x; b += 64)
    acc += __hip_atomic_load(sum.blk + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    float r = acc * sum.scale;
    if (sum.scale_dev != nullptr) r = r * sum.scale_dev[0];
    sum.out[0] = r;
  }
}

// ---- KvsAll: KL divergence against the normalised multi-hot labels of a row -------------------
// KLDivWithSoftmaxKgeLoss with a label matrix (kge/util/loss.py:208-213; train_KvsAll.py:274-294,
// no label smoothing): y_ij = 1/k_i on the k_i labels P_i of row i, so
//   loss_i = sum_j y_ij (log y_ij - log softmax_ij) = lse_i - (1/k_i) sum_{j in P_i} score(i,j) - log k_i
//   d loss_i / d score(i,j) = softmax_ij - y_ij
// (a row without labels has y = 0: loss 0, gradient 0).  lse_i comes from the V3_LSE kernel; the
// few label scores are evaluated here (one wave per row: q_i in registers, bf16-rounded like the
// matrix-core kernel's fragments, exact bf16 products, f32 accumulation in lane-partial +
// butterfly order -- equal to the kernel's scores up to f32 summation order); the backward
// subtracts g_i y_ij from the softmax gradient the V3_DS kernel wrote.
template <int SCORER>
__device__ __forceinline__ float kl_label_sum(const Operand& A, const Operand& R, const Operand& TG, int dir, int d,
                                              long long i, int lane, const long long* __restrict__ rowptr,
                                              const long long* __restrict__ col, long long col_lo, long long m,
                                              int& cnt) {
  // the sum of row i's label scores inside [col_lo, col_lo + m) (every lane of the wave returns it) and their number
  const int hp = d / 4;  // packed pairs (two bf16 per dword) per half
  const unsigned int* a = (const unsigned int*)((const unsigned short*)A.base + index_at(A.idx, i) * A.ld);
  const unsigned int* r = (const unsigned int*)((const unsigned short*)R.base + index_at(R.idx, i) * R.ld);
  unsigned int q0[2] = {0u, 0u}, q1[2] = {0u, 0u};  // d <= 512: at most 2 pairs per lane and half
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = lane + 64 * u;
    if (c < hp) bf16_qpair<SCORER>(dir, a[c], a[hp + c], r[c], r[hp + c], q0[u], q1[u]);
  }
  // four labels at a time: their row loads are in flight together (one label after the other was a chain of
  // dependent gathers, 11 us per launch for <= 8 labels per row); the sums are taken in label order as before --
  // the same bits
  float tsum = 0.0f;
  cnt = 0;
  const long long e1 = rowptr[i + 1];
  for (long long e = rowptr[i]; e < e1; e += 4) {
    bool ok[4];
    unsigned int t0[4][2] = {}, t1[4][2] = {};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long cl = e + k < e1 ? col[e + k] - col_lo : -1;
      ok[k] = cl >= 0 && cl < m;
      const unsigned int* t = (const unsigned int*)((const unsigned short*)TG.base + (ok[k] ? cl : 0) * TG.ld);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = lane + 64 * u;
        if (c < hp) {
          t0[k][u] = t[c];
          t1[k][u] = t[hp + c];
        }
      }
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = lane + 64 * u;
        if (c < hp) {
          acc[k] = __builtin_fmaf(__uint_as_float(q0[u] << 16), __uint_as_float(t0[k][u] << 16), acc[k]);
          acc[k] = __builtin_fmaf(__uint_as_float(q0[u] & 0xffff0000u), __uint_as_float(t0[k][u] & 0xffff0000u), acc[k]);
          acc[k] = __builtin_fmaf(__uint_as_float(q1[u] << 16), __uint_as_float(t1[k][u] << 16), acc[k]);
          acc[k] = __builtin_fmaf(__uint_as_float(q1[u] & 0xffff0000u), __uint_as_float(t1[k][u] & 0xffff0000u), acc[k]);
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ok[k]) {
        tsum += acc[k];
        ++cnt;
      }
    }
  }
  return tsum;
}

template <int SCORER>
__global__ __launch_bounds__(256) void kl_label_kernel(Operand A, Operand R, Operand TG, int dir, int d,
                                                       long long n, const long long* __restrict__ rowptr,
                                                       const long long* __restrict__ col,
                                                       float* __restrict__ label_sum, long long col_lo, long long m,
                                                       float* __restrict__ label_cnt) {
  // label columns are ids in [col_lo, col_lo + m) of the scored rows TG (entity-sharded training: GLOBAL ids, this
  // rank's shard starting at col_lo); ids outside the range belong to other shards and are skipped.  label_cnt
  // (may be NULL): the number of the row's labels inside the range
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int cnt;
  const float tsum = kl_label_sum<SCORER>(A, R, TG, dir, d, i, lane, rowptr, col, col_lo, m, cnt);
  if (lane == 0) {
    label_sum[i] = tsum;
    if (label_cnt != nullptr) label_cnt[i] = (float)cnt;
  }
}

__global__ __launch_bounds__(256) void kl_combine_kernel(const float* __restrict__ part, int ncg, long long n,
                                                         const float* __restrict__ label_sum,
                                                         const long long* __restrict__ rowptr,
                                                         float* __restrict__ loss_rows, float* __restrict__ lse,
                                                         const float* __restrict__ label_weight) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* p = part + i * ncg * 2;
  float M = -__builtin_inff();
  for (int c = lane; c < ncg; c += 64) M = fmaxf(M, p[2 * c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
  float L = 0.0f;
  for (int c = lane; c < ncg; c += 64) L += p[2 * c + 1] * expf(p[2 * c] - M);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) L += __shfl_xor(L, off, 64);
  if (lane == 0) {
    const float z = M + logf(L);
    const long long k = rowptr[i + 1] - rowptr[i];
    lse[i] = z;
    // label_weight (label smoothing, kge_kl_weighted_fwd): lse_i - w_i * (sum of the label scores), for
    // every row; the caller adds the terms that do not depend on the label scores
    if (label_weight != nullptr) loss_rows[i] = z - label_weight[i] * (k > 0 ? label_sum[i] : 0.0f);
    else loss_rows[i] = k > 0 ? z - label_sum[i] / (float)k - logf((float)k) : 0.0f;
  }
}

// kl_label_kernel + kl_combine_kernel in one launch (kge_kl_fwd on unsharded tables): the wave that merges row i's
// column groups evaluates its label scores first -- the same values, one launch (~4.5 us of a KvsAll step per query
// type) fewer.
template <int SCORER>
__global__ __launch_bounds__(256) void kl_label_combine_kernel(Operand A, Operand R, Operand TG, int dir, int d, long long n,
                                                               const long long* __restrict__ rowptr,
                                                               const long long* __restrict__ col, long long m,
                                                               const float* __restrict__ part, int ncg,
                                                               float* __restrict__ loss_rows, float* __restrict__ lse,
                                                               const float* __restrict__ label_weight) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int cnt;
  const float tsum = kl_label_sum<SCORER>(A, R, TG, dir, d, i, lane, rowptr, col, 0LL, m, cnt);
  const float* p = part + i * ncg * 2;
  float M = -__builtin_inff();
  for (int c = lane; c < ncg; c += 64) M = fmaxf(M, p[2 * c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
  float L = 0.0f;
  for (int c = lane; c < ncg; c += 64) L += p[2 * c + 1] * expf(p[2 * c] - M);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) L += __shfl_xor(L, off, 64);
  if (lane == 0) {
    const float z = M + logf(L);
    const long long k = rowptr[i + 1] - rowptr[i];
    lse[i] = z;
    if (label_weight != nullptr) loss_rows[i] = z - label_weight[i] * (k > 0 ? tsum : 0.0f);
    else loss_rows[i] = k > 0 ? z - tsum / (float)k - logf((float)k) : 0.0f;
  }
}

// G16[i, j] -= g_i / k_i for the labels j of row i (bf16 read-modify-write; labels unique per row)
__global__ __launch_bounds__(256) void kl_sub_kernel(unsigned short* __restrict__ g16, long long ld16, long long n,
                                                     const long long* __restrict__ rowptr,
                                                     const long long* __restrict__ col,
                                                     const float* __restrict__ g_rows, float g_scalar,
                                                     const float* __restrict__ label_weight, long long col_lo,
                                                     long long m, const float* __restrict__ g_dev = nullptr) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long long b = rowptr[i], e = rowptr[i + 1];
  if (e <= b) return;
  const float gi = g_rows != nullptr ? g_rows[i] : (g_dev != nullptr ? g_scalar * g_dev[0] : g_scalar);
  const float y = label_weight != nullptr ? gi * label_weight[i] : gi / (float)(e - b);
  for (long long x = b + lane; x < e; x += 64) {
    const long long cl = col[x] - col_lo;
    if (cl < 0 || cl >= m) continue;  // another shard's column
    unsigned short* p = g16 + i * ld16 + cl;
    const float v = __uint_as_float((unsigned int)*p << 16) - y;
    *p = (unsigned short)(bf16_pack(v, 0.0f) & 0xffffu);
  }
}

// ---- binary cross entropy with logits against multi-hot labels ---------------------------------
// BCEWithLogitsKgeLoss, bce_type None (kge/util/loss.py:137-159): sum over ALL entities j of
// BCEWithLogits(score(i,j) + offset, y_ij), y_ij = 1 on the row's labels.  With
// softplus(x) = log(1 + e^x):   loss_i = sum_j softplus(x_ij) - sum_{j in P_i} x_ij,  x = score + offset,
//   d loss_i / d score(i,j) = sigmoid(x_ij) - y_ij.
// The V3_SPLUS kernel accumulates the softplus sums (per row and column group), kl_label_kernel the
// label scores; the backward is the V3_DSIG kernel + the label subtraction + the two products.
__global__ __launch_bounds__(256) void bce_combine_kernel(const float* __restrict__ part, int ncg, long long n,
                                                          const float* __restrict__ label_sum,
                                                          const long long* __restrict__ rowptr, float offset,
                                                          float* __restrict__ loss_rows,
                                                          const float* __restrict__ label_cnt) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* p = part + i * ncg * 2;
  float S = 0.0f;
  for (int c = lane; c < ncg; c += 64) S += p[2 * c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) S += __shfl_xor(S, off, 64);
  if (lane == 0) {
    const float k = label_cnt != nullptr ? label_cnt[i] : (float)(rowptr[i + 1] - rowptr[i]);  // labels in range
    loss_rows[i] = S - (label_sum[i] + k * offset);
  }
}

// G16[i, j] -= g_i for the labels j of row i (y_ij = 1)
__global__ __launch_bounds__(256) void bce_sub_kernel(unsigned short* __restrict__ g16, long long ld16, long long n,
                                                      const long long* __restrict__ rowptr,
                                                      const long long* __restrict__ col,
                                                      const float* __restrict__ g_rows, float g_scalar,
                                                      long long col_lo, long long m,
                                                      const float* __restrict__ g_dev = nullptr) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float y = g_rows != nullptr ? g_rows[i] : (g_dev != nullptr ? g_scalar * g_dev[0] : g_scalar);
  for (long long x = rowptr[i] + lane; x < rowptr[i + 1]; x += 64) {
    const long long cl = col[x] - col_lo;
    if (cl < 0 || cl >= m) continue;  // another shard's column
    unsigned short* p = g16 + i * ld16 + cl;
    const float v = __uint_as_float((unsigned int)*p << 16) - y;
    *p = (unsigned short)(bf16_pack(v, 0.0f) & 0xffffu);
  }
}

// tools/ce_phases.py: per-workgroup s_memtime stamps of the next fused-loss launches (not part of the ABI)
static unsigned long long* g_ce_stamps = nullptr;

static inline long long al256(long long x) { return (x + 255) & ~255LL; }

// The V3_LSE pass (row statistics of softmax over all entities + the label's score): on the
// loader/consumer kernel (score_pairs_bf16_v4.hip) where it applies -- d in {256, 512}, one workgroup
// per CU -- else on the single-role kernel.  KGE_CE_V3=1 (tests, profiling) forces the latter.
int run_pairs_bf16_v4_epi(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R,
                          const Operand& TG, int dir, int d, long long n, long long m, hipStream_t st, void* ws,
                          long long ws_bytes, const CeArgs& ce, unsigned long long* dbg);

// Which kernel runs a loss pass over n rows per side:
//   pairs_bf16_v8_ce_kernel (ce_pairs_v8.hip; round 6) -- V3_LSE / V3_DS at d in {256, 512} for more than half a 256-row
//     chunk of rows: a query-build launch into the workspace's fragment area, then the persistent kernel;
//   pairs_bf16_v4_kernel -- the other shapes at d in {256, 512} and V3_DSIG; in-launch cooperative build;
//   pairs_bf16_v3_kernel -- d = 128, V3_SPLUS, and whatever v4 declines.
// Switches (kge_debug_set_switch): CE_V8 = 0 never the persistent kernel, 1 whenever its geometry allows; CE_V3 = 1
// forces the single-role kernel (tests, profiling).
static bool ce_takes_v8(int epi, int d, long long n, long long m, const Operand& TG) {
  if ((epi != V3_LSE && epi != V3_DS) || TG.idx.ptr != nullptr || TG.ld * 2 >= (1LL << 28) || sw(SW_CE_V3) == 1) return false;
  const long long e = sw(SW_CE_V8);
  if (e == 0) return false;
  if (e == 1) return (d == 256 || d == 512) && n >= 1 && m >= 1;
  return pairs_bf16_v8_ce_takes(d, n, m);
}

// column groups of the (max, sum exp) partials of a forward pass over n rows per side
static int ce_column_groups(int d, long long n, long long m, bool two_sided, const Operand& TG) {
  if (ce_takes_v8(V3_LSE, d, n, m, TG)) {
    const int ncg = pairs_bf16_v8_ce_column_groups(d, n, m, two_sided);
    if (ncg > 0) return ncg;
  }
  return pairs_bf16_v3_column_groups(two_sided ? 2 * ((n + 127) / 128) * 128 : n, m);
}
// the most column groups any kernel choice writes for this shape (workspace sizes)
static int ce_column_groups_max(int d, long long n, long long m, bool two_sided) {
  const int a = pairs_bf16_v3_column_groups(two_sided ? 2 * ((n + 127) / 128) * 128 : n, m);
  const int b = pairs_bf16_v8_ce_column_groups(d, n, m, two_sided);
  return a > b ? a : b;
}

// q16 != NULL (gradient pass): the launch that builds the fragments also writes the products' query matrix Q16 and
// clears `zero` (the relation-gradient accumulator) -- the products then skip their own build launch (*q16_done)
static int run_v8_ce_pass(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG,
                          int dir, int d, long long n, long long m, hipStream_t st, void* ws, long long coop,
                          const CeArgs& ce, unsigned short* q16 = nullptr, float* zero = nullptr,
                          long long zero_cnt = 0, bool* q16_done = nullptr) {
  // the fragment area of the workspace (behind the control block) holds whole 128-row groups of both sides
  if (ws == nullptr || coop < PAIRS_WS_CTRL_BYTES + (A2 ? 2 : 1) * ((n + 127) / 128) * 128 * (long long)d * 2)
    return KGE_ERR_UNSUPPORTED;
  void* const qf = (char*)ws + PAIRS_WS_CTRL_BYTES;
  int rc = q16 != nullptr ? run_query_build_q16(scorer, A, A2, R, dir, d, n, qf, q16, zero, zero_cnt, st)
                          : run_query_build(scorer, false, A, A2, R, dir, d, n, qf, st);
  if (rc != KGE_OK) return rc;
  if (q16 != nullptr && q16_done != nullptr) *q16_done = true;
  return run_pairs_bf16_v8_ce(epi, TG, d, n, m, A2 != nullptr, qf, ce, st);
}

// the G16 pass of a backward (V3_DS / V3_DSIG)
static int run_ds_pass(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG,
                       int dir, int d, long long n, long long m, hipStream_t st, void* ws, long long coop,
                       const CeArgs& ce, unsigned long long* dbg, unsigned short* q16 = nullptr, float* zero = nullptr,
                       long long zero_cnt = 0, bool* q16_done = nullptr) {
  if (ce_takes_v8(epi, d, n, m, TG)) {
    const int rc = run_v8_ce_pass(scorer, epi, A, A2, R, TG, dir, d, n, m, st, ws, coop, ce, q16, zero, zero_cnt, q16_done);
    if (rc != KGE_ERR_UNSUPPORTED || (q16_done != nullptr && *q16_done)) return rc == KGE_ERR_UNSUPPORTED ? KGE_ERR_LAUNCH : rc;
  }
  if (sw(SW_CE_V3) != 1) {
    CeArgs c4 = ce;
    if (A2 != nullptr) c4.rgn1 = 0;  // the v4 launcher takes the second side as an operand, not a marker
    const int rc = run_pairs_bf16_v4_epi(scorer, epi, A, A2, R, TG, dir, d, n, m, st, ws, coop, c4, dbg);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
  }
  return run_pairs_bf16_v3_ce(scorer, epi, A, R, TG, dir, d, n, m, st, ws, coop, ce, dbg);
}

// the V3_LSE pass (row statistics of softmax over all entities + the label's score); ce.part holds
// ce_column_groups(...) column groups per row
static int run_lse_pass(int scorer, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG, int dir,
                        int d, long long n, long long m, hipStream_t st, void* ws, long long coop, const CeArgs& ce,
                        unsigned long long* dbg, unsigned short* keep_q16 = nullptr) {
  // keep_q16 (KGE_FLAG_CE_KEEP_QUERIES): the build launch also writes the gradient products' query matrix there
  if (ce_takes_v8(V3_LSE, d, n, m, TG) && pairs_bf16_v8_ce_column_groups(d, n, m, A2 != nullptr) > 0) {
    const int rc = run_v8_ce_pass(scorer, V3_LSE, A, A2, R, TG, dir, d, n, m, st, ws, coop, ce, keep_q16);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
    return KGE_ERR_LAUNCH;  // (the caller sized ce.part for this kernel's column groups: no other kernel may fill it)
  }
  if (sw(SW_CE_V3) != 1) {
    CeArgs c4 = ce;
    if (A2 != nullptr) c4.rgn1 = 0;  // the v4 launcher takes the second side as an operand, not a marker
    const int rc = run_pairs_bf16_v4_lse(scorer, A, A2, R, TG, dir, d, n, m, st, ws, coop, c4, dbg);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
  }
  return run_pairs_bf16_v3_ce(scorer, V3_LSE, A, R, TG, dir, d, n, m, st, ws, coop, ce, dbg);
}

// scratch layout: [fragments + flags of the cooperative build][part n*ncg*2 f32][true n f32]   (forward)
//                 [fragments + flags][G16 n * ld16 bf16][Q16 n * d bf16]                          (backward)
static inline long long ce_ld16(long long m) { return (m + 63) & ~63LL; }

long long ce_workspace_bytes(int d, long long n, long long m) {
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const long long fwd = al256(n * ce_column_groups_max(d, n, m, false) * 8) + 2 * al256(n * 4);  // partials, label sums, counts
  const long long bwd = al256(n * ce_ld16(m) * 2) + al256(n * (long long)d * 2);
  return coop + (fwd > bwd ? fwd : bwd);
}

bool ce_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R, const Operand& TG) {
  return TG.idx.ptr == nullptr && pairs_bf16_v3_supported(scorer, dtype, d, A, R, TG);
}

int run_ce_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const Index& label, float* loss_rows, float* lse, void* ws, long long ws_bytes,
               hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const int ncg = ce_column_groups(d, n, m, false, TG);
  CeArgs ce{};
  ce.label = label;
  ce.part = (float*)((char*)ws + coop);
  ce.true_score = (float*)((char*)ws + coop + al256(n * ncg * 8));
  // (a label outside [0, num_ent) is found by no lane: ce_combine_kernel makes its row's loss NaN, not stale scratch)
  const int rc = run_lse_pass(scorer, A, nullptr, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps);
  if (rc != KGE_OK) return rc;
  hipLaunchKernelGGL(ce_combine_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ce.part, ncg, n,
                     ce.true_score, loss_rows, lse, label, label, n, m, CeSum{});
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_ce_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const Index& label, const float* lse, const float* g_rows, float g_scalar, float* g_a,
               float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const long long ld16 = ce_ld16(m);
  CeArgs ce{};
  ce.label = label;
  ce.lse = lse;
  ce.g_rows = g_rows;
  ce.g_scalar = g_scalar;
  ce.g16 = (unsigned short*)((char*)ws + coop);
  ce.ld16 = ld16;
  unsigned short* Q16 = (unsigned short*)((char*)ws + coop + al256(n * ld16 * 2));
  bool q16_done = false;
  const int rc = run_ds_pass(scorer, V3_DS, A, nullptr, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps, Q16, nullptr, 0,
                             &q16_done);
  if (rc != KGE_OK) return rc;
  return run_pairs_bwd_products16(scorer, dir, A, R, TG, d, n, m, ce.g16, ld16, Q16, g_a, g_p, g_tgt, st, q16_done);
}

int run_kl_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const long long* rowptr, const long long* col, float* loss_rows, float* lse, void* ws,
               long long ws_bytes, hipStream_t st, const float* label_weight, long long col_lo) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const int ncg = ce_column_groups(d, n, m, false, TG);
  CeArgs ce{};
  ce.part = (float*)((char*)ws + coop);
  ce.true_score = (float*)((char*)ws + coop + al256(n * ncg * 8));  // here: the rows' label-score sums
  const int rc = run_lse_pass(scorer, A, nullptr, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps);
  if (rc != KGE_OK) return rc;
  const dim3 grid((unsigned)((n + 3) / 4));
  if (col_lo == 0) {  // the whole table: label scores and the merge of the column groups in one launch
    if (scorer == KGE_COMPLEX)
      hipLaunchKernelGGL(kl_label_combine_kernel<KGE_COMPLEX>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col, m,
                         ce.part, ncg, loss_rows, lse, label_weight);
    else
      hipLaunchKernelGGL(kl_label_combine_kernel<KGE_DISTMULT>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col, m,
                         ce.part, ncg, loss_rows, lse, label_weight);
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
  }
  if (scorer == KGE_COMPLEX)
    hipLaunchKernelGGL(kl_label_kernel<KGE_COMPLEX>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col,
                       ce.true_score, col_lo, m, (float*)nullptr);
  else
    hipLaunchKernelGGL(kl_label_kernel<KGE_DISTMULT>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col,
                       ce.true_score, col_lo, m, (float*)nullptr);
  hipLaunchKernelGGL(kl_combine_kernel, grid, dim3(256), 0, st, ce.part, ncg, n, ce.true_score, rowptr, loss_rows,
                     lse, label_weight);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_kl_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const long long* rowptr, const long long* col, const float* lse, const float* g_rows,
               float g_scalar, float* g_a, float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st,
               const float* label_weight, const float* label_bias, long long col_lo) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const long long ld16 = ce_ld16(m);
  CeArgs ce{};
  // rows without labels: zero gradient (all-zero label rows normalise to zero) -- but not under label
  // smoothing (label_weight given), where every row's label distribution has mass on every entity
  ce.rowptr = label_weight != nullptr ? nullptr : rowptr;
  ce.row_bias = label_bias;  // the uniform mass of smoothed labels, subtracted at every column inside the kernel
  ce.lse = lse;
  ce.g_rows = g_rows;
  ce.g_scalar = g_scalar;
  ce.g16 = (unsigned short*)((char*)ws + coop);
  ce.ld16 = ld16;
  unsigned short* Q16 = (unsigned short*)((char*)ws + coop + al256(n * ld16 * 2));
  const int rc = run_ds_pass(scorer, V3_DS, A, nullptr, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps);
  if (rc != KGE_OK) return rc;
  hipLaunchKernelGGL(kl_sub_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ce.g16, ld16, n, rowptr, col,
                     g_rows, g_scalar, label_weight, col_lo, m, (const float*)nullptr);
  if (hipGetLastError() != hipSuccess) return KGE_ERR_LAUNCH;
  return run_pairs_bwd_products16(scorer, dir, A, R, TG, d, n, m, ce.g16, ld16, Q16, g_a, g_p, g_tgt, st);
}

int run_bce_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                long long m, const long long* rowptr, const long long* col, float offset, float* loss_rows, void* ws,
                long long ws_bytes, hipStream_t st, long long col_lo) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const int ncg = pairs_bf16_v3_column_groups(n, m);
  CeArgs ce{};
  ce.offset = offset;
  ce.part = (float*)((char*)ws + coop);
  ce.true_score = (float*)((char*)ws + coop + al256(n * ncg * 8));  // the rows' label-score sums
  const int rc = run_pairs_bf16_v3_ce(scorer, V3_SPLUS, A, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps);
  if (rc != KGE_OK) return rc;
  const dim3 grid((unsigned)((n + 3) / 4));
  float* cnt = ce.true_score + al256(n * 4) / 4;  // the rows' counts of labels in range, behind the label sums
  if (scorer == KGE_COMPLEX)
    hipLaunchKernelGGL(kl_label_kernel<KGE_COMPLEX>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col,
                       ce.true_score, col_lo, m, cnt);
  else
    hipLaunchKernelGGL(kl_label_kernel<KGE_DISTMULT>, grid, dim3(256), 0, st, A, R, TG, dir, d, n, rowptr, col,
                       ce.true_score, col_lo, m, cnt);
  hipLaunchKernelGGL(bce_combine_kernel, grid, dim3(256), 0, st, ce.part, ncg, n, ce.true_score, rowptr, offset,
                     loss_rows, (const float*)cnt);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_bce_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                long long m, const long long* rowptr, const long long* col, float offset, const float* g_rows,
                float g_scalar, float* g_a, float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st,
                long long col_lo) {
  if (n == 0) return KGE_OK;
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const long long ld16 = ce_ld16(m);
  CeArgs ce{};
  ce.offset = offset;
  ce.g_rows = g_rows;
  ce.g_scalar = g_scalar;
  ce.g16 = (unsigned short*)((char*)ws + coop);
  ce.ld16 = ld16;
  unsigned short* Q16 = (unsigned short*)((char*)ws + coop + al256(n * ld16 * 2));
  const int rc = run_ds_pass(scorer, V3_DSIG, A, nullptr, R, TG, dir, d, n, m, st, ws, coop, ce, g_ce_stamps);
  if (rc != KGE_OK) return rc;
  hipLaunchKernelGGL(bce_sub_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ce.g16, ld16, n, rowptr, col,
                     g_rows, g_scalar, col_lo, m, (const float*)nullptr);
  if (hipGetLastError() != hipSuccess) return KGE_ERR_LAUNCH;
  return run_pairs_bwd_products16(scorer, dir, A, R, TG, d, n, m, ce.g16, ld16, Q16, g_a, g_p, g_tgt, st);
}

// ---- both directions of a 1vsAll batch in one set of launches (kge_ce_sp_po_fwd / _bwd) ---------
// Rows [0, n) of every per-row array are the (s, p, ?) queries with labels o, rows [n, 2n) the
// (?, p, o) queries with labels s: what train_1vsAll.py:64-81 does in two passes.  One scoring
// launch covers both sides (one start-up, like kge_score_sp_po), the gradient products run once
// over 2n rows, and dT needs no second accumulation pass.
static inline long long ce2_rows(long long n) { return 2 * ((n + 127) / 128) * 128; }  // padded, for geometry

long long ce2_workspace_bytes(int d, long long n, long long m) {
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));  // sized for two sides
  // forward: column-group partials, label scores, the combine launch's per-workgroup sums (kge_ce_sp_po_fwd_sum)
  const long long fwd = al256(2 * n * ce_column_groups_max(d, n, m, true) * 8) + al256(2 * n * 4) +
                        al256((2 * n + 3) / 4 * 4);
  // backward: G16, Q16 and (kge_ce_sp_po_bwd_accum) the f32 dQ rows + the split-K partials of that product
  const long long bwd = al256(2 * n * ce_ld16(m) * 2) + al256(2 * n * (long long)d * 2) +
                        al256(2 * n * (long long)d * 4) + al256(gemm16_dq_scratch_bytes(d, 2 * n, m));
  return coop + (fwd > bwd ? fwd : bwd);
}

// Does the backward of this shape start from fragments the forward left (both calls under KGE_FLAG_CE_KEEP_QUERIES)?
// Only where BOTH passes run on the persistent kernel: its fragment area is written by the build launch alone.
static bool ce2_keeps(int d, long long n, long long m, const Operand& TG) {
  return ce_takes_v8(V3_LSE, d, n, m, TG) && ce_takes_v8(V3_DS, d, n, m, TG) &&
         pairs_bf16_v8_ce_column_groups(d, n, m, true) > 0;
}

int run_ce2_fwd(int scorer, const Operand& S, const Operand& O, const Operand& R, const Operand& TG, int d,
                long long n, long long m, float* loss_rows, float* lse, void* ws, long long ws_bytes,
                hipStream_t st, float* loss_sum, const float* scale_dev, float scale, bool keep) {
  // loss_sum != NULL (kge_ce_sp_po_fwd_sum): loss_sum[0] = scale * scale_dev[0] * sum of the 2n loss rows, from
  // the combine launch itself
  if (n == 0) {
    if (loss_sum != nullptr && !fill_words_async(loss_sum, 0, sizeof(float), st)) return KGE_ERR_LAUNCH;
    return KGE_OK;
  }
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce2_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const int ncg = ce_column_groups(d, n, m, true, TG);
  CeArgs ce{};
  ce.label = O.idx;   // side 1: true objects
  ce.label2 = S.idx;  // side 2: true subjects
  ce.rgn1 = -1;       // two-sided: the launcher fills in the row-group split
  ce.a2 = O;
  ce.side2_off = n;
  ce.part = (float*)((char*)ws + coop);
  ce.true_score = (float*)((char*)ws + coop + al256(2 * n * ncg * 8));
  // keep: the backward's Q16 (its place in the BACKWARD's layout of the same workspace: behind G16, beyond everything the
  // forward uses) is written by this call's build launch
  unsigned short* const keep_q16 =
      keep && ce2_keeps(d, n, m, TG) ? (unsigned short*)((char*)ws + coop + al256(2 * n * ce_ld16(m) * 2)) : nullptr;
  const int rc = run_lse_pass(scorer, S, &O, R, TG, KGE_SP_, d, n, m, st, ws, coop, ce, g_ce_stamps, keep_q16);
  if (rc != KGE_OK) return rc;
  CeSum sum{};
  if (loss_sum != nullptr) {
    sum.out = loss_sum;
    sum.blk = ce.true_score + al256(2 * n * 4) / 4;
    sum.counter = (unsigned int*)((char*)ws + PAIRS_WS_CE_COUNTER_OFF);
    sum.scale_dev = scale_dev;
    sum.scale = scale;
  }
  const int rpb = loss_sum != nullptr ? 16 : 4;  // rows per workgroup
  hipLaunchKernelGGL(ce_combine_kernel, dim3((unsigned)((2 * n + rpb - 1) / rpb)), dim3(64 * rpb), 0, st, ce.part, ncg,
                     2 * n, ce.true_score, loss_rows, lse, ce.label, ce.label2, n, m, sum);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_ce2_bwd(int scorer, const Operand& S, const Operand& O, const Operand& R, const Operand& TG, int d,
                long long n, long long m, const float* lse, const float* g_rows, float g_scalar, float* g_a,
                float* g_p, float* g_tgt, float* acc_rel, long long acc_rel_rows, long long acc_rel_ld, void* ws,
                long long ws_bytes, hipStream_t st, const float* g_dev, const float* g_dev2, bool kept) {
  // acc_rel != NULL (kge_ce_sp_po_bwd_accum): g_a / g_p are not returned; the row gradients are added
  // into g_tgt (on top of dT) and into acc_rel [acc_rel_rows, acc_rel_ld], which the query-build launch of the
  // products clears on its way (a launch of its own until round 5)
  if (n == 0) {
    if (acc_rel != nullptr && (!fill_words_async(acc_rel, 0, (size_t)acc_rel_rows * acc_rel_ld * sizeof(float), st) ||
                               !fill_words_async(g_tgt, 0, (size_t)m * d * sizeof(float), st)))
      return KGE_ERR_LAUNCH;
    return KGE_OK;
  }
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < ce2_workspace_bytes(d, n, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n));
  const long long ld16 = ce_ld16(m);
  float* const dq_rows = (float*)((char*)ws + coop + al256(2 * n * ld16 * 2) + al256(2 * n * (long long)d * 2));
  const long long dq_scratch_bytes = gemm16_dq_scratch_bytes(d, 2 * n, m);
  float* const dq_scratch = dq_scratch_bytes > 0 ? dq_rows + al256(2 * n * (long long)d * 4) / 4 : nullptr;
  if (acc_rel != nullptr) g_a = dq_rows;
  CeArgs ce{};
  ce.label = O.idx;
  ce.label2 = S.idx;
  ce.rgn1 = -1;
  ce.a2 = O;
  ce.side2_off = n;
  ce.lse = lse;
  ce.g_rows = g_rows;
  ce.g_scalar = g_scalar;
  ce.g_dev = g_dev;
  ce.g_dev2 = g_dev2;
  ce.g16 = (unsigned short*)((char*)ws + coop);
  ce.ld16 = ld16;
  unsigned short* Q16 = (unsigned short*)((char*)ws + coop + al256(2 * n * ld16 * 2));
  bool q16_done = false;
  if (kept && acc_rel != nullptr && ce2_keeps(d, n, m, TG)) {
    // KGE_FLAG_CE_KEEP_QUERIES: the fragments and Q16 are the forward's -- straight to the gradient pass; acc_rel is
    // cleared by the split-K sum's launch
    const int rc = run_pairs_bf16_v8_ce(V3_DS, TG, d, n, m, true, (char*)ws + PAIRS_WS_CTRL_BYTES, ce, st);
    if (rc != KGE_OK) return rc == KGE_ERR_UNSUPPORTED ? KGE_ERR_LAUNCH : rc;
    return run_pairs_bwd_products16_two(scorer, S, O, R, R, TG, d, n, n, m, ce.g16, ld16, Q16, g_a, g_p, g_tgt, acc_rel,
                                        acc_rel_rows, acc_rel_ld, dq_scratch, dq_scratch_bytes, st, true, true);
  }
  const int rc = run_ds_pass(scorer, V3_DS, S, &O, R, TG, KGE_SP_, d, n, m, st, ws, coop, ce, g_ce_stamps, Q16, acc_rel,
                             acc_rel != nullptr ? acc_rel_rows * acc_rel_ld : 0LL, &q16_done);
  if (rc != KGE_OK) return rc;
  return run_pairs_bwd_products16_two(scorer, S, O, R, R, TG, d, n, n, m, ce.g16, ld16, Q16, g_a, g_p, g_tgt, acc_rel,
                                      acc_rel_rows, acc_rel_ld, dq_scratch, dq_scratch_bytes, st, q16_done);
}

// ---- both query types of a KvsAll batch, backward (kge_kl2_bwd_accum / kge_bce2_bwd_accum) ------------------------
// TrainingJobKvsAll scores the sp_ queries and the _po queries of a batch one after the other and back-propagates each
// loss on its own (train_KvsAll.py:274-294): two d loss / d score passes, FOUR gradient products, and autograd's
// index_add / accumulate passes over the [E, d] gradient in between (a third of the step's kernel time at the
// FB15k-237 shape).  Here each type's d loss / d score pass writes its rows of ONE G16 matrix (sp_ rows first), and the
// two-sided products of the 1vsAll step run once over the n1 + n2 rows: dT sums both types inside the product, the
// chain launch scatters the row gradients on top of it and into the relation gradient -- complete table gradients,
// as kge_ce_sp_po_bwd_accum returns them.  kind 0: kl (lse given; label_weight / label_bias for smoothed labels),
// kind 1: bce with logits (offset).
long long multilabel2_workspace_bytes(int d, long long n1, long long n2, long long m) {
  const long long nmax = n1 > n2 ? n1 : n2, rows = n1 + n2;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, nmax));
  return coop + al256(rows * ce_ld16(m) * 2) + al256(rows * (long long)d * 2) + al256(rows * (long long)d * 4) +
         al256(gemm16_dq_scratch_bytes(d, rows, m));
}

int run_multilabel2_bwd_accum(int scorer, int kind, float offset, const LossSide& sp, const LossSide& po,
                              const Operand& TG, int d, long long m, float* grad_ent, float* grad_rel,
                              long long rel_rows, long long rel_ld, void* ws, long long ws_bytes, hipStream_t st) {
  const long long n1 = sp.n, n2 = po.n, rows = n1 + n2;
  if (rows == 0) {
    if (!fill_words_async(grad_rel, 0, (size_t)rel_rows * rel_ld * sizeof(float), st) ||
        !fill_words_async(grad_ent, 0, (size_t)m * d * sizeof(float), st))
      return KGE_ERR_LAUNCH;
    return KGE_OK;
  }
  if (ws == nullptr || ((uintptr_t)ws & 255) || ws_bytes < multilabel2_workspace_bytes(d, n1, n2, m)) return KGE_ERR_WORKSPACE;
  const long long coop = al256(pairs_bf16_v3_workspace_bytes(d, n1 > n2 ? n1 : n2));
  const long long ld16 = ce_ld16(m);
  unsigned short* const G16 = (unsigned short*)((char*)ws + coop);
  unsigned short* const Q16 = (unsigned short*)((char*)ws + coop + al256(rows * ld16 * 2));
  float* const dq_rows = (float*)((char*)Q16 + al256(rows * (long long)d * 2));
  const long long dq_scratch_bytes = gemm16_dq_scratch_bytes(d, rows, m);
  float* const dq_scratch = dq_scratch_bytes > 0 ? dq_rows + al256(rows * (long long)d * 4) / 4 : nullptr;
  for (int side = 0; side < 2; ++side) {
    const LossSide& x = side ? po : sp;
    if (x.n == 0) continue;
    CeArgs ce{};
    ce.g_rows = x.g_rows;
    ce.g_scalar = x.g_scalar;
    ce.g_dev = x.g_dev;  // (g_rows == NULL: every row's gradient is g_scalar * g_dev[0])
    ce.g16 = G16 + (side ? n1 : 0) * ld16;
    ce.ld16 = ld16;
    if (kind == 0) {
      ce.rowptr = x.label_weight != nullptr ? nullptr : x.rowptr;  // (see run_kl_bwd)
      ce.row_bias = x.label_bias;
      ce.lse = x.lse;
    } else {
      ce.offset = offset;
    }
    const int rc = run_ds_pass(scorer, kind == 0 ? V3_DS : V3_DSIG, x.A, nullptr, x.R, TG, side ? KGE_PO_ : KGE_SP_, d,
                               x.n, m, st, ws, coop, ce, g_ce_stamps);
    if (rc != KGE_OK) return rc;
    const dim3 grid((unsigned)((x.n + 3) / 4));
    if (kind == 0)
      hipLaunchKernelGGL(kl_sub_kernel, grid, dim3(256), 0, st, ce.g16, ld16, x.n, x.rowptr, x.col, x.g_rows, x.g_scalar,
                         x.label_weight, 0LL, m, x.g_dev);
    else
      hipLaunchKernelGGL(bce_sub_kernel, grid, dim3(256), 0, st, ce.g16, ld16, x.n, x.rowptr, x.col, x.g_rows,
                         x.g_scalar, 0LL, m, x.g_dev);
    if (hipGetLastError() != hipSuccess) return KGE_ERR_LAUNCH;
  }
  return run_pairs_bwd_products16_two(scorer, sp.A, po.A, sp.R, po.R, TG, d, n1, n2, m, G16, ld16, Q16, dq_rows, nullptr,
                                      grad_ent, grad_rel, rel_rows, rel_ld, dq_scratch, dq_scratch_bytes, st);
}

void ce_set_stamps(unsigned long long* p) { g_ce_stamps = p; }

}  // namespace kge
