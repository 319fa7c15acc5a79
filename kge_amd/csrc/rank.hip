// rank.hip -- filter-and-rank counts of EntityRankingJob on gfx950:
// _filter_and_rank + _get_ranks_and_num_ties (kge/job/eval_entity_ranking.py:533-596).
//
// The reference makes ~10 full passes over the [n, 2c] score matrix per ranking (clone,
// isnan, subtract dense 0/inf labels, isclose, gt, and, two sums) plus a sparse->dense
// label scatter.  Here: ONE streaming pass over the scores (HBM scan bound, n*c*4 bytes)
// counting `close` and `greater & !close` against the row's true score, then a sparse
// correction over the row's CSR filter labels: a filtered column becomes -inf
// (x - inf = -inf, NaN -> -inf, :565-566,583-586), so its raw contribution is taken out
// and the contribution of -inf is put in.  Integer work: bit-exact vs oracle/.
#include "common.hpp"

namespace kge {

constexpr int RK_THREADS = 256;

__global__ __launch_bounds__(RK_THREADS) void rank_kernel(
    const float* __restrict__ scores, long long lds, long long n, long long c,
    const float* __restrict__ true_scores, const long long* __restrict__ rowptr,
    const long long* __restrict__ lcol, long long col_offset,
    const long long* __restrict__ true_col, float atol, float rtol,
    unsigned long long* __restrict__ rank, unsigned long long* __restrict__ ties,
    long long cols_per_block) {
  const long long i = blockIdx.y;
  const long long jb = (long long)blockIdx.x * cols_per_block;
  long long je = jb + cols_per_block;
  if (je > c) je = c;
  float t = true_scores[i];
  if (t != t) t = -__builtin_inff();
  const float* row = scores + i * lds;
  int gt = 0, cl = 0;

  // scalar head up to 16-byte alignment, float4 body, scalar tail
  long long j0 = jb;
  const uintptr_t addr = (uintptr_t)(row + jb);
  long long head = ((16 - (addr & 15)) & 15) >> 2;
  if (head > je - jb) head = je - jb;
  if ((long long)threadIdx.x < head) count_one(row[jb + threadIdx.x], t, atol, rtol, gt, cl);
  j0 = jb + head;
  const long long nvec = (je - j0) >> 2;
  const f32x4* vrow = reinterpret_cast<const f32x4*>(row + j0);
  for (long long v = threadIdx.x; v < nvec; v += RK_THREADS) {
    f32x4 x = vrow[v];
    count_one(x[0], t, atol, rtol, gt, cl);
    count_one(x[1], t, atol, rtol, gt, cl);
    count_one(x[2], t, atol, rtol, gt, cl);
    count_one(x[3], t, atol, rtol, gt, cl);
  }
  const long long jt = j0 + (nvec << 2);
  if (jt + (long long)threadIdx.x < je) count_one(row[jt + threadIdx.x], t, atol, rtol, gt, cl);

  // sparse filter correction (labels of this row that fall into this block's columns)
  if (rowptr != nullptr) {
    const long long tc = true_col ? true_col[i] : -1;
    for (long long e = rowptr[i] + threadIdx.x; e < rowptr[i + 1]; e += RK_THREADS) {
      const long long g = lcol[e];
      if (true_col && g == tc) continue;  // the positive itself stays (:288-290)
      const long long j = g - col_offset;
      if (j < jb || j >= je) continue;
      int rg = 0, rc = 0;
      count_one(row[j], t, atol, rtol, rg, rc);
      const bool fc = is_close(-__builtin_inff(), t, atol, rtol);
      gt -= rg;  // -inf is never greater
      cl += (fc ? 1 : 0) - rc;
    }
  }

  // block reduction
  __shared__ int sg[RK_THREADS / 64], sc[RK_THREADS / 64];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    gt += __shfl_xor(gt, off, 64);
    cl += __shfl_xor(cl, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sg[threadIdx.x >> 6] = gt;
    sc[threadIdx.x >> 6] = cl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long G = 0, C = 0;
#pragma unroll
    for (int w = 0; w < RK_THREADS / 64; ++w) {
      G += sg[w];
      C += sc[w];
    }
    if (G != 0) atomicAdd(&rank[i], (unsigned long long)G);
    if (C != 0) atomicAdd(&ties[i], (unsigned long long)C);
  }
}

int run_rank(const float* scores, long long lds, long long n, long long c,
             const float* true_scores, const long long* rowptr, const long long* lcol,
             long long col_offset, const long long* true_col, float atol, float rtol,
             long long* rank, long long* ties, hipStream_t st) {
  if (n == 0 || c == 0) return KGE_OK;
  // enough blocks to fill the chip: ~2048 blocks, at least 4096 columns per block
  long long splits = (2048 + n - 1) / n;
  long long cpb = (c + splits - 1) / splits;
  if (cpb < 4096) cpb = 4096;
  cpb = (cpb + 3) & ~3LL;
  splits = (c + cpb - 1) / cpb;
  dim3 grid((unsigned)splits, (unsigned)n);
  hipLaunchKernelGGL(rank_kernel, grid, dim3(RK_THREADS), 0, st, scores, lds, n, c, true_scores,
                     rowptr, lcol, col_offset, true_col, atol, rtol,
                     (unsigned long long*)rank, (unsigned long long*)ties, cpb);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- evaluation without host round trips -------------------------------------
// The reference builds, per batch, sparse label tensors from numba dict lookups on the host
// (KvsAllIndex: kge/indexing.py:10-194; job/util.py:6-29; _collate: eval_entity_ranking.py:77-101),
// densifies them per chunk (:489-531) and ranks raw / filtered / filtered-with-test in three
// passes.  Here the index lives in HBM as sorted arrays; a batch needs
//   filter_lookup_kernel   per-row binary search: key -> [begin, end) into the index's value array
//   rank_multi_kernel      ONE scan of the scores for the raw counts + a sparse correction per
//                          filter set: all rankings of a direction from one pass
//   rank_hist_kernel       tie policy + rank histogram (:598-618, 665-687)
// and no device -> host synchronisation.

// One WAVE per row: a 64-ary search.  A thread-per-row binary search over the ~3e5 keys of an FB15k-237-sized
// index is 18 DEPENDENT loads (8 us of pure latency for 512 rows, and the evaluation loop does four lookups
// per batch: they were 31 % of its kernel time); with 64 probes per step the range shrinks 64-fold per round
// trip: 3-4 steps.
__global__ __launch_bounds__(256) void filter_lookup_kernel(const long long* __restrict__ keys, long long num_keys,
                                                            const long long* __restrict__ starts, Index a,
                                                            Index b, long long mult, long long n,
                                                            long long* __restrict__ begin,
                                                            long long* __restrict__ end) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const long long key = index_at(a, i) * mult + index_at(b, i);
  long long lo = 0, hi = num_keys;  // invariant: the first position with keys[pos] >= key lies in [lo, hi]
  while (hi - lo > 64) {
    const long long step = (hi - lo + 63) >> 6;
    const long long pos = lo + lane * step;  // probes, ascending
    const bool below = pos < hi && keys[pos] < key;
    const int c = __popcll(__ballot(below));  // the probes below the key are a prefix
    if (c == 0) {
      hi = lo;
    } else {
      const long long nhi = lo + c * step;  // the first probe that is not below (or past the end)
      lo = lo + (c - 1) * step + 1;
      hi = nhi < hi ? nhi : hi;
    }
  }
  const long long pos = lo + lane;
  const bool below = pos < hi && keys[pos] < key;
  lo += __popcll(__ballot(below));
  if (lane == 0) {
    const bool hit = lo < num_keys && keys[lo] == key;
    begin[i] = hit ? starts[lo] : 0;
    end[i] = hit ? starts[lo + 1] : 0;
  }
}

// Up to four lookups (the sp / po keys of the filtered and the filtered-with-test index: what one evaluation
// batch needs) in ONE launch: blockIdx.y = query.
constexpr int FLQ_MAX = 4;
struct FilterQueries {
  const long long* keys[FLQ_MAX];
  long long num_keys[FLQ_MAX];
  const long long* starts[FLQ_MAX];
  Index a[FLQ_MAX], b[FLQ_MAX];
  long long mult[FLQ_MAX];
  long long* begin[FLQ_MAX];
  long long* end[FLQ_MAX];
};

__global__ __launch_bounds__(256) void filter_lookup_multi_kernel(FilterQueries Q, long long n) {
  const int q = blockIdx.y;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const long long* __restrict__ keys = Q.keys[q];
  const long long num_keys = Q.num_keys[q];
  const long long key = index_at(Q.a[q], i) * Q.mult[q] + index_at(Q.b[q], i);
  long long lo = 0, hi = num_keys;  // the 64-ary search of filter_lookup_kernel
  while (hi - lo > 64) {
    const long long step = (hi - lo + 63) >> 6;
    const long long pos = lo + lane * step;
    const bool below = pos < hi && keys[pos] < key;
    const int c = __popcll(__ballot(below));
    if (c == 0) {
      hi = lo;
    } else {
      const long long nhi = lo + c * step;
      lo = lo + (c - 1) * step + 1;
      hi = nhi < hi ? nhi : hi;
    }
  }
  const long long pos = lo + lane;
  const bool below = pos < hi && keys[pos] < key;
  lo += __popcll(__ballot(below));
  if (lane == 0) {
    const bool hit = lo < num_keys && keys[lo] == key;
    Q.begin[q][i] = hit ? Q.starts[q][lo] : 0;
    Q.end[q][i] = hit ? Q.starts[q][lo + 1] : 0;
  }
}

constexpr int RK_MAXF = 4;
struct RankFilters {
  int K;
  const long long* begin[RK_MAXF];  // [n] ranges into col[k]
  const long long* end[RK_MAXF];
  const long long* col[RK_MAXF];    // global entity ids, unique within a range
};

__device__ __forceinline__ void rk_block_sum(int& a, int& b, int* sa, int* sb) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  __syncthreads();  // previous use of sa / sb is over
  if ((threadIdx.x & 63) == 0) {
    sa[threadIdx.x >> 6] = a;
    sb[threadIdx.x >> 6] = b;
  }
  __syncthreads();
  a = b = 0;
#pragma unroll
  for (int w = 0; w < RK_THREADS / 64; ++w) {
    a += sa[w];
    b += sb[w];
  }
}

// rank / ties: [K + 1][n], row 0 = raw, row k + 1 = filter set k; ACCUMULATED
__global__ __launch_bounds__(RK_THREADS) void rank_multi_kernel(
    const float* __restrict__ scores, long long lds, long long n, long long c,
    const float* __restrict__ true_scores, RankFilters F, long long col_offset,
    const long long* __restrict__ true_col, float atol, float rtol, unsigned long long* __restrict__ rank,
    unsigned long long* __restrict__ ties, long long cols_per_block) {
  __shared__ int sg[RK_THREADS / 64], sc[RK_THREADS / 64];
  const long long i = blockIdx.y;
  const long long jb = (long long)blockIdx.x * cols_per_block;
  long long je = jb + cols_per_block;
  if (je > c) je = c;
  float t = true_scores[i];
  if (t != t) t = -__builtin_inff();
  const float* row = scores + i * lds;
  int gt = 0, cl = 0;
  const uintptr_t addr = (uintptr_t)(row + jb);
  long long head = ((16 - (addr & 15)) & 15) >> 2;
  if (head > je - jb) head = je - jb;
  if ((long long)threadIdx.x < head) count_one(row[jb + threadIdx.x], t, atol, rtol, gt, cl);
  const long long j0 = jb + head;
  const long long nvec = (je - j0) >> 2;
  const f32x4* vrow = reinterpret_cast<const f32x4*>(row + j0);
  for (long long v = threadIdx.x; v < nvec; v += RK_THREADS) {
    f32x4 x = vrow[v];
    count_one(x[0], t, atol, rtol, gt, cl);
    count_one(x[1], t, atol, rtol, gt, cl);
    count_one(x[2], t, atol, rtol, gt, cl);
    count_one(x[3], t, atol, rtol, gt, cl);
  }
  const long long jt = j0 + (nvec << 2);
  if (jt + (long long)threadIdx.x < je) count_one(row[jt + threadIdx.x], t, atol, rtol, gt, cl);
  rk_block_sum(gt, cl, sg, sc);
  const int G = gt, C = cl;  // raw counts of this block's columns (every thread)
  if (threadIdx.x == 0) {
    if (G != 0) atomicAdd(&rank[i], (unsigned long long)G);
    if (C != 0) atomicAdd(&ties[i], (unsigned long long)C);
  }
  const long long tc = true_col ? true_col[i] : -1;
  const int fc = is_close(-__builtin_inff(), t, atol, rtol) ? 1 : 0;
  for (int k = 0; k < F.K; ++k) {
    int dg = 0, dc = 0;  // a filtered column becomes -inf: take its raw contribution out, put -inf's in
    const long long* lcol = F.col[k];
    for (long long e = F.begin[k][i] + threadIdx.x; e < F.end[k][i]; e += RK_THREADS) {
      const long long g = lcol[e];
      if (true_col && g == tc) continue;  // the positive itself stays (:288-290)
      const long long j = g - col_offset;
      if (j < jb || j >= je) continue;
      int rg = 0, rc = 0;
      count_one(row[j], t, atol, rtol, rg, rc);
      dg -= rg;
      dc += fc - rc;
    }
    rk_block_sum(dg, dc, sg, sc);
    if (threadIdx.x == 0) {
      const long long Gk = (long long)G + dg, Ck = (long long)C + dc;
      if (Gk != 0) atomicAdd(&rank[(k + 1) * n + i], (unsigned long long)Gk);
      if (Ck != 0) atomicAdd(&ties[(k + 1) * n + i], (unsigned long long)Ck);
    }
  }
}

// hist[m][rank_of(rank[m][i], ties[m][i])] += 1 (float32 histogram, like the reference's)
__global__ __launch_bounds__(256) void rank_hist_kernel(const long long* __restrict__ rank,
                                                        const long long* __restrict__ ties, long long total,
                                                        long long n, int policy, float* __restrict__ hist,
                                                        long long ldh, long long num_ent,
                                                        long long* __restrict__ ranks_out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long r0 = rank[t], ti = ties[t];
  // EntityRankingJob._get_ranks (:598-618): 0 rounded_mean_rank, 1 best_rank, 2 worst_rank
  const long long r = policy == 0 ? r0 + ti / 2 : (policy == 1 ? r0 : r0 + ti - 1);
  if (ranks_out) ranks_out[t] = r;
  if (r >= 0 && r < num_ent) unsafeAtomicAdd(hist + (t / n) * ldh + r, 1.0f);
}

int run_filter_lookup(const long long* keys, long long num_keys, const long long* starts, const Index& a,
                      const Index& b, long long mult, long long n, long long* begin, long long* end,
                      hipStream_t st) {
  if (n == 0) return KGE_OK;
  hipLaunchKernelGGL(filter_lookup_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, keys, num_keys,
                     starts, a, b, mult, n, begin, end);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_filter_lookup_multi(int nq, const long long* const* keys, const long long* num_keys,
                            const long long* const* starts, const Index* a, const Index* b, const long long* mult,
                            long long n, long long* const* begin, long long* const* end, hipStream_t st) {
  if (n == 0 || nq == 0) return KGE_OK;
  if (nq < 0 || nq > FLQ_MAX) return KGE_ERR_UNSUPPORTED;
  FilterQueries Q{};
  for (int q = 0; q < nq; ++q) {
    Q.keys[q] = keys[q]; Q.num_keys[q] = num_keys[q]; Q.starts[q] = starts[q];
    Q.a[q] = a[q]; Q.b[q] = b[q]; Q.mult[q] = mult[q];
    Q.begin[q] = begin[q]; Q.end[q] = end[q];
  }
  hipLaunchKernelGGL(filter_lookup_multi_kernel, dim3((unsigned)((n + 3) / 4), (unsigned)nq), dim3(256), 0, st, Q, n);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_rank_multi(const float* scores, long long lds, long long n, long long c, const float* true_scores,
                   int K, const long long* const* begin, const long long* const* end,
                   const long long* const* col, long long col_offset, const long long* true_col, float atol,
                   float rtol, long long* rank, long long* ties, hipStream_t st) {
  if (n == 0 || c == 0) return KGE_OK;
  if (K < 0 || K > RK_MAXF) return KGE_ERR_UNSUPPORTED;
  RankFilters F{};
  F.K = K;
  for (int k = 0; k < K; ++k) {
    F.begin[k] = begin[k];
    F.end[k] = end[k];
    F.col[k] = col[k];
  }
  long long splits = (2048 + n - 1) / n;
  long long cpb = (c + splits - 1) / splits;
  if (cpb < 4096) cpb = 4096;
  cpb = (cpb + 3) & ~3LL;
  splits = (c + cpb - 1) / cpb;
  hipLaunchKernelGGL(rank_multi_kernel, dim3((unsigned)splits, (unsigned)n), dim3(RK_THREADS), 0, st, scores, lds,
                     n, c, true_scores, F, col_offset, true_col, atol, rtol, (unsigned long long*)rank,
                     (unsigned long long*)ties, cpb);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- kge_score_rank_sp_po: the filter sets as per-row column bit masks for the counting epilogue of the scoring
// kernel (score_pairs_bf16_v4.hip, V3_RANK).  One wave per (row, list); set = 1 sets the bits of the row's filter
// columns that fall into the scored slice [col_begin, col_begin + m) -- except the row's own true column, which
// is never filtered (eval_entity_ranking.py:288-290) --, set = 0 clears the words again (the buffer is all-zero
// between calls, so no pass over it ever touches more than the listed entries).
__global__ __launch_bounds__(256) void rank_bits_kernel(RankBitLists B, long long n, long long col_begin,
                                                        long long m, long long rs, long long us, int set) {
  const int q = blockIdx.y;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  rank_bits_row(B, q, i, threadIdx.x & 63, col_begin, m, rs, us, set);
}

int run_rank_bits(int lists, const long long* const* begin, const long long* const* end, const long long* const* col,
                  const Index* keep, unsigned int* const* bits, long long n, long long col_begin, long long m,
                  long long rs, long long us, int set, hipStream_t st) {
  if (n == 0 || lists == 0) return KGE_OK;
  if (lists < 0 || lists > 4) return KGE_ERR_UNSUPPORTED;
  RankBitLists B{};
  for (int q = 0; q < lists; ++q) {
    B.begin[q] = begin[q]; B.end[q] = end[q]; B.col[q] = col[q];
    B.keep[q] = keep[q]; B.bits[q] = bits[q];
  }
  hipLaunchKernelGGL(rank_bits_kernel, dim3((unsigned)((n + 3) / 4), (unsigned)lists), dim3(256), 0, st, B, n,
                     col_begin, m, rs, us, set);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- one evaluation batch in four launches (kge_eval_batch): the launch in front of the scoring and the one behind
// it.  EvalLists: per list q = side * K + k (side 0: the sp_ ranking, key s * R + p, values = objects, the row's own
// o never filtered; side 1: _po, key p * E + o, values = subjects, keep s): the filter index of set k.

// blockIdx.y < nq: one WAVE per (row, list): the 64-ary search of filter_lookup_kernel, then the wave sets the bits
// of the row's filtered columns.  blockIdx.y == nq: the target list of the true-score launch, (o | s) as int64.
__global__ __launch_bounds__(256) void eval_begin_kernel(EvalLists L, Index s, Index o, long long n, long long m,
                                                         long long rs, long long us, long long* __restrict__ tgt) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  eval_begin_row(L, s, o, n, m, rs, us, tgt, (int)blockIdx.y, i, threadIdx.x & 63);
}

// Blocks [0, nq * ceil(n / 4)): the filter bits of (row, list) cleared again (the words that were set: the buffer is
// all-zero between batches).  The rest: _get_ranks (tie policy) + hist_all of both directions (rank_hist_kernel) on
// the counters [2 (o | s)][2 (rank | ties)][M][n], which are zeroed for the next batch on the way.
__global__ __launch_bounds__(256) void eval_end_kernel(EvalLists L, long long n, long long m, long long rs, long long us, int M,
                                                       int policy, long long* __restrict__ counts,
                                                       float* __restrict__ hist, long long ldh, long long num_ent,
                                                       long long* __restrict__ ranks_o, long long* __restrict__ ranks_s) {
  const long long rb = (n + 3) / 4;
  const long long clear_blocks = (long long)L.nq * rb;
  if ((long long)blockIdx.x < clear_blocks) {
    const int q = (int)(blockIdx.x / rb);
    const long long i = ((long long)blockIdx.x % rb) * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const long long b = L.range[q][i], e = L.range[q][n + i];
    const long long keep = index_at(L.keep[q], i);
    const long long* __restrict__ col = L.values[q];
    unsigned int* row = L.bits[q] + i * rs;
    for (long long x = b + lane; x < e; x += 64) {
      const long long g = col[x];
      if (g == keep || g < 0 || g >= m) continue;
      row[(g >> 5) * us] = 0u;
    }
    return;
  }
  const long long t = ((long long)blockIdx.x - clear_blocks) * 256 + threadIdx.x;
  const long long per = (long long)M * n;
  if (t >= 2 * per) return;
  const int dir = t >= per ? 1 : 0;
  const long long u = t - dir * per;  // ranking * n + row
  long long* rank = counts + (long long)(dir * 2) * per + u;
  long long* ties = counts + (long long)(dir * 2 + 1) * per + u;
  const long long r0 = *rank, ti = *ties;
  *rank = 0;
  *ties = 0;
  const long long r = policy == 0 ? r0 + ti / 2 : (policy == 1 ? r0 : r0 + ti - 1);
  long long* ro = dir ? ranks_s : ranks_o;
  if (ro) ro[u] = r;
  if (r >= 0 && r < num_ent) unsafeAtomicAdd(hist + (u / n) * ldh + r, 1.0f);
}

int run_eval_begin(const EvalLists& L, const Index& s, const Index& o, long long n, long long m, long long rs,
                   long long us, long long* tgt, hipStream_t st) {
  if (n == 0) return KGE_OK;
  hipLaunchKernelGGL(eval_begin_kernel, dim3((unsigned)((n + 3) / 4), (unsigned)(L.nq + 1)), dim3(256), 0, st, L, s, o,
                     n, m, rs, us, tgt);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_eval_end(const EvalLists& L, long long n, long long m, long long rs, long long us, int M, int policy, long long* counts,
                 float* hist, long long ldh, long long num_ent, long long* ranks_o, long long* ranks_s, hipStream_t st) {
  if (n == 0) return KGE_OK;
  const long long blocks = (long long)L.nq * ((n + 3) / 4) + (2LL * M * n + 255) / 256;
  hipLaunchKernelGGL(eval_end_kernel, dim3((unsigned)blocks), dim3(256), 0, st, L, n, m, rs, us, M, policy, counts, hist,
                     ldh, num_ent, ranks_o, ranks_s);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_rank_hist(const long long* rank, const long long* ties, int M, long long n, int policy, float* hist,
                  long long ldh, long long num_ent, long long* ranks_out, hipStream_t st) {
  const long long total = (long long)M * n;
  if (total == 0) return KGE_OK;
  hipLaunchKernelGGL(rank_hist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rank, ties, total,
                     n, policy, hist, ldh, num_ent, ranks_out);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
