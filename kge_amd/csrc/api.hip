// api.hip -- the C ABI of libkge_amd.so (include/kge_amd.h): argument validation and
// kernel selection.  No torch, no allocation, no global state; every call enqueues on the
// caller's hipStream_t and returns a kge_status.
#include <dlfcn.h>
#include "common.hpp"
#include "bf16_queries.hpp"
#include <cstdlib>

namespace kge {
int run_spo(int scorer, int dtype, bool neg_mode, const Operand& S, const Operand& R,
            const Operand& O, int d, int dr, long long n, int slot, const void* neg,
            int neg_itype, long long neg_ld, long long K, float lp, float* out,
            long long ldo, hipStream_t st);
int run_pairs_exact(int scorer, int dtype, bool use_mfma, const Operand& A, const Operand& R,
                    const Operand& TG, int dir, int d, int dr, long long n, long long m,
                    float lp, float* out, long long ldo, hipStream_t st, bool round_query = true,
                    const RankArgs* rk = nullptr);
bool pairs_bf16_v3_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                             const Operand& TG);
int run_pairs_bf16_v3(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir,
                      int d, long long n, long long m, float* out, long long ldo, hipStream_t st,
                      unsigned long long* dbg, void* ws, long long ws_bytes);
long long pairs_bf16_v3_workspace_bytes(int d, long long n);
bool pairs_bf16_v4_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                             const Operand& TG);
int run_pairs_bf16_v4(int scorer, const Operand& A, const Operand* A2, const Operand& R,
                      const Operand& TG, int dir, int d, long long n, long long m, float* out,
                      long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                      void* ws, long long ws_bytes, int reserve_cus);
long long pairs_bf16_v4_query_bytes(int d, long long n, bool two_sided, bool split);
int run_pairs_bf16_true(int scorer, bool split, const Operand& TG, int d, long long n, const void* qf, const Index& t_sp,
                        const Index& t_po, float* true_sp, float* true_po, hipStream_t st);
int run_eval_begin_build(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                         long long n, void* qf, const EvalLists& L, const Index& s, const Index& o, long long m,
                         long long rs, long long us, long long* tgt, hipStream_t st);
int run_query_build_bits(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                         long long n, void* qf, const RankBitLists& B, int lists, long long col_begin, long long m,
                         long long rs, long long us, hipStream_t st);
int run_query_build(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                    long long n, void* qf, hipStream_t st);
int run_pairs_bf16_v4_prepared(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R,
                               const Operand& TG, int dir, int d, long long n, long long m, float* out,
                               long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                               const void* ready, void* ws, long long ws_bytes, int reserve_cus, const Operand* nA,
                               const Operand* nA2, const Operand* nR, long long nn, void* nqf);
int run_query_build_multi(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                          long long n, int nbatch, void* qf, long long qstride_bytes, hipStream_t st);
int run_pairs_bf16_v8(int scorer, bool split, const Operand& TG, bool two_sided, int d, long long n, long long m,
                      int nbatch, const void* qf, long long q_stride_bytes, float* out, long long out_stride,
                      long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg, const NextQ& nx,
                      int reserve_cus);
NextQ pairs_bf16_nextq(bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, long long n,
                       int nbatch, void* qf, long long qstride_bytes);
int run_table_max_norm(const Operand& TG, long long m, int d, float* out, hipStream_t st);
long long pairs_bf16_band_list_bytes(long long n);
int run_pairs_bf16_rescore(const Operand& TG, int d, long long n, long long m, const void* qf, const CeArgs& ce,
                           long long nlists, hipStream_t st);
int run_pairs_bf16_v8_rank(int scorer, bool split, const Operand& TG, int d, long long n, long long m, const void* qf,
                           const CeArgs& ce, hipStream_t st, unsigned long long* dbg, int reserve_cus, bool band = false,
                           long long* band_lists = nullptr);
int v8_launch_count(int which);
int run_embed2(const EmbedJob& a, const EmbedJob& b, int rowbytes, int esize, hipStream_t st);
int run_ns_bce(int kind, const float* scores, long long ld, long long n, long long c, float offset, float temp,
               float* loss_rows, float* grad, long long ldg, hipStream_t st);
int run_shard_rows(const ShardJob& a, const ShardJob& b, const ShardJob& c, int rowbytes01, int rowbytes2, int esize,
                   hipStream_t st);
int run_rank(const float* scores, long long lds, long long n, long long c,
             const float* true_scores, const long long* rowptr, const long long* lcol,
             long long col_offset, const long long* true_col, float atol, float rtol,
             long long* rank, long long* ties, hipStream_t st);
int run_filter_lookup(const long long* keys, long long num_keys, const long long* starts, const Index& a,
                      const Index& b, long long mult, long long n, long long* begin, long long* end,
                      hipStream_t st);
int run_filter_lookup_multi(int nq, const long long* const* keys, const long long* num_keys,
                            const long long* const* starts, const Index* a, const Index* b, const long long* mult,
                            long long n, long long* const* begin, long long* const* end, hipStream_t st);
int run_rank_multi(const float* scores, long long lds, long long n, long long c, const float* true_scores,
                   int K, const long long* const* begin, const long long* const* end,
                   const long long* const* col, long long col_offset, const long long* true_col, float atol,
                   float rtol, long long* rank, long long* ties, hipStream_t st);
int run_rank_hist(const long long* rank, const long long* ties, int M, long long n, int policy, float* hist,
                  long long ldh, long long num_ent, long long* ranks_out, hipStream_t st);
double run_mfma_rate(const void* rnd, int iters, float* sink, hipStream_t st);
int run_eval_begin(const EvalLists& L, const Index& s, const Index& o, long long n, long long m, long long rs, long long us,
                   long long* tgt, hipStream_t st);
int run_eval_end(const EvalLists& L, long long n, long long m, long long rs, long long us, int M, int policy, long long* counts,
                 float* hist, long long ldh, long long num_ent, long long* ranks_o, long long* ranks_s, hipStream_t st);
int run_rank_bits(int lists, const long long* const* begin, const long long* const* end, const long long* const* col,
                  const Index* keep, unsigned int* const* bits, long long n, long long col_begin, long long m,
                  long long rs, long long us, int set, hipStream_t st);
int run_pairs_bf16_v4_epi(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R,
                          const Operand& TG, int dir, int d, long long n, long long m, hipStream_t st, void* ws,
                          long long ws_bytes, const CeArgs& ce, unsigned long long* dbg);
int run_pairs_bwd_gemm16(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                         int dr, long long n, long long m, const float* gout, long long ldg, float* g_a,
                         float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st);
long long pairs_bwd_workspace_bytes(int dtype, int scorer, int d, long long n, long long m);
int run_pairs_bwd(int scorer, float lp, int dir, const Operand& A, const Operand& R,
                  const Operand& TG, int d, int dr, long long n, long long m, const float* gout,
                  long long ldg, const float* scores, long long lds, float* g_a, float* g_p,
                  float* g_tgt, hipStream_t st, bool self_contained);
int run_spo_bwd(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                int dr, long long n, const float* gout, const float* scores, float* g_s, float* g_p,
                float* g_o, hipStream_t st);
int run_neg_bwd_accum(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                      int dr, long long n, int slot, const void* neg, int neg_itype, long long neg_ld,
                      long long K, const float* gout, long long ldg, const float* scores, long long lds,
                      float* ge, long long ge_ld, float* gr, long long gr_ld, hipStream_t st);
int run_neg_order(const void* neg, int neg_itype, long long neg_ld, long long n, long long K, long long num_ent,
                  long long* cursor, long long* order, hipStream_t st);
int run_neg_bwd_accum_sorted(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d, int dr,
                             long long n, int slot, const void* neg, int neg_itype, long long neg_ld, long long K,
                             const long long* order, const float* gout, long long ldg, const float* scores,
                             long long lds, float* ge, long long ge_ld, float* gr, long long gr_ld, float* rot,
                             long long num_rel, hipStream_t st);
int run_spo_bwd_accum(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                      int dr, long long n, const float* gout, const float* scores, float* ge, long long ge_ld,
                      float* gr, long long gr_ld, hipStream_t st);
long long ce_workspace_bytes(int d, long long n, long long m);
void ce_set_stamps(unsigned long long* p);
int run_bce_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                long long m, const long long* rowptr, const long long* col, float offset, float* loss_rows, void* ws,
                long long ws_bytes, hipStream_t st, long long col_lo = 0);
int run_bce_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                long long m, const long long* rowptr, const long long* col, float offset, const float* g_rows,
                float g_scalar, float* g_a, float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st,
                long long col_lo = 0);
long long ce2_workspace_bytes(int d, long long n, long long m);
int run_ce2_fwd(int scorer, const Operand& S, const Operand& O, const Operand& R, const Operand& TG, int d,
                long long n, long long m, float* loss_rows, float* lse, void* ws, long long ws_bytes,
                hipStream_t st, float* loss_sum = nullptr, const float* scale_dev = nullptr, float scale = 1.0f,
                bool keep = false);
int run_ce2_bwd(int scorer, const Operand& S, const Operand& O, const Operand& R, const Operand& TG, int d,
                long long n, long long m, const float* lse, const float* g_rows, float g_scalar, float* g_a,
                float* g_p, float* g_tgt, float* acc_rel, long long acc_rel_rows, long long acc_rel_ld, void* ws,
                long long ws_bytes, hipStream_t st, const float* g_dev = nullptr, const float* g_dev2 = nullptr,
                bool kept = false);
int run_adagrad_multi(const kge_adagrad_seg* segs, int num, hipStream_t st);
int run_adagrad_multi_pen(const kge_adagrad_seg* segs, const kge_penalty_seg* pens, int num, hipStream_t st);
long long multilabel2_workspace_bytes(int d, long long n1, long long n2, long long m);
int run_multilabel2_bwd_accum(int scorer, int kind, float offset, const LossSide& sp, const LossSide& po,
                              const Operand& TG, int d, long long m, float* grad_ent, float* grad_rel,
                              long long rel_rows, long long rel_ld, void* ws, long long ws_bytes, hipStream_t st);
int run_adagrad(float* param, const float* grad, float* sum, long long count, float minus_clr, float weight_decay,
                float eps, unsigned short* copy16, hipStream_t st);
int run_adam(float* param, const float* grad, float* m1, float* m2, long long count, float step_size, float bc2_sqrt,
             float omb1, float beta2, float omb2, float weight_decay, float eps, unsigned short* copy16, hipStream_t st);
int run_adagrad_rows(float* param, long long param_ld, const float* grows, long long g_ld, float* sum, long long sum_ld,
                     const long long* rows, long long nrows, int dim, float minus_clr, float eps,
                     unsigned short* copy16, long long c_ld, hipStream_t st);
int run_kl_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const long long* rowptr, const long long* col, float* loss_rows, float* lse, void* ws,
               long long ws_bytes, hipStream_t st, const float* label_weight = nullptr, long long col_lo = 0);
int run_kl_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const long long* rowptr, const long long* col, const float* lse, const float* g_rows,
               float g_scalar, float* g_a, float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st,
               const float* label_weight = nullptr, const float* label_bias = nullptr, long long col_lo = 0);
void set_g16_dbg(unsigned long long* p);
void v6_set_stamps(unsigned long long* p);
bool pairs_bf16_v4_rank_launchable(int d, long long n, long long m, long long ws_bytes);
int run_debug_gemm16(int which, int lib, int d, long long rows, long long m, const unsigned short* X, long long ldx,
                     const unsigned short* G16, long long mp, float* out, float* scratch, long long scratch_bytes,
                     hipStream_t st);
bool ce_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R, const Operand& TG);
int run_ce_fwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const Index& label, float* loss_rows, float* lse, void* ws, long long ws_bytes,
               hipStream_t st);
int run_ce_bwd(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
               long long m, const Index& label, const float* lse, const float* g_rows, float g_scalar, float* g_a,
               float* g_p, float* g_tgt, void* ws, long long ws_bytes, hipStream_t st);
}  // namespace kge

using namespace kge;

namespace {

// ---- profiler ranges: KGE_ROCTX=1 puts a roctx range around every entry point of the C ABI, so
// that a `rocprofv3 --marker-trace --kernel-trace` of a LibKGE job attributes kernels and host time per call the way the
// reference's own timing buckets do per batch (kge/job/train.py:347-350, 536-557: prepare / forward / backward /
// optimizer).  The roctx library is opened on first use; without the variable (or the library) a range is two loads.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    const char* e = getenv("KGE_ROCTX");
    if (!e || e[0] != '1') return;
    // rocprofv3 records the ranges of the rocprofiler-sdk's roctx library; roctracer's libroctx64 (rocprof v1 / v2) is the
    // fallback
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1",
                             "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "libroctx64.so.4",
                             "/opt/rocm/lib/libroctx64.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static const RoctxApi& roctx_api() {
  static const RoctxApi api;
  return api;
}
struct RoctxRange {
  const bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) {
    if (on) roctx_api().push(name);
  }
  ~RoctxRange() {
    if (on) roctx_api().pop();
  }
};
#define KGE_RANGE() RoctxRange kge_roctx_range_(__func__)


int check_tables(const kge_tables* t, bool need_ptrs) {
  if (!t) return KGE_ERR_INVALID_ARG;
  if (t->scorer < KGE_COMPLEX || t->scorer > KGE_ROTATE) return KGE_ERR_INVALID_ARG;
  if (t->dtype != KGE_F32 && t->dtype != KGE_BF16) return KGE_ERR_INVALID_ARG;
  if (t->dim <= 0 || t->rel_dim <= 0 || t->dim > (1 << 20)) return KGE_ERR_INVALID_ARG;
  const bool cplx = t->scorer == KGE_COMPLEX || t->scorer == KGE_ROTATE;
  if (cplx && (t->dim % 2)) return KGE_ERR_INVALID_ARG;  // rotate.py:82-86
  if (t->scorer == KGE_ROTATE) {
    if (t->rel_dim != t->dim / 2) return KGE_ERR_INVALID_ARG;  // rotate.py:87-93
  } else if (t->rel_dim != t->dim) {
    return KGE_ERR_INVALID_ARG;
  }
  if ((t->scorer == KGE_TRANSE || t->scorer == KGE_ROTATE) && !(t->l_norm > 0.0f))
    return KGE_ERR_INVALID_ARG;
  if (need_ptrs) {
    if (!t->ent || !t->rel) return KGE_ERR_INVALID_ARG;
    if (t->num_ent <= 0 || t->num_rel <= 0) return KGE_ERR_INVALID_ARG;
    if (t->ent_ld < t->dim || t->rel_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  }
  return KGE_OK;
}

int check_index(const kge_index& ix, bool allow_null, int64_t len = 1) {
  if (ix.start != 0) return KGE_ERR_INVALID_ARG;  // (a range of targets: check_targets below, kge_score_sp / _po / _sp_po only)
  if (!ix.ptr) return (allow_null || len == 0) ? KGE_OK : KGE_ERR_INVALID_ARG;
  if (ix.itype != KGE_I32 && ix.itype != KGE_I64) return KGE_ERR_INVALID_ARG;
  if (ix.stride < 1) return KGE_ERR_INVALID_ARG;
  return KGE_OK;
}

Operand ent_op(const kge_tables* t, const kge_index& ix) {
  return Operand{t->ent, t->ent_ld, make_index(ix)};
}
// The `targets` of kge_score_sp / _po / _sp_po: a listed subset, all entities (ptr == NULL, start == 0, m == num_ent) or
// the contiguous range [start, start + m) (ptr == NULL): the reference's entity chunk, torch.arange(chunk_start,
// chunk_end) (eval_entity_ranking.py:216-229) -- rows of the table itself, streamed without an index.
int check_targets(const kge_tables* t, const kge_index& ix, int64_t m) {
  if (ix.ptr) {
    kge_index k = ix;
    if (k.start != 0) return KGE_ERR_INVALID_ARG;
    return check_index(k, true);
  }
  if (ix.start < 0 || (int64_t)ix.start + m > t->num_ent) return KGE_ERR_INVALID_ARG;
  return KGE_OK;
}
Operand tgt_op(const kge_tables* t, const kge_index& ix) {
  if (ix.ptr) return ent_op(t, ix);
  const int64_t esize = t->dtype == KGE_BF16 ? 2 : 4;
  return Operand{(const char*)t->ent + (int64_t)ix.start * t->ent_ld * esize, t->ent_ld, Index{nullptr, 1, KGE_I64}};
}
Operand rel_op(const kge_tables* t, const kge_index& ix) {
  return Operand{t->rel, t->rel_ld, make_index(ix)};
}

// One-call entry points (kge_score_sp / _po / _sp_po) at d = 512 against all entities (or a contiguous slice): a
// query_build_kernel launch + the direct-store kernel on prepared queries instead of the cooperative in-launch build
// (tools/one_call_probe.py, profiles/r4_one_call.txt).  KGE_ONE_CALL_PREPARED=0/1 forces either.
static bool one_call_prepared(const kge_tables* t, const Operand& TG, int64_t n, int64_t m, int64_t ws_bytes, bool two_sided) {
  const long long e = sw(SW_ONE_CALL_PREPARED);
  if (e == 0) return false;
  if (t->dim != 512 || TG.idx.ptr != nullptr) return false;
  if (ws_bytes < PAIRS_WS_CTRL_BYTES + pairs_bf16_v4_query_bytes((int)t->dim, n, two_sided, false)) return false;
  if (e == 1) return true;
  // measured, FB15k-237 shape: two-sided n = 512 27.3 -> 24.9 us, n = 1024 41.8 -> 40.9; one-sided equal (16.8 / 16.7);
  // n <= 128: the second launch costs more than the round trips it saves (13.1 -> 14.6)
  return two_sided && m >= 2048 && n >= 256;
}

// ---- a one-call entry with MANY rows = a group of 512-row batches of the persistent kernel ------------------------
// KgeModel.score_sp / score_po / score_sp_po with a large batch (kge_model.py:682-702, 749-789), the sub-batches of one
// training batch (kge/job/train.py:595-610) and two-step evaluation (eval_entity_ranking.py:143-229) all arrive here as
// ONE call.  At d = 512 against all entities (or a contiguous slice) rows [0, 512 L) are scored as L batches of ONE
// launch of pairs_bf16_v8_kernel (score_pairs_bf16_v8.hip: persistent grid, two consumer waves per SIMD, each XCD
// streaming its own table slice once for all L batches) behind ONE query-build launch; the n % 512 rows left go the way
// a call of that size goes.  Same fragments, same chains: the bits of the single-batch kernels (tests:
// test_gpu_queries.py::test_one_call_entry_with_many_rows_takes_the_persistent_kernel).  KGE_ONE_CALL_V8=0 switches the
// route off, KGE_ONE_CALL_V8_MIN_ROWS (default 1024) moves the threshold.
constexpr long long ONE_CALL_V8_ROWS = 512;

static long long one_call_v8_min_rows() {
  const long long v = sw(SW_ONE_CALL_V8_MIN_ROWS);
  if (v >= 2 * ONE_CALL_V8_ROWS) return v;
  return 2 * ONE_CALL_V8_ROWS;
}

static Operand operand_from(const Operand& X, long long k, int esize) {
  Operand r = X;
  if (X.idx.ptr != nullptr)
    r.idx.ptr = (const char*)X.idx.ptr + k * X.idx.stride * (X.idx.itype == KGE_I32 ? 4 : 8);
  else
    r.base = (const char*)X.base + k * X.ld * esize;  // dense rows (kge_score_emb): the rows themselves
  return r;
}

// rows [0, 512 L) of the call; *done = 512 L.  KGE_ERR_UNSUPPORTED: not this route's case (nothing was launched).
static int one_call_v8(const kge_tables* t, int dir, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG,
                       int64_t n, int64_t m, float* out, int64_t ldo, int64_t b2, void* ws, int64_t ws_bytes, hipStream_t st,
                       int64_t* done) {
  *done = 0;
  if (sw(SW_ONE_CALL_V8) == 0) return KGE_ERR_UNSUPPORTED;
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0, two = A2 != nullptr;
  const int d = (int)t->dim;
  // d = 512, or d = 256 (round 6: the persistent structure's score-store epilogue, ce_pairs_v8.hip)
  if (t->dtype != KGE_BF16 || (d != 512 && d != 256) || TG.idx.ptr != nullptr || ws == nullptr ||
      n < one_call_v8_min_rows())
    return KGE_ERR_UNSUPPORTED;
  // (d = 256: the group kernel is 2 x the single-batch kernels at the FB15k-237 shape -- 5.4-7.0 against 11-14.5 us per
  // one-sided batch -- and on a par with them on a Wikidata5M shard, 291-326 against 305-339 us, where a "group" is two
  // batches of 1.2 GB each: tools/d256_store_probe.py, profiles/r6_d256_store_policies.txt)
  if (d == 256 && m > 65536) return KGE_ERR_UNSUPPORTED;
  if (t->flags & (KGE_FLAG_EXACT | KGE_FLAG_BF16_V3)) return KGE_ERR_UNSUPPORTED;
  if (t->scorer != KGE_COMPLEX && t->scorer != KGE_DISTMULT) return KGE_ERR_UNSUPPORTED;
  if (!pairs_bf16_v4_supported(t->scorer, t->dtype, d, A, R, TG) ||
      (two && !pairs_bf16_v4_supported(t->scorer, t->dtype, d, *A2, R, TG)))
    return KGE_ERR_UNSUPPORTED;
  const long long L = n / ONE_CALL_V8_ROWS;
  const long long per = pairs_bf16_v4_query_bytes(d, ONE_CALL_V8_ROWS, two, split);
  if (L < 2 || L > (1 << 20) || ws_bytes < PAIRS_WS_CTRL_BYTES + L * per || (((uintptr_t)ws + PAIRS_WS_CTRL_BYTES) & 15))
    return KGE_ERR_UNSUPPORTED;
  void* const qf = (char*)ws + PAIRS_WS_CTRL_BYTES;
  int rc = run_query_build_multi(t->scorer, split, A, A2, R, dir, d, ONE_CALL_V8_ROWS, (int)L, qf, per, st);
  if (rc != KGE_OK) return rc;
  rc = run_pairs_bf16_v8(t->scorer, split, TG, two, d, ONE_CALL_V8_ROWS, m, (int)L, qf, per, out, ONE_CALL_V8_ROWS * ldo,
                         ldo, two ? b2 : 0, st, nullptr, NextQ{}, (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255);
  if (rc == KGE_OK) *done = L * ONE_CALL_V8_ROWS;
  return rc;  // (UNSUPPORTED behind the build launch: the fragments are simply not used)
}

// ---- the bf16 matrix-core STORE path of ComplEx / DistMult: four routes, tried in this order -----------------------
//   1  many rows against all entities: ONE persistent launch over the call's 512-row batches (one_call_v8 above:
//      pairs_bf16_v8_kernel at d = 512, pairs_bf16_v8_ce_kernel<V3_STORE> at d = 256); the n % 512 rows left re-enter
//   2  prepared queries: a query-build launch + the direct-store kernel (pairs_bf16_v6 / v7_kernel) -- split queries
//      (q = q_hi + q_lo) always, single-pass queries where one_call_prepared measured the two launches ahead
//   3  the loader/consumer kernel with its in-launch cooperative query build (pairs_bf16_v4_kernel), one- or two-sided
//   4  one-sided calls only: the single-role kernel (pairs_bf16_v3_kernel): d = 128, no workspace, more row groups or
//      fewer CUs than route 3 takes, KGE_FLAG_BF16_V3
// All four score a pair with the same products in the same K order: the same bits (tests/test_gpu_parity.py,
// test_gpu_fuzz_shapes.py, test_gpu_queries.py).  A2 != NULL: both blocks of a two-sided call (the second `b2` floats
// into a row).  KGE_ERR_UNSUPPORTED = no route applies and nothing was launched: a one-sided caller goes on to the
// exact f32 chain, a two-sided caller side by side.  (Until round 6 two more generations sat in this ladder: the
// workgroup-local-build kernel "v5" behind route 3 and the tile-per-workgroup kernel "v1" for dim % 64 == 0 behind
// route 4; what they took now runs on route 4 and on the exact chain -- DESIGN 13.)
static int bf16_store_dispatch(const kge_tables* t, int dir, const Operand& A, const Operand* A2, const Operand& R,
                               const Operand& TG, int64_t n, int64_t m, float* out, int64_t ldo, int64_t b2, void* ws,
                               int64_t ws_bytes, hipStream_t st) {
  if (t->dtype != KGE_BF16 || (t->flags & KGE_FLAG_EXACT)) return KGE_ERR_UNSUPPORTED;
  const int d = (int)t->dim;
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0, two = A2 != nullptr;
  const int reserve = (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255;
  {  // route 1
    int64_t done = 0;
    const int rc8 = one_call_v8(t, dir, A, A2, R, TG, n, m, out, ldo, b2, ws, ws_bytes, st, &done);
    if (rc8 != KGE_ERR_UNSUPPORTED) {
      if (rc8 != KGE_OK || done == n) return rc8;
      const Operand Ar = operand_from(A, done, 2), Rr = operand_from(R, done, 2);
      const Operand A2r = two ? operand_from(*A2, done, 2) : Operand{};
      float* const outr = out + done * ldo;
      const int rcr = bf16_store_dispatch(t, dir, Ar, two ? &A2r : nullptr, Rr, TG, n - done, m, outr, ldo, b2, ws, ws_bytes, st);
      if (rcr != KGE_ERR_UNSUPPORTED || !two) return rcr;
      // two-sided rows left that no route takes as one launch: side by side, HERE (the first rows are scored already)
      const int rc = bf16_store_dispatch(t, KGE_SP_, Ar, nullptr, Rr, TG, n - done, m, outr, ldo, 0, ws, ws_bytes, st);
      if (rc) return rc;
      return bf16_store_dispatch(t, KGE_PO_, A2r, nullptr, Rr, TG, n - done, m, outr + b2, ldo, 0, ws, ws_bytes, st);
    }
  }
  const bool v4_ok = ws != nullptr && !(t->flags & KGE_FLAG_BF16_V3) &&
                     pairs_bf16_v4_supported(t->scorer, t->dtype, d, A, R, TG) &&
                     (!two || pairs_bf16_v4_supported(t->scorer, t->dtype, d, *A2, R, TG));
  if (v4_ok && (split || one_call_prepared(t, TG, n, m, ws_bytes, two))) {  // route 2
    const int rcp = run_pairs_bf16_v4_prepared(t->scorer, split, A, A2, R, TG, dir, d, n, m, out, ldo, two ? b2 : 0, st,
                                               nullptr, nullptr, ws, ws_bytes, reserve, nullptr, nullptr, nullptr, 0, nullptr);
    if (rcp != KGE_ERR_UNSUPPORTED) return rcp;
  }
  // split queries stop here: f32 arithmetic on the table values, never a rounded query (the exact chain of the
  // one-sided caller keeps the query vector in f32)
  if (split) return KGE_ERR_UNSUPPORTED;
  if (v4_ok) {  // route 3
    const int rc = run_pairs_bf16_v4(t->scorer, A, A2, R, TG, dir, d, n, m, out, ldo, two ? b2 : 0, st, nullptr, ws, ws_bytes,
                                     reserve);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
  }
  if (!two && pairs_bf16_v3_supported(t->scorer, t->dtype, d, A, R, TG))  // route 4
    return run_pairs_bf16_v3(t->scorer, A, R, TG, dir, d, n, m, out, ldo, st, nullptr, ws, ws_bytes);
  return KGE_ERR_UNSUPPORTED;
}

int pairs_dispatch(const kge_tables* t, int dir, const Operand& A, const Operand& R,
                   const Operand& TG, int64_t n, int64_t m, float* out, int64_t ldo,
                   void* ws, int64_t ws_bytes, hipStream_t st) {
  const int rc = bf16_store_dispatch(t, dir, A, nullptr, R, TG, n, m, out, ldo, 0, ws, ws_bytes, st);
  if (rc != KGE_ERR_UNSUPPORTED) return rc;
  // the exact f32 chain: float32 tables, TransE / RotatE, every other dim, KGE_FLAG_EXACT (on bf16 tables the query
  // vector rounded to bf16: the bits of the single-pass semantics) and split queries that no matrix-core route took
  // (the query vector kept in f32)
  const bool keep_f32_query = (t->flags & KGE_FLAG_SPLIT_QUERY) && !(t->flags & KGE_FLAG_EXACT) && t->dtype == KGE_BF16;
  return run_pairs_exact(t->scorer, t->dtype, !(t->flags & KGE_FLAG_NO_MFMA), A, R, TG, dir, (int)t->dim, (int)t->rel_dim, n,
                         m, t->l_norm, out, ldo, st, /*round_query=*/!keep_f32_query);
}

int pairs_entry(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                kge_index targets, int64_t m, float* out, int64_t ldo, void* ws, int64_t ws_bytes,
                void* stream) {
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || m < 0 || (!out && n * m > 0) || ldo < m) return KGE_ERR_INVALID_ARG;
  if (n == 0 || m == 0) return KGE_OK;  // empty batch / empty subset: nothing to score
  if ((rc = check_index(a, false)) || (rc = check_index(p, false)) || (rc = check_targets(t, targets, m)))
    return rc;
  return pairs_dispatch(t, dir, ent_op(t, a), rel_op(t, p), tgt_op(t, targets), n, m, out, ldo,
                        ws, ws_bytes, (hipStream_t)stream);
}

}  // namespace

// ---- measurement switches (switches.hpp, include/kge_amd_debug.h) ----
namespace kge {
static long long g_switch[SW_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static_assert(SW_COUNT == 19, "g_switch's initialiser");
static const char* const g_switch_name[SW_COUNT] = {
    "ONE_CALL_PREPARED", "ONE_CALL_V8", "ONE_CALL_V8_MIN_ROWS", "V8_RANK", "RANK_FUSED_FRONT", "BWD_GEMM_LIB",
    "CE_V3", "CE_V8", "V4_OWN_BUILD", "V4_INTERLEAVE", "V4_STORE_SC1", "V6", "V7", "V7_NOSTORE", "V7_PROBE", "V8",
    "V8_VAR", "V8R_PROBE", "TRANSE_GENERIC"};
long long sw(Switch s) { return __atomic_load_n(&g_switch[(int)s], __ATOMIC_RELAXED); }
static int switch_index(const char* name) {
  if (name == nullptr) return -1;
  if (name[0] == 'K' && name[1] == 'G' && name[2] == 'E' && name[3] == '_') name += 4;  // (the old variable names)
  for (int i = 0; i < SW_COUNT; ++i) {
    const char *a = g_switch_name[i], *b = name;
    while (*a && *a == *b) ++a, ++b;
    if (*a == 0 && *b == 0) return i;
  }
  return -1;
}
}  // namespace kge

extern "C" {

int kge_abi_version(void) { return KGE_AMD_ABI_VERSION; }

const char* kge_status_string(int status) {
  switch (status) {
    case KGE_OK: return "ok";
    case KGE_ERR_INVALID_ARG: return "invalid argument";
    case KGE_ERR_UNSUPPORTED: return "unsupported configuration";
    case KGE_ERR_LAUNCH: return "HIP launch/runtime error";
    case KGE_ERR_NO_DEVICE: return "no gfx950 device";
    case KGE_ERR_WORKSPACE: return "workspace too small";
  }
  return "unknown status";
}

int kge_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int kge_score_spo(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                  float* out, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || (!out && n > 0)) return KGE_ERR_INVALID_ARG;
  if (n == 0) return KGE_OK;
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) ||
      (rc = check_index(o, false)))
    return rc;
  return run_spo(t->scorer, t->dtype, false, ent_op(t, s), rel_op(t, p), ent_op(t, o),
                 (int)t->dim, (int)t->rel_dim, n, 2, nullptr, 0, 0, 0, t->l_norm, out, 0,
                 (hipStream_t)stream);
}

int64_t kge_score_workspace_bytes(const kge_tables* t, int64_t n) {
  if (!t || n <= 0 || t->dtype != KGE_BF16) return 0;
  if (t->scorer != KGE_COMPLEX && t->scorer != KGE_DISTMULT) return 0;
  if (t->dim != 128 && t->dim != 256 && t->dim != 512) return 0;
  // bf16 query fragments of whole 128-row groups + the publication flags of the builders
  if ((t->flags & KGE_FLAG_SPLIT_QUERY) && (t->dim == 256 || t->dim == 512))  // q_hi and q_lo rows, both sides
    return PAIRS_WS_CTRL_BYTES + pairs_bf16_v4_query_bytes((int)t->dim, n, true, true);
  return pairs_bf16_v3_workspace_bytes((int)t->dim, n);
}

// ---- prepared queries (include/kge_amd.h) -----------------------------------------------------------------------
static bool queries_supported(const kge_tables* t) {
  return t->dtype == KGE_BF16 && (t->scorer == KGE_COMPLEX || t->scorer == KGE_DISTMULT) &&
         (t->dim == 256 || t->dim == 512) && !(t->flags & (KGE_FLAG_EXACT |
                                                           KGE_FLAG_BF16_V3));
}

int64_t kge_queries_bytes(const kge_tables* t, int combine, int64_t n) {
  if (!t || n <= 0 || check_tables(t, false) != KGE_OK || !queries_supported(t)) return 0;
  if (combine != KGE_SP_ && combine != KGE_PO_ && combine != KGE_SP_PO) return 0;
  return pairs_bf16_v4_query_bytes((int)t->dim, n, combine == KGE_SP_PO, (t->flags & KGE_FLAG_SPLIT_QUERY) != 0);
}

// the operands of a batch: entity rows of the first side, of the second side (SP_PO), relation rows
static int query_operands(const kge_tables* t, int combine, const kge_index& s, const kge_index& p,
                          const kge_index& o, Operand& A, Operand& A2, Operand& R, int& dir) {
  int rc;
  if (combine == KGE_SP_ || combine == KGE_SP_PO) {
    if ((rc = check_index(s, false))) return rc;
  }
  if (combine == KGE_PO_ || combine == KGE_SP_PO) {
    if ((rc = check_index(o, false))) return rc;
  }
  if ((rc = check_index(p, false))) return rc;
  dir = combine == KGE_PO_ ? KGE_PO_ : KGE_SP_;
  A = ent_op(t, combine == KGE_PO_ ? o : s);
  A2 = ent_op(t, o);
  R = rel_op(t, p);
  return KGE_OK;
}

int kge_build_queries(const kge_tables* t, int combine, kge_index s, kge_index p, kge_index o, int64_t n,
                      void* queries, int64_t queries_bytes, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (combine != KGE_SP_ && combine != KGE_PO_ && combine != KGE_SP_PO) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && !queries)) return KGE_ERR_INVALID_ARG;
  if (n == 0) return KGE_OK;
  if (!queries_supported(t)) return KGE_ERR_UNSUPPORTED;
  if (queries_bytes < kge_queries_bytes(t, combine, n)) return KGE_ERR_WORKSPACE;
  Operand A, A2, R;
  int dir;
  if ((rc = query_operands(t, combine, s, p, o, A, A2, R, dir))) return rc;
  if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, A, R, A)) return KGE_ERR_UNSUPPORTED;
  return run_query_build(t->scorer, (t->flags & KGE_FLAG_SPLIT_QUERY) != 0, A, combine == KGE_SP_PO ? &A2 : nullptr, R,
                         dir, (int)t->dim, n, queries, (hipStream_t)stream);
}

int kge_score_queries(const kge_tables* t, int combine, const void* queries, int64_t n, kge_index targets, int64_t m,
                      float* out, int64_t ldo, int64_t block2_offset, const kge_next_queries* next, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (combine != KGE_SP_ && combine != KGE_PO_ && combine != KGE_SP_PO) return KGE_ERR_INVALID_ARG;
  const int64_t b2 = combine == KGE_SP_PO ? (block2_offset > 0 ? block2_offset : m) : 0;
  const int64_t width = combine == KGE_SP_PO ? b2 + m : m;
  if (n < 0 || m < 0 || (!out && n * m > 0) || ldo < width || b2 < 0 || (combine == KGE_SP_PO && b2 < m) ||
      (n > 0 && !queries))
    return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(targets, true))) return rc;
  if (!targets.ptr && m != t->num_ent) return KGE_ERR_INVALID_ARG;
  if (!queries_supported(t)) return KGE_ERR_UNSUPPORTED;
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0;
  Operand TG = ent_op(t, targets);
  if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, TG, TG, TG)) return KGE_ERR_UNSUPPORTED;
  Operand nA{}, nA2{}, nR{};
  int ndir = KGE_SP_;
  const bool has_next = next != nullptr && next->n > 0 && next->queries != nullptr;
  if (has_next) {
    if (next->queries == queries) return KGE_ERR_INVALID_ARG;  // (this launch still reads `queries`)
    if (next->queries_bytes < kge_queries_bytes(t, combine, next->n)) return KGE_ERR_WORKSPACE;
    if ((rc = query_operands(t, combine, next->s, next->p, next->o, nA, nA2, nR, ndir))) return rc;
    if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, nA, nR, nA)) return KGE_ERR_UNSUPPORTED;
  }
  if (n == 0 || m == 0) {  // nothing to score: the next batch still wants its queries
    if (!has_next) return KGE_OK;
    return run_query_build(t->scorer, split, nA, combine == KGE_SP_PO ? &nA2 : nullptr, nR, ndir, (int)t->dim,
                           next->n, next->queries, (hipStream_t)stream);
  }
  const Operand none{t->ent, t->ent_ld, Index{nullptr, 1, 1}};  // (never read: the queries are prepared)
  return run_pairs_bf16_v4_prepared(t->scorer, split, none, combine == KGE_SP_PO ? &none : nullptr, none, TG,
                                    combine == KGE_PO_ ? KGE_PO_ : KGE_SP_, (int)t->dim, n, m, out, ldo, b2,
                                    (hipStream_t)stream, nullptr, queries, nullptr, 0,
                                    (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255, has_next ? &nA : nullptr,
                                    has_next && combine == KGE_SP_PO ? &nA2 : nullptr, has_next ? &nR : nullptr,
                                    has_next ? next->n : 0, has_next ? next->queries : nullptr);
}

// ---- groups of batches: one launch per group (score_pairs_bf16_v8.hip) ---------------------------------------------
int kge_build_queries_multi(const kge_tables* t, int combine, kge_index s, kge_index p, kge_index o, int64_t n,
                            int64_t num_batches, void* queries, int64_t queries_stride, int64_t queries_bytes,
                            void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (combine != KGE_SP_ && combine != KGE_PO_ && combine != KGE_SP_PO) return KGE_ERR_INVALID_ARG;
  if (n < 0 || num_batches < 0 || num_batches > (1 << 20) || (n * num_batches > 0 && !queries) || queries_stride < 0)
    return KGE_ERR_INVALID_ARG;
  if (n == 0 || num_batches == 0) return KGE_OK;
  if (!queries_supported(t)) return KGE_ERR_UNSUPPORTED;
  const int64_t per = kge_queries_bytes(t, combine, n);
  if ((queries_stride & 15) || queries_stride < per) return KGE_ERR_INVALID_ARG;
  if (queries_bytes < (num_batches - 1) * queries_stride + per) return KGE_ERR_WORKSPACE;
  Operand A, A2, R;
  int dir;
  if ((rc = query_operands(t, combine, s, p, o, A, A2, R, dir))) return rc;
  if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, A, R, A)) return KGE_ERR_UNSUPPORTED;
  return run_query_build_multi(t->scorer, (t->flags & KGE_FLAG_SPLIT_QUERY) != 0, A, combine == KGE_SP_PO ? &A2 : nullptr,
                               R, dir, (int)t->dim, n, (int)num_batches, queries, queries_stride, (hipStream_t)stream);
}

int kge_score_queries_multi(const kge_tables* t, int combine, const void* queries, int64_t queries_stride, int64_t n,
                            int64_t num_batches, kge_index targets, int64_t m, float* out, int64_t out_stride,
                            int64_t ldo, int64_t block2_offset, const kge_next_queries* next, int64_t next_stride,
                            void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (combine != KGE_SP_ && combine != KGE_PO_ && combine != KGE_SP_PO) return KGE_ERR_INVALID_ARG;
  const int64_t b2 = combine == KGE_SP_PO ? (block2_offset > 0 ? block2_offset : m) : 0;
  const int64_t width = combine == KGE_SP_PO ? b2 + m : m;
  if (n < 0 || m < 0 || num_batches < 0 || num_batches > (1 << 20) || ldo < width || (combine == KGE_SP_PO && b2 < m) ||
      queries_stride < 0 || out_stride < 0)
    return KGE_ERR_INVALID_ARG;
  const bool work = n > 0 && m > 0 && num_batches > 0;
  if (work && (!out || !queries)) return KGE_ERR_INVALID_ARG;
  if (work && num_batches > 1 && out_stride < (n - 1) * ldo + width) return KGE_ERR_INVALID_ARG;  // blocks overlap
  if ((rc = check_index(targets, true))) return rc;
  if (!targets.ptr && m != t->num_ent) return KGE_ERR_INVALID_ARG;
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0;
  // groups: d = 512 (pairs_bf16_v8_kernel), d = 256 (its parametric sibling's store epilogue)
  if (!queries_supported(t) || (t->dim != 512 && t->dim != 256) || targets.ptr) return KGE_ERR_UNSUPPORTED;
  const int64_t per = kge_queries_bytes(t, combine, n);
  if (work && num_batches > 1 && ((queries_stride & 15) || queries_stride < per)) return KGE_ERR_INVALID_ARG;
  Operand TG = ent_op(t, targets);
  if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, TG, TG, TG)) return KGE_ERR_UNSUPPORTED;
  NextQ nx{};
  const bool has_next = next != nullptr && next->n > 0 && next->queries != nullptr;
  if (has_next) {  // the NEXT group: num_batches batches of next->n rows, fragments next_stride bytes apart
    Operand nA, nA2, nR;
    int ndir;
    const int64_t nper = kge_queries_bytes(t, combine, next->n);
    if (next->queries == queries || (next_stride & 15) || (num_batches > 1 && next_stride < nper)) return KGE_ERR_INVALID_ARG;
    if (next->queries_bytes < (num_batches > 0 ? num_batches - 1 : 0) * next_stride + nper) return KGE_ERR_WORKSPACE;
    if ((rc = query_operands(t, combine, next->s, next->p, next->o, nA, nA2, nR, ndir))) return rc;
    if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, nA, nR, nA)) return KGE_ERR_UNSUPPORTED;
    if (!work)
      return run_query_build_multi(t->scorer, split, nA, combine == KGE_SP_PO ? &nA2 : nullptr, nR, ndir, (int)t->dim,
                                   next->n, (int)(num_batches > 0 ? num_batches : 1), next->queries, next_stride,
                                   (hipStream_t)stream);
    if (t->dim == 256 && num_batches > 1) {
      // d = 256 groups: the store kernel has no in-launch build -- the next group's fragments by a launch of their own,
      // in front of the scoring launch (independent work: the fragments live in another buffer)
      rc = run_query_build_multi(t->scorer, split, nA, combine == KGE_SP_PO ? &nA2 : nullptr, nR, ndir, (int)t->dim, next->n,
                                 (int)num_batches, next->queries, next_stride, (hipStream_t)stream);
      if (rc != KGE_OK) return rc;
    } else {
      nx = pairs_bf16_nextq(split, nA, combine == KGE_SP_PO ? &nA2 : nullptr, nR, ndir, next->n, (int)num_batches,
                            next->queries, next_stride);
    }
  }
  if (!work) return KGE_OK;
  if (num_batches == 1)  // a group of one: the single-batch entry (its kernel choice: pairs_bf16_v7 / v6 / v8)
    return kge_score_queries(t, combine, queries, n, targets, m, out, ldo, block2_offset, next, stream);
  return run_pairs_bf16_v8(t->scorer, split, TG, combine == KGE_SP_PO, (int)t->dim, n, m, (int)num_batches, queries,
                           queries_stride, out, out_stride, ldo, b2, (hipStream_t)stream, nullptr, nx,
                           (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255);
}

int kge_score_sp(const kge_tables* t, kge_index s, kge_index p, int64_t n, kge_index targets,
                 int64_t m, float* out, int64_t ldo, void* workspace, int64_t workspace_bytes,
                 void* stream) {
  KGE_RANGE();
  return pairs_entry(t, KGE_SP_, s, p, n, targets, m, out, ldo, workspace, workspace_bytes, stream);
}

int kge_score_po(const kge_tables* t, kge_index p, kge_index o, int64_t n, kge_index targets,
                 int64_t m, float* out, int64_t ldo, void* workspace, int64_t workspace_bytes,
                 void* stream) {
  KGE_RANGE();
  return pairs_entry(t, KGE_PO_, o, p, n, targets, m, out, ldo, workspace, workspace_bytes, stream);
}

int kge_score_sp_po(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                    kge_index targets, int64_t m, float* out, int64_t ldo, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  if (ldo < 2 * m) return KGE_ERR_INVALID_ARG;
  // both score blocks from ONE launch where a route of bf16_store_dispatch takes the call two-sided: the query build,
  // the kernel start-up and the launch overhead are paid once
  if (t && workspace && n > 0 && m > 0 && out && check_tables(t, true) == KGE_OK && check_index(s, false) == KGE_OK &&
      check_index(p, false) == KGE_OK && check_index(o, false) == KGE_OK &&
      check_targets(t, targets, m) == KGE_OK) {
    const Operand S = ent_op(t, s), O = ent_op(t, o), P = rel_op(t, p), TG = tgt_op(t, targets);
    const int rc2 = bf16_store_dispatch(t, KGE_SP_, S, &O, P, TG, n, m, out, ldo, m, workspace, workspace_bytes,
                                        (hipStream_t)stream);
    if (rc2 != KGE_ERR_UNSUPPORTED) return rc2;
  }
  int rc = pairs_entry(t, KGE_SP_, s, p, n, targets, m, out, ldo, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return pairs_entry(t, KGE_PO_, o, p, n, targets, m, out ? out + m : out, ldo, workspace,
                     workspace_bytes, stream);
}

int kge_score_neg(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                  int slot, const void* neg, int32_t neg_itype, int64_t neg_ld,
                  int64_t num_neg, float* out, int64_t ldo, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || num_neg < 0 || (slot != 0 && slot != 2)) return KGE_ERR_INVALID_ARG;
  if (n * num_neg > 0 && (!out || !neg)) return KGE_ERR_INVALID_ARG;
  if (neg_ld < num_neg || ldo < num_neg) return KGE_ERR_INVALID_ARG;
  if (neg_itype != KGE_I32 && neg_itype != KGE_I64) return KGE_ERR_INVALID_ARG;
  if (n == 0 || num_neg == 0) return KGE_OK;
  if (n > 65535) return KGE_ERR_UNSUPPORTED;  // grid.y; callers sub-batch far below this
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) ||
      (rc = check_index(o, false)))
    return rc;
  return run_spo(t->scorer, t->dtype, true, ent_op(t, s), rel_op(t, p), ent_op(t, o),
                 (int)t->dim, (int)t->rel_dim, n, slot, neg, neg_itype, neg_ld, num_neg,
                 t->l_norm, out, ldo, (hipStream_t)stream);
}

// ---- LookupEmbedder.embed: gathered entity rows and relation rows of a batch, one launch
int kge_embed(const kge_tables* t, kge_index ent_idx, int64_t n_ent, void* ent_out, int64_t ent_ldo,
              kge_index rel_idx, int64_t n_rel, void* rel_out, int64_t rel_ldo, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n_ent < 0 || n_rel < 0 || (n_ent > 0 && !ent_out) || (n_rel > 0 && !rel_out)) return KGE_ERR_INVALID_ARG;
  if ((n_ent > 0 && (rc = check_index(ent_idx, false))) || (n_rel > 0 && (rc = check_index(rel_idx, false))))
    return rc;
  const int es = t->dtype == KGE_BF16 ? 2 : 4;
  const long long eb = t->dim * es, rb = t->rel_dim * es;  // bytes per row
  auto ok = [&](const void* p, long long ld, long long rowb) {
    return ((uintptr_t)p & 15) == 0 && (ld * es) % 16 == 0 && rowb % 16 == 0 && rowb < (1LL << 31);
  };
  if ((n_ent && (!ok(t->ent, t->ent_ld, eb) || !ok(ent_out, ent_ldo, eb))) ||
      (n_rel && (!ok(t->rel, t->rel_ld, rb) || !ok(rel_out, rel_ldo, rb))))
    return KGE_ERR_UNSUPPORTED;
  EmbedJob a{t->ent, t->ent_ld, make_index(ent_idx), n_ent, ent_out, ent_ldo};
  EmbedJob b{t->rel, t->rel_ld, make_index(rel_idx), n_rel, rel_out, rel_ldo};
  const EmbedJob none{nullptr, 0, Index{}, 0, nullptr, 0};
  if (eb == rb || n_ent == 0 || n_rel == 0)
    return run_embed2(n_ent ? a : none, n_rel ? b : none, (int)(n_ent ? eb : rb), es, (hipStream_t)stream);
  rc = run_embed2(a, none, (int)eb, es, (hipStream_t)stream);  // RotatE: rows of different length
  return rc ? rc : run_embed2(none, b, (int)rb, es, (hipStream_t)stream);
}

int kge_ns_bce_loss(const float* scores, int64_t ld, int64_t n, int64_t c, int kind, float offset, float temperature,
                    float* loss_rows, float* grad, int64_t ldg, void* stream) {
  KGE_RANGE();
  if (n < 0 || c < 1 || kind < 0 || kind > 2 || ld < c || (grad && ldg < c)) return KGE_ERR_INVALID_ARG;
  if (n > 0 && (!scores || !loss_rows)) return KGE_ERR_INVALID_ARG;
  if (kind != 0 && c < 2) return KGE_ERR_INVALID_ARG;  // the mean / adversarial forms need a negative
  return run_ns_bce(kind, scores, ld, n, c, offset, temperature, loss_rows, grad, ldg, (hipStream_t)stream);
}

int kge_shard_gather(const kge_tables* t, int64_t lo, const kge_index* ids, int num_ids, int64_t n, void* send,
                     int64_t send_ld, kge_index rel_idx, void* rel_out, int64_t rel_ldo, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (num_ids < 1 || num_ids > 2 || !ids || n < 0 || (n > 0 && !send) || lo < 0) return KGE_ERR_INVALID_ARG;
  for (int j = 0; j < num_ids; ++j)
    if (n > 0 && (rc = check_index(ids[j], false))) return rc;
  const bool with_rel = rel_out != nullptr;
  if (with_rel && n > 0 && (rc = check_index(rel_idx, false))) return rc;
  const int es = t->dtype == KGE_BF16 ? 2 : 4;
  const long long eb = t->dim * es, rb = t->rel_dim * es;
  auto ok = [&](const void* p, long long ld, long long rowb) {
    return ((uintptr_t)p & 15) == 0 && (ld * es) % 16 == 0 && rowb % 16 == 0 && rowb < (1LL << 31);
  };
  if (!ok(t->ent, t->ent_ld, eb) || !ok(send, send_ld, eb) ||
      (with_rel && (!ok(t->rel, t->rel_ld, rb) || !ok(rel_out, rel_ldo, rb))))
    return KGE_ERR_UNSUPPORTED;
  if (n == 0) return KGE_OK;
  if (t->num_ent < 1) return KGE_ERR_INVALID_ARG;
  const ShardJob none{nullptr, 0, Index{}, 0, nullptr, 0, 0, 0, 0, 0, 0};
  ShardJob a{t->ent, t->ent_ld, make_index(ids[0]), n, send, send_ld, lo, t->num_ent - 1, 0, 0, 0};
  ShardJob b = none;
  if (num_ids == 2) {
    b = a;
    b.idx = make_index(ids[1]);
    b.out = (char*)send + n * send_ld * es;
  }
  ShardJob c = none;
  if (with_rel) c = ShardJob{t->rel, t->rel_ld, make_index(rel_idx), n, rel_out, rel_ldo, 0, t->num_rel - 1, 0, 0, 0};
  return run_shard_rows(a, b, c, (int)eb, with_rel ? (int)rb : 0, es, (hipStream_t)stream);
}

int kge_shard_pick(const void* gathered, int64_t ld, int dtype, int64_t dim, int64_t shard_rows, int world,
                   const kge_index* ids, int num_ids, int64_t n, void* rows, int64_t rows_ld, void* stream) {
  KGE_RANGE();
  if (num_ids < 1 || num_ids > 2 || !ids || n < 0 || world < 1 || shard_rows < 1 || dim < 1 ||
      (dtype != KGE_BF16 && dtype != KGE_F32) || (n > 0 && (!gathered || !rows)))
    return KGE_ERR_INVALID_ARG;
  int rc;
  for (int j = 0; j < num_ids; ++j)
    if (n > 0 && (rc = check_index(ids[j], false))) return rc;
  const int es = dtype == KGE_BF16 ? 2 : 4;
  const long long eb = dim * es;
  if (((uintptr_t)gathered & 15) || ((uintptr_t)rows & 15) || (ld * es) % 16 || (rows_ld * es) % 16 || eb % 16 ||
      eb >= (1LL << 31))
    return KGE_ERR_UNSUPPORTED;
  if (n == 0) return KGE_OK;
  const ShardJob none{nullptr, 0, Index{}, 0, nullptr, 0, 0, 0, 0, 0, 0};
  const long long kn = (long long)num_ids * n;  // rows per rank's block of the all-gather
  ShardJob a{gathered, ld, make_index(ids[0]), n, rows, rows_ld, 0, 0, shard_rows, kn, 0};
  ShardJob b = none;
  if (num_ids == 2) {
    b = a;
    b.idx = make_index(ids[1]);
    b.out = (char*)rows + n * rows_ld * es;
    b.add = n;
  }
  return run_shard_rows(a, b, none, (int)eb, 0, es, (hipStream_t)stream);
}

int kge_score_emb(const kge_tables* t, int combine, const void* s_emb, int64_t s_ld,
                  const void* p_emb, int64_t p_ld, const void* o_emb, int64_t o_ld, int64_t n,
                  int64_t m, float* out, int64_t ldo, void* workspace, int64_t workspace_bytes,
                  void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, false);
  if (rc) return rc;
  if (!s_emb || !p_emb || !o_emb || n < 0 || m < 0) return KGE_ERR_INVALID_ARG;
  if (s_ld < t->dim || o_ld < t->dim || p_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  const Index ident{nullptr, 1, KGE_I64};
  Operand S{s_emb, s_ld, ident}, P{p_emb, p_ld, ident}, O{o_emb, o_ld, ident};
  hipStream_t st = (hipStream_t)stream;
  if (combine == KGE_SPO) {
    if (!out && n > 0) return KGE_ERR_INVALID_ARG;
    return run_spo(t->scorer, t->dtype, false, S, P, O, (int)t->dim, (int)t->rel_dim, n, 2,
                   nullptr, 0, 0, 0, t->l_norm, out, 0, st);
  }
  if ((!out && n * m > 0) || ldo < m) return KGE_ERR_INVALID_ARG;
  if (combine == KGE_SP_)
    return pairs_dispatch(t, KGE_SP_, S, P, O, n, m, out, ldo, workspace, workspace_bytes, st);
  if (combine == KGE_PO_)
    return pairs_dispatch(t, KGE_PO_, O, P, S, n, m, out, ldo, workspace, workspace_bytes, st);
  return KGE_ERR_INVALID_ARG;
}

int kge_score_emb_sp_po(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb, int64_t p_ld,
                        const void* o_emb, int64_t o_ld, int64_t n, const void* tgt_emb, int64_t tgt_ld, int64_t m,
                        float* out, int64_t ldo, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  return kge_score_emb_sp_po_blocks(t, s_emb, s_ld, p_emb, p_ld, o_emb, o_ld, n, tgt_emb, tgt_ld, m, out, ldo, m, workspace,
                                    workspace_bytes, stream);
}

int kge_score_emb_sp_po_blocks(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb, int64_t p_ld,
                               const void* o_emb, int64_t o_ld, int64_t n, const void* tgt_emb, int64_t tgt_ld, int64_t m,
                               float* out, int64_t ldo, int64_t block2_offset, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, false);
  if (rc) return rc;
  if (!s_emb || !p_emb || !o_emb || !tgt_emb || n < 0 || m < 0) return KGE_ERR_INVALID_ARG;
  if (s_ld < t->dim || o_ld < t->dim || tgt_ld < t->dim || p_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  const int64_t b2 = block2_offset;
  if ((!out && n * m > 0) || b2 < m || ldo < b2 + m) return KGE_ERR_INVALID_ARG;
  const Index ident{nullptr, 1, KGE_I64};
  Operand S{s_emb, s_ld, ident}, P{p_emb, p_ld, ident}, O{o_emb, o_ld, ident}, TG{tgt_emb, tgt_ld, ident};
  hipStream_t st = (hipStream_t)stream;
  if (workspace && n > 0 && m > 0) {  // as kge_score_sp_po, on dense rows (the per-rank scoring launch of the sharded step)
    const int rc2 = bf16_store_dispatch(t, KGE_SP_, S, &O, P, TG, n, m, out, ldo, b2, workspace, workspace_bytes, st);
    if (rc2 != KGE_ERR_UNSUPPORTED) return rc2;
  }
  rc = pairs_dispatch(t, KGE_SP_, S, P, TG, n, m, out, ldo, workspace, workspace_bytes, st);
  if (rc) return rc;
  return pairs_dispatch(t, KGE_PO_, O, P, TG, n, m, out ? out + b2 : out, ldo, workspace, workspace_bytes, st);
}

int kge_rank_counts(const float* scores, int64_t lds, int64_t n, int64_t c,
                    const float* true_scores, const int64_t* lbl_rowptr,
                    const int64_t* lbl_col, int64_t col_offset, const int64_t* true_col,
                    float atol, float rtol, int64_t* rank, int64_t* ties, void* stream) {
  KGE_RANGE();
  if (n < 0 || c < 0 || lds < c) return KGE_ERR_INVALID_ARG;
  if (n * c > 0 && (!scores || !true_scores || !rank || !ties)) return KGE_ERR_INVALID_ARG;
  if (lbl_rowptr && !lbl_col) return KGE_ERR_INVALID_ARG;
  if (n > 65535) return KGE_ERR_UNSUPPORTED;
  return run_rank(scores, lds, n, c, true_scores, (const long long*)lbl_rowptr,
                  (const long long*)lbl_col, col_offset, (const long long*)true_col, atol, rtol,
                  (long long*)rank, (long long*)ties, (hipStream_t)stream);
}

int kge_filter_lookup(const int64_t* sorted_keys, int64_t num_keys, const int64_t* starts, kge_index a,
                      kge_index b, int64_t mult, int64_t n, int64_t* begin, int64_t* end, void* stream) {
  KGE_RANGE();
  if (n < 0 || num_keys < 0) return KGE_ERR_INVALID_ARG;
  if (n > 0 && (!begin || !end || (num_keys > 0 && (!sorted_keys || !starts)))) return KGE_ERR_INVALID_ARG;
  int rc;
  if ((rc = check_index(a, false, n)) || (rc = check_index(b, false, n))) return rc;
  return run_filter_lookup((const long long*)sorted_keys, num_keys, (const long long*)starts, make_index(a),
                           make_index(b), mult, n, (long long*)begin, (long long*)end, (hipStream_t)stream);
}

int kge_filter_lookup_multi(const kge_filter_query* queries, int num_queries, int64_t n, void* stream) {
  KGE_RANGE();
  if (n < 0 || num_queries < 0 || (num_queries > 0 && !queries)) return KGE_ERR_INVALID_ARG;
  if (num_queries > KGE_MAX_FILTER_QUERIES) return KGE_ERR_UNSUPPORTED;
  const long long *keys[KGE_MAX_FILTER_QUERIES], *starts[KGE_MAX_FILTER_QUERIES];
  long long nk[KGE_MAX_FILTER_QUERIES], mult[KGE_MAX_FILTER_QUERIES];
  long long *begin[KGE_MAX_FILTER_QUERIES], *end[KGE_MAX_FILTER_QUERIES];
  Index a[KGE_MAX_FILTER_QUERIES], b[KGE_MAX_FILTER_QUERIES];
  for (int q = 0; q < num_queries; ++q) {
    const kge_filter_query& x = queries[q];
    if (x.num_keys < 0) return KGE_ERR_INVALID_ARG;
    if (n > 0 && (!x.begin || !x.end || (x.num_keys > 0 && (!x.sorted_keys || !x.starts)))) return KGE_ERR_INVALID_ARG;
    int rc;
    if ((rc = check_index(x.a, false, n)) || (rc = check_index(x.b, false, n))) return rc;
    keys[q] = (const long long*)x.sorted_keys; starts[q] = (const long long*)x.starts; nk[q] = x.num_keys;
    mult[q] = x.mult; begin[q] = (long long*)x.begin; end[q] = (long long*)x.end;
    a[q] = make_index(x.a); b[q] = make_index(x.b);
  }
  return run_filter_lookup_multi(num_queries, keys, nk, starts, a, b, mult, n, begin, end, (hipStream_t)stream);
}

int kge_rank_counts_multi(const float* scores, int64_t lds, int64_t n, int64_t c, const float* true_scores,
                          int num_filters, const int64_t* const* lbl_begin, const int64_t* const* lbl_end,
                          const int64_t* const* lbl_col, int64_t col_offset, const int64_t* true_col,
                          float atol, float rtol, int64_t* rank, int64_t* ties, void* stream) {
  KGE_RANGE();
  if (n < 0 || c < 0 || lds < c || num_filters < 0) return KGE_ERR_INVALID_ARG;
  if (num_filters > KGE_MAX_FILTERS) return KGE_ERR_UNSUPPORTED;
  if (n * c > 0 && (!scores || !true_scores || !rank || !ties)) return KGE_ERR_INVALID_ARG;
  if (num_filters > 0 && n > 0) {
    if (!lbl_begin || !lbl_end || !lbl_col) return KGE_ERR_INVALID_ARG;
    for (int k = 0; k < num_filters; ++k)
      if (!lbl_begin[k] || !lbl_end[k] || !lbl_col[k]) return KGE_ERR_INVALID_ARG;
  }
  if (n > 65535) return KGE_ERR_UNSUPPORTED;
  return run_rank_multi(scores, lds, n, c, true_scores, num_filters, (const long long* const*)lbl_begin,
                        (const long long* const*)lbl_end, (const long long* const*)lbl_col, col_offset,
                        (const long long*)true_col, atol, rtol, (long long*)rank, (long long*)ties,
                        (hipStream_t)stream);
}

// ---- scoring + rank counting in one kernel (no [n, 2m] score matrix)
// The filter bits of one (side, filter set k): bit (j & 31) of the 32-bit word [k + i * rs + (j >> 5) * us] of the
// side's block says column j of the scored slice is filtered for row i.  WORD-major, the K sets of a side interleaved
// (rs = K, us = K x pitch; the pitch: n rounded up to 64 rows): the words of columns [32 u, 32 u + 32) of all rows
// and sets lie side by side.  The counting kernels hold one query row per lane and walk the columns in units of
// 32: a wave's load of its 32 rows' words (both sets: one 8-byte load) touches one or two lines; row-major (one 72 KB
// row of bits per query at the Wikidata5M shard) it touched 32 lines in 32 pages, a fifth of that kernel's time.
struct RankBitsLayout {
  int64_t rs, us, words;  // words per side (all its sets)
};
static inline RankBitsLayout rank_bits_layout(int64_t n, int64_t m, int num_filters) {
  const int64_t pitch = (n + 63) / 64 * 64, cols = (m + 63) / 64 * 2, K = num_filters;
  return RankBitsLayout{K, K * pitch, K * pitch * cols};
}

int64_t kge_score_rank_bits_bytes(int64_t n, int64_t m, int num_filters) {
  if (n <= 0 || m <= 0 || num_filters <= 0) return 0;
  return 2 * rank_bits_layout(n, m, num_filters).words * 4 + 16;  // (+ 16: the last row's 8-byte load of one set)
}

// S / O / P: the query operands (table + index, or dense rows); TG: the scored entity rows, m of them, whose global
// ids start at col_begin; keep_o / keep_s: the rows' true object / subject ids (never filtered)
static int score_rank_core(const kge_tables* t, const Operand& S, const Operand& O, const Operand& P,
                           const Operand& TG, const Index& keep_o, const Index& keep_s, int64_t n, int64_t col_begin,
                           int64_t m, const float* true_sp, const float* true_po, int num_filters,
                           const int64_t* const* sp_begin, const int64_t* const* sp_end, const int64_t* const* sp_col,
                           const int64_t* const* po_begin, const int64_t* const* po_end, const int64_t* const* po_col,
                           float atol, float rtol, int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po,
                           int64_t* ties_po, int64_t ld, void* filter_bits, int64_t filter_bits_bytes, void* workspace,
                           int64_t workspace_bytes, void* stream, int64_t true_stride = 1, bool manage_bits = true,
                           bool queries_ready = false, const kge_rank_band* band = nullptr) {
  // queries_ready (kge_eval_batch): the batch's query fragments already sit in the workspace (behind its control block)
  // manage_bits = false (kge_eval_batch): the filter bits are set already and are cleared by the caller; the lists
  // are not looked at
  if (!true_sp || !true_po || !rank_sp || !ties_sp || !rank_po || !ties_po || ld < n) return KGE_ERR_INVALID_ARG;
  if (num_filters < 0) return KGE_ERR_INVALID_ARG;
  if (num_filters > 2) return KGE_ERR_UNSUPPORTED;
  for (int k = 0; manage_bits && k < num_filters; ++k)
    if (!sp_begin || !sp_end || !sp_col || !po_begin || !po_end || !po_col || !sp_begin[k] || !sp_end[k] ||
        !sp_col[k] || !po_begin[k] || !po_end[k] || !po_col[k])
      return KGE_ERR_INVALID_ARG;
  // Which kernel counts -- always the one whose store path kge_score_sp_po would take for these tables, so that the
  // counts are those of the two-step path bit for bit:
  //   bf16 ComplEx / DistMult, d in {256, 512}, default flags: the loader/consumer kernel's counting epilogue;
  //   float32 tables (any scorer), TransE / RotatE (any dtype), bf16 under KGE_FLAG_EXACT: the exact kernels'
  //   (score_pairs.hip, score_pairs_f32.hip: rank_tile_rows), one launch per side;
  //   everything else (bf16 at other dims, split queries -- whose partial scores sit in two consumer waves):
  //   KGE_ERR_UNSUPPORTED, the caller scores and scans.
  const bool dot = t->scorer == KGE_COMPLEX || t->scorer == KGE_DISTMULT;
  const bool exact_path = t->dtype == KGE_F32 || !dot || (t->flags & KGE_FLAG_EXACT);
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0 && !exact_path;
  if (t->flags & KGE_FLAG_BF16_V3) return KGE_ERR_UNSUPPORTED;
  if ((t->flags & KGE_FLAG_SPLIT_QUERY) && exact_path) return KGE_ERR_UNSUPPORTED;
  // split queries are counted by pairs_bf16_v8_rank_kernel only (their two partial scores meet in one lane there)
  const bool v8_rank = !exact_path && sw(SW_V8_RANK) != 0 && (t->dim == 256 || t->dim == 512) && TG.idx.ptr == nullptr &&
                       workspace_bytes >= PAIRS_WS_CTRL_BYTES + pairs_bf16_v4_query_bytes((int)t->dim, n, true, split);
  if (split && !v8_rank) return KGE_ERR_UNSUPPORTED;
  // band-and-rescore: the split counts from a single-pass launch over the q_hi blocks + a small second launch over the
  // pairs it lists (pairs_bf16_v8_rank_kernel<BAND>, pairs_bf16_rescore_kernel)
  if (band != nullptr) {
    if (!split) return KGE_ERR_INVALID_ARG;  // (a single-pass count has nothing to resolve)
    if (!band->table_max_norm || !band->list || ((uintptr_t)band->list & 15) || ((uintptr_t)band->status & 3))
      return KGE_ERR_INVALID_ARG;
    if (band->list_bytes < pairs_bf16_band_list_bytes(n)) return KGE_ERR_WORKSPACE;
  }
  if (!exact_path) {
    if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, S, P, TG) ||
        !pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, O, P, TG))
      return KGE_ERR_UNSUPPORTED;
    if (!workspace || ((uintptr_t)workspace & 15)) return KGE_ERR_WORKSPACE;
  }
  const RankBitsLayout bl = rank_bits_layout(n, m, num_filters);
  if (num_filters > 0 && (!filter_bits || ((uintptr_t)filter_bits & 7) ||
                          filter_bits_bytes < kge_score_rank_bits_bytes(n, m, num_filters)))
    return KGE_ERR_WORKSPACE;
  CeArgs ce{};
  ce.rk_true[0] = true_sp;
  ce.rk_true[1] = true_po;
  ce.rk_true_stride = true_stride;
  ce.rk_rank[0] = (unsigned long long*)rank_sp;
  ce.rk_ties[0] = (unsigned long long*)ties_sp;
  ce.rk_rank[1] = (unsigned long long*)rank_po;
  ce.rk_ties[1] = (unsigned long long*)ties_po;
  ce.rk_ld = ld;
  ce.rk_atol = atol;
  ce.rk_rtol = rtol;
  ce.rk_nfilt = num_filters;
  ce.rk_bits_rs = bl.rs;
  ce.rk_bits_us = bl.us;
  if (band != nullptr) {
    ce.rk_tmax = band->table_max_norm;
    ce.rk_list = (u32x4*)band->list;
    ce.rk_list_bytes = band->list_bytes;
    ce.rk_status = band->status;
  }
  // lists: [sp side: filter sets][po side: filter sets]; the true column of the sp ranking is o, of the po ranking s
  const long long *lb[4], *le[4], *lc[4];
  Index keep[4];
  unsigned int* bits[4];
  for (int k = 0; k < num_filters; ++k) {
    for (int side = 0; side < 2; ++side) {
      const int q = side * num_filters + k;
      if (manage_bits) {
        lb[q] = (const long long*)(side ? po_begin[k] : sp_begin[k]);
        le[q] = (const long long*)(side ? po_end[k] : sp_end[k]);
        lc[q] = (const long long*)(side ? po_col[k] : sp_col[k]);
      }
      keep[q] = side ? keep_s : keep_o;
      bits[q] = (unsigned int*)filter_bits + (int64_t)side * bl.words + k;
      ce.rk_bits[side][k] = bits[q];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const int nlists = manage_bits ? 2 * num_filters : 0;
  // The persistent counting kernel takes its bits from ONE launch together with its query fragments
  // (query_build_bits_kernel) and clears every word it has read itself: two launches instead of four.
  const bool fused_front = v8_rank && !exact_path && nlists > 0 && !queries_ready && n > 0 &&
                           sw(SW_RANK_FUSED_FRONT) != 0;
  int rc = fused_front ? KGE_OK : run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 1, st);
  if (rc) return rc;
  if (exact_path) {
    for (int side = 0; side < 2 && rc == KGE_OK; ++side) {
      RankArgs rk{};
      rk.tru = ce.rk_true[side];
      rk.tru_stride = true_stride;
      rk.rank = ce.rk_rank[side];
      rk.ties = ce.rk_ties[side];
      rk.ld = ld;
      rk.atol = atol;
      rk.rtol = rtol;
      rk.nfilt = num_filters;
      rk.bits_rs = bl.rs;
      rk.bits_us = bl.us;
      for (int k = 0; k < num_filters; ++k) rk.bits[k] = ce.rk_bits[side][k];
      rc = run_pairs_exact(t->scorer, t->dtype, !(t->flags & KGE_FLAG_NO_MFMA), side ? O : S, P, TG,
                           side ? KGE_PO_ : KGE_SP_, (int)t->dim, (int)t->rel_dim, n, m, t->l_norm, nullptr, 1, st,
                           /*round_query=*/true, &rk);
    }
    const int rcb = run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 0, st);
    return rc != KGE_OK ? rc : rcb;
  }
  if (v8_rank) {
    // query fragments into the workspace (one small launch), then the persistent counting kernel: any n, no co-residency
    void* qf = (char*)workspace + PAIRS_WS_CTRL_BYTES;
    if (fused_front) {
      RankBitLists B{};
      for (int q = 0; q < nlists; ++q) {
        B.begin[q] = lb[q]; B.end[q] = le[q]; B.col[q] = lc[q];
        B.keep[q] = keep[q]; B.bits[q] = bits[q];
      }
      rc = run_query_build_bits(t->scorer, split, S, &O, P, KGE_SP_, (int)t->dim, n, qf, B, nlists, col_begin, m, bl.rs,
                                bl.us, st);
      if (rc == KGE_ERR_UNSUPPORTED) {  // (not a shape of the fused front: the two launches)
        rc = run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 1, st);
        if (rc == KGE_OK) rc = run_query_build(t->scorer, split, S, &O, P, KGE_SP_, (int)t->dim, n, qf, st);
      }
    } else if (!queries_ready) {
      rc = run_query_build(t->scorer, split, S, &O, P, KGE_SP_, (int)t->dim, n, qf, st);
    }
    // the kernel clears the filter words it reads (every word of rows < n exactly once) when the bits are this
    // call's to manage: no clearing launch behind it
    ce.rk_clear_bits = manage_bits && num_filters > 0 ? 1 : 0;
    long long band_lists = 0;
    if (rc == KGE_OK)
      rc = run_pairs_bf16_v8_rank(t->scorer, split, TG, (int)t->dim, n, m, qf, ce, st, nullptr,
                                  (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255, band != nullptr, &band_lists);
    if (rc == KGE_OK && band != nullptr)
      rc = run_pairs_bf16_rescore(TG, (int)t->dim, n, m, qf, ce, band_lists, st);
    ce.rk_clear_bits = 0;
    if (rc == KGE_OK) return KGE_OK;  // counted, bits cleared by the kernel
    if (rc != KGE_ERR_UNSUPPORTED || split) {
      const int rcb = run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 0, st);
      return rc != KGE_OK ? rc : rcb;
    }
    rc = KGE_OK;  // declined (nothing counted, bits still set): the round-3 kernel below
  }
  // one launch holds at most 32 row groups (one workgroup per CU and XCD-aligned column groups): 2,048 rows per
  // side; larger batches go through in row blocks
  const int esize = t->dtype == KGE_BF16 ? 2 : 4;
  auto rows_from = [&](const Operand& x, int64_t r0) {
    Operand y = x;
    if (y.idx.ptr == nullptr) y.base = (const char*)y.base + r0 * y.ld * esize;
    else y.idx.ptr = (const char*)y.idx.ptr + r0 * y.idx.stride * (y.idx.itype ? 8 : 4);
    return y;
  };
  constexpr int64_t BLOCK = 2048;
  for (int64_t r0 = 0; r0 < n; r0 += BLOCK)  // every block's launch geometry first: decline before anything counts
    if (!pairs_bf16_v4_rank_launchable((int)t->dim, n - r0 < BLOCK ? n - r0 : BLOCK, m, workspace_bytes)) {
      const int rcb = run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 0, st);
      return rcb != KGE_OK ? rcb : KGE_ERR_UNSUPPORTED;
    }
  for (int64_t r0 = 0; r0 < n && rc == KGE_OK; r0 += BLOCK) {
    const int64_t nb = n - r0 < BLOCK ? n - r0 : BLOCK;
    CeArgs cb = ce;
    for (int side = 0; side < 2; ++side) {
      cb.rk_true[side] += r0 * true_stride;
      cb.rk_rank[side] += r0;
      cb.rk_ties[side] += r0;
      for (int k = 0; k < num_filters; ++k) cb.rk_bits[side][k] += r0 * bl.rs;
    }
    const Operand Sb = rows_from(S, r0), Ob = rows_from(O, r0), Pb = rows_from(P, r0);
    rc = run_pairs_bf16_v4_epi(t->scorer, V3_RANK, Sb, &Ob, Pb, TG, KGE_SP_, (int)t->dim, nb, m, st, workspace,
                               workspace_bytes, cb, nullptr);
  }
  // (also after a declined launch: the bits must not outlive the call)
  const int rc2 = run_rank_bits(nlists, lb, le, lc, keep, bits, n, col_begin, m, bl.rs, bl.us, 0, st);
  return rc ? rc : rc2;
}

int kge_score_rank_sp_po(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int64_t col_begin,
                         int64_t m, const float* true_sp, const float* true_po, int num_filters,
                         const int64_t* const* sp_begin, const int64_t* const* sp_end, const int64_t* const* sp_col,
                         const int64_t* const* po_begin, const int64_t* const* po_end, const int64_t* const* po_col,
                         float atol, float rtol, int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po,
                         int64_t* ties_po, int64_t ld, void* filter_bits, int64_t filter_bits_bytes, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || m < 0 || col_begin < 0 || col_begin + m > t->num_ent) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false, n)) || (rc = check_index(p, false, n)) || (rc = check_index(o, false, n))) return rc;
  if (n == 0 || m == 0) return KGE_OK;
  const kge_index all = {nullptr, 0, 0, 1};
  Operand S = ent_op(t, s), O = ent_op(t, o), P = rel_op(t, p), TG = ent_op(t, all);
  TG.base = (const char*)TG.base + col_begin * TG.ld * (t->dtype == KGE_BF16 ? 2 : 4);
  return score_rank_core(t, S, O, P, TG, make_index(o), make_index(s), n, col_begin, m, true_sp, true_po, num_filters,
                         sp_begin, sp_end, sp_col, po_begin, po_end, po_col, atol, rtol, rank_sp, ties_sp, rank_po,
                         ties_po, ld, filter_bits, filter_bits_bytes, workspace, workspace_bytes, stream);
}

int kge_score_rank_sp_po_band(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int64_t col_begin,
                              int64_t m, const float* true_sp, const float* true_po, int num_filters,
                              const int64_t* const* sp_begin, const int64_t* const* sp_end,
                              const int64_t* const* sp_col, const int64_t* const* po_begin,
                              const int64_t* const* po_end, const int64_t* const* po_col, float atol, float rtol,
                              int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po, int64_t* ties_po, int64_t ld,
                              void* filter_bits, int64_t filter_bits_bytes, void* workspace, int64_t workspace_bytes,
                              void* stream, const kge_rank_band* band) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || m < 0 || col_begin < 0 || col_begin + m > t->num_ent) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false, n)) || (rc = check_index(p, false, n)) || (rc = check_index(o, false, n))) return rc;
  if (n == 0 || m == 0) return KGE_OK;
  const kge_index all = {nullptr, 0, 0, 1};
  Operand S = ent_op(t, s), O = ent_op(t, o), P = rel_op(t, p), TG = ent_op(t, all);
  TG.base = (const char*)TG.base + col_begin * TG.ld * (t->dtype == KGE_BF16 ? 2 : 4);
  return score_rank_core(t, S, O, P, TG, make_index(o), make_index(s), n, col_begin, m, true_sp, true_po, num_filters,
                         sp_begin, sp_end, sp_col, po_begin, po_end, po_col, atol, rtol, rank_sp, ties_sp, rank_po,
                         ties_po, ld, filter_bits, filter_bits_bytes, workspace, workspace_bytes, stream, 1, true, false,
                         band);
}

// 1: this library was built with -DKGE_STALL_INJECT (random sleeps in front of the barriers and LDS-DMA pieces of the
// hand-synchronised kernels: tools/gpu_stall_inject.sh); 0: the product build
int kge_debug_stall_build() {
#ifdef KGE_STALL_INJECT
  return 1;
#else
  return 0;
#endif
}

int64_t kge_rank_band_list_bytes(int64_t n) { return pairs_bf16_band_list_bytes(n); }

int kge_table_max_row_norm(const kge_tables* t, int64_t row_begin, int64_t m, float* out, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (!out || row_begin < 0 || m < 0 || row_begin + m > t->num_ent) return KGE_ERR_INVALID_ARG;
  if (t->dtype != KGE_BF16) return KGE_ERR_UNSUPPORTED;
  const kge_index all = {nullptr, 0, 0, 1};
  Operand TG = ent_op(t, all);
  TG.base = (const char*)TG.base + row_begin * TG.ld * 2;
  return run_table_max_norm(TG, m, (int)t->dim, out, (hipStream_t)stream);
}

int kge_score_rank_emb_sp_po(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb, int64_t p_ld,
                             const void* o_emb, int64_t o_ld, kge_index s_ids, kge_index o_ids, int64_t n,
                             const void* tgt_emb, int64_t tgt_ld, int64_t col_begin, int64_t m, const float* true_sp,
                             const float* true_po, int num_filters, const int64_t* const* sp_begin,
                             const int64_t* const* sp_end, const int64_t* const* sp_col,
                             const int64_t* const* po_begin, const int64_t* const* po_end,
                             const int64_t* const* po_col, float atol, float rtol, int64_t* rank_sp, int64_t* ties_sp,
                             int64_t* rank_po, int64_t* ties_po, int64_t ld, void* filter_bits,
                             int64_t filter_bits_bytes, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, false);
  if (rc) return rc;
  if (n < 0 || m < 0 || col_begin < 0) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s_ids, false, n)) || (rc = check_index(o_ids, false, n))) return rc;
  if (n == 0 || m == 0) return KGE_OK;
  if (!s_emb || !p_emb || !o_emb || !tgt_emb) return KGE_ERR_INVALID_ARG;
  if (s_ld < t->dim || o_ld < t->dim || tgt_ld < t->dim || p_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  const Index ident{nullptr, 1, KGE_I64};
  Operand S{s_emb, s_ld, ident}, P{p_emb, p_ld, ident}, O{o_emb, o_ld, ident}, TG{tgt_emb, tgt_ld, ident};
  return score_rank_core(t, S, O, P, TG, make_index(o_ids), make_index(s_ids), n, col_begin, m, true_sp, true_po,
                         num_filters, sp_begin, sp_end, sp_col, po_begin, po_end, po_col, atol, rtol, rank_sp, ties_sp,
                         rank_po, ties_po, ld, filter_bits, filter_bits_bytes, workspace, workspace_bytes, stream);
}

// ---- one evaluation batch in four launches
static void eval_scratch_layout(int64_t n, int K, int64_t& off_tgt, int64_t& off_true, int64_t& total) {
  auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
  off_tgt = up((int64_t)2 * K * 2 * n * 8);  // behind the ranges
  off_true = off_tgt + up(2 * n * 8);
  total = off_true + up(n * 4 * n * 4);
}

int64_t kge_eval_batch_scratch_bytes(const kge_tables* t, int64_t n, int num_filters) {
  if (!t || n <= 0 || num_filters < 0 || num_filters > 2) return 0;
  int64_t a, b, total;
  eval_scratch_layout(n, num_filters, a, b, total);
  return total;
}

int kge_eval_batch(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int num_filters,
                   const kge_eval_filter* filters, float atol, float rtol, int tie_policy, int64_t* counts,
                   float* hist, int64_t ldh, int64_t* ranks_o, int64_t* ranks_s, void* filter_bits,
                   int64_t filter_bits_bytes, void* scratch, int64_t scratch_bytes, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  return kge_eval_batch_band(t, s, p, o, n, num_filters, filters, atol, rtol, tie_policy, counts, hist, ldh, ranks_o,
                             ranks_s, filter_bits, filter_bits_bytes, scratch, scratch_bytes, workspace,
                             workspace_bytes, stream, nullptr);
}

int kge_eval_batch_band(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int num_filters,
                        const kge_eval_filter* filters, float atol, float rtol, int tie_policy, int64_t* counts,
                        float* hist, int64_t ldh, int64_t* ranks_o, int64_t* ranks_s, void* filter_bits,
                        int64_t filter_bits_bytes, void* scratch, int64_t scratch_bytes, void* workspace,
                        int64_t workspace_bytes, void* stream, const kge_rank_band* band) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (n < 0 || num_filters < 0 || ldh < t->num_ent) return KGE_ERR_INVALID_ARG;
  if (num_filters > 2) return KGE_ERR_UNSUPPORTED;
  if (tie_policy < KGE_TIES_ROUNDED_MEAN || tie_policy > KGE_TIES_WORST) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false, n)) || (rc = check_index(p, false, n)) || (rc = check_index(o, false, n))) return rc;
  if (n == 0) return KGE_OK;
  if (!counts || !hist || (num_filters > 0 && !filters)) return KGE_ERR_INVALID_ARG;
  const int64_t E = t->num_ent, R = t->num_rel, m = E;
  const RankBitsLayout bl = rank_bits_layout(n, m, num_filters);
  int64_t off_tgt, off_true, total;
  eval_scratch_layout(n, num_filters, off_tgt, off_true, total);
  if (!scratch || ((uintptr_t)scratch & 15) || scratch_bytes < total) return KGE_ERR_WORKSPACE;
  const int64_t bits_need = kge_score_rank_bits_bytes(n, m, num_filters);
  if (num_filters > 0 && (!filter_bits || ((uintptr_t)filter_bits & 7) || filter_bits_bytes < bits_need))
    return KGE_ERR_WORKSPACE;
  char* sc = (char*)scratch;
  long long* tgt = (long long*)(sc + off_tgt);
  float* trueblk = (float*)(sc + off_true);
  EvalLists L{};
  L.nq = 2 * num_filters;
  const Index si = make_index(s), pi = make_index(p), oi = make_index(o);
  for (int k = 0; k < num_filters; ++k)
    for (int side = 0; side < 2; ++side) {
      const int q = side * num_filters + k;
      const kge_eval_filter& f = filters[k];
      const int64_t nk = side ? f.po_num_keys : f.sp_num_keys;
      const int64_t *keys = side ? f.po_keys : f.sp_keys, *starts = side ? f.po_starts : f.sp_starts,
                    *values = side ? f.po_values : f.sp_values;
      if (nk < 0 || (nk > 0 && (!keys || !starts || !values))) return KGE_ERR_INVALID_ARG;
      L.keys[q] = (const long long*)keys;
      L.num_keys[q] = nk;
      L.starts[q] = (const long long*)starts;
      L.values[q] = (const long long*)values;
      L.mult[q] = side ? E : R;        // sp: key s * R + p;  po: key p * E + o
      L.a[q] = side ? pi : si;
      L.b[q] = side ? oi : pi;
      L.keep[q] = side ? si : oi;      // the row's own true column is never filtered
      L.range[q] = (long long*)sc + (int64_t)q * 2 * n;
      L.bits[q] = (unsigned int*)filter_bits + (int64_t)side * bl.words + k;
    }
  hipStream_t st = (hipStream_t)stream;
  // (2) the true scores: the batch against its own targets, [n, 4 n] = (sp_ vs o | s, _po vs o | s); the diagonals
  // (i, i) and (i, 3 n + i) are elements of the score matrix bit for bit (each score is its own chain)
  kge_index tgi{tgt, KGE_I64, 0, 1};
  kge_index all{nullptr, KGE_I64, 0, 1};
  const Operand S = ent_op(t, s), O = ent_op(t, o), P = rel_op(t, p), TG = ent_op(t, all);
  // bf16 ComplEx / DistMult at dim 256 / 512: the batch's query fragments are built ONCE (one small launch) and serve
  // both the true scores and the counting launch (pairs_bf16_v8_rank_kernel), which then start on prepared queries
  const bool dot = t->scorer == KGE_COMPLEX || t->scorer == KGE_DISTMULT;
  const bool split = (t->flags & KGE_FLAG_SPLIT_QUERY) != 0;
  bool ready = t->dtype == KGE_BF16 && dot && !(t->flags & (KGE_FLAG_EXACT | KGE_FLAG_BF16_V3)) &&
               (t->dim == 256 || t->dim == 512) && sw(SW_V8_RANK) != 0 && workspace && !((uintptr_t)workspace & 15) &&
               workspace_bytes >= PAIRS_WS_CTRL_BYTES + pairs_bf16_v4_query_bytes((int)t->dim, n, true, split) &&
               pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, S, P, TG) &&
               pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, O, P, TG);
  // (1) filter lookup + filter bits + the target list (o | s) -- with the query fragments of a bf16 batch built by
  // spare blocks of the same launch (eval_begin_build_kernel)
  bool built = false;
  if (ready) {
    rc = run_eval_begin_build(t->scorer, split, S, &O, P, KGE_SP_, (int)t->dim, n, (char*)workspace + PAIRS_WS_CTRL_BYTES, L,
                              si, oi, m, bl.rs, bl.us, tgt, st);
    built = rc == KGE_OK;
    if (rc != KGE_OK && rc != KGE_ERR_UNSUPPORTED) return rc;
  }
  if (!built && (rc = run_eval_begin(L, si, oi, n, m, bl.rs, bl.us, tgt, st))) return rc;
  // true_sp[i] at trueblk[i * tstride], true_po[i] at true_po0[i * tstride]
  const float* true_po0 = trueblk + 3 * n;
  int64_t tstride = 4 * n + 1;
  if (ready) {
    void* qf = (char*)workspace + PAIRS_WS_CTRL_BYTES;
    rc = built ? KGE_OK : run_query_build(t->scorer, split, S, &O, P, KGE_SP_, (int)t->dim, n, qf, st);
    if (rc == KGE_OK) {
      // one chain per triple from the prepared fragments (pairs_bf16_true_kernel): [n] + [n] floats, no score block
      rc = run_pairs_bf16_true(t->scorer, split, TG, (int)t->dim, n, qf, oi, si, trueblk, trueblk + n, st);
      if (rc == KGE_OK) {
        true_po0 = trueblk + n;
        tstride = 1;
      } else if (rc == KGE_ERR_UNSUPPORTED) {  // the batch against its listed targets with a scoring launch
        const Operand TL = ent_op(t, tgi);
        rc = run_pairs_bf16_v4_prepared(t->scorer, split, S, &O, P, TL, KGE_SP_, (int)t->dim, n, 2 * n, trueblk, 4 * n,
                                        2 * n, st, nullptr, qf, workspace, workspace_bytes, 0, nullptr, nullptr, nullptr, 0,
                                        nullptr);
        if (rc == KGE_ERR_UNSUPPORTED) {  // (a launch the loader/consumer kernel declines: the one-call path)
          ready = false;
          rc = kge_score_sp_po(t, s, p, o, n, tgi, 2 * n, trueblk, 4 * n, workspace, workspace_bytes, stream);
        }
      }
    }
  } else {
    rc = kge_score_sp_po(t, s, p, o, n, tgi, 2 * n, trueblk, 4 * n, workspace, workspace_bytes, stream);
  }
  // (3) scores + counts against all entities, no score matrix
  const int M = num_filters + 1;
  const int64_t per = (int64_t)M * n;
  if (rc == KGE_OK)
    rc = score_rank_core(t, S, O, P, TG, oi, si, n, 0, m, trueblk, true_po0, num_filters, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, atol, rtol, counts, counts + per, counts + 2 * per,
                         counts + 3 * per, n, filter_bits, filter_bits_bytes, workspace, workspace_bytes, stream,
                         tstride, /*manage_bits=*/false, /*queries_ready=*/ready, band);
  // (4) bits cleared, tie policy + histograms, counters back to zero.  Behind a declined step (3) (counters untouched)
  // and behind ANY failure of steps (2) / (3): the bits are cleared and the counters zeroed all the same -- the buffers
  // are persistent and the next batch relies on finding them all-zero (advisor, round 3)
  if (rc != KGE_OK) {
    EvalLists Lc = L;  // clear only
    const int rc2 = run_eval_end(Lc, n, m, bl.rs, bl.us, 0, tie_policy, (long long*)counts, hist, ldh, E, nullptr, nullptr, st);
    if (rc != KGE_ERR_UNSUPPORTED)  // a failed launch may have left partial counts behind
      (void)kge::fill_words_async(counts, 0, (size_t)(4 * per) * sizeof(int64_t), st);
    return rc == KGE_ERR_UNSUPPORTED && rc2 != KGE_OK ? rc2 : rc;
  }
  return run_eval_end(L, n, m, bl.rs, bl.us, M, tie_policy, (long long*)counts, hist, ldh, E, (long long*)ranks_o,
                      (long long*)ranks_s, st);
}

int kge_rank_hist(const int64_t* rank, const int64_t* ties, int num_rankings, int64_t n, int tie_policy,
                  float* hist, int64_t ldh, int64_t num_ent, int64_t* ranks_out, void* stream) {
  KGE_RANGE();
  if (num_rankings < 0 || n < 0 || num_ent < 0 || ldh < num_ent) return KGE_ERR_INVALID_ARG;
  if (tie_policy < KGE_TIES_ROUNDED_MEAN || tie_policy > KGE_TIES_WORST) return KGE_ERR_INVALID_ARG;
  if ((int64_t)num_rankings * n > 0 && (!rank || !ties || !hist)) return KGE_ERR_INVALID_ARG;
  return run_rank_hist((const long long*)rank, (const long long*)ties, num_rankings, n, tie_policy, hist, ldh,
                       num_ent, (long long*)ranks_out, (hipStream_t)stream);
}

int64_t kge_score_bwd_workspace_bytes(const kge_tables* t, int64_t n, int64_t m) {
  if (!t || n <= 0 || m <= 0) return 0;
  return pairs_bwd_workspace_bytes(t->dtype, t->scorer, (int)t->dim, n, m);
}

int kge_score_pairs_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                        kge_index targets, int64_t m, const float* gout, int64_t ldg,
                        const float* scores, int64_t lds, float* g_a, float* g_p, float* g_tgt,
                        void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (t->dtype == KGE_BF16) {  // mixed precision: ComplEx / DistMult on the bf16 matrix cores
    if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
    if (n < 0 || m < 0 || ldg < m) return KGE_ERR_INVALID_ARG;
    if (n * m > 0 && (!gout || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
    if ((rc = check_index(a, false)) || (rc = check_index(p, false)) || (rc = check_index(targets, true)))
      return rc;
    if (!targets.ptr && m != t->num_ent) return KGE_ERR_INVALID_ARG;
    return run_pairs_bwd_gemm16(t->scorer, dir, ent_op(t, a), rel_op(t, p), ent_op(t, targets), (int)t->dim,
                                (int)t->rel_dim, n, m, gout, ldg, g_a, g_p, g_tgt, workspace, workspace_bytes,
                                (hipStream_t)stream);
  }
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || m < 0 || ldg < m || (scores && lds < m)) return KGE_ERR_INVALID_ARG;
  if (n * m > 0 && (!gout || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false)) || (rc = check_index(p, false)) ||
      (rc = check_index(targets, true)))
    return rc;
  if (!targets.ptr && m != t->num_ent) return KGE_ERR_INVALID_ARG;
  if (n > 65535LL * 64 || m > 65535LL * 64) return KGE_ERR_UNSUPPORTED;
  return run_pairs_bwd(t->scorer, t->l_norm, dir, ent_op(t, a), rel_op(t, p), ent_op(t, targets),
                       (int)t->dim, (int)t->rel_dim, n, m, gout, ldg, scores, lds, g_a, g_p, g_tgt,
                       (hipStream_t)stream, (t->flags & KGE_FLAG_EXACT) != 0);
}

// ---- fused 1vsAll loss (ce_loss.hip) ------------------------------------------------------
namespace {
int ce_check(const kge_tables* t, int dir, const kge_index& a, const kge_index& p, const kge_index& label,
             int64_t n) {
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n)) || (rc = check_index(label, false, n)))
    return rc;
  const kge_index all = {nullptr, 0, 0, 1};
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, all)))
    return KGE_ERR_UNSUPPORTED;
  return KGE_OK;
}
}  // namespace

int64_t kge_ce_workspace_bytes(const kge_tables* t, int64_t n) {
  if (check_tables(t, false) != KGE_OK || n <= 0 || t->num_ent <= 0) return 0;
  if (t->dtype != KGE_BF16 || (t->scorer != KGE_COMPLEX && t->scorer != KGE_DISTMULT)) return 0;
  if (t->dim != 128 && t->dim != 256 && t->dim != 512) return 0;
  return ce_workspace_bytes((int)t->dim, n, t->num_ent);
}

int kge_ce_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, kge_index label, int64_t n,
               float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const int rc = ce_check(t, dir, a, p, label, n);
  if (rc) return rc;
  if (n > 0 && (!loss_rows || !lse)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce_fwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, all), dir, (int)t->dim, n, t->num_ent,
                    make_index(label), loss_rows, lse, workspace, workspace_bytes, (hipStream_t)stream);
}

int kge_ce_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, kge_index label, int64_t n,
               const float* lse, const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt,
               void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const int rc = ce_check(t, dir, a, p, label, n);
  if (rc) return rc;
  if (n > 0 && (!lse || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce_bwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, all), dir, (int)t->dim, n, t->num_ent,
                    make_index(label), lse, g_rows, g_scalar, g_a, g_p, g_tgt, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

// ---- the same with DENSE query rows (entity-sharded training: the query rows of a batch come out of
// an exchange between the shards, the targets are this rank's rows, the label is a local row id or
// none) -------------------------------------------------------------------------------------------
namespace {
int ce_emb_check(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                 const kge_index& label, int64_t n, Operand& A, Operand& R, Operand& TG) {
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!a_rows || !p_rows)) || a_ld < t->dim || p_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(label, true, n))) return rc;
  const Index ident{nullptr, 1, KGE_I64};
  A = Operand{a_rows, a_ld, ident};
  R = Operand{p_rows, p_ld, ident};
  TG = Operand{t->ent, t->ent_ld, ident};
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, A, R, TG)) return KGE_ERR_UNSUPPORTED;
  return KGE_OK;
}
}  // namespace

int kge_ce_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                   kge_index label, int64_t n, float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes,
                   void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, label, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!loss_rows || !lse)) return KGE_ERR_INVALID_ARG;
  return run_ce_fwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, make_index(label), loss_rows, lse, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

int kge_ce_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                   kge_index label, int64_t n, const float* lse, const float* g_rows, float g_scalar, float* g_a,
                   float* g_p, float* g_tgt, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, label, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!lse || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  return run_ce_bwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, make_index(label), lse, g_rows, g_scalar, g_a,
                    g_p, g_tgt, workspace, workspace_bytes, (hipStream_t)stream);
}

// KvsAll losses on dense query rows against this rank's shard (entity-sharded training): label columns are GLOBAL
// entity ids, the shard's rows being the ids [col_lo, col_lo + t->num_ent); labels of other shards are skipped.
int kge_kl_weighted_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                            int64_t p_ld, int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo,
                            const float* label_weight, float* loss_rows, float* lse, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const kge_index none = {nullptr, 0, 0, 1};
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, none, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!lbl_rowptr || !lbl_col || !label_weight || !loss_rows || !lse)) return KGE_ERR_INVALID_ARG;
  return run_kl_fwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, (const long long*)lbl_rowptr,
                    (const long long*)lbl_col, loss_rows, lse, workspace, workspace_bytes, (hipStream_t)stream,
                    label_weight, col_lo);
}

int kge_kl_weighted_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                            int64_t p_ld, int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo,
                            const float* label_weight, const float* label_bias, const float* lse, const float* g_rows,
                            float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const kge_index none = {nullptr, 0, 0, 1};
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, none, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!lbl_rowptr || !lbl_col || !label_weight || !lse || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  return run_kl_bwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, (const long long*)lbl_rowptr,
                    (const long long*)lbl_col, lse, g_rows, g_scalar, g_a, g_p, g_tgt, workspace, workspace_bytes,
                    (hipStream_t)stream, label_weight, label_bias, col_lo);
}

int kge_bce_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                    int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo, float offset,
                    float* loss_rows, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const kge_index none = {nullptr, 0, 0, 1};
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, none, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!lbl_rowptr || !lbl_col || !loss_rows)) return KGE_ERR_INVALID_ARG;
  return run_bce_fwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, (const long long*)lbl_rowptr,
                     (const long long*)lbl_col, offset, loss_rows, workspace, workspace_bytes, (hipStream_t)stream,
                     col_lo);
}

int kge_bce_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                    int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo, float offset,
                    const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  Operand A, R, TG;
  const kge_index none = {nullptr, 0, 0, 1};
  const int rc = ce_emb_check(t, dir, a_rows, a_ld, p_rows, p_ld, none, n, A, R, TG);
  if (rc) return rc;
  if (n > 0 && (!lbl_rowptr || !lbl_col || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  return run_bce_bwd(t->scorer, A, R, TG, dir, (int)t->dim, n, t->num_ent, (const long long*)lbl_rowptr,
                     (const long long*)lbl_col, offset, g_rows, g_scalar, g_a, g_p, g_tgt, workspace, workspace_bytes,
                     (hipStream_t)stream, col_lo);
}

int64_t kge_ce_sp_po_workspace_bytes(const kge_tables* t, int64_t n) {
  if (kge_ce_workspace_bytes(t, n) <= 0) return 0;
  return ce2_workspace_bytes((int)t->dim, n, t->num_ent);
}

int kge_ce_sp_po_fwd(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, float* loss_rows,
                     float* lse, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = ce_check(t, KGE_SP_, s, p, o, n);
  if (rc) return rc;
  if (n > 0 && (!loss_rows || !lse)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce2_fwd(t->scorer, ent_op(t, s), ent_op(t, o), rel_op(t, p), ent_op(t, all), (int)t->dim, n,
                     t->num_ent, loss_rows, lse, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr, 1.0f,
                     (t->flags & KGE_FLAG_CE_KEEP_QUERIES) != 0);
}

int kge_ce_sp_po_bwd(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, const float* lse,
                     const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                     int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = ce_check(t, KGE_SP_, s, p, o, n);
  if (rc) return rc;
  if (n > 0 && (!lse || !g_a || !g_p || !g_tgt)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce2_bwd(t->scorer, ent_op(t, s), ent_op(t, o), rel_op(t, p), ent_op(t, all), (int)t->dim, n,
                     t->num_ent, lse, g_rows, g_scalar, g_a, g_p, g_tgt, nullptr, 0, 0, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

int kge_ce_sp_po_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, const float* lse,
                           const float* g_rows, float g_scalar, float* grad_ent, float* grad_rel, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = ce_check(t, KGE_SP_, s, p, o, n);
  if (rc) return rc;
  if (!grad_ent || !grad_rel || (n > 0 && !lse)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce2_bwd(t->scorer, ent_op(t, s), ent_op(t, o), rel_op(t, p), ent_op(t, all), (int)t->dim, n,
                     t->num_ent, lse, g_rows, g_scalar, nullptr, nullptr, grad_ent, grad_rel, t->num_rel,
                     t->rel_dim, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr,
                     (t->flags & KGE_FLAG_CE_KEEP_QUERIES) != 0);
}

int kge_ce_sp_po_fwd_sum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, float* loss_rows,
                         float* lse, const float* scale_dev, float scale, float* loss_sum, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  int rc = ce_check(t, KGE_SP_, s, p, o, n);
  if (rc) return rc;
  if (!loss_sum || (n > 0 && (!loss_rows || !lse))) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce2_fwd(t->scorer, ent_op(t, s), ent_op(t, o), rel_op(t, p), ent_op(t, all), (int)t->dim, n,
                     t->num_ent, loss_rows, lse, workspace, workspace_bytes, (hipStream_t)stream, loss_sum, scale_dev,
                     scale, (t->flags & KGE_FLAG_CE_KEEP_QUERIES) != 0);
}

int kge_ce_sp_po_bwd_accum_sum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                               const float* lse, const float* g_dev, const float* scale_dev, float scale,
                               float* grad_ent, float* grad_rel, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  KGE_RANGE();
  int rc = ce_check(t, KGE_SP_, s, p, o, n);
  if (rc) return rc;
  if (!grad_ent || !grad_rel || (n > 0 && !lse)) return KGE_ERR_INVALID_ARG;
  const kge_index all = {nullptr, 0, 0, 1};
  return run_ce2_bwd(t->scorer, ent_op(t, s), ent_op(t, o), rel_op(t, p), ent_op(t, all), (int)t->dim, n,
                     t->num_ent, lse, nullptr, scale, nullptr, nullptr, grad_ent, grad_rel, t->num_rel, t->rel_dim,
                     workspace, workspace_bytes, (hipStream_t)stream, g_dev, scale_dev,
                     (t->flags & KGE_FLAG_CE_KEEP_QUERIES) != 0);
}

int kge_kl_weighted_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                        const int64_t* lbl_rowptr, const int64_t* lbl_col, const float* label_weight,
                        float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !label_weight || !loss_rows || !lse))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_kl_fwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                    (const long long*)lbl_rowptr, (const long long*)lbl_col, loss_rows, lse, workspace,
                    workspace_bytes, (hipStream_t)stream, label_weight);
}

int kge_kl_weighted_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                        const int64_t* lbl_rowptr, const int64_t* lbl_col, const float* label_weight,
                        const float* label_bias, const float* lse, const float* g_rows, float g_scalar, float* g_a, float* g_p,
                        float* g_tgt, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !label_weight || !lse || !g_a || !g_p || !g_tgt)))
    return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_kl_bwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                    (const long long*)lbl_rowptr, (const long long*)lbl_col, lse, g_rows, g_scalar, g_a, g_p, g_tgt,
                    workspace, workspace_bytes, (hipStream_t)stream, label_weight, label_bias);
}

int64_t kge_multilabel2_workspace_bytes(const kge_tables* t, int64_t n_sp, int64_t n_po) {
  if (n_sp < 0 || n_po < 0 || kge_ce_workspace_bytes(t, n_sp > n_po ? n_sp : n_po) <= 0) return 0;
  return multilabel2_workspace_bytes((int)t->dim, n_sp, n_po, t->num_ent);
}

int kge_multilabel2_bwd_accum(const kge_tables* t, int loss, float offset, const kge_label_queries* sp,
                              const kge_label_queries* po, float* grad_ent, float* grad_rel, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if ((loss != KGE_LOSS_KL && loss != KGE_LOSS_BCE) || !sp || !po || !grad_ent || !grad_rel) return KGE_ERR_INVALID_ARG;
  LossSide side[2];
  for (int k = 0; k < 2; ++k) {
    const kge_label_queries& q = k ? *po : *sp;
    if (q.n < 0 || (q.n > 0 && (!q.lbl_rowptr || !q.lbl_col || (loss == KGE_LOSS_KL && !q.lse)))) return KGE_ERR_INVALID_ARG;
    if ((rc = check_index(q.a, false, q.n)) || (rc = check_index(q.p, false, q.n))) return rc;
    if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, q.a), rel_op(t, q.p), ent_op(t, none)))
      return KGE_ERR_UNSUPPORTED;
    side[k] = LossSide{ent_op(t, q.a), rel_op(t, q.p), q.n, (const long long*)q.lbl_rowptr, (const long long*)q.lbl_col,
                       q.lse, q.g_rows, q.g_scalar, nullptr, nullptr, q.g_dev};
  }
  return run_multilabel2_bwd_accum(t->scorer, loss, offset, side[0], side[1], ent_op(t, none), (int)t->dim, t->num_ent,
                                   grad_ent, grad_rel, t->num_rel, t->rel_dim, workspace, workspace_bytes,
                                   (hipStream_t)stream);
}

int kge_kl_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n, const int64_t* lbl_rowptr,
               const int64_t* lbl_col, float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes,
               void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !loss_rows || !lse))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_kl_fwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                    (const long long*)lbl_rowptr, (const long long*)lbl_col, loss_rows, lse, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

int kge_kl_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n, const int64_t* lbl_rowptr,
               const int64_t* lbl_col, const float* lse, const float* g_rows, float g_scalar, float* g_a,
               float* g_p, float* g_tgt, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !lse || !g_a || !g_p || !g_tgt))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_kl_bwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                    (const long long*)lbl_rowptr, (const long long*)lbl_col, lse, g_rows, g_scalar, g_a, g_p, g_tgt,
                    workspace, workspace_bytes, (hipStream_t)stream);
}

int kge_adagrad_step(float* param, const float* grad, float* state_sum, int64_t count, float minus_clr,
                     float weight_decay, float eps, void* bf16_copy, void* stream) {
  KGE_RANGE();
  if (count < 0 || (count > 0 && (!param || !grad || !state_sum))) return KGE_ERR_INVALID_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)state_sum) & 15) return KGE_ERR_INVALID_ARG;
  if (bf16_copy && ((uintptr_t)bf16_copy & 7)) return KGE_ERR_INVALID_ARG;
  return run_adagrad(param, grad, state_sum, count, minus_clr, weight_decay, eps, (unsigned short*)bf16_copy,
                     (hipStream_t)stream);
}

int kge_adagrad_step_multi(const kge_adagrad_seg* segs, int num_segs, void* stream) {
  KGE_RANGE();
  if (num_segs < 0 || num_segs > KGE_ADAGRAD_MAX_SEGS || (num_segs > 0 && !segs)) return KGE_ERR_INVALID_ARG;
  for (int j = 0; j < num_segs; ++j) {
    const kge_adagrad_seg& g = segs[j];
    if (g.count < 0 || (g.count > 0 && (!g.param || !g.grad || !g.state_sum))) return KGE_ERR_INVALID_ARG;
    if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.state_sum) & 15) return KGE_ERR_INVALID_ARG;
    if (g.bf16_copy && ((uintptr_t)g.bf16_copy & 7)) return KGE_ERR_INVALID_ARG;
  }
  return run_adagrad_multi(segs, num_segs, (hipStream_t)stream);
}

int kge_adagrad_step_multi_penalty(const kge_adagrad_seg* segs, const kge_penalty_seg* pens, int num_segs,
                                   void* stream) {
  if (!pens) return kge_adagrad_step_multi(segs, num_segs, stream);
  KGE_RANGE();
  if (num_segs < 0 || num_segs > KGE_ADAGRAD_MAX_SEGS || (num_segs > 0 && !segs)) return KGE_ERR_INVALID_ARG;
  for (int j = 0; j < num_segs; ++j) {
    const kge_adagrad_seg& g = segs[j];
    const kge_penalty_seg& q = pens[j];
    if (g.count < 0 || (g.count > 0 && (!g.param || !g.grad || !g.state_sum))) return KGE_ERR_INVALID_ARG;
    if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.state_sum) & 15) return KGE_ERR_INVALID_ARG;
    if (g.bf16_copy && ((uintptr_t)g.bf16_copy & 7)) return KGE_ERR_INVALID_ARG;
    if (q.kind < 0 || q.kind > 2) return KGE_ERR_INVALID_ARG;
    if (q.kind != 0 && (!q.value || ((uintptr_t)q.value & 7))) return KGE_ERR_INVALID_ARG;
    if (q.kind == 1 && (q.p < 1 || q.p > 3)) return KGE_ERR_UNSUPPORTED;
    if (q.kind == 2 && (q.row_dim <= 0 || q.row_dim % 8 != 0 || g.count % q.row_dim != 0)) return KGE_ERR_INVALID_ARG;
  }
  return run_adagrad_multi_pen(segs, pens, num_segs, (hipStream_t)stream);
}

int kge_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float step_size,
                  float bias_correction2_sqrt, double beta1, double beta2, float weight_decay, float eps,
                  void* bf16_copy, void* stream) {
  KGE_RANGE();
  if (count < 0 || (count > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return KGE_ERR_INVALID_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return KGE_ERR_INVALID_ARG;
  if (bf16_copy && ((uintptr_t)bf16_copy & 7)) return KGE_ERR_INVALID_ARG;
  if (!(bias_correction2_sqrt > 0.0f)) return KGE_ERR_INVALID_ARG;
  // 1 - beta in double, then rounded: the weights torch derives from the Python floats
  return run_adam(param, grad, exp_avg, exp_avg_sq, count, step_size, bias_correction2_sqrt, (float)(1.0 - beta1),
                  (float)beta2, (float)(1.0 - beta2),
                  weight_decay, eps, (unsigned short*)bf16_copy, (hipStream_t)stream);
}

int kge_adagrad_step_rows(float* param, int64_t param_ld, const float* grad_rows, int64_t grad_ld, float* state_sum,
                          int64_t sum_ld, const int64_t* rows, int64_t num_rows, int64_t dim, float minus_clr,
                          float eps, void* bf16_copy, int64_t copy_ld, void* stream) {
  KGE_RANGE();
  if (num_rows < 0 || dim < 0 || dim > 0x7fffffff) return KGE_ERR_INVALID_ARG;
  if (num_rows > 0 && dim > 0 && (!param || !grad_rows || !state_sum || !rows)) return KGE_ERR_INVALID_ARG;
  if (param_ld < dim || grad_ld < dim || sum_ld < dim || (bf16_copy && copy_ld < dim)) return KGE_ERR_INVALID_ARG;
  return run_adagrad_rows(param, param_ld, grad_rows, grad_ld, state_sum, sum_ld, (const long long*)rows, num_rows,
                          (int)dim, minus_clr, eps, (unsigned short*)bf16_copy, copy_ld, (hipStream_t)stream);
}

int kge_bce_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n, const int64_t* lbl_rowptr,
                const int64_t* lbl_col, float offset, float* loss_rows, void* workspace, int64_t workspace_bytes,
                void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !loss_rows))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_bce_fwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                     (const long long*)lbl_rowptr, (const long long*)lbl_col, offset, loss_rows, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

int kge_bce_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n, const int64_t* lbl_rowptr,
                const int64_t* lbl_col, float offset, const float* g_rows, float g_scalar, float* g_a, float* g_p,
                float* g_tgt, void* workspace, int64_t workspace_bytes, void* stream) {
  KGE_RANGE();
  const kge_index none = {nullptr, 0, 0, 1};
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (dir != KGE_SP_ && dir != KGE_PO_) return KGE_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!lbl_rowptr || !lbl_col || !g_a || !g_p || !g_tgt))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(a, false, n)) || (rc = check_index(p, false, n))) return rc;
  if (!ce_supported(t->scorer, t->dtype, (int)t->dim, ent_op(t, a), rel_op(t, p), ent_op(t, none)))
    return KGE_ERR_UNSUPPORTED;
  return run_bce_bwd(t->scorer, ent_op(t, a), rel_op(t, p), ent_op(t, none), dir, (int)t->dim, n, t->num_ent,
                     (const long long*)lbl_rowptr, (const long long*)lbl_col, offset, g_rows, g_scalar, g_a, g_p,
                     g_tgt, workspace, workspace_bytes, (hipStream_t)stream);
}

int kge_score_spo_bwd(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                      const float* gout, const float* scores, float* g_s, float* g_p, float* g_o,
                      void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (n < 0 || (n > 0 && (!gout || !g_s || !g_p || !g_o))) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) ||
      (rc = check_index(o, false)))
    return rc;
  return run_spo_bwd(t->scorer, t->l_norm, ent_op(t, s), rel_op(t, p), ent_op(t, o), (int)t->dim,
                     (int)t->rel_dim, n, gout, scores, g_s, g_p, g_o, (hipStream_t)stream);
}

int kge_score_spo_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                            const float* gout, const float* scores, float* grad_ent, int64_t grad_ent_ld,
                            float* grad_rel, int64_t grad_rel_ld, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (n < 0 || (n > 0 && (!gout || !grad_ent || !grad_rel))) return KGE_ERR_INVALID_ARG;
  if (grad_ent_ld < t->dim || grad_rel_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) || (rc = check_index(o, false)))
    return rc;
  return run_spo_bwd_accum(t->scorer, t->l_norm, ent_op(t, s), rel_op(t, p), ent_op(t, o), (int)t->dim,
                           (int)t->rel_dim, n, gout, scores, grad_ent, grad_ent_ld, grad_rel, grad_rel_ld,
                           (hipStream_t)stream);
}

int kge_score_neg_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                            int slot, const void* neg, int32_t neg_itype, int64_t neg_ld,
                            int64_t num_neg, const float* gout, int64_t ldg, const float* scores,
                            int64_t lds, float* grad_ent, int64_t grad_ent_ld, float* grad_rel,
                            int64_t grad_rel_ld, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (n < 0 || num_neg < 0 || (slot != 0 && slot != 2)) return KGE_ERR_INVALID_ARG;
  if (n * num_neg > 0 && (!gout || !neg || !grad_ent || !grad_rel)) return KGE_ERR_INVALID_ARG;
  if (neg_ld < num_neg || ldg < num_neg || (scores && lds < num_neg)) return KGE_ERR_INVALID_ARG;
  if (neg_itype != KGE_I32 && neg_itype != KGE_I64) return KGE_ERR_INVALID_ARG;
  if (grad_ent_ld < t->dim || grad_rel_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) || (rc = check_index(o, false)))
    return rc;
  return run_neg_bwd_accum(t->scorer, t->l_norm, ent_op(t, s), rel_op(t, p), ent_op(t, o), (int)t->dim,
                           (int)t->rel_dim, n, slot, neg, neg_itype, neg_ld, num_neg, gout, ldg, scores,
                           lds, grad_ent, grad_ent_ld, grad_rel, grad_rel_ld, (hipStream_t)stream);
}

int kge_neg_order(const void* neg, int32_t neg_itype, int64_t neg_ld, int64_t n, int64_t num_neg, int64_t num_ent,
                  int64_t* cursor, int64_t* order, void* stream) {
  KGE_RANGE();
  if (n < 0 || num_neg < 0 || num_ent <= 0 || neg_ld < num_neg) return KGE_ERR_INVALID_ARG;
  if (neg_itype != KGE_I32 && neg_itype != KGE_I64) return KGE_ERR_INVALID_ARG;
  if (n * num_neg > 0 && (!neg || !cursor)) return KGE_ERR_INVALID_ARG;  // (order == NULL: the histogram step)
  return run_neg_order(neg, neg_itype, neg_ld, n, num_neg, num_ent, (long long*)cursor, (long long*)order,
                       (hipStream_t)stream);
}

int kge_score_neg_bwd_accum_sorted(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                                   int slot, const void* neg, int32_t neg_itype, int64_t neg_ld,
                                   int64_t num_neg, const int64_t* order, const float* gout, int64_t ldg,
                                   const float* scores, int64_t lds, float* grad_ent, int64_t grad_ent_ld,
                                   float* grad_rel, int64_t grad_rel_ld, float* rel_scratch, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, true);
  if (rc) return rc;
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (n < 0 || num_neg < 0 || (slot != 0 && slot != 2)) return KGE_ERR_INVALID_ARG;
  if (n * num_neg > 0 && (!gout || !neg || !grad_ent || !grad_rel || !order)) return KGE_ERR_INVALID_ARG;
  if (neg_ld < num_neg || ldg < num_neg || (scores && lds < num_neg)) return KGE_ERR_INVALID_ARG;
  if (neg_itype != KGE_I32 && neg_itype != KGE_I64) return KGE_ERR_INVALID_ARG;
  if (grad_ent_ld < t->dim || grad_rel_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  if ((rc = check_index(s, false)) || (rc = check_index(p, false)) || (rc = check_index(o, false)))
    return rc;
  return run_neg_bwd_accum_sorted(t->scorer, t->l_norm, ent_op(t, s), rel_op(t, p), ent_op(t, o), (int)t->dim,
                                  (int)t->rel_dim, n, slot, neg, neg_itype, neg_ld, num_neg, (const long long*)order,
                                  gout, ldg, scores, lds, grad_ent, grad_ent_ld, grad_rel, grad_rel_ld, rel_scratch,
                                  t->num_rel, (hipStream_t)stream);
}

int kge_score_emb_bwd(const kge_tables* t, int combine, const void* s_emb, int64_t s_ld,
                      const void* p_emb, int64_t p_ld, const void* o_emb, int64_t o_ld, int64_t n,
                      int64_t m, const float* gout, int64_t ldg, const float* scores, int64_t lds,
                      float* g_s, float* g_p, float* g_o, void* stream) {
  KGE_RANGE();
  int rc = check_tables(t, false);
  if (rc) return rc;
  if (t->dtype != KGE_F32) return KGE_ERR_UNSUPPORTED;
  if (!s_emb || !p_emb || !o_emb || !gout || !g_s || !g_p || !g_o || n < 0 || m < 0)
    return KGE_ERR_INVALID_ARG;
  if (s_ld < t->dim || o_ld < t->dim || p_ld < t->rel_dim) return KGE_ERR_INVALID_ARG;
  const Index ident{nullptr, 1, KGE_I64};
  Operand S{s_emb, s_ld, ident}, P{p_emb, p_ld, ident}, O{o_emb, o_ld, ident};
  hipStream_t st = (hipStream_t)stream;
  const int d = (int)t->dim, dr = (int)t->rel_dim;
  if (combine == KGE_SPO)
    return run_spo_bwd(t->scorer, t->l_norm, S, P, O, d, dr, n, gout, scores, g_s, g_p, g_o, st);
  if (ldg < m || (scores && lds < m)) return KGE_ERR_INVALID_ARG;
  if (combine == KGE_SP_)
    return run_pairs_bwd(t->scorer, t->l_norm, KGE_SP_, S, P, O, d, dr, n, m, gout, ldg, scores,
                         lds, g_s, g_p, g_o, st, (t->flags & KGE_FLAG_EXACT) != 0);
  if (combine == KGE_PO_)
    return run_pairs_bwd(t->scorer, t->l_norm, KGE_PO_, O, P, S, d, dr, n, m, gout, ldg, scores,
                         lds, g_o, g_p, g_s, st, (t->flags & KGE_FLAG_EXACT) != 0);
  return KGE_ERR_INVALID_ARG;
}

// Not part of the public ABI: timestamp buffer (64 x u64 per workgroup) for the next
// kge_ce_fwd / kge_ce_bwd scoring launches (tools/ce_phases.py); NULL switches it off.
void kge_debug_ce_stamps(unsigned long long* stamps) { kge::ce_set_stamps(stamps); }

// Not part of the public ABI: timestamp buffer (64 x u64 per workgroup) for the next pairs_bf16_v6_kernel launches
// that carry none of their own (tools/v6_probe.py: stamps of two-sided / pipelined launches); NULL switches it off.
void kge_debug_v6_stamps(unsigned long long* stamps) { kge::v6_set_stamps(stamps); }
namespace kge {
__global__ void sqrt_check_kernel(unsigned int lo, unsigned long long count, unsigned long long* nbad, unsigned int* first) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const unsigned int b = lo + (unsigned int)i;
    const float x = __builtin_bit_cast(float, b);
    const float a = sqrt_rn_fast(x), r = __builtin_sqrtf(x);
    if (__builtin_bit_cast(unsigned int, a) != __builtin_bit_cast(unsigned int, r)) {
      const unsigned long long k = atomicAdd(nbad, 1ull);
      if (k < 16) first[k] = b;
    }
  }
}
}  // namespace kge
int kge_debug_sqrt_check(uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first16, void* stream) {
  if (mismatches == nullptr || first16 == nullptr || count == 0 || count > (1ull << 32)) return KGE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(kge::sqrt_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, first_bits,
                     (unsigned long long)count, (unsigned long long*)mismatches, first16);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}
int kge_debug_launch_count(int which) { return kge::v8_launch_count(which); }

int kge_debug_set_switch(const char* name, int64_t value) {
  const int i = kge::switch_index(name);
  if (i < 0) return KGE_ERR_INVALID_ARG;
  __atomic_store_n(&kge::g_switch[i], value < 0 ? -1LL : (long long)value, __ATOMIC_RELAXED);
  return KGE_OK;
}
int64_t kge_debug_get_switch(const char* name) {
  const int i = kge::switch_index(name);
  return i < 0 ? -2 : (int64_t)kge::sw((kge::Switch)i);
}

double kge_debug_mfma_rate(const void* operands, int iters, float* sink, void* stream) {
  return kge::run_mfma_rate(operands, iters, sink, (hipStream_t)stream);
}

// Not part of the public ABI: one gradient contraction of the bf16 backward on its own
// (tests/test_gpu_bwd_gemm16.py, tools/gemm16_probe.py); see run_debug_gemm16 in bwd_gemm.hip.
void kge_debug_gemm16_stamps(unsigned long long* stamps) { kge::set_g16_dbg(stamps); }

int kge_debug_gemm16(int which, int lib, int d, int64_t rows, int64_t m, const void* x, int64_t ldx, const void* g16,
                     int64_t mp, float* out, float* scratch, int64_t scratch_bytes, void* stream) {
  return kge::run_debug_gemm16(which, lib, d, rows, m, (const unsigned short*)x, ldx, (const unsigned short*)g16, mp,
                               out, scratch, scratch_bytes, (hipStream_t)stream);
}

// Not part of the public ABI (include/kge_amd.h): score_sp of the bf16 matrix-core kernels with a per-workgroup
// timestamp buffer (64 x u64 per workgroup) for tools/v2_phases.py / tools/prep_probe.py.  ablate 100 / 101:
// prepared / prepared split queries (builder launch, then the stamped scoring launch); 0: the one-call path.
int kge_debug_score_sp_bf16_v2(const kge_tables* t, kge_index s, kge_index p, int64_t n,
                               int64_t m, float* out, int64_t ldo, unsigned long long* stamps,
                               int ablate, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  int rc = check_tables(t, true);
  if (rc) return rc;
  kge_index all{nullptr, KGE_I64, 0, 1};
  Operand A = ent_op(t, s), R = rel_op(t, p), TG = ent_op(t, all);
  if (!pairs_bf16_v3_supported(t->scorer, t->dtype, (int)t->dim, A, R, TG)) return KGE_ERR_UNSUPPORTED;
  if (ablate == 100 || ablate == 101) {
    if (!pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, A, R, TG)) return KGE_ERR_UNSUPPORTED;
    return run_pairs_bf16_v4_prepared(t->scorer, ablate == 101, A, nullptr, R, TG, KGE_SP_, (int)t->dim, n, m, out, ldo, 0,
                                      (hipStream_t)stream, stamps, nullptr, workspace, workspace_bytes, 0, nullptr,
                                      nullptr, nullptr, 0, nullptr);
  }
  if (ablate) return KGE_ERR_UNSUPPORTED;
  if (!(t->flags & KGE_FLAG_BF16_V3) && workspace != nullptr &&
      pairs_bf16_v4_supported(t->scorer, t->dtype, (int)t->dim, A, R, TG)) {
    const int rc4 = run_pairs_bf16_v4(t->scorer, A, nullptr, R, TG, KGE_SP_, (int)t->dim, n, m, out, ldo,
                                      0, (hipStream_t)stream, stamps, workspace, workspace_bytes,
                                      (t->flags >> KGE_FLAG_RESERVE_CUS_SHIFT) & 255);
    if (rc4 != KGE_ERR_UNSUPPORTED) return rc4;
  }
  return run_pairs_bf16_v3(t->scorer, A, R, TG, KGE_SP_, (int)t->dim, n, m, out, ldo,
                           (hipStream_t)stream, stamps, workspace, workspace_bytes);
}

}  // extern "C"
