/*
 * kge_amd.h -- C ABI of the MI355X (gfx950) KGE scoring engine.
 *
 * This is the drop-in boundary of the hot path.  The
 * reference (uma-pi1/kge, "LibKGE") has no native code; its boundary for this
 * path is the index-level Python API of KgeModel.  Every entry point below
 * replaces one reference function and cites it (paths relative to the
 * reference tree):
 *
 *   kge_score_spo      KgeModel.score_spo        kge/model/kge_model.py:663-680
 *   kge_score_sp       KgeModel.score_sp         kge/model/kge_model.py:682-702
 *   kge_score_po       KgeModel.score_po         kge/model/kge_model.py:704-725
 *   kge_score_sp_po    KgeModel.score_sp_po      kge/model/kge_model.py:749-789
 *   kge_score_emb      RelationalScorer.score_emb kge/model/kge_model.py:151-213
 *                      (ComplEx complex.py:18-43, DistMult distmult.py:13-25,
 *                       TransE transe.py:15-37, RotatE rotate.py:20-69)
 *   kge_score_neg      BatchNegativeSample.score (impl "triple")
 *                                                kge/util/sampler.py:263-306
 *   kge_rank_counts    EntityRankingJob._filter_and_rank /
 *                      _get_ranks_and_num_ties   kge/job/eval_entity_ranking.py:533-596
 *   kge_score_*_bwd    autograd backward of the above (implicit in the
 *                      reference: train_1vsAll.py:70,81; train_KvsAll.py:293;
 *                      train_negative_sampling.py:161)
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  All pointers are DEVICE
 *     pointers (HBM) unless the name says host.  The library never allocates
 *     or frees device memory and keeps no global state: outputs and
 *     workspaces are caller-provided.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     All work is enqueued on that stream; calls are asynchronous and
 *     re-entrant.
 *   - every function returns KGE_OK (0) or a negative kge_status; nothing
 *     throws across the ABI.  kge_status_string() gives a static message.
 *   - embedding tables are row-major [rows, dim] with leading dimension `ld`
 *     (elements); element type f32 or bf16.  All scores are f32.
 *   - index vectors may be int32 or int64 with an element stride (the
 *     reference trainers pass stride-3 views triples[:,0] of an [n,3] tensor:
 *     kge/job/train_1vsAll.py:64; eval passes int32: eval_entity_ranking.py:164).
 *   - ComplEx/RotatE entity rows are [real half | imaginary half]
 *     (complex.py:24-27, rotate.py:24-25); RotatE relation rows hold dim/2
 *     phases in radians (rotate.py:28,88-93).
 *   - arithmetic is specified operation by operation in DESIGN.md ("canonical
 *     arithmetic"); the f32 paths are bit-reproducible against oracle/.
 */
#ifndef KGE_AMD_H
#define KGE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGE_AMD_ABI_VERSION 1

typedef enum kge_status {
  KGE_OK = 0,
  KGE_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, bad enum          */
  KGE_ERR_UNSUPPORTED = -2,   /* valid request this build has no kernel for     */
  KGE_ERR_LAUNCH = -3,        /* hipLaunchKernel / hip runtime error            */
  KGE_ERR_NO_DEVICE = -4,     /* no gfx950 device visible                       */
  KGE_ERR_WORKSPACE = -5      /* caller workspace too small                     */
} kge_status;

typedef enum kge_scorer {
  KGE_COMPLEX = 0,            /* kge/model/complex.py                            */
  KGE_DISTMULT = 1,           /* kge/model/distmult.py                           */
  KGE_TRANSE = 2,             /* kge/model/transe.py                             */
  KGE_ROTATE = 3              /* kge/model/rotate.py                             */
} kge_scorer;

typedef enum kge_dtype { KGE_F32 = 0, KGE_BF16 = 1 } kge_dtype;
typedef enum kge_itype { KGE_I32 = 0, KGE_I64 = 1 } kge_itype;

/* combine modes of RelationalScorer.score_emb (kge_model.py:151-213) */
typedef enum kge_combine {
  KGE_SPO = 0, KGE_SP_ = 1, KGE_PO_ = 2,
  KGE_SP_PO = 3               /* both blocks of KgeModel.score_sp_po (prepared-query entry points only) */
} kge_combine;

/* Entity + relation lookup tables (LookupEmbedder._embeddings.weight,
 * kge/model/embedder/lookup_embedder.py:44-46). */
typedef struct kge_tables {
  const void* ent;            /* [num_ent, dim]      */
  const void* rel;            /* [num_rel, rel_dim]  */
  int32_t dtype;              /* kge_dtype of both tables                        */
  int32_t scorer;             /* kge_scorer                                      */
  int64_t num_ent;
  int64_t num_rel;
  int64_t dim;                /* entity embedding size d                         */
  int64_t rel_dim;            /* d, except RotatE: d/2                           */
  int64_t ent_ld;             /* leading dimensions in elements                  */
  int64_t rel_ld;
  float l_norm;               /* TransE/RotatE `l_norm` option (transe.yaml:11)  */
  int32_t flags;              /* kge_flags, 0 = default                           */
} kge_tables;

/* Per-call kernel selection (no global state).  Default: fastest kernel.        */
typedef enum kge_flags {
  KGE_FLAG_EXACT = 1,         /* ComplEx/DistMult on bf16 tables: use the bit-reproducible
                                 f32-chain kernel instead of the bf16 MFMA kernel            */
  KGE_FLAG_NO_MFMA = 2,       /* f32 ComplEx/DistMult: VALU fmaf chain instead of the f32
                                 MFMA (same bits; used to cross-check the MFMA mapping)      */
                              /* (4, 8: retired with the tile-per-workgroup kernels v1 and v2 -- bf16 tables of a
                                 dim outside {128, 256, 512} run the f32 chain; the bits are ignored)       */
  KGE_FLAG_BF16_V3 = 16,      /* bf16 ComplEx/DistMult with a workspace: the single-role
                                 kernel (v3) instead of the loader/consumer kernel (v4)      */
  KGE_FLAG_SPLIT_QUERY = 32   /* bf16 ComplEx/DistMult scoring: the query vector q = s (x) r is NOT rounded
                                 to one bf16 but carried as q_hi + q_lo (two bf16 pieces, two MFMA chains,
                                 one f32 add): products stay exact, only the f32 summation order differs from
                                 f32 arithmetic on the same bf16 tables (+ a 2^-17 relative residue of q for
                                 ComplEx, none for DistMult): f32-level parity at about half the speed of the
                                 single-pass kernel instead of the seventh the f32 kernels run at.  (KGE_FLAG_EXACT
                                 is something else: the single-pass semantics -- q rounded to bf16 -- as an
                                 order-specified f32 chain.)  Evaluation
                                 (rank parity with the reference: eval_entity_ranking.py:590-595 ties at
                                 rtol 1e-4) wants this; training does not.  Applies to kge_score_sp / _po /
                                 _sp_po / _emb / _emb_sp_po and the prepared-query entry points; shapes the
                                 matrix-core kernel does not take run the f32 chain on the widened tables with
                                 an unrounded query (bit-exact f32 arithmetic on the table values).            */
} kge_flags;
/* Bits 8..15 of `flags`: number of compute units the persistent bf16 scoring kernel leaves
 * free (it launches one workgroup per remaining CU), so that kernels on other streams -- the
 * RCCL kernels of an overlapped exchange -- run beside it.  0 = all.  */
/* kge_ce_sp_po_fwd(_sum) / kge_ce_sp_po_bwd_accum(_sum) only -- the caller's promise that the backward call follows
 * the forward call with the SAME tables, index vectors, n and workspace, and that no other call used that workspace in
 * between (a captured training step; kge_amd.model checks it with a per-workspace generation count): the forward then
 * also leaves the gradient products' query matrix in the workspace and the backward starts from the forward's query
 * fragments instead of building them again (one launch less per step).  Without the promise kept the backward reads
 * stale fragments: set it on BOTH calls or on neither. */
#define KGE_FLAG_CE_KEEP_QUERIES (1 << 16)
#define KGE_FLAG_RESERVE_CUS_SHIFT 8
#define KGE_FLAG_RESERVE_CUS(n) (((n) & 255) << KGE_FLAG_RESERVE_CUS_SHIFT)

/* An index vector: element i is ptr[i*stride] of type itype.
 * ptr == NULL means the identity 0,1,2,... (used for "all entities") -- or, as the `targets` of kge_score_sp / _po /
 * _sp_po only, the contiguous range start, start+1, ..., start+m-1 (0 <= start, start + m <= num_ent): the entity chunk
 * the reference's EntityRankingJob scores against, passed there as torch.arange(chunk_start, chunk_end)
 * (kge/job/eval_entity_ranking.py:216-229) -- the kernels stream rows [start, start + m) of the table itself, no index
 * is read.  `start` must be 0 with a non-NULL ptr and everywhere else. */
typedef struct kge_index {
  const void* ptr;
  int32_t itype;              /* kge_itype */
  int32_t start;              /* see above; 0 otherwise (until round 6 this field was `reserved`, always 0) */
  int64_t stride;             /* in elements */
} kge_index;

/* ---- library ---------------------------------------------------------- */
int kge_abi_version(void);
const char* kge_status_string(int status);
/* Number of gfx950 devices visible to the HIP runtime (0 if none). */
int kge_device_count(void);

/* ---- index-level scoring (fused gather + score) ------------------------ */

/* Optional device workspace for the scoring calls below.  The library never allocates:
 * a caller that passes `workspace_bytes >= kge_score_workspace_bytes(t, n)` lets the
 * bf16 ComplEx/DistMult path build the n query vectors ONCE -- cooperatively inside the one
 * scoring kernel: a few workgroups build, publish through the workspace, all workgroups
 * consume -- instead of once per workgroup.  workspace == NULL (or too small) selects the
 * path where every workgroup builds its own copy.  Results are identical bit for bit.  The
 * workspace must be ZEROED ONCE (hipMemset) before its first use and left to the library from
 * then on: besides scratch it holds the builders' flag lines and a "degraded" word.  A consumer
 * workgroup waits for its builders only for a bounded time (they may not be running: other
 * streams' kernels, a second process, CU masking); after a time-out it builds its own query
 * vectors and sets that word to a count of launches; the next few thousand calls on the workspace
 * skip the hand-off (slower, never wrong, never a hang), then it is tried again (non-zero garbage
 * in a fresh workspace has the same effect, for as long as it takes to count it down).  The
 * workspace is only accessed during a call (stream order; calls may be captured into a hipGraph
 * and replayed) and must not be shared by calls that may run concurrently on different streams;
 * 16-byte aligned.  ONE workspace may serve calls with different n (size it for the largest): the
 * control block -- flag lines and the degraded word -- lies at its start, at an offset that does not
 * depend on n; the query vectors follow.  The same holds for the workspaces of the fused-loss entry points below. */
int64_t kge_score_workspace_bytes(const kge_tables* t, int64_t n);

/* out[i] = score(s[i], p[i], o[i]), i < n.        KgeModel.score_spo */
int kge_score_spo(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                  int64_t n, float* out, void* stream);

/* out[i*ldo + j] = score(s[i], p[i], targets[j]), j < m.  KgeModel.score_sp
 * targets.ptr == NULL: all entities, m must equal t->num_ent. */
int kge_score_sp(const kge_tables* t, kge_index s, kge_index p, int64_t n,
                 kge_index targets, int64_t m, float* out, int64_t ldo,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* out[i*ldo + j] = score(targets[j], p[i], o[i]).  KgeModel.score_po */
int kge_score_po(const kge_tables* t, kge_index p, kge_index o, int64_t n,
                 kge_index targets, int64_t m, float* out, int64_t ldo,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* out[i*ldo + j] = sp score, out[i*ldo + m + j] = po score (j < m); ldo >= 2m.
 * KgeModel.score_sp_po (cat of score_sp and score_po over one entity subset). */
int kge_score_sp_po(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                    int64_t n, kge_index targets, int64_t m, float* out,
                    int64_t ldo, void* workspace, int64_t workspace_bytes,
                    void* stream);

/* ---- prepared queries: the query build taken out of the scoring launch -------------------------------
 * KgeModel.score_sp / score_po / score_sp_po (kge/model/kge_model.py:682-789) compute, per call,
 * q_i = embed(s_i) (x) embed(p_i) (complex.py:30-37, distmult.py:17-21: the "sp_" / "_po" halves of score_emb) and
 * then q . E^T.  Inside ONE launch the first half is a chain of five dependent memory round trips before the
 * first score can be stored -- 40 % of a launch at the FB15k-237 shape.  A caller that knows its
 * batches ahead (EntityRankingJob._evaluate iterates a DataLoader: eval_entity_ranking.py:158-170; so does every
 * training epoch) can take it off the critical path:
 *
 *   kge_build_queries(batch 0)                               -- one small launch, once
 *   kge_score_queries(batch k, next = batch k + 1) ...       -- ONE launch per batch: scores batch k from its
 *                                                               prepared queries, while workgroups on the compute
 *                                                               units the launch leaves idle build batch k + 1's
 *
 * `queries` is an opaque device buffer of kge_queries_bytes(t, combine, n) bytes (16-byte aligned; bf16 MFMA
 * operand fragments of the n query vectors; with KGE_FLAG_SPLIT_QUERY the q_hi and q_lo pieces), valid for the
 * tables, flags, combine and n it was built with; the caller double-buffers (`next->queries` must differ from
 * `queries`).  combine: KGE_SP_ (s, p given; o ignored), KGE_PO_ (p, o given; s ignored) or KGE_SP_PO (all three;
 * out[i, :m] = sp scores, out[i, b2:b2+m] = po scores, b2 = block2_offset or m).  Scores are bit-identical to kge_score_sp / _po /
 * _sp_po on the same tables and flags.  bf16 ComplEx / DistMult, dim 256 / 512; otherwise KGE_ERR_UNSUPPORTED.
 * No workspace, no flags to zero, no co-residency requirement: capture-safe, any n. */
typedef struct kge_next_queries {
  kge_index s, p, o;          /* the next batch (as for kge_build_queries)                       */
  int64_t n;                  /* 0: nothing to build                                             */
  void* queries;              /* destination, kge_queries_bytes(t, combine, n) bytes             */
  int64_t queries_bytes;
} kge_next_queries;
int64_t kge_queries_bytes(const kge_tables* t, int combine, int64_t n);
int kge_build_queries(const kge_tables* t, int combine, kge_index s, kge_index p, kge_index o,
                      int64_t n, void* queries, int64_t queries_bytes, void* stream);
/* block2_offset (KGE_SP_PO only): column at which the po block of a row starts, >= m and <= ldo - m; 0 = m
 * (the reference's cat layout).  With ldo and block2_offset multiples of 8 floats on a 32-byte aligned `out`
 * every store of the kernel covers whole 32-byte memory sectors (8 % per launch at the
 * FB15k-237 shape, 28 % on a Wikidata5M shard). */
int kge_score_queries(const kge_tables* t, int combine, const void* queries, int64_t n,
                      kge_index targets, int64_t m, float* out, int64_t ldo, int64_t block2_offset,
                      const kge_next_queries* next /* may be NULL */, void* stream);

/* GROUPS of batches in one launch.  A caller that iterates a split knows more than the next batch: an epoch's
 * DataLoader (kge/job/train_1vsAll.py:31-41) or EntityRankingJob._evaluate's loop (eval_entity_ranking.py:143-229)
 * can hand over `num_batches` equally shaped batches at once.  The scoring launch is then ONE persistent kernel that
 * streams the table through every compute unit once per batch and writes batch l's block at out + l * out_stride
 * (floats; each block laid out as kge_score_queries lays out its one): the launch gap, the cold start of a launch and
 * the compute units a single batch's grid cannot fill are paid once per group, not once per batch.
 *   kge_build_queries_multi   batch l = rows [l n, (l + 1) n) of s / p / o; its fragments at queries + l *
 *                             queries_stride (bytes; a multiple of 16, >= kge_queries_bytes(t, combine, n));
 *   kge_score_queries_multi   scores the group; `next` (may be NULL) describes the NEXT group -- num_batches batches of
 *                             next->n rows each, index vectors of num_batches * next->n entries, fragments next_stride
 *                             bytes apart in next->queries -- built by the same launch behind its last unit.
 * Scores are bit-identical to num_batches kge_score_queries calls.  bf16 ComplEx / DistMult at dim 512 against all
 * entities; anything else KGE_ERR_UNSUPPORTED (loop over kge_score_queries instead). */
int kge_build_queries_multi(const kge_tables* t, int combine, kge_index s, kge_index p, kge_index o, int64_t n,
                            int64_t num_batches, void* queries, int64_t queries_stride, int64_t queries_bytes,
                            void* stream);
int kge_score_queries_multi(const kge_tables* t, int combine, const void* queries, int64_t queries_stride, int64_t n,
                            int64_t num_batches, kge_index targets, int64_t m, float* out, int64_t out_stride,
                            int64_t ldo, int64_t block2_offset, const kge_next_queries* next /* may be NULL */,
                            int64_t next_stride, void* stream);

/* Negative-sampling scores, the "triple" implementation without building the
 * [n*K,3] index tensor: slot 0/2 = corrupt s / o.
 * out[i*ldo + k] = score of triple i with slot replaced by neg[i*neg_ld + k].
 * BatchNegativeSample.score, kge/util/sampler.py:291-306. */
int kge_score_neg(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                  int64_t n, int slot, const void* neg, int32_t neg_itype,
                  int64_t neg_ld, int64_t num_neg, float* out, int64_t ldo,
                  void* stream);

/* ---- embedding-level scoring (dense inputs, no gather) ----------------- */
/* RelationalScorer.score_emb.  `t` supplies scorer, dtype, dim, rel_dim and
 * l_norm only (t->ent/rel ignored).  SPO: all three [n, *] -> out[n];
 * SP_: s,p [n,*], o [m,dim] -> out[n,m]; PO_: p,o [n,*], s [m,dim] -> out[n,m]. */
int kge_score_emb(const kge_tables* t, int combine, const void* s_emb,
                  int64_t s_ld, const void* p_emb, int64_t p_ld,
                  const void* o_emb, int64_t o_ld, int64_t n, int64_t m,
                  float* out, int64_t ldo, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* score_sp_po on dense rows (KgeModel.score_sp_po, kge/model/kge_model.py:749-789, after its
 * embed() calls): s_emb / p_emb / o_emb are the n query rows, tgt_emb the m target rows;
 * out[i, :m] = sp_ scores, out[i, m:2m] = _po scores (ldo >= 2m), one two-sided launch when the
 * bf16 matrix-core kernel applies (workspace as for kge_score_sp_po).  Used by the entity-sharded
 * path, whose query rows arrive by all-gather. */
int kge_score_emb_sp_po(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb,
                        int64_t p_ld, const void* o_emb, int64_t o_ld, int64_t n,
                        const void* tgt_emb, int64_t tgt_ld, int64_t m, float* out, int64_t ldo,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* The same with the _po block `block2_offset` (>= m) floats behind the sp_ block of a row instead of right behind it
 * (ldo >= block2_offset + m): with both blocks on whole 256-byte lines -- block2_offset = a padded pitch, ldo twice
 * that -- the direct-store kernel takes its aligned store path (the per-rank launch of the sharded step). */
int kge_score_emb_sp_po_blocks(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb,
                               int64_t p_ld, const void* o_emb, int64_t o_ld, int64_t n,
                               const void* tgt_emb, int64_t tgt_ld, int64_t m, float* out, int64_t ldo,
                               int64_t block2_offset, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- ranking ------------------------------------------------------------ */
/* For each row i of scores[n, c] (leading dim lds) and its true score:
 *   x = NaN -> -inf; filtered columns -> -inf; t = NaN -> -inf
 *   close   = (x == t) || (isfinite(|x-t|) && |x-t| <= atol + |rtol*t|)
 *   ties[i] += #close ;  rank[i] += #(x > t && !close)
 * Filtered columns of row i are lbl_col[lbl_rowptr[i] .. lbl_rowptr[i+1]) minus
 * col_offset; entries outside [0,c) and the entry equal to true_col[i] (global
 * id, may be NULL) are ignored.  Columns must be unique within a row.
 * lbl_rowptr == NULL: raw ranking.  rank/ties are int64 and ACCUMULATED
 * (counts are additive over entity chunks: eval_entity_ranking.py:310-313).
 * EntityRankingJob._filter_and_rank + _get_ranks_and_num_ties,
 * kge/job/eval_entity_ranking.py:533-596. */
int kge_rank_counts(const float* scores, int64_t lds, int64_t n, int64_t c,
                    const float* true_scores, const int64_t* lbl_rowptr,
                    const int64_t* lbl_col, int64_t col_offset,
                    const int64_t* true_col, float atol, float rtol,
                    int64_t* rank, int64_t* ties, void* stream);

/* ---- evaluation without host round trips --------------- */
/* The filter index of a set of splits lives on the device as sorted arrays:
 *   sorted_keys[num_keys]   unique keys (s*num_rel + p for the sp index, p*num_ent + o for po)
 *   starts[num_keys + 1]    range of key k in the index's value array
 *   values[...]             the known answers of each key, unique per key
 * (replaces the numba dict KvsAllIndex, kge/indexing.py:10-194).
 *
 * kge_filter_lookup: begin[i], end[i] = range of key a[i]*mult + b[i] (0, 0 if absent): what
 * get_sp_po_coords_from_spo_batch (kge/job/util.py:6-29) + _collate
 * (eval_entity_ranking.py:77-101) look up per batch on the host. */
int kge_filter_lookup(const int64_t* sorted_keys, int64_t num_keys, const int64_t* starts,
                      kge_index a, kge_index b, int64_t mult, int64_t n, int64_t* begin,
                      int64_t* end, void* stream);

/* Up to KGE_MAX_FILTER_QUERIES lookups in ONE launch: an evaluation batch needs four (the (s, p) and the
 * (p, o) keys of the filtered and of the filtered-with-test index), each a few dependent round trips of pure
 * latency on its own. */
#define KGE_MAX_FILTER_QUERIES 4
typedef struct kge_filter_query {
  const int64_t* sorted_keys;
  int64_t num_keys;
  const int64_t* starts;
  kge_index a, b;
  int64_t mult;
  int64_t* begin;
  int64_t* end;
} kge_filter_query;
int kge_filter_lookup_multi(const kge_filter_query* queries, int num_queries, int64_t n, void* stream);

/* kge_rank_counts for the raw ranking and `num_filters` (<= KGE_MAX_FILTERS) filtered
 * rankings from ONE scan of the scores: rank/ties are [num_filters + 1][n] int64, row 0 raw,
 * row k + 1 filtered by the columns lbl_col[k][lbl_begin[k][i] .. lbl_end[k][i]) of row i
 * (global ids, minus col_offset; the one equal to true_col[i] stays).  ACCUMULATED over
 * entity chunks.  lbl_begin / lbl_end / lbl_col are HOST arrays of device pointers.
 * eval_entity_ranking.py:233-313 runs _filter_and_rank once per ranking. */
#define KGE_MAX_FILTERS 4
int kge_rank_counts_multi(const float* scores, int64_t lds, int64_t n, int64_t c,
                          const float* true_scores, int num_filters,
                          const int64_t* const* lbl_begin, const int64_t* const* lbl_end,
                          const int64_t* const* lbl_col, int64_t col_offset,
                          const int64_t* true_col, float atol, float rtol, int64_t* rank,
                          int64_t* ties, void* stream);

/* Scoring and rank counting in ONE kernel: the counts kge_score_sp_po + kge_rank_counts_multi produce for the
 * entity slice [col_begin, col_begin + m) (bit for bit: the same score chains, the same tie arithmetic),
 * without the [n, 2m] score matrix ever being written or read -- EntityRankingJob._evaluate's
 * score_sp_po -> _filter_and_rank -> _get_ranks_and_num_ties (eval_entity_ranking.py:227-313) per entity chunk.
 *   true_sp[i] / true_po[i]   score of triple i as the sp_ / _po scoring sees it (an element of the score
 *                             matrix: e.g. the diagonal of kge_score_sp_po against targets = o resp. s);
 *   sp_* / po_*               num_filters (<= 2) filter sets per direction, as for kge_rank_counts_multi:
 *                             HOST arrays of device pointers; columns are global entity ids, the row's own
 *                             o[i] (sp_) / s[i] (_po) is never filtered;
 *   rank_* / ties_*           [num_filters + 1][ld] int64, row 0 raw, ACCUMULATED (over entity chunks);
 *   filter_bits               >= kge_score_rank_bits_bytes(n, m, num_filters) bytes, 8-byte aligned, ZEROED ONCE
 *                             before its first use: the call sets one bit per filtered (row, column), counts,
 *                             and clears the same words again (all-zero between calls);
 *   workspace                 as for kge_score_sp_po (kge_score_workspace_bytes(t, n)).
 * KGE_ERR_UNSUPPORTED (tables other than bf16 ComplEx / DistMult with dim 256 / 512, a launch the
 * loader/consumer kernel declines): use kge_score_sp_po + kge_rank_counts_multi. */
int64_t kge_score_rank_bits_bytes(int64_t n, int64_t m, int num_filters);
int kge_score_rank_sp_po(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                         int64_t col_begin, int64_t m, const float* true_sp, const float* true_po,
                         int num_filters, const int64_t* const* sp_begin, const int64_t* const* sp_end,
                         const int64_t* const* sp_col, const int64_t* const* po_begin,
                         const int64_t* const* po_end, const int64_t* const* po_col, float atol, float rtol,
                         int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po, int64_t* ties_po, int64_t ld,
                         void* filter_bits, int64_t filter_bits_bytes, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* Band-and-rescore: the counts of kge_score_rank_sp_po / kge_eval_batch under KGE_FLAG_SPLIT_QUERY --
 * the parity-compliant evaluation mode, EntityRankingJob._get_ranks_and_num_ties' tie band honoured on full-precision
 * query vectors (eval_entity_ranking.py:571-596) -- at close to the price of the SINGLE-PASS counting kernel.  The
 * single-pass score x_hi is the hi half of the split score x = fl(x_hi + x_lo), and |x_lo| <= ||q_lo_i|| * max_j ||t_j||:
 * with row i's tolerance widened by that bound, a score outside the widened band compares with the true score the same
 * way under x_hi and under x.  Launch 1 (the single-pass counting kernel on the q_hi fragments) counts those and LISTS
 * the (row, column) pairs inside the band -- per wave, no atomics --; launch 2 gathers the listed columns' table rows,
 * scores them with both chains (the bits the split kernel counts) and finishes the counts.  Identical (rank, ties) to
 * the split kernel by construction and by test WHEN NO PAIR WAS DROPPED; on a trained model (true scores in the tail of
 * their rows) ~2e-5 of the pairs are listed, on random tables ~1.5 % -- far more than the lists hold.
 *   table_max_norm  [1] device float: kge_table_max_row_norm of the scored rows [col_begin, col_begin + m);
 *   list            device scratch, 16-byte aligned, >= kge_rank_band_list_bytes(n) bytes, ZEROED ONCE by the caller
 *                   before the first call (the calls leave every list empty); 255 pairs per wave and 256-row chunk;
 *   status          [2] device uint32 or NULL, zeroed by the caller: [0] += pairs listed, [1] += pairs DROPPED because
 *                   a wave's list was full (sticky).
 * status[1] != 0 means some call's counts are INCOMPLETE: the caller must redo those batches without `band` (the
 * evaluator reads the words once at the end of a run -- no host wait per batch -- and falls back to the split kernel
 * for the run; it also probes the first batch and drops the band when it lists too much: nothing to gain there).
 * band == NULL: the plain entry points.  KGE_ERR_INVALID_ARG: band without KGE_FLAG_SPLIT_QUERY, missing pieces;
 * KGE_ERR_WORKSPACE: list too small for n; KGE_ERR_UNSUPPORTED as for kge_score_rank_sp_po. */
typedef struct kge_rank_band {
  const float* table_max_norm;
  void* list;
  int64_t list_bytes;
  uint32_t* status;
} kge_rank_band;
int64_t kge_rank_band_list_bytes(int64_t n);
/* 1.001 x the largest Euclidean norm of the rows [row_begin, row_begin + m) of the bf16 entity table, into out[0]
 * (device).  Once per table state (an evaluation run), not per batch. */
int kge_table_max_row_norm(const kge_tables* t, int64_t row_begin, int64_t m, float* out, void* stream);
int kge_score_rank_sp_po_band(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                              int64_t col_begin, int64_t m, const float* true_sp, const float* true_po,
                              int num_filters, const int64_t* const* sp_begin, const int64_t* const* sp_end,
                              const int64_t* const* sp_col, const int64_t* const* po_begin,
                              const int64_t* const* po_end, const int64_t* const* po_col, float atol, float rtol,
                              int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po, int64_t* ties_po, int64_t ld,
                              void* filter_bits, int64_t filter_bits_bytes, void* workspace,
                              int64_t workspace_bytes, void* stream, const kge_rank_band* band);

/* The same for dense query rows (the row-sharded multi-GPU path: s / p / o rows come out of the exchange, the
 * scored rows tgt_emb[m] are this rank's shard, whose global entity ids start at col_begin; s_ids / o_ids =
 * the global ids of the rows' true subject / object, which the filters never remove).  Counts of the shard's
 * columns only: the caller sums them over the ranks (one int64 all-reduce). */
int kge_score_rank_emb_sp_po(const kge_tables* t, const void* s_emb, int64_t s_ld, const void* p_emb,
                             int64_t p_ld, const void* o_emb, int64_t o_ld, kge_index s_ids, kge_index o_ids,
                             int64_t n, const void* tgt_emb, int64_t tgt_ld, int64_t col_begin, int64_t m,
                             const float* true_sp, const float* true_po, int num_filters,
                             const int64_t* const* sp_begin, const int64_t* const* sp_end,
                             const int64_t* const* sp_col, const int64_t* const* po_begin,
                             const int64_t* const* po_end, const int64_t* const* po_col, float atol, float rtol,
                             int64_t* rank_sp, int64_t* ties_sp, int64_t* rank_po, int64_t* ties_po, int64_t ld,
                             void* filter_bits, int64_t filter_bits_bytes, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* One evaluation batch of EntityRankingJob._evaluate (eval_entity_ranking.py:103-481: label lookup, score_sp_po,
 * _filter_and_rank per ranking, _get_ranks, hist_all) against ALL entities in FOUR launches, one call:
 *   (1) the filter ranges of the batch's (s, p) / (p, o) keys in up to two filter indexes (kge_filter_lookup's
 *       search) and one bit per filtered (row, column) -- the row's own o / s never --, plus the target list (o | s);
 *   (2) the true scores: the batch against its own targets (kge_score_sp_po on 2 n target rows), diagonals kept;
 *   (3) scoring + counting (kge_score_rank_sp_po's kernels: no [n, 2E] score matrix);
 *   (4) the bits cleared again, tie policy + rank histograms of both directions (kge_rank_hist), counters zeroed.
 * counts: int64 [2 (o | s)][2 (rank | ties)][num_filters + 1][n], ALL-ZERO on entry and again on return; hist:
 * float [num_filters + 1][ldh] accumulated (row 0 raw); ranks_o / ranks_s: int64 [num_filters + 1][n] or NULL;
 * filter_bits: as for kge_score_rank_sp_po (>= kge_score_rank_bits_bytes(n, num_entities, num_filters) bytes, ZEROED
 * ONCE before its first use, all-zero between calls); scratch: kge_eval_batch_scratch_bytes(t, n, num_filters) bytes,
 * 16-byte aligned, contents irrelevant; workspace as for kge_score_sp_po.  KGE_ERR_UNSUPPORTED: tables without a
 * counting kernel (see kge_score_rank_sp_po) -- nothing was counted, nothing left behind.  No host wait. */
typedef struct kge_eval_filter {
  const int64_t* sp_keys;   int64_t sp_num_keys; const int64_t* sp_starts; const int64_t* sp_values;  /* key s*R + p -> o's */
  const int64_t* po_keys;   int64_t po_num_keys; const int64_t* po_starts; const int64_t* po_values;  /* key p*E + o -> s's */
} kge_eval_filter;
int64_t kge_eval_batch_scratch_bytes(const kge_tables* t, int64_t n, int num_filters);
int kge_eval_batch(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int num_filters,
                   const kge_eval_filter* filters, float atol, float rtol, int tie_policy, int64_t* counts,
                   float* hist, int64_t ldh, int64_t* ranks_o, int64_t* ranks_s, void* filter_bits,
                   int64_t filter_bits_bytes, void* scratch, int64_t scratch_bytes, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* kge_eval_batch with band-and-rescore in step (3) (see kge_score_rank_sp_po_band); band == NULL: kge_eval_batch. */
int kge_eval_batch_band(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n, int num_filters,
                        const kge_eval_filter* filters, float atol, float rtol, int tie_policy, int64_t* counts,
                        float* hist, int64_t ldh, int64_t* ranks_o, int64_t* ranks_s, void* filter_bits,
                        int64_t filter_bits_bytes, void* scratch, int64_t scratch_bytes, void* workspace,
                        int64_t workspace_bytes, void* stream, const kge_rank_band* band);

/* hist[m*ldh + r] += 1.0f with r = rank of the tie policy, for all [num_rankings][n] counts;
 * ranks_out (may be NULL) receives r.  EntityRankingJob._get_ranks (:598-618) + hist_all
 * (:665-687); the float32 histogram is the reference's. */
enum { KGE_TIES_ROUNDED_MEAN = 0, KGE_TIES_BEST = 1, KGE_TIES_WORST = 2 };
int kge_rank_hist(const int64_t* rank, const int64_t* ties, int num_rankings, int64_t n,
                  int tie_policy, float* hist, int64_t ldh, int64_t num_ent,
                  int64_t* ranks_out, void* stream);

/* ---- LookupEmbedder.embed ------------------------------------------------ */
/* ent_out[i, :] = ent[ent_idx[i], :] (n_ent rows, leading dimension ent_ldo elements) and
 * rel_out[i, :] = rel[rel_idx[i], :] in ONE launch (table dtype; rows of 16-byte multiples).
 * kge/model/embedder/lookup_embedder.py:96-105.  The scoring entry points gather on the fly
 * and never need this; the entity-sharded path does: a rank gathers the query rows it owns
 * before the single all-gather of a batch.  Either count may be 0. */
int kge_embed(const kge_tables* t, kge_index ent_idx, int64_t n_ent, void* ent_out,
              int64_t ent_ldo, kge_index rel_idx, int64_t n_rel, void* rel_out,
              int64_t rel_ldo, void* stream);

/* BCEWithLogitsKgeLoss over the [n, c = 1 + K] score block of one negative-sampling slot (kge/util/loss.py:136-189 on
 * the block TrainingJobNegativeSampling._process_subbatch assembles, train_negative_sampling.py:120-151: column 0 the
 * positive, columns 1.. its negatives), forward and gradient in one pass:
 *   kind 0 "bce": sum_j l(x_j, y_j);  1 "bce_mean": (l(x_0, 1) + sum_{j>=1} l(x_j, 0) / K) / 2;
 *   2 "bce_self_adversarial": (l(x_0, 1) + sum_{j>=1} w_j l(x_j, 0)) / 2, w = softmax_j(temperature * x_j), not
 *   differentiated -- x = score + offset (train.loss_arg), l = torch.nn.BCEWithLogitsLoss's element.
 * loss_rows[i] = row i's term (the job's loss is their sum, divided by the batch size by the job);
 * grad (may be NULL) [n, c], leading dimension ldg: d (sum_i loss_rows[i]) / d scores[i, j]. */
int kge_ns_bce_loss(const float* scores, int64_t ld, int64_t n, int64_t c, int kind, float offset, float temperature,
                    float* loss_rows, float* grad, int64_t ldg, void* stream);

/* The two row moves of the entity-sharded exchange (kge_amd/sharded.py: ShardedEntityTable.exchange_rows),
 * the id arithmetic evaluated inside the kernel, one launch each:
 *   kge_shard_gather   t->ent = THIS RANK's rows [lo, lo + t->num_ent) of the entity table.  For j < num_ids (1 or 2
 *                      id vectors of GLOBAL entity ids, e.g. the s and the o column of a batch), i < n:
 *                        send[(j*n + i) * send_ld ..] = t->ent[clamp(ids[j][i] - lo, 0, t->num_ent - 1)]
 *                      (an id this rank does not own reads some local row: that entry of the rank's block is never
 *                      picked) and, if rel_out != NULL, rel_out[i] = t->rel[rel_idx[i]] (replicated relation table).
 *   kge_shard_pick     after the all-gather of the ranks' blocks (`gathered` = world x [num_ids * n] rows, leading
 *                      dimension ld):  rows[j*n + i] = gathered[(ids[j][i] / shard_rows) * num_ids * n + j*n + i],
 *                      the owner's copy of every row (shard_rows = ceil(E / world), the rows per rank).
 * dtype: kge_dtype of the rows.  Rows must be 16-byte multiples on 16-byte aligned pitches (KGE_ERR_UNSUPPORTED). */
int kge_shard_gather(const kge_tables* t, int64_t lo, const kge_index* ids, int num_ids, int64_t n, void* send,
                     int64_t send_ld, kge_index rel_idx, void* rel_out, int64_t rel_ldo, void* stream);
int kge_shard_pick(const void* gathered, int64_t ld, int dtype, int64_t dim, int64_t shard_rows, int world,
                   const kge_index* ids, int num_ids, int64_t n, void* rows, int64_t rows_ld, void* stream);

/* ---- 1vsAll loss fused with the scoring ---------------- */
/* loss_rows[i] = logsumexp_j score(i, j) - score(i, label[i]),  lse[i] = logsumexp_j score(i, j),
 * j over ALL entities; score(i, .) = the kge_score_sp row (dir = KGE_SP_, a = s, label = o) or
 * the kge_score_po row (dir = KGE_PO_, a = o, label = s).  sum_i loss_rows[i] is what
 * TrainingJob1vsAll computes with the default `train.loss: kl`: score_sp / score_po
 * (kge/job/train_1vsAll.py:64, 75) followed by KLDivWithSoftmaxKgeLoss with index labels =
 * CrossEntropyLoss(reduction="sum") (kge/util/loss.py:192-207, train_1vsAll.py:65, 76).
 * The [n, num_ent] score matrix is never written: the scoring kernel folds every tile into a
 * per-row running (max, sum exp).  The scores inside are bit-identical to kge_score_sp /
 * kge_score_po on the same tables.
 *
 * A label outside [0, num_ent) gives loss_rows[i] = NaN (the reference raises an index error).
 *
 * kge_ce_bwd: gradients of sum_i g_i * loss_rows[i] (g_i = g_rows[i], or g_scalar if g_rows is
 * NULL -- the reference's 1 / batch_size), laid out as for kge_score_pairs_bwd:
 *   g_a [n, dim], g_p [n, rel_dim], g_tgt [num_ent, dim]   (f32, OVERWRITTEN)
 * The kernel recomputes the score tiles, writes d loss / d score in bf16 to the workspace and
 * runs the two gradient products of the mixed-precision backward on it.
 *
 * ComplEx / DistMult, bf16 tables, dim in {128, 256, 512}; anything else:
 * KGE_ERR_UNSUPPORTED (compose kge_score_sp + the loss instead).  Both calls need
 * `workspace_bytes >= kge_ce_workspace_bytes(t, n)` of 256-byte aligned device scratch (no
 * initialisation; stream-ordered use; not shared by concurrent calls).  Not capturable into a
 * hipGraph (the gradient products go through hipBLASLt). */
int64_t kge_ce_workspace_bytes(const kge_tables* t, int64_t n);
int kge_ce_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, kge_index label,
               int64_t n, float* loss_rows, float* lse, void* workspace,
               int64_t workspace_bytes, void* stream);
int kge_ce_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, kge_index label,
               int64_t n, const float* lse, const float* g_rows, float g_scalar, float* g_a,
               float* g_p, float* g_tgt, void* workspace, int64_t workspace_bytes,
               void* stream);

/* The same pair with DENSE query rows (a_rows [n, dim], p_rows [n, rel_dim], row-major, bf16) scored
 * against ALL rows of t->ent: the per-shard step of entity-sharded 1vsAll training:
 * the query rows of a batch come out of the exchange between the shards, t->ent is this rank's shard,
 * `label[i]` is the LOCAL row id of row i's true entity or any value outside [0, t->num_ent) (also a
 * NULL vector) if another shard owns it -- loss_rows[i] is then NaN and lse[i] the shard's log-sum-exp;
 * the caller merges the shards' lse and the owner's score and passes the GLOBAL lse to the backward,
 * which returns this shard's part of the query-row gradients (g_a, g_p: summed over the shards by the
 * caller) and the gradient of its own rows (g_tgt).  Workspace as for kge_ce_fwd. */
int kge_ce_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                   int64_t p_ld, kge_index label, int64_t n, float* loss_rows, float* lse,
                   void* workspace, int64_t workspace_bytes, void* stream);
int kge_ce_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                   int64_t p_ld, kge_index label, int64_t n, const float* lse, const float* g_rows,
                   float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* KvsAll training over an entity-SHARDED table (BASELINE configs[3]; TrainingJobKvsAll._process_subbatch,
 * kge/job/train_KvsAll.py:216-294, with the entity table row-sharded as SURVEY.md 8e lays out): kge_kl_weighted_fwd /
 * _bwd and kge_bce_fwd / _bwd with DENSE query rows (a_rows [n, dim], p_rows [n, rel_dim]: they come out of the
 * exchange between the shards) against ALL rows of t->ent = this rank's shard, whose rows are the GLOBAL entity ids
 * [col_lo, col_lo + t->num_ent).  lbl_col holds GLOBAL ids: labels outside the shard are skipped (another rank adds
 * them), so that
 *   kl:   loss_rows[i] = lse_shard[i] - w_i * sum_{labels of row i IN this shard} score;  the caller merges the shards'
 *         lse (log-sum-exp) and sums the label terms; the backward takes the GLOBAL lse;
 *   bce:  loss_rows[i] = this shard's part of the sum over all entities: the caller adds the shards' values.
 * Gradients as kge_ce_emb_bwd: g_a / g_p = this shard's part of the query-row gradients, g_tgt = the shard's rows. */
int kge_kl_weighted_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                            int64_t p_ld, int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo,
                            const float* label_weight, float* loss_rows, float* lse, void* workspace,
                            int64_t workspace_bytes, void* stream);
int kge_kl_weighted_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows,
                            int64_t p_ld, int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo,
                            const float* label_weight, const float* label_bias /* [n] or NULL */, const float* lse,
                            const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                            int64_t workspace_bytes, void* stream);
int kge_bce_emb_fwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                    int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo, float offset,
                    float* loss_rows, void* workspace, int64_t workspace_bytes, void* stream);
int kge_bce_emb_bwd(const kge_tables* t, int dir, const void* a_rows, int64_t a_ld, const void* p_rows, int64_t p_ld,
                    int64_t n, const int64_t* lbl_rowptr, const int64_t* lbl_col, int64_t col_lo, float offset,
                    const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* Both directions of a 1vsAll batch at once (train_1vsAll.py:64-81 in one pass): rows [0, n) of
 * loss_rows / lse / g_rows are the (s, p, ?) queries with labels o, rows [n, 2n) the (?, p, o)
 * queries with labels s.  One scoring launch for both sides; the gradient products run once
 * over the 2n rows.  kge_ce_sp_po_bwd: g_a [2n, dim] (rows [0, n): gradient of the s rows,
 * [n, 2n): of the o rows), g_p [2n, rel_dim], g_tgt [num_ent, dim], OVERWRITTEN.  Values equal
 * the two one-sided calls up to f32 summation order (the same scores; the log-sum-exp merges a
 * different split of the columns, the products sum over 2n rows).
 * Workspace: kge_ce_sp_po_workspace_bytes(t, n). */
int64_t kge_ce_sp_po_workspace_bytes(const kge_tables* t, int64_t n);
int kge_ce_sp_po_fwd(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                     float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes,
                     void* stream);
int kge_ce_sp_po_bwd(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                     const float* lse, const float* g_rows, float g_scalar, float* g_a,
                     float* g_p, float* g_tgt, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* kge_ce_sp_po_bwd with the scatter-add of the row gradients done by the library:
 * grad_ent [num_ent, dim] and grad_rel [num_rel, rel_dim] (contiguous f32, OVERWRITTEN) receive the
 * complete gradients of both tables -- what autograd accumulates into `.grad` for the reference
 * (the dense target gradient, plus the gathered s / o rows, plus the gathered relation rows). */
int kge_ce_sp_po_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                           const float* lse, const float* g_rows, float g_scalar,
                           float* grad_ent, float* grad_rel, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* The batch loss as ONE device scalar, for a training step that is captured into a hipGraph and replayed
 * (kge_amd/train_graph.py): TrainingJob1vsAll sums the rows and divides by the batch size
 * (kge/job/train_1vsAll.py:64-82, kge/util/loss.py:192-207 reduction "sum"), three more launches and as many
 * autograd nodes around the two calls above.  kge_ce_sp_po_fwd_sum: kge_ce_sp_po_fwd, and in the same launches
 *   loss_sum[0] = scale * scale_dev[0] * sum_i loss_rows[i]      (scale_dev == NULL: factor 1; device float)
 * summed in a fixed order (bitwise reproducible).  kge_ce_sp_po_bwd_accum_sum: kge_ce_sp_po_bwd_accum with the
 * SAME gradient for every row, g = scale * g_dev[0] * scale_dev[0] (NULL: factor 1) -- the upstream gradient of
 * loss_sum and the scale as device scalars: nothing of the step is a host value, so a replay follows both.
 * The workspace's control block (its first 33,024 bytes: flags of the cooperative build, the arrival counter of the
 * summing launch) must be ZERO before
 * the first call with a given workspace; every call leaves them zero.  A control block that was never cleared gives a
 * wrong loss_sum on every call. */
int kge_ce_sp_po_fwd_sum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                         float* loss_rows, float* lse, const float* scale_dev, float scale,
                         float* loss_sum, void* workspace, int64_t workspace_bytes, void* stream);
int kge_ce_sp_po_bwd_accum_sum(const kge_tables* t, kge_index s, kge_index p, kge_index o, int64_t n,
                               const float* lse, const float* g_dev, const float* scale_dev, float scale,
                               float* grad_ent, float* grad_rel, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* KvsAll variant: KL divergence of softmax(score(i, .)) from the row's normalised multi-hot
 * labels, lbl_col[lbl_rowptr[i] .. lbl_rowptr[i+1]) (int64 CSR on the device, entity ids unique
 * per row), y_ij = 1/k_i:
 *   loss_rows[i] = lse[i] - (1/k_i) sum_{j in labels_i} score(i, j) - log k_i     (0 if k_i = 0)
 * = KLDivWithSoftmaxKgeLoss with a label matrix, no label smoothing (kge/util/loss.py:208-213),
 * as TrainingJobKvsAll computes it for sp_ / _po queries (kge/job/train_KvsAll.py:244-294).
 * The label scores use the same operands as the matrix-core kernel (query vector rounded to
 * bf16, exact products, f32 accumulation) in a different f32 summation order.  kge_kl_bwd:
 * gradients of sum_i g_i * loss_rows[i], outputs as kge_ce_bwd.  Same support matrix and
 * workspace as kge_ce_fwd / kge_ce_bwd. */
int kge_kl_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
               const int64_t* lbl_rowptr, const int64_t* lbl_col, float* loss_rows, float* lse,
               void* workspace, int64_t workspace_bytes, void* stream);
int kge_kl_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
               const int64_t* lbl_rowptr, const int64_t* lbl_col, const float* lse,
               const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt,
               void* workspace, int64_t workspace_bytes, void* stream);

/* The pair with a per-row LABEL WEIGHT w_i instead of 1/k_i -- what KvsAll label smoothing
 * (kge/job/train_KvsAll.py:34-49, 262-270: labels = (1 - eps) * labels + 1/E, then normalised) needs
 * from the fused kernels:
 *   loss_rows[i] = lse[i] - w_i * sum_{j in labels_i} score(i, j)              (rows without labels: lse[i])
 *   d / d score(i, j) of sum_i g_i loss_rows[i] = g_i * (softmax_ij - w_i [j in labels_i])
 *   (kge_kl_weighted_bwd with label_bias b != NULL: g_i * (softmax_ij - b_i - w_i [j in labels_i]), i.e. the
 *   gradient of loss_rows[i] - b_i * sum_j score(i, j): the uniform term below, taken inside the kernel)
 * With Z_i = (1 - eps) k_i + 1, a_i = ((1 - eps) + 1/E) / Z_i, b_i = (1/E) / Z_i the smoothed loss of row i is
 *   loss_rows[i] (w_i = a_i - b_i)  -  b_i * sum_j score(i, j)  +  k_i a_i log a_i + (E - k_i) b_i log b_i;
 * the middle term is linear in the entity table (sum_j score(i, j) = score of row i against the table's
 * column sum): kge_amd.model adds it and the constant on the host side of the ABI. */
int kge_kl_weighted_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                        const int64_t* lbl_rowptr, const int64_t* lbl_col, const float* label_weight,
                        float* loss_rows, float* lse, void* workspace, int64_t workspace_bytes,
                        void* stream);
int kge_kl_weighted_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                        const int64_t* lbl_rowptr, const int64_t* lbl_col, const float* label_weight,
                        const float* label_bias /* [n] or NULL */, const float* lse, const float* g_rows,
                        float g_scalar, float* g_a, float* g_p,
                        float* g_tgt, void* workspace, int64_t workspace_bytes, void* stream);

/* Binary cross entropy with logits against the rows' multi-hot labels (CSR as for kge_kl_fwd), summed
 * over ALL entities:  loss_rows[i] = sum_j BCEWithLogits(score(i, j) + offset, y_ij), y_ij = 1 on
 * the labels = BCEWithLogitsKgeLoss with bce_type None (kge/util/loss.py:137-159; `offset` =
 * train.loss_arg), as TrainingJobKvsAll / TrainingJob1vsAll use it with train.loss: bce.
 * kge_bce_bwd: gradients of sum_i g_i * loss_rows[i] (d/d score = sigmoid(score + offset) - y),
 * outputs as kge_ce_bwd.  Same support matrix and workspace as kge_ce_fwd / kge_ce_bwd. */
int kge_bce_fwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                const int64_t* lbl_rowptr, const int64_t* lbl_col, float offset,
                float* loss_rows, void* workspace, int64_t workspace_bytes, void* stream);
int kge_bce_bwd(const kge_tables* t, int dir, kge_index a, kge_index p, int64_t n,
                const int64_t* lbl_rowptr, const int64_t* lbl_col, float offset,
                const float* g_rows, float g_scalar, float* g_a, float* g_p, float* g_tgt,
                void* workspace, int64_t workspace_bytes, void* stream);

/* Both query types of a KvsAll batch, backward, with COMPLETE table gradients.  TrainingJobKvsAll scores the sp_ and
 * the _po queries of a batch one after the other and back-propagates each loss on its own
 * (kge/job/train_KvsAll.py:274-294): two d loss / d score passes, four gradient products, and autograd's index_add and
 * accumulation passes over the [num_ent, dim] gradient between them.  Here each type's pass (that of kge_kl_bwd /
 * kge_bce_bwd, no label smoothing) fills its rows of one gradient matrix and the two products run once over the
 * n_sp + n_po rows: grad_ent [num_ent, dim] and grad_rel [num_rel, rel_dim] (contiguous f32, OVERWRITTEN) receive what
 * `.grad` holds after the reference's two backward calls -- the dense target gradient of both types plus the
 * gathered entity and relation rows' gradients (float atomics: equal up to their order).
 *   sp: a = subjects, p = relations of the sp_ queries; po: a = OBJECTS, p = relations of the _po queries;
 *   lse: kge_kl_fwd's (KGE_LOSS_KL; unused for KGE_LOSS_BCE); g_rows / g_scalar / g_dev: upstream gradient of loss_rows;
 *   offset: kge_bce_fwd's (KGE_LOSS_BCE).  Either side may be empty (n = 0).
 * Workspace: kge_multilabel2_workspace_bytes(t, n_sp, n_po) (0: unsupported tables), 256-byte aligned, zeroed once. */
#define KGE_LOSS_KL 0
#define KGE_LOSS_BCE 1
typedef struct kge_label_queries {
  kge_index a, p;
  int64_t n;
  const int64_t* lbl_rowptr;   /* [n + 1] */
  const int64_t* lbl_col;
  const float* lse;            /* [n] or NULL (bce) */
  const float* g_rows;         /* [n] or NULL: g_scalar (x g_dev[0]) for every row */
  float g_scalar;
  const float* g_dev;          /* [1] device float or NULL: the upstream gradient of a SUMMED loss (a captured step
                                * holds no host value: kge_ce_sp_po_bwd_accum_sum) */
} kge_label_queries;
int64_t kge_multilabel2_workspace_bytes(const kge_tables* t, int64_t n_sp, int64_t n_po);
int kge_multilabel2_bwd_accum(const kge_tables* t, int loss, float offset, const kge_label_queries* sp,
                              const kge_label_queries* po, float* grad_ent, float* grad_rel,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* ---- optimizer step over a table ---------------------- */
/* One dense Adagrad step on `count` contiguous f32 elements (16-byte aligned arrays), in place:
 *   g = grad + weight_decay * param (if weight_decay != 0);  state_sum += g*g;
 *   param += (minus_clr * g) / (sqrt(state_sum) + eps)
 * with minus_clr = -lr / (1 + (step - 1) * lr_decay): torch.optim.Adagrad's update
 * (the optimizer LibKGE creates by default, kge/util/optimizer.py:15-20; config-default.yaml
 * train.optimizer), one pass instead of five element-wise kernels.  bf16_copy != NULL: also
 * writes RNE(param) there (8-byte aligned) -- the table copies the bf16 scoring kernels read. */
int kge_adagrad_step(float* param, const float* grad, float* state_sum, int64_t count,
                     float minus_clr, float weight_decay, float eps, void* bf16_copy,
                     void* stream);

/* kge_adagrad_step on up to KGE_ADAGRAD_MAX_SEGS tables in ONE launch (the entity and the relation table of a
 * training step; kge/job/train.py:471-474 optimizer.step() over all parameters): the same arithmetic per
 * element, one launch latency instead of one per parameter. */
#define KGE_ADAGRAD_MAX_SEGS 8
typedef struct kge_adagrad_seg {
  float* param;
  const float* grad;
  float* state_sum;
  void* bf16_copy;      /* or NULL */
  int64_t count;
  float minus_clr, weight_decay, eps;
} kge_adagrad_seg;
int kge_adagrad_step_multi(const kge_adagrad_seg* segs, int num_segs, void* stream);

/* kge_adagrad_step_multi with the embedders' UNWEIGHTED penalty terms folded into the pass
 * (LookupEmbedder.penalty, kge/model/embedder/lookup_embedder.py:122-147, back-propagated by TrainingJob.run_epoch
 * between the batch and optimizer.step(), kge/job/train.py:417-436): segment j adds the gradient of
 *   kind 1 (regularize lp):          weight / p * sum |x|^p             (p = 1, 2, 3)   -> weight * sign(x) |x|^(p-1)
 *   kind 2 (regularize n3, complex): weight / 3 * sum |z|^3, |z| = sqrt(re^2 + im^2 + 1e-14), im = col + row_dim / 2
 * to grad before the update (grad itself is not written), and adds sum |x|^p (sum |z|^3) of the PRE-step parameters
 * to *value (a device double the caller zeroed; the term the reference's trace shows is weight / p times it).
 * `weight` is the gradient's factor: regularize_weight, doubled for an entity embedder shared by the subject and
 * object slot (kge_model.py:620-625).  kind 0: the plain step.  kind 2 needs count % row_dim == 0 and
 * row_dim % 8 == 0.  pens == NULL: kge_adagrad_step_multi. */
typedef struct kge_penalty_seg {
  int32_t kind, p;
  float weight;
  int64_t row_dim;
  double* value;        /* [1] device, or NULL for kind 0 */
} kge_penalty_seg;
int kge_adagrad_step_multi_penalty(const kge_adagrad_seg* segs, const kge_penalty_seg* pens, int num_segs,
                                   void* stream);

/* One dense Adam step (torch.optim.Adam, amsgrad / maximize off) on `count` contiguous f32 elements,
 * in place, one pass:  g = grad + weight_decay * param (if != 0);  exp_avg += (g - exp_avg)(1 - beta1);
 * exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) g g;
 * param -= step_size * exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps),
 * step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the caller
 * (kge/util/optimizer.py:15-20 with train.optimizer.default.type: Adam).  bf16_copy as above. */
int kge_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                  float step_size, float bias_correction2_sqrt, double beta1, double beta2,
                  float weight_decay, float eps, void* bf16_copy, void* stream);

/* Row-sparse Adagrad step: only the `num_rows` listed rows (unique ids, int64) of a [*, dim] table
 * have a gradient, grad_rows[r, :] belonging to row rows[r] -- torch.optim.Adagrad's update for the
 * sparse gradients of lookup_embedder.sparse: True (kge/model/embedder/lookup_embedder.yaml:78-81):
 *   state_sum[row] += g*g;  param[row] += minus_clr * g / (sqrt(state_sum[row]) + eps).
 * Untouched rows are neither read nor written.  bf16_copy: the same rows of the bf16 table copy. */
int kge_adagrad_step_rows(float* param, int64_t param_ld, const float* grad_rows, int64_t grad_ld,
                          float* state_sum, int64_t sum_ld, const int64_t* rows, int64_t num_rows,
                          int64_t dim, float minus_clr, float eps, void* bf16_copy, int64_t copy_ld,
                          void* stream);

/* ---- backward (autograd twins) ------------------------------------------ */
/* All gradients are f32 and OVERWRITTEN; tables/embeddings must be f32, except
 * kge_score_pairs_bwd for ComplEx/DistMult, which also takes bf16 tables (mixed-precision
 * training: gout rounded to bf16, both products on the bf16 matrix cores, f32 accumulation)
 * and then needs `workspace_bytes >= kge_score_bwd_workspace_bytes(t, n, m)` of 256-byte
 * aligned device scratch (0 / NULL for f32 tables).
 * `scores` is the forward output (needed by TransE/RotatE with l_norm != 1 to
 * recover the distance; may be NULL otherwise).
 *
 * kge_score_pairs_bwd: gradients of sum_ij gout[i,j]*score(i,j) for
 * kge_score_sp / kge_score_po (dir = KGE_SP_ / KGE_PO_):
 *   g_a   [n, dim]      grad of the gathered entity query rows (s for SP_, o for PO_)
 *   g_p   [n, rel_dim]  grad of the gathered relation rows
 *   g_tgt [m, dim]      grad of the target rows (dense; caller scatter-adds)   */
int64_t kge_score_bwd_workspace_bytes(const kge_tables* t, int64_t n, int64_t m);
int kge_score_pairs_bwd(const kge_tables* t, int dir, kge_index a, kge_index p,
                        int64_t n, kge_index targets, int64_t m,
                        const float* gout, int64_t ldg, const float* scores,
                        int64_t lds, float* g_a, float* g_p, float* g_tgt,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Gradients of sum_i gout[i]*score(s_i,p_i,o_i): g_s,g_o [n,dim], g_p [n,rel_dim]. */
int kge_score_spo_bwd(const kge_tables* t, kge_index s, kge_index p,
                      kge_index o, int64_t n, const float* gout,
                      const float* scores, float* g_s, float* g_p, float* g_o,
                      void* stream);

/* The same gradients ACCUMULATED (float atomics) into dense table gradients grad_ent [num_ent,
 * dim] and grad_rel [num_rel, rel_dim] (caller zeroes or pre-loads them): what autograd's
 * scatter-add of the gathered rows produces, without the three [n, dim] row-gradient tensors.
 * Runs of equal s / p indices (negative sampling: n*K triples, s and p repeated K times in a
 * row, kge/util/sampler.py:291-306) are summed in registers first.  dim <= 1024. */
int kge_score_spo_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                            int64_t n, const float* gout, const float* scores,
                            float* grad_ent, int64_t grad_ent_ld, float* grad_rel,
                            int64_t grad_rel_ld, void* stream);

/* Backward of kge_score_neg (replaces autograd through BatchNegativeSample.score,
 * kge/util/sampler.py:263-306, called at kge/job/train_negative_sampling.py:142-163): gradients
 * of sum_{i,k} gout[i*ldg + k] * score(triple i with `slot` replaced by neg[i*neg_ld + k])
 * ACCUMULATED into the dense table gradients like kge_score_spo_bwd_accum.  The relation row and
 * the uncorrupted entity row of a positive are read once and their gradients summed in registers
 * over 64 negatives; only the corrupted rows stream.  `scores` = the forward output [n, num_neg]
 * (row pitch lds; needed for TransE / RotatE with l_norm != 1, may be NULL otherwise).
 * f32 tables, dim <= 1024. */
int kge_score_neg_bwd_accum(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                            int64_t n, int slot, const void* neg, int32_t neg_itype,
                            int64_t neg_ld, int64_t num_neg, const float* gout, int64_t ldg,
                            const float* scores, int64_t lds, float* grad_ent,
                            int64_t grad_ent_ld, float* grad_rel, int64_t grad_rel_ld,
                            void* stream);
/* The same backward with the [n, num_neg] occurrences SORTED by the entity they corrupt: `order` = int64 [n * num_neg]
 * (device), the positions i * num_neg + k in ascending neg[i, k] (any order among equal ids: kge_neg_order above).  kge_score_neg_bwd_accum issues one float atomic per element and occurrence into
 * grad_ent (262 M per slot at the WN18RR shape with 512 x 1000 negatives: 73 % of that training step); here a wave keeps
 * the running sum of an entity's gradient row in registers over consecutive occurrences and writes it where the entity
 * changes: ~num_entities / 32 + n * num_neg / 32 row flushes instead of n * num_neg.  Same sums in another order
 * (float rounding apart).  Worth the sort from a few occurrences per entity on.  rel_scratch (may be NULL): RotatE only,
 * num_rel x 2 rel_dim floats of scratch for the relations' cos / sin (computed once per call instead of per occurrence).  kge/util/sampler.py:263-306 backward,
 * called at kge/job/train_negative_sampling.py:161-163. */
/* `order` for the call below: a counting sort of the [n, num_neg] samples by entity id in two calls, no host wait
 * (capturable).  order == NULL: the histogram -- cursor[e] += the number of samples with id e (int64 [num_ent], zeroed
 * by the caller).  The caller turns it into exclusive prefix sums (a cumsum: the torch front end's), then order != NULL:
 * the scatter -- on return order[cursor_in[e] .. cursor_in[e + 1]) holds the positions i * num_neg + k of entity e's
 * samples in any order; cursor is clobbered. */
int kge_neg_order(const void* neg, int32_t neg_itype, int64_t neg_ld, int64_t n, int64_t num_neg,
                  int64_t num_ent, int64_t* cursor, int64_t* order, void* stream);
int kge_score_neg_bwd_accum_sorted(const kge_tables* t, kge_index s, kge_index p, kge_index o,
                                   int64_t n, int slot, const void* neg, int32_t neg_itype,
                                   int64_t neg_ld, int64_t num_neg, const int64_t* order,
                                   const float* gout, int64_t ldg, const float* scores, int64_t lds,
                                   float* grad_ent, int64_t grad_ent_ld, float* grad_rel,
                                   int64_t grad_rel_ld, float* rel_scratch, void* stream);

/* Backward of kge_score_emb (dense embeddings).  SPO: g_s,g_o [n,dim], g_p [n,rel_dim].
 * SP_: g_s [n,dim], g_p [n,rel_dim], g_o [m,dim].  PO_: g_o [n,dim], g_p, g_s [m,dim]. */
int kge_score_emb_bwd(const kge_tables* t, int combine, const void* s_emb,
                      int64_t s_ld, const void* p_emb, int64_t p_ld,
                      const void* o_emb, int64_t o_ld, int64_t n, int64_t m,
                      const float* gout, int64_t ldg, const float* scores,
                      int64_t lds, float* g_s, float* g_p, float* g_o,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KGE_AMD_H */
