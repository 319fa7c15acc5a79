/*
 * kge_oracle.c -- CPU restatement of the LibKGE scoring + rank-count hot path.
 *
 * TEST INFRASTRUCTURE.  This file is the parity oracle for the HIP kernels in
 * kge_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; nothing under kge_amd/ links or imports it.
 *
 * Parity pin: the oracle is checked against golden vectors produced by the
 * live reference (tests/golden/make_golden.py imports /root/reference and runs
 * the reference's own KgeModel.score_* and EntityRankingJob code); see
 * tests/test_oracle_golden.py and tests/test_oracle_vs_reference.py.
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference tree).  The reference computes with torch ops whose reduction
 * order is implementation defined; this restatement fixes one order ("canonical
 * arithmetic") so that the GPU kernels can be compared
 * bit for bit:
 *   - f32 everywhere, no FMA contraction unless written as fmaf();
 *   - sp_/_po ("pair") scores: one sequential chain per (query,target) pair
 *     over the complex index c (two-half interleave, see pair_*());
 *   - spo scores: 64 strided partial sums (lane = (k/8)%64) + xor butterfly;
 *   - bf16 tables are widened exactly to f32; the ComplEx/DistMult bf16 pair
 *     path rounds the query vector q to bf16 (RNE) like a bf16 GEMM operand.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { KO_COMPLEX = 0, KO_DISTMULT = 1, KO_TRANSE = 2, KO_ROTATE = 3 };
enum { KO_F32 = 0, KO_BF16 = 1 };
enum { KO_SPLIT_QUERY = 1 }; /* bit of ko_tables.reserved: the split-query semantics of the bf16 pair path */
enum { KO_SP = 1, KO_PO = 2 };

typedef struct ko_tables {
  const void* ent;
  const void* rel;
  int32_t dtype;
  int32_t scorer;
  int64_t num_ent, num_rel, dim, rel_dim, ent_ld, rel_ld;
  float l_norm;
  int32_t reserved;
} ko_tables;

/* ---- bf16 <-> f32 ------------------------------------------------------- */
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
uint16_t ko_f32_to_bf16(float f) { return f32_to_bf16_rne(f); }
float ko_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }

static inline float ld_elem(const void* base, int dtype, int64_t off) {
  return dtype == KO_BF16 ? bf16_to_f32(((const uint16_t*)base)[off])
                          : ((const float*)base)[off];
}
static void load_row(const void* base, int dtype, int64_t ld, int64_t row,
                     int64_t dim, float* dst) {
  for (int64_t k = 0; k < dim; ++k) dst[k] = ld_elem(base, dtype, row * ld + k);
}
static inline int64_t idx_at(const void* p, int itype, int64_t stride, int64_t i) {
  if (!p) return i;
  return itype ? ((const int64_t*)p)[i * stride] : (int64_t)((const int32_t*)p)[i * stride];
}

/* ---- canonical sin/cos (rotate.py:28 torch.cos/torch.sin) ----------------
 * Cody-Waite reduction by pi/2 + cephes-style minimax polynomials, written
 * with explicit fmaf so CPU and GPU produce identical bits. */
void ko_sincosf(float x, float* sn, float* cs) {
  const float TWO_OVER_PI = 0.63661977236758134308f;
  const float P1 = 1.5707855224609375f;        /* pi/2 split in three parts */
  const float P2 = 1.0804334124e-5f;
  const float P3 = 6.0770999344e-11f;
  float j = rintf(x * TWO_OVER_PI);
  float r = fmaf(-j, P1, x);
  r = fmaf(-j, P2, r);
  r = fmaf(-j, P3, r);
  float z = r * r;
  /* sin(r) */
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float s = fmaf(ps * z, r, r);
  /* cos(r) */
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  int q = (int)j & 3;
  float so, co;
  switch (q) {
    case 0: so = s; co = c; break;
    case 1: so = c; co = -s; break;
    case 2: so = -s; co = -c; break;
    default: so = -c; co = s; break;
  }
  *sn = so;
  *cs = co;
}

/* ---- query vector q(i) ----------------------------------------------------
 * dir SP: from subject row a=s and relation row r; dir PO: from object row a=o.
 *   DistMult  distmult.py:17-21   q = a*r
 *   ComplEx   complex.py:30-39    SP: q = s (x) r, PO: q = conj(r) (x) o
 *   TransE    transe.py:19-34     SP: q = s+r,   PO: q = o-r
 *   RotatE    rotate.py:43-64     SP: q = s (x) e^{i th}, PO: q = e^{-i th} (x) o
 * Products are rounded individually (torch elementwise semantics). */
static void build_q(int scorer, int dir, const float* a, const float* r,
                    int64_t d, float* q) {
  int64_t h = d / 2;
  switch (scorer) {
    case KO_DISTMULT:
      for (int64_t k = 0; k < d; ++k) q[k] = a[k] * r[k];
      break;
    case KO_TRANSE:
      for (int64_t k = 0; k < d; ++k) q[k] = (dir == KO_SP) ? (a[k] + r[k]) : (a[k] - r[k]);
      break;
    case KO_COMPLEX:
      for (int64_t c = 0; c < h; ++c) {
        float are = a[c], aim = a[h + c], rre = r[c], rim = r[h + c];
        if (dir == KO_SP) {
          q[c] = are * rre - aim * rim;
          q[h + c] = aim * rre + are * rim;
        } else {
          q[c] = rre * are + rim * aim;
          q[h + c] = rre * aim - rim * are;
        }
      }
      break;
    case KO_ROTATE:
      for (int64_t c = 0; c < h; ++c) {
        float sn, cs;
        ko_sincosf(r[c], &sn, &cs);
        float are = a[c], aim = a[h + c];
        if (dir == KO_SP) {
          q[c] = are * cs - aim * sn;
          q[h + c] = are * sn + aim * cs;
        } else {
          q[c] = cs * are + sn * aim;
          q[h + c] = cs * aim - sn * are;
        }
      }
      break;
  }
}

/* ---- canonical pair reduction (one query vector vs one target row) --------
 * Two-half interleave over c in [0,hh): element c of the first half, then
 * element hh+c of the second half (hh = ceil(d/2)).  For ComplEx/RotatE these
 * are the real and imaginary part of complex coordinate c. */
static float pair_score(int scorer, float l_norm, const float* q, const float* t,
                        int64_t d) {
  int64_t hh = (d + 1) / 2;
  float acc = 0.0f;
  if (scorer == KO_COMPLEX || scorer == KO_DISTMULT) {
    for (int64_t c = 0; c < hh; ++c) {
      acc = fmaf(q[c], t[c], acc);
      if (hh + c < d) acc = fmaf(q[hh + c], t[hh + c], acc);
    }
    return acc;
  }
  if (scorer == KO_TRANSE) {
    /* transe.py:22-34: -cdist(q, t, p)  (no eps on this path) */
    for (int64_t c = 0; c < hh; ++c) {
      float d0 = q[c] - t[c];
      if (l_norm == 1.0f) acc = acc + fabsf(d0);
      else if (l_norm == 2.0f) acc = fmaf(d0, d0, acc);
      else acc = acc + powf(fabsf(d0), l_norm);
      if (hh + c < d) {
        float d1 = q[hh + c] - t[hh + c];
        if (l_norm == 1.0f) acc = acc + fabsf(d1);
        else if (l_norm == 2.0f) acc = fmaf(d1, d1, acc);
        else acc = acc + powf(fabsf(d1), l_norm);
      }
    }
  } else { /* KO_ROTATE rotate.py:48-52,198-213 */
    for (int64_t c = 0; c < hh; ++c) {
      float dre = q[c] - t[c], dim_ = q[hh + c] - t[hh + c];
      float ab = sqrtf(fmaf(dim_, dim_, dre * dre));
      if (l_norm == 1.0f) acc = acc + ab;
      else if (l_norm == 2.0f) acc = fmaf(ab, ab, acc);
      else acc = acc + powf(ab, l_norm);
    }
  }
  if (l_norm == 1.0f) return -acc;
  if (l_norm == 2.0f) return -sqrtf(acc);
  return -powf(acc, 1.0f / l_norm);
}

/* ---- KgeModel.score_sp / score_po (kge_model.py:682-725) ----------------- */
int ko_score_pairs(const ko_tables* t, int dir, const void* a_idx, int a_itype,
                   int64_t a_stride, const void* p_idx, int p_itype,
                   int64_t p_stride, int64_t n, const void* tgt_idx, int t_itype,
                   int64_t t_stride, int64_t m, float* out, int64_t ldo) {
  int64_t d = t->dim, dr = t->rel_dim;
  float* a = (float*)malloc(sizeof(float) * (size_t)(2 * d + dr + 8));
  float *r = a + d, *q = r + dr;
  float* ql = (float*)malloc(sizeof(float) * (size_t)(d + 8));
  float* T = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m * d : 1));
  int bf16_q = (t->dtype == KO_BF16) &&
               (t->scorer == KO_COMPLEX || t->scorer == KO_DISTMULT);
  for (int64_t j = 0; j < m; ++j)
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(tgt_idx, t_itype, t_stride, j), d, T + j * d);
  for (int64_t i = 0; i < n; ++i) {
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(a_idx, a_itype, a_stride, i), d, a);
    load_row(t->rel, t->dtype, t->rel_ld, idx_at(p_idx, p_itype, p_stride, i), dr, r);
    build_q(t->scorer, dir, a, r, d, q);
    if (bf16_q && (t->reserved & KO_SPLIT_QUERY)) {
      /* KGE_FLAG_SPLIT_QUERY (include/kge_amd.h): q = q_hi + q_lo, two bf16 pieces, each with its own
       * canonical f32 chain, one f32 add -- f32 arithmetic on the bf16 table values up to the
       * 2^-17 relative residue of q and the summation order (SURVEY.md 8(c) gate 4) */
      for (int64_t k = 0; k < d; ++k) {
        float hi = bf16_to_f32(f32_to_bf16_rne(q[k]));
        ql[k] = bf16_to_f32(f32_to_bf16_rne(q[k] - hi));
        q[k] = hi;
      }
      for (int64_t j = 0; j < m; ++j)
        out[i * ldo + j] = pair_score(t->scorer, t->l_norm, q, T + j * d, d) +
                           pair_score(t->scorer, t->l_norm, ql, T + j * d, d);
      continue;
    }
    if (bf16_q)
      for (int64_t k = 0; k < d; ++k) q[k] = bf16_to_f32(f32_to_bf16_rne(q[k]));
    for (int64_t j = 0; j < m; ++j)
      out[i * ldo + j] = pair_score(t->scorer, t->l_norm, q, T + j * d, d);
  }
  free(T);
  free(a);
  free(ql);
  return 0;
}

/* ---- canonical row reduction for spo: 64 strided partials + butterfly ----- */
static float butterfly64(float* P) {
  for (int off = 32; off >= 1; off >>= 1) {
    float T[64];
    for (int l = 0; l < 64; ++l) T[l] = P[l] + P[l ^ off];
    memcpy(P, T, sizeof(T));
  }
  return P[0];
}

/* KgeModel.score_spo (kge_model.py:663-680) + scorer "spo" branches:
 * complex.py:35, distmult.py:15, transe.py:18 (pairwise_distance, eps=1e-6),
 * rotate.py:30-42. */
static float spo_score(int scorer, float l_norm, const float* s, const float* r,
                       const float* o, int64_t d) {
  float P[64];
  for (int l = 0; l < 64; ++l) P[l] = 0.0f;
  int64_t h = d / 2;
  if (scorer == KO_DISTMULT) {
    for (int64_t k = 0; k < d; ++k) {
      int l = (int)((k / 8) % 64);
      P[l] = fmaf(s[k] * r[k], o[k], P[l]);
    }
    return butterfly64(P);
  }
  if (scorer == KO_COMPLEX) {
    for (int64_t c = 0; c < h; ++c) {
      int l = (int)((c / 8) % 64);
      float qre = s[c] * r[c] - s[h + c] * r[h + c];
      float qim = s[h + c] * r[c] + s[c] * r[h + c];
      P[l] = fmaf(qre, o[c], P[l]);
      P[l] = fmaf(qim, o[h + c], P[l]);
    }
    return butterfly64(P);
  }
  if (scorer == KO_TRANSE) {
    for (int64_t k = 0; k < d; ++k) {
      int l = (int)((k / 8) % 64);
      float df = ((s[k] + r[k]) - o[k]) + 1e-6f; /* F.pairwise_distance eps */
      if (l_norm == 1.0f) P[l] = P[l] + fabsf(df);
      else if (l_norm == 2.0f) P[l] = fmaf(df, df, P[l]);
      else P[l] = P[l] + powf(fabsf(df), l_norm);
    }
  } else { /* ROTATE */
    for (int64_t c = 0; c < h; ++c) {
      int l = (int)((c / 8) % 64);
      float sn, cs;
      ko_sincosf(r[c], &sn, &cs);
      float qre = s[c] * cs - s[h + c] * sn;
      float qim = s[c] * sn + s[h + c] * cs;
      float dre = qre - o[c], dim_ = qim - o[h + c];
      float ab = sqrtf(fmaf(dim_, dim_, dre * dre));
      if (l_norm == 1.0f) P[l] = P[l] + ab;
      else if (l_norm == 2.0f) P[l] = fmaf(ab, ab, P[l]);
      else P[l] = P[l] + powf(ab, l_norm);
    }
  }
  float acc = butterfly64(P);
  if (l_norm == 1.0f) return -acc;
  if (l_norm == 2.0f) return -sqrtf(acc);
  return -powf(acc, 1.0f / l_norm);
}

int ko_score_spo(const ko_tables* t, const void* s_idx, int s_itype, int64_t s_stride,
                 const void* p_idx, int p_itype, int64_t p_stride, const void* o_idx,
                 int o_itype, int64_t o_stride, int64_t n, float* out) {
  int64_t d = t->dim, dr = t->rel_dim;
  float* s = (float*)malloc(sizeof(float) * (size_t)(2 * d + dr + 8));
  float *o = s + d, *r = o + d;
  for (int64_t i = 0; i < n; ++i) {
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(s_idx, s_itype, s_stride, i), d, s);
    load_row(t->rel, t->dtype, t->rel_ld, idx_at(p_idx, p_itype, p_stride, i), dr, r);
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(o_idx, o_itype, o_stride, i), d, o);
    out[i] = spo_score(t->scorer, t->l_norm, s, r, o, d);
  }
  free(s);
  return 0;
}

/* BatchNegativeSample.score, implementation "triple" (sampler.py:291-306):
 * triple i with slot (0=s, 2=o) replaced by each of its K negatives, scored
 * with score_spo. */
int ko_score_neg(const ko_tables* t, const void* s_idx, int s_itype, int64_t s_stride,
                 const void* p_idx, int p_itype, int64_t p_stride, const void* o_idx,
                 int o_itype, int64_t o_stride, int64_t n, int slot, const void* neg,
                 int neg_itype, int64_t neg_ld, int64_t K, float* out, int64_t ldo) {
  int64_t d = t->dim, dr = t->rel_dim;
  float* s = (float*)malloc(sizeof(float) * (size_t)(3 * d + dr + 8));
  float *o = s + d, *r = o + d, *x = r + dr;
  for (int64_t i = 0; i < n; ++i) {
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(s_idx, s_itype, s_stride, i), d, s);
    load_row(t->rel, t->dtype, t->rel_ld, idx_at(p_idx, p_itype, p_stride, i), dr, r);
    load_row(t->ent, t->dtype, t->ent_ld, idx_at(o_idx, o_itype, o_stride, i), d, o);
    for (int64_t k = 0; k < K; ++k) {
      load_row(t->ent, t->dtype, t->ent_ld, idx_at(neg, neg_itype, 1, i * neg_ld + k), d, x);
      out[i * ldo + k] = (slot == 0) ? spo_score(t->scorer, t->l_norm, x, r, o, d)
                                     : spo_score(t->scorer, t->l_norm, s, r, x, d);
    }
  }
  free(s);
  return 0;
}

/* ---- EntityRankingJob._filter_and_rank / _get_ranks_and_num_ties ----------
 * eval_entity_ranking.py:533-596.  Filtering subtracts +inf from the labelled
 * columns (:565-566) so they end up -inf (or NaN -> -inf, :583-586).
 * torch.isclose(x, t, rtol, atol) in f32:
 *   close = (x == t) | (isfinite(|x-t|) & (|x-t| <= atol + |rtol*t|)). */
static inline int is_close(float x, float t, float atol, float rtol) {
  if (x == t) return 1;
  float err = fabsf(x - t);
  float allowed = atol + fabsf(rtol * t);
  return isfinite(err) && err <= allowed;
}

int ko_rank_counts(const float* scores, int64_t lds, int64_t n, int64_t c,
                   const float* true_scores, const int64_t* lbl_rowptr,
                   const int64_t* lbl_col, int64_t col_offset, const int64_t* true_col,
                   float atol, float rtol, int64_t* rank, int64_t* ties) {
  unsigned char* filt = (unsigned char*)malloc((size_t)(c > 0 ? c : 1));
  for (int64_t i = 0; i < n; ++i) {
    memset(filt, 0, (size_t)c);
    if (lbl_rowptr) {
      for (int64_t e = lbl_rowptr[i]; e < lbl_rowptr[i + 1]; ++e) {
        int64_t g = lbl_col[e];
        if (true_col && g == true_col[i]) continue; /* :288-290 remove the positive */
        int64_t j = g - col_offset;
        if (j >= 0 && j < c) filt[j] = 1;
      }
    }
    float t = true_scores[i];
    if (isnan(t)) t = -INFINITY;
    int64_t rk = 0, ti = 0;
    for (int64_t j = 0; j < c; ++j) {
      float x = scores[i * lds + j];
      if (filt[j] || isnan(x)) x = -INFINITY;
      int cl = is_close(x, t, atol, rtol);
      ti += cl;
      rk += (x > t) && !cl;
    }
    rank[i] += rk;
    ties[i] += ti;
  }
  free(filt);
  return 0;
}
