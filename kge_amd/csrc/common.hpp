// common.hpp -- device helpers shared by the gfx950 kernels of the KGE scoring engine.
//
// Canonical arithmetic: every f32 operation sequence below is
// specified so that oracle/kge_oracle.c reproduces it bit for bit.  The translation
// unit is compiled with -ffp-contract=off; fused multiply-adds appear only where
// __builtin_fmaf is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kge_amd.h"
#include "switches.hpp"

namespace kge {

// control block at the start of the cooperative-build workspace: 512 flag lines of 64 B + the degraded word
constexpr long long PAIRS_WS_CTRL_BYTES = 512 * 8 * 8 + 256;
// inside its last 256 bytes: +0 the degraded word (8 B), +64 the arrival counter of ce_combine_kernel's sum (4 B)
constexpr long long PAIRS_WS_CE_COUNTER_OFF = 512 * 8 * 8 + 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Index vector as passed over the C ABI, flattened for kernel arguments.
struct Index {
  const void* ptr;   // NULL = identity
  long long stride;  // elements
  int itype;         // 0 = int32, 1 = int64
};

__host__ inline Index make_index(const kge_index& k) {
  Index r;
  r.ptr = k.ptr;
  r.stride = k.stride;
  r.itype = k.itype;
  return r;
}

__device__ __forceinline__ long long index_at(const Index& ix, long long i) {
  if (ix.ptr == nullptr) return i;
  if (ix.itype) return ((const long long*)ix.ptr)[i * ix.stride];
  return (long long)((const int*)ix.ptr)[i * ix.stride];
}

// One kernel operand: rows `idx[i]` of a row-major matrix (table or dense embeddings).
struct Operand {
  const void* base;
  long long ld;  // elements
  Index idx;
};

// One gather job of embed.hip: out[i, :] = table[idx[i], :]
struct EmbedJob {
  const void* table;
  long long ld;  // elements
  Index idx;
  long long n;
  void* out;
  long long ldo;  // elements
};

// One row move of the sharded exchange (embed.hip): out[i] = table[map(idx[i], i)].
struct ShardJob {
  const void* table;
  long long ld;  // elements
  Index idx;
  long long n;
  void* out;
  long long ldo;             // elements
  long long sub, hi;         // div == 0: row = clamp(id - sub, 0, hi)
  long long div, mul, add;   // div > 0:  row = (id / div) * mul + add + i
};

// Flattened table descriptor for kernels.
struct Tables {
  const void* ent;
  const void* rel;
  long long num_ent, num_rel;
  int dim, rel_dim;
  long long ent_ld, rel_ld;
  float l_norm;
};

__host__ inline Tables make_tables(const kge_tables* t) {
  Tables r;
  r.ent = t->ent;
  r.rel = t->rel;
  r.num_ent = t->num_ent;
  r.num_rel = t->num_rel;
  r.dim = (int)t->dim;
  r.rel_dim = (int)t->rel_dim;
  r.ent_ld = t->ent_ld;
  r.rel_ld = t->rel_ld;
  r.l_norm = t->l_norm;
  return r;
}

// ---- bf16 ---------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even, NaN -> quiet NaN (same bit recipe as the oracle)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// ---- element loads (T = float or unsigned short(bf16)) --------------------------
template <typename T>
__device__ __forceinline__ float ld1(const T* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<unsigned short>(const unsigned short* p) {
  return bf16_to_f32(*p);
}

// 4 consecutive elements -> 4 floats (pointer must be 16 B (f32) / 8 B (bf16) aligned)
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 ld4<unsigned short>(const unsigned short* p) {
  u32x2 v = *reinterpret_cast<const u32x2*>(p);
  f32x4 r;
  r[0] = __uint_as_float(v[0] << 16);
  r[1] = __uint_as_float(v[0] & 0xffff0000u);
  r[2] = __uint_as_float(v[1] << 16);
  r[3] = __uint_as_float(v[1] & 0xffff0000u);
  return r;
}

// 8 consecutive elements -> 8 floats (32 B (f32) / 16 B (bf16) aligned)
struct f32x8 {
  float v[8];
};
template <typename T>
__device__ __forceinline__ f32x8 ld8(const T* p);
template <>
__device__ __forceinline__ f32x8 ld8<float>(const float* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
template <>
__device__ __forceinline__ f32x8 ld8<unsigned short>(const unsigned short* p) {
  u32x4 v = *reinterpret_cast<const u32x4*>(p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __uint_as_float(v[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
  return r;
}

// ---- canonical sin/cos (oracle: ko_sincosf) ---------------------------------------
__device__ __forceinline__ void sincos_canon(float x, float& sn, float& cs) {
  const float TWO_OVER_PI = 0.63661977236758134308f;
  const float P1 = 1.5707855224609375f;
  const float P2 = 1.0804334124e-5f;
  const float P3 = 6.0770999344e-11f;
  float j = __builtin_rintf(x * TWO_OVER_PI);
  float r = __builtin_fmaf(-j, P1, x);
  r = __builtin_fmaf(-j, P2, r);
  r = __builtin_fmaf(-j, P3, r);
  float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  float s = __builtin_fmaf(ps * z, r, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  float c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  int q = ((int)j) & 3;
  float so = (q & 1) ? c : s;
  float co = (q & 1) ? s : c;
  so = (q & 2) ? -so : so;
  co = ((q + 1) & 2) ? -co : co;
  sn = so;
  cs = co;
}

// IEEE-correct sqrt (the oracle uses glibc sqrtf, correctly rounded).
__device__ __forceinline__ float sqrt_rn(float x) { return __fsqrt_rn(x); }

// The same value in fewer issue slots, for the kernels that take one square root per scored coordinate (RotatE): the
// compiler's sequence for a correctly rounded float sqrt (no fast-math) is input scaling + v_sqrt_f32 + two one-ulp
// candidates with their residuals + selects + rescaling -- ~16 operations besides the quarter-rate v_sqrt.  Here:
// v_rsq_f32 (1 ulp), one coupled Newton step on g ~ sqrt(x) and h ~ 1 / (2 sqrt(x)), and Markstein's final correction
// g + h (x - g g) with the residual from ONE fma: seven operations + the quarter-rate v_rsq.  Correct rounding holds for
// x in [2^-96, 2^126] -- checked EXHAUSTIVELY against the compiler's sequence over all 2^32 bit patterns on the device
// (tools/ubench/sqrt_exhaustive.hip, 0 mismatches; profiles/r4_sqrt_exhaustive.txt); outside that range (zeros,
// denormal-sized squares, huge values, inf, NaN, negatives: the residual would leave the normal range) the IEEE form.
constexpr float SQRT_FAST_LO = 0x1p-96f, SQRT_FAST_HI = 0x1p+126f;
// (the range check is the caller's: the pair kernel checks the minimum and maximum of a 4 x 4 micro-tile once)
__device__ __forceinline__ float sqrt_rn_core(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-g, h, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_rn_fast(float x) {
  if (__builtin_expect(!(x >= SQRT_FAST_LO && x <= SQRT_FAST_HI), 0)) return __builtin_sqrtf(x);
  return sqrt_rn_core(x);
}

enum { NORM_L1 = 1, NORM_L2 = 2, NORM_LP = 3 };
__host__ inline int norm_mode(float p) { return p == 1.0f ? NORM_L1 : (p == 2.0f ? NORM_L2 : NORM_LP); }

// accumulate one non-negative distance component
template <int NORM>
__device__ __forceinline__ float norm_acc(float acc, float a, float p) {
  if (NORM == NORM_L1) return acc + a;
  if (NORM == NORM_L2) return __builtin_fmaf(a, a, acc);
  return acc + powf(a, p);  // general p: libm pow, tolerance-level only
}

// q halves for 4 coordinates.  a0/a1: entity row halves, r0/r1: relation row halves
// (RotatE: r0 = phases).  dir: KGE_SP_ or KGE_PO_.
template <int SCORER>
__device__ __forceinline__ void build_q4(int dir, const f32x4& a0, const f32x4& a1,
                                         const f32x4& r0, const f32x4& r1, f32x4& q0,
                                         f32x4& q1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (SCORER == KGE_DISTMULT) {
      q0[i] = a0[i] * r0[i];
      q1[i] = a1[i] * r1[i];
    } else if (SCORER == KGE_TRANSE) {
      q0[i] = (dir == KGE_SP_) ? (a0[i] + r0[i]) : (a0[i] - r0[i]);
      q1[i] = (dir == KGE_SP_) ? (a1[i] + r1[i]) : (a1[i] - r1[i]);
    } else if (SCORER == KGE_COMPLEX) {
      if (dir == KGE_SP_) {
        q0[i] = a0[i] * r0[i] - a1[i] * r1[i];
        q1[i] = a1[i] * r0[i] + a0[i] * r1[i];
      } else {
        q0[i] = r0[i] * a0[i] + r1[i] * a1[i];
        q1[i] = r0[i] * a1[i] - r1[i] * a0[i];
      }
    } else {  // ROTATE
      float sn, cs;
      sincos_canon(r0[i], sn, cs);
      if (dir == KGE_SP_) {
        q0[i] = a0[i] * cs - a1[i] * sn;
        q1[i] = a0[i] * sn + a1[i] * cs;
      } else {
        q0[i] = cs * a0[i] + sn * a1[i];
        q1[i] = cs * a1[i] - sn * a0[i];
      }
    }
  }
}



// ---- helpers shared by the bf16 matrix-core pair kernels (score_pairs_bf16_v2/v3/v4.hip) ----

// RNE pack of two f32 into one dword of two bf16 (lo in bits 0-15)
__device__ __forceinline__ unsigned int bf16_pack(float lo, float hi) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// Query vector of ComplEx / DistMult on bf16 tables, two coordinates per dword: (a0,a1) entity
// halves, (r0,r1) relation halves -> (q0,q1).  Every product and sum is rounded on its own
// (no contraction: -ffp-contract=off), then RNE to bf16: the bits of build_q() of the oracle.
template <int SCORER>
__device__ __forceinline__ void bf16_qpair(int dir, unsigned int a0, unsigned int a1,
                                           unsigned int r0, unsigned int r1, unsigned int& q0,
                                           unsigned int& q1) {
  const float a0l = __uint_as_float(a0 << 16), a0h = __uint_as_float(a0 & 0xffff0000u);
  const float a1l = __uint_as_float(a1 << 16), a1h = __uint_as_float(a1 & 0xffff0000u);
  const float r0l = __uint_as_float(r0 << 16), r0h = __uint_as_float(r0 & 0xffff0000u);
  const float r1l = __uint_as_float(r1 << 16), r1h = __uint_as_float(r1 & 0xffff0000u);
  float q0l, q0h, q1l, q1h;
  if (SCORER == KGE_DISTMULT) {
    q0l = a0l * r0l; q0h = a0h * r0h; q1l = a1l * r1l; q1h = a1h * r1h;
  } else if (dir == KGE_SP_) {
    q0l = a0l * r0l - a1l * r1l; q0h = a0h * r0h - a1h * r1h;
    q1l = a1l * r0l + a0l * r1l; q1h = a1h * r0h + a0h * r1h;
  } else {
    q0l = r0l * a0l + r1l * a1l; q0h = r0h * a0h + r1h * a1h;
    q1l = r0l * a1l - r1l * a0l; q1h = r0h * a1h - r1h * a0h;
  }
  q0 = bf16_pack(q0l, q0h);
  q1 = bf16_pack(q1l, q1h);
}

// The same values with a third of the instructions (gfx950): packed f32 multiplies / fused
// multiply-adds and the hardware's RNE pack.  bf16 x bf16 products are EXACT in f32 (8 + 8 mantissa
// bits), so fl(fl(ab) - fl(cd)) = fl(ab - cd) = fma(a, b, -(cd)) bit for bit -- the fma is not a
// contraction that changes results here; v_cvt_pk_bf16_f32 rounds to nearest even like bf16_pack (NaN
// payloads aside).  3.5 VALU ops per output instead of ~11.
typedef float f32x2q __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int bf16_pack_hw(f32x2q v) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2q));
}
template <int SCORER>
__device__ __forceinline__ void bf16_qpair_fast(int dir, unsigned int a0, unsigned int a1, unsigned int r0,
                                                unsigned int r1, unsigned int& q0, unsigned int& q1) {
  const f32x2q A0 = {__uint_as_float(a0 << 16), __uint_as_float(a0 & 0xffff0000u)};
  const f32x2q A1 = {__uint_as_float(a1 << 16), __uint_as_float(a1 & 0xffff0000u)};
  const f32x2q R0 = {__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u)};
  const f32x2q R1 = {__uint_as_float(r1 << 16), __uint_as_float(r1 & 0xffff0000u)};
  f32x2q Q0, Q1;
  if (SCORER == KGE_DISTMULT) {
    Q0 = A0 * R0;
    Q1 = A1 * R1;
  } else if (dir == KGE_SP_) {
    Q0 = __builtin_elementwise_fma(A0, R0, -(A1 * R1));
    Q1 = __builtin_elementwise_fma(A1, R0, A0 * R1);
  } else {
    Q0 = __builtin_elementwise_fma(R0, A0, R1 * A1);
    Q1 = __builtin_elementwise_fma(R0, A1, -(R1 * A0));
  }
  q0 = bf16_pack_hw(Q0);
  q1 = bf16_pack_hw(Q1);
}

// row index through an index vector; MODE 0 = identity, 1 = int32, 2 = int64 (no branches)
template <int MODE>
__device__ __forceinline__ long long index_mode(const Index& ix, long long i) {
  if (MODE == 0) return i;
  if (MODE == 1) return (long long)((const int*)ix.ptr)[i * ix.stride];
  return ((const long long*)ix.ptr)[i * ix.stride];
}

// ---- fused 1vsAll loss epilogues of pairs_bf16_v3_kernel (score_pairs_bf16_v3.hip, ce_loss.hip) ----
// What happens to a finished 128 x 64 score tile (template parameter EPI):
//   V3_STORE  written to out[n, m] (f32)                                    -- kge_score_sp/_po
//   V3_LSE    folded into per-row running (max, sum exp) and the label's score picked out; the
//             score matrix is never written                                 -- kge_ce_fwd
//   V3_DS     d loss / d score = g_i * (exp(score - lse_i) - [j == label_i]) rounded to bf16 and
//             written to G16[n, ld16] (the left operand of both gradient GEMMs) -- kge_ce_bwd
// The MFMA chain, its operands and their order are the same in all three: the scores inside
// V3_LSE / V3_DS are bit-identical to what V3_STORE writes.
//   V3_SPLUS  folded into a per-row running sum of softplus(score + offset) (binary cross entropy
//             with logits against all-zero labels; the label terms are added outside)  -- kge_bce_fwd
//   V3_DSIG   d loss / d score = g_i * (sigmoid(score + offset) - [j == label_i]) as bf16 -> G16
//                                                                                    -- kge_bce_bwd
//   V3_RANK   counted against the row's true score (close / greater-and-not-close, the arithmetic of
//             rank.hip) right on the accumulators, with up to two filter sets given as per-row column
//             bit masks: kge_score_rank_sp_po (loader/consumer kernel only; nothing is written per score)
constexpr int V3_STORE = 0, V3_LSE = 1, V3_DS = 2, V3_SPLUS = 3, V3_DSIG = 4, V3_RANK = 5;
constexpr float V3_LOG2E = 1.44269504088896340736f;
constexpr float V3_LN2 = 0.69314718055994530942f;

// log(1 + exp(x)) without overflow: max(x, 0) + log1p(exp(-|x|)); the series below 2^-7 keeps the
// relative error of the small terms at ~1e-5 where 1 + e would round e away
__device__ __forceinline__ float v3_softplus(float x) {
  const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(x) * V3_LOG2E);  // (0, 1]
  const float l = e < 0.0078125f ? e * (1.0f - 0.5f * e) : __builtin_amdgcn_logf(1.0f + e) * V3_LN2;
  return __builtin_fmaxf(x, 0.0f) + l;
}

struct CeArgs {
  Index label;              // [n] entity ids (the true target of row i); ptr == NULL: no single label
  const long long* rowptr;  // multi-label loss (kge_kl_*): [n + 1] label CSR; rows without labels get
                            // a zero gradient (V3_DS); NULL otherwise
  float* part;              // V3_LSE: [n][ncg][2] per-column-group (max, sum exp)
  float* true_score;        // V3_LSE: [n] score(i, label_i)
  const float* lse;         // V3_DS: [n] logsumexp of row i
  const float* row_bias;    // V3_DS: [n] or NULL: d loss_i / d score_ij = g_i * (softmax_ij - row_bias[i]) - (labels);
                            // the uniform part of smoothed labels (kge_kl_weighted_bwd)
  const float* g_rows;      // V3_DS / V3_DSIG: [n] upstream gradient of row i's loss, or NULL: g_scalar
  float g_scalar;
  // g_rows == NULL: every row's gradient is g_scalar * g_dev[0] * g_dev2[0] (each factor 1 where the pointer is
  // NULL) -- the gradient of scale * sum(loss_rows) with the upstream gradient and the scale as DEVICE scalars
  // (kge_ce_sp_po_bwd_accum_sum: nothing of a captured training step is a host value)
  const float* g_dev;
  const float* g_dev2;
  float offset;             // V3_SPLUS / V3_DSIG: added to every score (train.loss_arg of the bce loss)
  unsigned short* g16;      // V3_DS: [n][ld16] bf16, ld16 % 64 == 0 and ld16 >= 64 * ntiles
  long long ld16;
  // two-sided launch (sp_ queries in row groups [0, rgn1), _po queries behind them): the entity
  // operand, labels and row offset (into lse / g_rows / part / true_score / g16) of the second side
  int rgn1;                 // 0: one-sided
  Operand a2;
  Index label2;
  long long side2_off;
  // V3_RANK: per side (index 1 = the _po side of a two-sided launch)
  const float* rk_true[2];           // true score of row i at [i * rk_true_stride]
  long long rk_true_stride;          // 1: a vector; 4 n + 1: the diagonals of the [n, 4 n] block of kge_eval_batch
  unsigned long long* rk_rank[2];    // [rk_nfilt + 1][rk_ld] ACCUMULATED: row 0 raw, row k + 1 filter set k
  unsigned long long* rk_ties[2];
  long long rk_ld;
  float rk_atol, rk_rtol;
  int rk_nfilt;                      // <= 2
  // bit (j & 31) of the 32-bit word [i * rk_bits_rs + (j >> 5) * rk_bits_us]: column j of the scored slice is
  // filtered for row i.  The library's layout (api.hip: rank_bits_layout) is WORD-major -- rs = 1, us = the row pitch:
  // the 32 rows of a wave read 32 neighbouring words (two cache lines) per unit of 32 columns, not one line per row
  const unsigned int* rk_bits[2][2];  // [side][filter set]
  long long rk_bits_rs, rk_bits_us;
  int rk_clear_bits;                  // pairs_bf16_v8_rank_kernel: store zero over every filter word it has read
  // band-and-rescore (pairs_bf16_v8_rank_kernel<BAND>)
  const float* rk_tmax;               // [1] the table's largest row norm (kge_table_max_row_norm)
  u32x4* rk_list;                     // the waves' pair lists (1 + 255 records of 16 bytes each; headers zero between calls)
  long long rk_list_bytes;
  unsigned int* rk_status;            // [2] or NULL: [0] += pairs listed, [1] += pairs DROPPED (a full list)
};

// one query type of a multi-label (KvsAll) batch: ce_loss.hip run_multilabel2_bwd_accum
struct LossSide {
  Operand A, R;             // entity rows (s of sp_ / o of _po queries), relation rows
  long long n;
  const long long* rowptr;  // [n + 1] label CSR
  const long long* col;
  const float* lse;         // kl: [n]
  const float* g_rows;      // [n] upstream gradients, or NULL: g_scalar (x g_dev[0])
  float g_scalar;
  const float* label_weight;  // kl with label smoothing: [n], and the rows' uniform mass; else NULL
  const float* label_bias;
  const float* g_dev;       // [1] device scalar or NULL: with g_rows == NULL every row's gradient is g_scalar * g_dev[0]
};

__device__ __forceinline__ float ce_row_gradient(const CeArgs& ce, long long row) {
  if (ce.g_rows != nullptr) return ce.g_rows[row];
  float g = ce.g_scalar;
  if (ce.g_dev != nullptr) g = g * ce.g_dev[0];
  if (ce.g_dev2 != nullptr) g = g * ce.g_dev2[0];
  return g;
}

// ---- the tie arithmetic of EntityRankingJob._get_ranks_and_num_ties (eval_entity_ranking.py:571-596), shared by
// rank.hip and the counting epilogue of the scoring kernel
__device__ __forceinline__ bool is_close(float x, float t, float atol, float rtol) {
  // torch.isclose in f32: (x == t) | (isfinite(|x-t|) & (|x-t| <= atol + |rtol*t|))
  if (x == t) return true;
  float err = __builtin_fabsf(x - t);
  float allowed = atol + __builtin_fabsf(rtol * t);
  return __builtin_isfinite(err) && err <= allowed;
}

__device__ __forceinline__ void count_one(float x, float t, float atol, float rtol, int& gt,
                                          int& cl) {
  if (x != x) x = -__builtin_inff();
  bool c = is_close(x, t, atol, rtol);
  cl += c ? 1 : 0;
  gt += (x > t && !c) ? 1 : 0;
}

// ---- counting epilogue of the exact pair kernels (score_pairs.hip, score_pairs_f32.hip): kge_score_rank_sp_po for
// float32 tables and for TransE / RotatE.  One side of the batch per launch.
struct RankArgs {
  const float* tru;                   // true score of row i at [i * tru_stride]
  long long tru_stride;
  unsigned long long* rank;           // [nfilt + 1][ld] ACCUMULATED: row 0 raw, row k + 1 filter set k
  unsigned long long* ties;
  long long ld;
  float atol, rtol;
  int nfilt;                          // <= 2
  // bit (j & 31) of the 32-bit word [i * bits_rs + (j >> 5) * bits_us]: column j is filtered for row i (FilterBits)
  const unsigned int* bits[2];
  long long bits_rs, bits_us;
  int col_tiles;                      // column tiles a workgroup walks (its counts leave it once)
};

// A finished ROWS x COLS score tile sits in LDS (`tile`, row pitch LDT floats).  256 threads: thread t takes the W =
// COLS * ROWS / 256 columns [seg * W, ...) of tile row t / SEGS -- W in {16, 64} divides 64 and the tile's first
// column is a multiple of COLS, so the thread's columns lie inside ONE 64-bit filter word per filter set -- and
// counts them against the row's true score with the arithmetic of rank.hip (count_one; a filtered column scores
// -inf) into its RankAcc.  A workgroup walks several column tiles of its rows (rank_acc_add per tile) and then
// rank_acc_flush adds the SEGS lanes of a row together and issues the row's int64 atomics once: per row, ranking
// and workgroup at most two atomics leave the workgroup (one tile per workgroup and one flush per segment was 750 k
// atomics per evaluation batch at the FB15k-237 shape, more time than the scoring); no score does.
struct RankAcc {
  int G, C;          // raw: greater-and-not-close, close
  int Gf[2], Cf[2];  // per filter set: the same with the filtered columns taken out (+ fc per filtered column)
};

template <int ROWS, int COLS, int LDT>
__device__ __forceinline__ void rank_acc_add(RankAcc& a, const float* tile, long long row0, long long col0,
                                             long long n, long long m, const RankArgs& rk, int tid) {
  constexpr int SEGS = 256 / ROWS, W = COLS / SEGS;
  static_assert(W == 16 || W == 64, "segment = 16 or 64 columns");
  const int row = tid / SEGS, seg = tid % SEGS;
  const long long orow = row0 + row;
  const long long c0 = col0 + seg * W;  // first column of the segment (relative to the scored slice)
  if (orow >= n || c0 >= m) return;
  float t = rk.tru[orow * rk.tru_stride];
  if (t != t) t = -__builtin_inff();
  unsigned long long gm = 0ull, cm = 0ull, vm = 0ull;
  const float* src = tile + row * LDT + seg * W;
#pragma unroll 8
  for (int c = 0; c < W; ++c) {
    if (c0 + c < m) {
      int g = 0, cl = 0;
      count_one(src[c], t, rk.atol, rk.rtol, g, cl);
      gm |= (unsigned long long)g << c;
      cm |= (unsigned long long)cl << c;
      vm |= 1ull << c;
    }
  }
  const int G = __builtin_popcountll(gm), C = __builtin_popcountll(cm);
  a.G += G;
  a.C += C;
  const int fc = t == -__builtin_inff() ? 1 : 0;  // is -inf (a filtered column's score) close to the true score
  const int sh = (int)(c0 & 31);  // (W = 64: c0 is a multiple of 64)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k < rk.nfilt) {
      const unsigned int* bw = rk.bits[k] + orow * rk.bits_rs + (c0 >> 5) * rk.bits_us;
      unsigned long long w = bw[0];
      if constexpr (W == 64) w |= (unsigned long long)bw[rk.bits_us] << 32;
      w = (w >> sh) & vm;
      a.Gf[k] += G - __builtin_popcountll(gm & w);
      a.Cf[k] += C - __builtin_popcountll(cm & w) + fc * __builtin_popcountll(w);
    }
  }
}

// every thread of the workgroup calls this (wave shuffles)
template <int ROWS>
__device__ __forceinline__ void rank_acc_flush(RankAcc a, long long row0, long long n, const RankArgs& rk, int tid) {
  constexpr int SEGS = 256 / ROWS;
#pragma unroll
  for (int off = 1; off < SEGS; off <<= 1) {
    a.G += __shfl_xor(a.G, off, 64);
    a.C += __shfl_xor(a.C, off, 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      a.Gf[k] += __shfl_xor(a.Gf[k], off, 64);
      a.Cf[k] += __shfl_xor(a.Cf[k], off, 64);
    }
  }
  const long long orow = row0 + tid / SEGS;
  if (orow >= n || (tid % SEGS) != 0) return;
  if (a.G) atomicAdd(rk.rank + orow, (unsigned long long)a.G);
  if (a.C) atomicAdd(rk.ties + orow, (unsigned long long)a.C);
  for (int k = 0; k < rk.nfilt; ++k) {
    if (a.Gf[k]) atomicAdd(rk.rank + (k + 1) * rk.ld + orow, (unsigned long long)a.Gf[k]);
    if (a.Cf[k]) atomicAdd(rk.ties + (k + 1) * rk.ld + orow, (unsigned long long)a.Cf[k]);
  }
}

// the filter lists of one evaluation batch (kge_eval_batch: rank.hip's eval_begin_kernel / eval_end_kernel)
constexpr int EV_MAXQ = 4;
struct EvalLists {
  int nq;
  const long long* keys[EV_MAXQ];
  long long num_keys[EV_MAXQ];
  const long long* starts[EV_MAXQ];
  const long long* values[EV_MAXQ];
  long long mult[EV_MAXQ];
  Index a[EV_MAXQ], b[EV_MAXQ], keep[EV_MAXQ];
  long long* range[EV_MAXQ];             // [2][n]: begin, end of row i's values (kept for the clearing pass)
  unsigned int* bits[EV_MAXQ];           // the layout of CeArgs::rk_bits
};

// ---- the filter sets as per-row column bit masks (kge_score_rank_sp_po): one wave per (row, list) sets (set = 1) the
// bits of the row's filter columns that fall into the scored slice [col_begin, col_begin + m) -- except the row's own
// true column, which is never filtered (eval_entity_ranking.py:288-290) -- or clears the words again (set = 0).
struct RankBitLists {
  const long long* begin[4];
  const long long* end[4];
  const long long* col[4];
  Index keep[4];
  unsigned int* bits[4];
};
__device__ __forceinline__ void rank_bits_row(const RankBitLists& B, int q, long long i, int lane, long long col_begin,
                                              long long m, long long rs, long long us, int set) {
  const long long keep = index_at(B.keep[q], i);
  const long long* __restrict__ col = B.col[q];
  unsigned int* row = B.bits[q] + i * rs;
  for (long long e = B.begin[q][i] + lane; e < B.end[q][i]; e += 64) {
    const long long g = col[e];
    const long long j = g - col_begin;
    if (g == keep || j < 0 || j >= m) continue;
    if (set) atomicOr(row + (j >> 5) * us, 1u << (j & 31));
    else row[(j >> 5) * us] = 0u;
  }
}

// ---- kge_eval_batch, the work in front of the scoring: one WAVE per (row, list q < nq): the 64-ary search of the
// filter index, then the wave sets the bits of the row's filtered columns; q == nq: the batch's target list (o | s).
__device__ __forceinline__ void eval_begin_row(const EvalLists& L, const Index& s, const Index& o, long long n, long long m,
                                               long long rs, long long us, long long* __restrict__ tgt, int q, long long i,
                                               int lane) {
  if (q == L.nq) {
    if (lane == 0) {
      tgt[i] = index_at(o, i);
      tgt[n + i] = index_at(s, i);
    }
    return;
  }
  const long long* __restrict__ keys = L.keys[q];
  const long long num_keys = L.num_keys[q];
  const long long key = index_at(L.a[q], i) * L.mult[q] + index_at(L.b[q], i);
  long long lo = 0, hi = num_keys;
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
  const bool hit = lo < num_keys && keys[lo] == key;
  const long long b = hit ? L.starts[q][lo] : 0, e = hit ? L.starts[q][lo + 1] : 0;
  if (lane == 0) {
    L.range[q][i] = b;
    L.range[q][n + i] = e;
  }
  const long long keep = index_at(L.keep[q], i);
  const long long* __restrict__ col = L.values[q];
  unsigned int* row = L.bits[q] + i * rs;
  for (long long x = b + lane; x < e; x += 64) {
    const long long g = col[x];
    if (g == keep || g < 0 || g >= m) continue;
    atomicOr(row + (g >> 5) * us, 1u << (g & 31));
  }
}

// ---- Stall injection (builds with -DKGE_STALL_INJECT only; tools/gpu_stall_inject.sh; SURVEY.md 5 "Race detection").
// The matrix-core kernels synchronise their LDS rings by hand: one workgroup barrier per unit and counted
// `s_waitcnt vmcnt(N)`, N = the vector-memory operations a wave has issued since the piece it waits for.  Both silent-
// corruption bugs of earlier rounds lived there and both needed a particular TIMING (a table from HBM instead of L2).
// This build perturbs the timing: in front of every workgroup barrier and every LDS-DMA piece of those kernels a wave
// sleeps, with probability 1/4, for a pseudo-random 0 .. ~2,000 cycles (a hash of the cycle counter, the wave and the
// site) -- producers fall behind consumers and the other way round, waves of a workgroup drift apart by whole units.
// s_sleep is not a memory operation: the counted waits see the same counts.  The bit-equality tests (every kernel
// against the oracle and against its sibling kernels) must pass unchanged under it.
#ifdef KGE_STALL_INJECT
__device__ __forceinline__ void kge_stall(unsigned int site) {
  const unsigned long long c = __builtin_readcyclecounter();
  unsigned int h = (unsigned int)c * 2654435761u ^ site * 40503u ^ ((blockIdx.x << 3) + (threadIdx.x >> 6)) * 9176u;
  h ^= h >> 13;
  h *= 0x5bd1e995u;
  h ^= h >> 15;
  h = (unsigned int)__builtin_amdgcn_readfirstlane((int)h);  // one decision per wave
  if ((h & 3u) == 0u) {
    const unsigned int reps = (h >> 2) & 31u;
    for (unsigned int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(1);  // 64 cycles each
  }
}
#define KGE_STALL(site) kge_stall((unsigned int)(site))
#else
#define KGE_STALL(site) do { } while (0)
#endif
#define KGE_BARRIER() do { KGE_STALL(__LINE__); __builtin_amdgcn_s_barrier(); } while (0)

// ---- fill_words_async: what the library uses INSTEAD of hipMemsetAsync.  A hipMemsetAsync captured into a hipGraph
// becomes a memset node that ROCm replays with its blit fill kernel (__amd_rocclr_fillBufferAligned) from a 16-byte
// pattern the graph does not own: after ~100 replays of a captured training step the relation-gradient buffer came
// back filled with a repeating 16-byte pattern of which one dword was garbage (round 4: tools/graph_step_probe.py,
// DESIGN.md 10.7) -- the zeroed accumulator was not zero, every replayed step added noise to the relation table.
// A kernel of our own carries its value as a launch argument, which the graph does own.  `bytes` is a multiple of 4.
__global__ static void fill_words_kernel(unsigned int* __restrict__ p, unsigned int v, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
static inline bool fill_words_async(void* ptr, unsigned char byte, size_t bytes, hipStream_t st) {
  if (bytes == 0) return true;
  const long long n = (long long)(bytes / 4);
  const unsigned int v = 0x01010101u * byte;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned int*)ptr, v, n);
  return hipGetLastError() == hipSuccess;
}

}  // namespace kge
