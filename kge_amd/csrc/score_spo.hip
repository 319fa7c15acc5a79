// score_spo.hip -- row-wise triple scoring: KgeModel.score_spo (kge_model.py:663-680)
// and the negative-sampling "triple" path (sampler.py:291-306), gfx950.
//
// HBM-gather bound: per scored triple one corrupted-slot row (d*elt bytes) is
// algorithmically required.  A group of G lanes (G = power of two
// >= D/8, D = number of reduction coordinates) owns one triple; lane g of the group
// owns coordinates [8g, 8g+8) (+ 8*64*t when D > 512) and reads them with 16/32-byte
// vector loads, so a group reads whole contiguous row segments.  The row reduction is
// the canonical "64 strided partials + xor butterfly" of oracle/kge_oracle.c
// (spo_score): levels with offset >= G add an exact +0 and are skipped.
#include <type_traits>

#include "common.hpp"

namespace kge {

// per-lane chunk of 8 coordinates of one row (first half and, for complex scorers,
// second half)
template <typename T, bool VEC>
__device__ __forceinline__ f32x8 load_chunk(const T* row, int c0, int limit) {
  if (VEC) return ld8<T>(row + c0);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (c0 + i < limit) ? ld1<T>(row + c0 + i) : 0.0f;
  return r;
}

template <int SCORER>
struct IsComplex {
  static constexpr bool value = (SCORER == KGE_COMPLEX || SCORER == KGE_ROTATE);
};

// Per-lane state that does not depend on the corrupted slot ("fixed side").
struct Fixed {
  f32x8 f0, f1, f2, f3;
};

// slot == 2 (object varies; also plain score_spo): fixed side = query vector q(s, r).
// slot == 0 (subject varies): fixed side = relation (or its cos/sin) and the object.
// e0/e1: halves of the fixed ENTITY row, r0/r1: relation row halves (RotatE: r0 = phases).
template <int SCORER>
__device__ __forceinline__ Fixed prep_chunk(int slot, const f32x8& e0, const f32x8& e1,
                                            const f32x8& r0, const f32x8& r1) {
  Fixed F;
  F.f0 = r0; F.f1 = r1; F.f2 = e0; F.f3 = e1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (slot != 0) {
      if (SCORER == KGE_DISTMULT) {
        F.f0.v[i] = e0.v[i] * r0.v[i];
      } else if (SCORER == KGE_COMPLEX) {
        F.f0.v[i] = e0.v[i] * r0.v[i] - e1.v[i] * r1.v[i];
        F.f1.v[i] = e1.v[i] * r0.v[i] + e0.v[i] * r1.v[i];
      } else if (SCORER == KGE_TRANSE) {
        F.f0.v[i] = e0.v[i] + r0.v[i];
      } else {
        float sn, cs;
        sincos_canon(r0.v[i], sn, cs);
        F.f0.v[i] = e0.v[i] * cs - e1.v[i] * sn;
        F.f1.v[i] = e0.v[i] * sn + e1.v[i] * cs;
      }
    } else if (SCORER == KGE_ROTATE) {
      float sn, cs;
      sincos_canon(r0.v[i], sn, cs);
      F.f0.v[i] = cs;
      F.f1.v[i] = sn;
    }
  }
  return F;
}

// accumulate one chunk (x0/x1 = halves of the varying entity row) into the lane partial P
template <int SCORER, int NORM>
__device__ __forceinline__ float apply_chunk(int slot, float P, const Fixed& F,
                                             const f32x8& x0, const f32x8& x1, int cnt,
                                             float lp) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < cnt) {
      if (SCORER == KGE_DISTMULT) {
        if (slot != 0) P = __builtin_fmaf(F.f0.v[i], x0.v[i], P);
        else P = __builtin_fmaf(x0.v[i] * F.f0.v[i], F.f2.v[i], P);
      } else if (SCORER == KGE_COMPLEX) {
        float qre, qim, ore, oim;
        if (slot != 0) {
          qre = F.f0.v[i]; qim = F.f1.v[i]; ore = x0.v[i]; oim = x1.v[i];
        } else {
          qre = x0.v[i] * F.f0.v[i] - x1.v[i] * F.f1.v[i];
          qim = x1.v[i] * F.f0.v[i] + x0.v[i] * F.f1.v[i];
          ore = F.f2.v[i]; oim = F.f3.v[i];
        }
        P = __builtin_fmaf(qre, ore, P);
        P = __builtin_fmaf(qim, oim, P);
      } else if (SCORER == KGE_TRANSE) {
        float df;  // F.pairwise_distance adds eps=1e-6 to every component (transe.py:18)
        if (slot != 0) df = (F.f0.v[i] - x0.v[i]) + 1e-6f;
        else df = ((x0.v[i] + F.f0.v[i]) - F.f2.v[i]) + 1e-6f;
        P = norm_acc<NORM>(P, __builtin_fabsf(df), lp);
      } else {  // ROTATE
        float qre, qim, ore, oim;
        if (slot != 0) {
          qre = F.f0.v[i]; qim = F.f1.v[i]; ore = x0.v[i]; oim = x1.v[i];
        } else {
          qre = x0.v[i] * F.f0.v[i] - x1.v[i] * F.f1.v[i];
          qim = x0.v[i] * F.f1.v[i] + x1.v[i] * F.f0.v[i];
          ore = F.f2.v[i]; oim = F.f3.v[i];
        }
        float dre = qre - ore, dim_ = qim - oim;
        float ab = sqrt_rn_fast(__builtin_fmaf(dim_, dim_, dre * dre));  // (correctly rounded: common.hpp)
        P = norm_acc<NORM>(P, ab, lp);
      }
    }
  }
  return P;
}

// load the fixed side of chunk c0 for (entity row erow, relation row rrow)
template <int SCORER, typename T, bool VEC>
__device__ __forceinline__ Fixed load_fixed(int slot, const T* erow, const T* rrow, int c0,
                                            int D, int h) {
  constexpr bool CPLX = (SCORER == KGE_COMPLEX || SCORER == KGE_ROTATE);
  f32x8 e0 = load_chunk<T, VEC>(erow, c0, D);
  f32x8 r0 = load_chunk<T, VEC>(rrow, c0, D);
  f32x8 e1 = e0, r1 = r0;
  if (CPLX) e1 = load_chunk<T, VEC>(erow + h, c0, D);
  if (SCORER == KGE_COMPLEX) r1 = load_chunk<T, VEC>(rrow + h, c0, D);
  return prep_chunk<SCORER>(slot, e0, e1, r0, r1);
}

template <int SCORER, int NORM>
__device__ __forceinline__ float finalize(float acc, float lp) {
  if (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT) return acc;
  if (NORM == NORM_L1) return -acc;
  if (NORM == NORM_L2) return -__builtin_sqrtf(acc);
  return -powf(acc, 1.0f / lp);
}

template <int G>
__device__ __forceinline__ float group_butterfly(float P) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) P = P + __shfl_xor(P, off, 64);
  return P;
}

// ---- score_spo ----------------------------------------------------------------------
template <int SCORER, typename T, int NORM, bool VEC, int G>
__global__ __launch_bounds__(256) void spo_kernel(Operand S, Operand R, Operand O, int d,
                                                  int dr, long long n, float lp,
                                                  float* __restrict__ out) {
  constexpr bool CPLX = IsComplex<SCORER>::value;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int RPW = 64 / G;  // rows per wave
  const int lg = lane & (G - 1);
  const long long row = ((long long)blockIdx.x * 4 + wave) * RPW + (lane / G);
  const bool valid = row < n;
  const int h = d / 2;
  const int D = CPLX ? h : d;
  const int nchunks = (D + 7) / 8;
  float P = 0.0f;
  if (valid) {
    const T* srow = (const T*)S.base + index_at(S.idx, row) * S.ld;
    const T* rrow = (const T*)R.base + index_at(R.idx, row) * R.ld;
    const T* orow = (const T*)O.base + index_at(O.idx, row) * O.ld;
    for (int ci = lg; ci < nchunks; ci += 64) {  // G < 64 implies nchunks <= G: one trip
      const int c0 = ci * 8;
      const int cnt = (D - c0 < 8) ? (D - c0) : 8;
      Fixed F = load_fixed<SCORER, T, VEC>(2, srow, rrow, c0, D, h);
      f32x8 x0 = load_chunk<T, VEC>(orow, c0, D);
      f32x8 x1 = x0;
      if (CPLX) x1 = load_chunk<T, VEC>(orow + h, c0, D);
      P = apply_chunk<SCORER, NORM>(2, P, F, x0, x1, VEC ? 8 : cnt, lp);
    }
  }
  P = group_butterfly<G>(P);
  if (valid && lg == 0) out[row] = finalize<SCORER, NORM>(P, lp);
}

// ---- negative sampling, "triple" semantics ---------------------------------------------
// grid.y = positive row i; the block's groups stride over that row's K negatives.  The
// non-corrupted entity row and the relation row are loaded ONCE per group into registers
// (one chunk per lane; D <= 512) so only the corrupted-slot row streams from HBM.
template <int SCORER, typename T, int NORM, bool VEC, int G>
__global__ __launch_bounds__(256, 4) void neg_kernel(Operand S, Operand R, Operand O, int d,
                                                  int dr, int slot, const void* neg,
                                                  int neg_itype, long long neg_ld,
                                                  long long K, float lp,
                                                  float* __restrict__ out, long long ldo) {
  constexpr bool CPLX = IsComplex<SCORER>::value;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int GPW = 64 / G;
  constexpr int GPB = 4 * GPW;  // groups per block
  const int lg = lane & (G - 1);
  const int gid = wave * GPW + lane / G;
  const long long i = blockIdx.y;
  const int h = d / 2;
  const int D = CPLX ? h : d;
  const int nchunks = (D + 7) / 8;
  const T* ent = (const T*)S.base;  // S.base == O.base == entity table
  const T* fixrow = ent + index_at(slot == 0 ? O.idx : S.idx, i) * S.ld;
  const T* rrow = (const T*)R.base + index_at(R.idx, i) * R.ld;
  const Index nix = {neg, 1, neg_itype};

  // first chunk of the fixed side lives in registers for the whole negative loop
  Fixed F0;
  const int c00 = lg * 8;
  const int cnt0 = (D - c00 < 8) ? (D - c00) : 8;
  if (lg < nchunks) F0 = load_fixed<SCORER, T, VEC>(slot, fixrow, rrow, c00, D, h);

  // The slot is uniform over the launch: one specialised copy of the loop per side keeps the
  // side selection out of the per-coordinate arithmetic; on the vector path every chunk is full
  // (D % 8 == 0), so the per-coordinate tail test folds away as well.
  auto sweep = [&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    const long long step = (long long)gridDim.x * GPB;
    const bool act = lg < nchunks;
    auto row_of = [&](long long kk) { return ent + index_at(nix, i * neg_ld + kk) * S.ld; };
    auto load_x = [&](const T* xr, f32x8& a, f32x8& b) {
      a = load_chunk<T, VEC>(xr, c00, D);
      b = a;
      if (CPLX) b = load_chunk<T, VEC>(xr + h, c00, D);
    };
    // No software pipeline: prefetching the next negative's index (RotatE 159 -> 167 us) or its
    // index and row (159 -> 270 us, TransE 138 -> 145 us) was slower on MI355X than leaving the
    // latency to the 4..8 resident waves per SIMD -- the extra live registers cost occupancy.
    for (long long k = (long long)blockIdx.x * GPB + gid; k < K; k += step) {
      const T* rc = row_of(k);
      float P = 0.0f;
      if (act) {
        f32x8 x0, x1;
        load_x(rc, x0, x1);
        P = apply_chunk<SCORER, NORM>(SLOT, P, F0, x0, x1, VEC ? 8 : cnt0, lp);
      }
      for (int ci = lg + 64; ci < nchunks; ci += 64) {  // only when D > 512
        const int c0 = ci * 8;
        const int cnt = (D - c0 < 8) ? (D - c0) : 8;
        Fixed F = load_fixed<SCORER, T, VEC>(SLOT, fixrow, rrow, c0, D, h);
        f32x8 y0 = load_chunk<T, VEC>(rc, c0, D);
        f32x8 y1 = y0;
        if (CPLX) y1 = load_chunk<T, VEC>(rc + h, c0, D);
        P = apply_chunk<SCORER, NORM>(SLOT, P, F, y0, y1, VEC ? 8 : cnt, lp);
      }
      P = group_butterfly<G>(P);
      if (lg == 0) out[i * ldo + k] = finalize<SCORER, NORM>(P, lp);
    }
  };
  if (slot == 0) sweep(std::integral_constant<int, 0>{});
  else sweep(std::integral_constant<int, 2>{});
}

// ---- host-side dispatch -------------------------------------------------------------------
static inline int group_size(int D) {
  int nchunks = (D + 7) / 8;
  int G = 8;  // groups narrower than 8 lanes are not instantiated (idle lanes add +0)
  while (G < nchunks && G < 64) G <<= 1;
  return G;
}

template <int SCORER, typename T, int NORM, bool VEC>
static int launch_spo_g(int G, const Operand& S, const Operand& R, const Operand& O, int d,
                        int dr, long long n, float lp, float* out, hipStream_t st) {
#define KGE_SPO_CASE(GG)                                                                 \
  case GG: {                                                                             \
    long long rows_per_block = 4LL * (64 / GG);                                          \
    unsigned grid = (unsigned)((n + rows_per_block - 1) / rows_per_block);               \
    hipLaunchKernelGGL((spo_kernel<SCORER, T, NORM, VEC, GG>), dim3(grid), dim3(256), 0, \
                       st, S, R, O, d, dr, n, lp, out);                                  \
    break;                                                                               \
  }
  switch (G) {
    KGE_SPO_CASE(8)
    KGE_SPO_CASE(16)
    KGE_SPO_CASE(32)
    KGE_SPO_CASE(64)
    default:
      return KGE_ERR_UNSUPPORTED;
  }
#undef KGE_SPO_CASE
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

template <int SCORER, typename T, int NORM, bool VEC>
static int launch_neg_g(int G, const Operand& S, const Operand& R, const Operand& O, int d,
                        int dr, long long n, int slot, const void* neg, int neg_itype,
                        long long neg_ld, long long K, float lp, float* out, long long ldo,
                        hipStream_t st) {
#define KGE_NEG_CASE(GG)                                                                   \
  case GG: {                                                                               \
    long long gpb = 4LL * (64 / GG);                                                       \
    /* negatives per group: 16 amortises the group's fixed side (two row loads, RotatE: */  \
    /* sin/cos), 4 when that would leave fewer than ~2048 workgroups */                    \
    long long per = 16;                                                                    \
    while (per > 4 && n * ((K + gpb * per - 1) / (gpb * per)) < 2048) per >>= 1;           \
    long long bx = (K + gpb * per - 1) / (gpb * per);                                      \
    if (bx < 1) bx = 1;                                                                    \
    if (bx > 64) bx = 64;                                                                  \
    hipLaunchKernelGGL((neg_kernel<SCORER, T, NORM, VEC, GG>), dim3((unsigned)bx, (unsigned)n), \
                       dim3(256), 0, st, S, R, O, d, dr, slot, neg, neg_itype, neg_ld, K,  \
                       lp, out, ldo);                                                      \
    break;                                                                                 \
  }
  switch (G) {
    KGE_NEG_CASE(8)
    KGE_NEG_CASE(16)
    KGE_NEG_CASE(32)
    KGE_NEG_CASE(64)
    default:
      return KGE_ERR_UNSUPPORTED;
  }
#undef KGE_NEG_CASE
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// can every 8-coordinate chunk be read with aligned vector loads?
static bool vec_ok(int scorer, int dtype, int d, int dr, const Operand& S, const Operand& R,
                   const Operand& O) {
  const int es = dtype == KGE_BF16 ? 2 : 4;
  const bool cplx = scorer == KGE_COMPLEX || scorer == KGE_ROTATE;
  const int D = cplx ? d / 2 : d;
  if (D % 8) return false;
  if (!aligned16(S.base) || !aligned16(R.base) || !aligned16(O.base)) return false;
  if ((S.ld * es) % 16 || (R.ld * es) % 16 || (O.ld * es) % 16) return false;
  if (cplx && ((long long)(d / 2) * es) % 16) return false;
  return true;
}

template <int SCORER, typename T>
static int dispatch_spo(bool neg_mode, int norm, bool vec, int G, const Operand& S,
                        const Operand& R, const Operand& O, int d, int dr, long long n,
                        int slot, const void* neg, int neg_itype, long long neg_ld,
                        long long K, float lp, float* out, long long ldo, hipStream_t st) {
#define KGE_GO(NORM, VEC)                                                                \
  return neg_mode ? launch_neg_g<SCORER, T, NORM, VEC>(G, S, R, O, d, dr, n, slot, neg,  \
                                                        neg_itype, neg_ld, K, lp, out,   \
                                                        ldo, st)                         \
                  : launch_spo_g<SCORER, T, NORM, VEC>(G, S, R, O, d, dr, n, lp, out, st)
  if constexpr (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT) {
    (void)norm;  // no norm: a single instantiation
    if (vec) { KGE_GO(NORM_L1, true); } else { KGE_GO(NORM_L1, false); }
  } else {
    if (norm == NORM_L1) {
      if (vec) { KGE_GO(NORM_L1, true); } else { KGE_GO(NORM_L1, false); }
    } else if (norm == NORM_L2) {
      if (vec) { KGE_GO(NORM_L2, true); } else { KGE_GO(NORM_L2, false); }
    } else {
      if (vec) { KGE_GO(NORM_LP, true); } else { KGE_GO(NORM_LP, false); }
    }
  }
#undef KGE_GO
}

int run_spo(int scorer, int dtype, bool neg_mode, const Operand& S, const Operand& R,
            const Operand& O, int d, int dr, long long n, int slot, const void* neg,
            int neg_itype, long long neg_ld, long long K, float lp, float* out,
            long long ldo, hipStream_t st) {
  if (n == 0 || (neg_mode && K == 0)) return KGE_OK;
  const bool cplx = scorer == KGE_COMPLEX || scorer == KGE_ROTATE;
  if (cplx && (d % 2)) return KGE_ERR_INVALID_ARG;
  const int D = cplx ? d / 2 : d;
  const int G = group_size(D);
  const bool vec = vec_ok(scorer, dtype, d, dr, S, R, O);
  const int norm = norm_mode(lp);
#define KGE_DT(SC)                                                                        \
  return dtype == KGE_BF16                                                                \
             ? dispatch_spo<SC, unsigned short>(neg_mode, norm, vec, G, S, R, O, d, dr, n, \
                                                slot, neg, neg_itype, neg_ld, K, lp, out,  \
                                                ldo, st)                                   \
             : dispatch_spo<SC, float>(neg_mode, norm, vec, G, S, R, O, d, dr, n, slot,    \
                                       neg, neg_itype, neg_ld, K, lp, out, ldo, st)
  switch (scorer) {
    case KGE_COMPLEX: KGE_DT(KGE_COMPLEX);
    case KGE_DISTMULT: KGE_DT(KGE_DISTMULT);
    case KGE_TRANSE: KGE_DT(KGE_TRANSE);
    case KGE_ROTATE: KGE_DT(KGE_ROTATE);
  }
#undef KGE_DT
  return KGE_ERR_INVALID_ARG;
}

}  // namespace kge
