// score_pairs.hip -- "one query row vs many target rows" scoring, gfx950:
// KgeModel.score_sp / score_po (kge_model.py:682-725) and the sp_/_po branches of the
// four scorers (complex.py:36-39, distmult.py:17-21, transe.py:19-34, rotate.py:43-64),
// with the embedding gather (lookup_embedder.py:96-112) fused in.
//
// This is the exact f32 path: every output element is ONE sequential chain over the
// coordinate pairs c = 0..ceil(d/2)-1 (first-half element c, then second-half element
// ceil(d/2)+c), identical to pair_score() in oracle/kge_oracle.c, so results are
// bit-reproducible for f32 AND bf16 tables (bf16 values are widened exactly; for
// ComplEx/DistMult on bf16 tables the query vector q is rounded to bf16 first, like a
// bf16 GEMM operand).
//
//   tile      64 query rows x 64 target rows per 256-thread workgroup
//   staging   gather + query build (q = s(x)r, s+r, s*e^{i th}, ...) into LDS as
//             Qs[half][c][row], Ts[half][c][row] (row fastest: conflict-free reads)
//   compute   ComplEx/DistMult: v_mfma_f32_32x32x2_f32 -- k=0 is the first-half element,
//             k=1 the second-half element of coordinate c; the f32 MFMA is an exact
//             k-ordered fmaf chain (cdna guide section 3), i.e. the canonical order.
//             TransE/RotatE: 4x4 register micro-tile per thread on the VALU.
#include "common.hpp"
#include "switches.hpp"

namespace kge {

constexpr int PT_BM = 64, PT_BN = 64, PT_KC = 16, PT_LD = 68;

template <typename T, bool VEC>
__device__ __forceinline__ f32x4 load4(const T* row, int c, int limit) {
  if (VEC) return ld4<T>(row + c);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (c + i < limit) ? ld1<T>(row + c + i) : 0.0f;
  return r;
}

// RANK: counts against the rows' true scores instead of stored scores (rank_tile_rows, common.hpp; the finished
// tile goes through the operand buffers: 64 x 68 floats, exactly their size).
template <int SCORER, typename T, int NORM, bool VEC, bool MFMA, bool RANK>
__global__ __launch_bounds__(256) void pairs_kernel(Operand A, Operand R, Operand TG, int dir,
                                                    int d, int dr, long long n, long long m,
                                                    float lp, int round_q,
                                                    float* __restrict__ out, long long ldo, RankArgs rk) {
  constexpr bool DOT = (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT);
  __shared__ __attribute__((aligned(16))) float QT[2][2][PT_KC][PT_LD];  // one object: the RANK epilogue reuses it
  auto& Qs = QT[0];
  auto& Ts = QT[1];

  const int tid = threadIdx.x;
  // RANK: this workgroup walks CT consecutive column tiles of its 64 rows and keeps the counts in registers
  const int CT = RANK ? rk.col_tiles : 1;
  const long long row0 = (long long)blockIdx.y * PT_BM;
  RankAcc racc{};
  for (int ct = 0; ct < CT; ++ct) {
  const long long col0 = ((long long)blockIdx.x * CT + ct) * PT_BN;
  if (col0 >= m) break;
  if (ct > 0) __syncthreads();  // the previous tile's epilogue is done with the operand buffers
  const int hh = (d + 1) / 2;  // coordinate pairs
  const int lim1 = d - hh;     // valid second-half elements
  const int nchunk = (hh + PT_KC - 1) / PT_KC;

  // staging role: row sr, coordinates 4*scq .. +3 of the chunk
  const int sr = tid >> 2, scq = tid & 3;
  long long qrow = row0 + sr;
  if (qrow >= n) qrow = n - 1;  // clamp: rows beyond n are computed but never stored
  long long trow = col0 + sr;
  if (trow >= m) trow = m - 1;
  const T* arow = (const T*)A.base + index_at(A.idx, qrow) * A.ld;
  const T* rrow = (const T*)R.base + index_at(R.idx, qrow) * R.ld;
  const T* tgrow = (const T*)TG.base + index_at(TG.idx, trow) * TG.ld;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;  // valid relation first-half elements
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;

  f32x4 a0, a1, r0, r1, t0, t1;
  auto gload = [&](int ch) {
    const int c = ch * PT_KC + scq * 4;
    if (VEC && c >= hh) {  // chunk tail beyond the row (hh % 4 == 0 on this path): zeros
      a0 = a1 = r0 = r1 = t0 = t1 = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    a0 = load4<T, VEC>(arow, c, hh);
    a1 = load4<T, VEC>(arow + hh, c, lim1);
    r0 = load4<T, VEC>(rrow, c, rl0);
    if (SCORER != KGE_ROTATE) r1 = load4<T, VEC>(rrow + hh, c, rl1);
    else r1 = r0;
    t0 = load4<T, VEC>(tgrow, c, hh);
    t1 = load4<T, VEC>(tgrow + hh, c, lim1);
  };
  auto sstore = [&]() {
    f32x4 q0, q1;
    build_q4<SCORER>(dir, a0, a1, r0, r1, q0, q1);
    if (round_q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q0[i] = round_bf16(q0[i]);
        q1[i] = round_bf16(q1[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Qs[0][scq * 4 + i][sr] = q0[i];
      Qs[1][scq * 4 + i][sr] = q1[i];
      Ts[0][scq * 4 + i][sr] = t0[i];
      Ts[1][scq * 4 + i][sr] = t1[i];
    }
  };

  const int lane = tid & 63, wave = tid >> 6;
  f32x16 macc;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) macc[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

  const int tx = tid & 15, ty = tid >> 4;
  const int mrow = 32 * (wave >> 1) + (lane & 31), mcol = 32 * (wave & 1) + (lane & 31);
  const int mh = lane >> 5;

  gload(0);
  sstore();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk) gload(ch + 1);
    if (DOT && MFMA) {
#pragma unroll
      for (int cc = 0; cc < PT_KC; ++cc) {
        float av = Qs[mh][cc][mrow];
        float bv = Ts[mh][cc][mcol];
        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, macc, 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int cc = 0; cc < PT_KC; ++cc) {
        f32x4 q0 = *reinterpret_cast<const f32x4*>(&Qs[0][cc][ty * 4]);
        f32x4 q1 = *reinterpret_cast<const f32x4*>(&Qs[1][cc][ty * 4]);
        f32x4 t0v = *reinterpret_cast<const f32x4*>(&Ts[0][cc][tx * 4]);
        f32x4 t1v = *reinterpret_cast<const f32x4*>(&Ts[1][cc][tx * 4]);
        if constexpr (!DOT && SCORER != KGE_TRANSE) {
          // RotatE: |q - t| of 16 complex coordinates.  The correctly rounded square root in its short form
          // (common.hpp: sqrt_rn_core, checked exhaustively) wherever all 16 squares lie in its range -- ONE check
          // of their minimum and maximum per micro-tile instead of a branch per root; zeros, denormal-sized or huge
          // squares, inf and NaN (the maximum of a set with a NaN may hide it: NaN in, NaN out either way) send the
          // micro-tile through the IEEE sequence.
          float x[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float dre = q0[i] - t0v[j], dim_ = q1[i] - t1v[j];
              x[i][j] = __builtin_fmaf(dim_, dim_, dre * dre);
            }
          float mn = x[0][0], mx = x[0][0];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              mn = __builtin_fminf(mn, x[i][j]);
              mx = __builtin_fmaxf(mx, x[i][j]);
            }
          if (__builtin_expect(mn >= SQRT_FAST_LO && mx <= SQRT_FAST_HI, 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = norm_acc<NORM>(acc[i][j], sqrt_rn_core(x[i][j]), lp);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = norm_acc<NORM>(acc[i][j], __builtin_sqrtf(x[i][j]), lp);
          }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (DOT) {
              acc[i][j] = __builtin_fmaf(q0[i], t0v[j], acc[i][j]);
              acc[i][j] = __builtin_fmaf(q1[i], t1v[j], acc[i][j]);
            } else {
              acc[i][j] = norm_acc<NORM>(acc[i][j], __builtin_fabsf(q0[i] - t0v[j]), lp);
              acc[i][j] = norm_acc<NORM>(acc[i][j], __builtin_fabsf(q1[i] - t1v[j]), lp);
            }
          }
        }
      }
    }
    __syncthreads();
    if (ch + 1 < nchunk) sstore();
    __syncthreads();
  }

  if constexpr (RANK) {
    static_assert(sizeof(QT) >= PT_BM * PT_LD * 4, "the score tile fits the operand buffers");
    float* const tile = &QT[0][0][0][0];  // [64][PT_LD]; the loop ended with a barrier
    if (DOT && MFMA) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(32 * (wave >> 1) + (r & 3) + 8 * (r >> 2) + 4 * mh) * PT_LD + 32 * (wave & 1) + (lane & 31)] = macc[r];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[i][j];
          if (!DOT) {
            if (NORM == NORM_L1) v = -v;
            else if (NORM == NORM_L2) v = -__builtin_sqrtf(v);
            else v = -powf(v, 1.0f / lp);
          }
          tile[(ty * 4 + i) * PT_LD + tx * 4 + j] = v;
        }
    }
    __syncthreads();
    rank_acc_add<PT_BM, PT_BN, PT_LD>(racc, tile, row0, col0, n, m, rk, tid);
    continue;
  }
  if (DOT && MFMA) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      long long orow = row0 + 32 * (wave >> 1) + (r & 3) + 8 * (r >> 2) + 4 * mh;
      long long ocol = col0 + 32 * (wave & 1) + (lane & 31);
      if (orow < n && ocol < m) out[orow * ldo + ocol] = macc[r];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long long orow = row0 + ty * 4 + i;
      if (orow >= n) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        long long ocol = col0 + tx * 4 + j;
        if (ocol >= m) continue;
        float v = acc[i][j];
        if (!DOT) {
          if (NORM == NORM_L1) v = -v;
          else if (NORM == NORM_L2) v = -__builtin_sqrtf(v);
          else v = -powf(v, 1.0f / lp);
        }
        out[orow * ldo + ocol] = v;
      }
    }
  }
  }  // column tiles
  if constexpr (RANK) rank_acc_flush<PT_BM>(racc, row0, n, rk, tid);
}

// ---- TransE store path (round 6) ----------------------------------------------------------------
// score = -|| q - t ||_p with q = s + r (sp_) / o - r (_po), transe.py:22-34: per element ONE subtract and ONE
// accumulate -- `acc + |x|` (L1; the absolute value is a source modifier) or fma(x, x, acc) (L2).  The generic kernel
// above spends 2.7 instruction slots per element on it (4 x 4 micro-tile: an LDS read per 8 elements, scalar
// subtracts).  Here: 128 query rows x 64 targets per workgroup, 8 x 4 outputs per thread, the subtracts of two rows as
// one packed instruction (float2 arithmetic: v_pk_add_f32 with a negated operand; L2: v_pk_fma_f32 as well), six
// ds_read_b128 per 64 elements.  The chain of every output is the canonical one (coordinate pairs in order, first-half
// element then second-half element: oracle/kge_oracle.c pair_score) -- packing changes which LANE-SLOT computes an
// element, not its operands or its order: the same bits as pairs_kernel<KGE_TRANSE>.
constexpr int TE_BM = 128, TE_BN = 64, TE_KC = 16, TE_LDQ = 132, TE_LDT = 68;
typedef float te_f2 __attribute__((ext_vector_type(2)));

template <typename T, int NORM>
__global__ __launch_bounds__(256, 4) void pairs_transe_kernel(Operand A, Operand R, Operand TG, int dir, int d,
                                                           long long n, long long m, float* __restrict__ out,
                                                           long long ldo) {
  static_assert(NORM == NORM_L1 || NORM == NORM_L2, "general p keeps the generic kernel (libm pow)");
  __shared__ __attribute__((aligned(16))) float Qs[2][TE_KC][TE_LDQ];
  __shared__ __attribute__((aligned(16))) float Ts[2][TE_KC][TE_LDT];
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.y * TE_BM, col0 = (long long)blockIdx.x * TE_BN;
  const int hh = d / 2;  // d % 8 == 0 on this path: hh % 4 == 0
  const int nchunk = (hh + TE_KC - 1) / TE_KC;
  // staging roles.  Queries: row qs_r, the coordinate quads qs_c and qs_c + 2 of the chunk's four (the two lanes of a
  // row write LDS rows four apart: different banks).  Targets: row ts_r, quad ts_c.
  const int qs_r = tid >> 1, qs_c = tid & 1;
  const int ts_r = tid >> 2, ts_c = tid & 3;
  long long qrow = row0 + qs_r;
  if (qrow >= n) qrow = n - 1;  // rows beyond n are computed and never stored
  long long trow = col0 + ts_r;
  if (trow >= m) trow = m - 1;
  const T* arow = (const T*)A.base + index_at(A.idx, qrow) * A.ld;
  const T* rrow = (const T*)R.base + index_at(R.idx, qrow) * R.ld;
  const T* tgrow = (const T*)TG.base + index_at(TG.idx, trow) * TG.ld;
  f32x4 a0[2], a1[2], r0[2], r1[2], t0, t1;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int ch) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = ch * TE_KC + (qs_c + 2 * k) * 4;
      if (c >= hh) {  // chunk tail beyond the row: zeros (|0 - 0| adds nothing)
        a0[k] = a1[k] = r0[k] = r1[k] = zero4;
      } else {
        a0[k] = ld4<T>(arow + c);
        a1[k] = ld4<T>(arow + hh + c);
        r0[k] = ld4<T>(rrow + c);
        r1[k] = ld4<T>(rrow + hh + c);
      }
    }
    const int c = ch * TE_KC + ts_c * 4;
    if (c >= hh) {
      t0 = t1 = zero4;
    } else {
      t0 = ld4<T>(tgrow + c);
      t1 = ld4<T>(tgrow + hh + c);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      f32x4 q0, q1;
      build_q4<KGE_TRANSE>(dir, a0[k], a1[k], r0[k], r1[k], q0, q1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Qs[0][(qs_c + 2 * k) * 4 + i][qs_r] = q0[i];
        Qs[1][(qs_c + 2 * k) * 4 + i][qs_r] = q1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Ts[0][ts_c * 4 + i][ts_r] = t0[i];
      Ts[1][ts_c * 4 + i][ts_r] = t1[i];
    }
  };
  const int tx = tid & 15, ty = tid >> 4;  // outputs: rows ty * 8 .. + 7, columns tx * 4 .. + 3
  // L1: scalar accumulators (v_add_f32 with the |x| source modifier: the packed add has no such modifier);
  // L2: packed ones (v_pk_fma_f32).  acc2[i][j] / (accs[2 i][j], accs[2 i + 1][j]) = rows (2 i, 2 i + 1) x column j
  te_f2 acc2[4][4];
  float accs[8][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc2[i][j] = te_f2{0.0f, 0.0f};
      accs[2 * i][j] = accs[2 * i + 1][j] = 0.0f;
    }
  gload(0);
  sstore();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk) gload(ch + 1);
#pragma unroll 4
    for (int cc = 0; cc < TE_KC; ++cc) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // first-half element, then second-half element: the canonical order
        const f32x4 qa = *reinterpret_cast<const f32x4*>(&Qs[h][cc][ty * 8]);
        const f32x4 qb = *reinterpret_cast<const f32x4*>(&Qs[h][cc][ty * 8 + 4]);
        const f32x4 tv = *reinterpret_cast<const f32x4*>(&Ts[h][cc][tx * 4]);
        const te_f2 q2[4] = {te_f2{qa[0], qa[1]}, te_f2{qa[2], qa[3]}, te_f2{qb[0], qb[1]}, te_f2{qb[2], qb[3]}};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const te_f2 x = q2[i] - te_f2{tv[j], tv[j]};
            if (NORM == NORM_L1) {
              // acc + |x| as ONE instruction each (the source modifier): written as asm because the vectoriser pairs the
              // two adds into a v_pk_add_f32 behind two v_and_b32 -- four slots per two elements instead of three
              asm("v_add_f32 %0, |%1|, %0" : "+v"(accs[2 * i][j]) : "v"(x[0]));
              asm("v_add_f32 %0, |%1|, %0" : "+v"(accs[2 * i + 1][j]) : "v"(x[1]));
            } else {
              acc2[i][j] = __builtin_elementwise_fma(x, x, acc2[i][j]);
            }
          }
      }
    }
    __syncthreads();
    if (ch + 1 < nchunk) sstore();
    __syncthreads();
  }
  const bool vec_out = (ldo % 4) == 0 && (((uintptr_t)out) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long orow = row0 + ty * 8 + i;
    if (orow >= n) continue;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = NORM == NORM_L1 ? accs[i][j] : acc2[i >> 1][j][i & 1];
      v[j] = NORM == NORM_L1 ? -a : -__builtin_sqrtf(a);
    }
    const long long ocol = col0 + tx * 4;
    if (vec_out && ocol + 4 <= m) {
      *reinterpret_cast<f32x4*>(out + orow * ldo + ocol) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ocol + j < m) out[orow * ldo + ocol + j] = v[j];
    }
  }
}

// ---- host dispatch ------------------------------------------------------------------------
static inline bool aligned16p(const void* p) { return ((uintptr_t)p & 15) == 0; }

static bool pairs_vec_ok(int dtype, int d, int dr, int scorer, const Operand& A,
                         const Operand& R, const Operand& TG) {
  if (d % 8) return false;  // hh % 4 == 0 and d even
  if (scorer == KGE_ROTATE && dr != d / 2) return false;
  const int es = dtype == KGE_BF16 ? 2 : 4;
  const int al = dtype == KGE_BF16 ? 8 : 16;  // bytes per 4-element load
  if (!aligned16p(A.base) || !aligned16p(R.base) || !aligned16p(TG.base)) return false;
  if ((A.ld * es) % al || (R.ld * es) % al || (TG.ld * es) % al) return false;
  return true;
}

template <int SCORER, typename T, int NORM>
static int launch_pairs(bool vec, bool mfma, const Operand& A, const Operand& R,
                        const Operand& TG, int dir, int d, int dr, long long n, long long m,
                        float lp, int round_q, float* out, long long ldo, hipStream_t st, const RankArgs* rk) {
  if constexpr (SCORER == KGE_TRANSE && (NORM == NORM_L1 || NORM == NORM_L2)) {
    // the store path on aligned rows: the 8 x 4 packed kernel (same bits; the counting epilogue and ragged rows keep
    // the generic kernel).  L1 (transe.yaml's default l_norm) by default: 316 -> 244 us (float32) / 292 -> 217 us (bf16)
    // at the FB15k-237 shape; L2 gains nothing (its subtract + fma pair is not what bounds the generic kernel) and
    // takes this kernel only under SW_TRANSE_GENERIC = 0 (tests).  SW_TRANSE_GENERIC = 1 (tests): never.
    const long long tsw = sw(SW_TRANSE_GENERIC);
    if (vec && rk == nullptr && tsw != 1 && (NORM == NORM_L1 || tsw == 0)) {
      dim3 tgrid((unsigned)((m + TE_BN - 1) / TE_BN), (unsigned)((n + TE_BM - 1) / TE_BM));
      hipLaunchKernelGGL((pairs_transe_kernel<T, NORM>), tgrid, dim3(256), 0, st, A, R, TG, dir, d, n, m, out, ldo);
      return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
    }
  }
  dim3 grid((unsigned)((m + PT_BN - 1) / PT_BN), (unsigned)((n + PT_BM - 1) / PT_BM));
  constexpr bool DOT = (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT);
  // counting launch: column tiles per workgroup -- as many as still leave four workgroups per compute unit (<= 16)
  RankArgs r2 = rk != nullptr ? *rk : RankArgs{};
  long long ctn = (long long)grid.x * grid.y / 1024;
  r2.col_tiles = (int)(ctn < 1 ? 1 : (ctn > 16 ? 16 : ctn));
  dim3 rgrid((grid.x + r2.col_tiles - 1) / r2.col_tiles, grid.y);
#define KGE_PL(VEC, MF)                                                                               \
  do {                                                                                                \
    if (rk != nullptr)                                                                                \
      hipLaunchKernelGGL((pairs_kernel<SCORER, T, NORM, VEC, MF, true>), rgrid, dim3(256), 0, st, A, R, \
                         TG, dir, d, dr, n, m, lp, round_q, out, ldo, r2);                            \
    else                                                                                              \
      hipLaunchKernelGGL((pairs_kernel<SCORER, T, NORM, VEC, MF, false>), grid, dim3(256), 0, st, A, R, \
                         TG, dir, d, dr, n, m, lp, round_q, out, ldo, RankArgs{});                     \
  } while (0)
  if constexpr (DOT) {
    if (mfma) {
      if (vec) KGE_PL(true, true); else KGE_PL(false, true);
    } else {
      if (vec) KGE_PL(true, false); else KGE_PL(false, false);
    }
  } else {
    if (vec) KGE_PL(true, false); else KGE_PL(false, false);
  }
#undef KGE_PL
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

template <int SCORER, typename T>
static int pairs_norm(int norm, bool vec, bool mfma, const Operand& A, const Operand& R,
                      const Operand& TG, int dir, int d, int dr, long long n, long long m,
                      float lp, int round_q, float* out, long long ldo, hipStream_t st, const RankArgs* rk) {
  if constexpr (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT) {
    return launch_pairs<SCORER, T, NORM_L1>(vec, mfma, A, R, TG, dir, d, dr, n, m, lp, round_q,
                                            out, ldo, st, rk);
  } else {
    if (norm == NORM_L1)
      return launch_pairs<SCORER, T, NORM_L1>(vec, mfma, A, R, TG, dir, d, dr, n, m, lp, 0, out,
                                              ldo, st, rk);
    if (norm == NORM_L2)
      return launch_pairs<SCORER, T, NORM_L2>(vec, mfma, A, R, TG, dir, d, dr, n, m, lp, 0, out,
                                              ldo, st, rk);
    return launch_pairs<SCORER, T, NORM_LP>(vec, mfma, A, R, TG, dir, d, dr, n, m, lp, 0, out,
                                            ldo, st, rk);
  }
}

int run_pairs_f32(int scorer, int dtype, const Operand& A, const Operand& R, const Operand& TG, int dir, int d,
                  long long n, long long m, int round_q, float* out, long long ldo, hipStream_t st,
                  const RankArgs* rk);

// exact (canonical f32) pair scoring for every scorer / dtype / dimension
// round_query = false (KGE_FLAG_SPLIT_QUERY's fallback): bf16 tables widened, the query vector kept in f32 --
// f32 arithmetic on the bf16 table values, the bits of the oracle on the widened tables
// rk != NULL: the counting epilogue instead of the score store (`out` is not touched)
int run_pairs_exact(int scorer, int dtype, bool use_mfma, const Operand& A, const Operand& R,
                    const Operand& TG, int dir, int d, int dr, long long n, long long m,
                    float lp, float* out, long long ldo, hipStream_t st, bool round_query, const RankArgs* rk) {
  if (n == 0 || m == 0) return KGE_OK;
  const bool cplx = scorer == KGE_COMPLEX || scorer == KGE_ROTATE;
  if (cplx && (d % 2)) return KGE_ERR_INVALID_ARG;
  const bool vec = pairs_vec_ok(dtype, d, dr, scorer, A, R, TG);
  const int norm = norm_mode(lp);
  const int round_q =
      (round_query && dtype == KGE_BF16 && (scorer == KGE_COMPLEX || scorer == KGE_DISTMULT)) ? 1 : 0;
  // ComplEx / DistMult on the f32 matrix cores: 128 x 128 tiles (score_pairs_f32.hip) unless the
  // batch is too small to fill them; same chain order, same bits
  if ((scorer == KGE_COMPLEX || scorer == KGE_DISTMULT) && use_mfma && vec && n > 64 && m > 64) {
    const int rc = run_pairs_f32(scorer, dtype, A, R, TG, dir, d, n, m, round_q, out, ldo, st, rk);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
  }
#define KGE_DT(SC)                                                                            \
  return dtype == KGE_BF16                                                                    \
             ? pairs_norm<SC, unsigned short>(norm, vec, use_mfma, A, R, TG, dir, d, dr, n, m, \
                                              lp, round_q, out, ldo, st, rk)                   \
             : pairs_norm<SC, float>(norm, vec, use_mfma, A, R, TG, dir, d, dr, n, m, lp,      \
                                     round_q, out, ldo, st, rk)
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
