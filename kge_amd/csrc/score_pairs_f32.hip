// score_pairs_f32.hip -- ComplEx / DistMult sp_/_po on the f32 matrix cores, exact canonical chain
// (f32 tables, and bf16 tables under KGE_FLAG_EXACT): the default path of a LibKGE model, whose
// parameters are float32 (kge/model/embedder/lookup_embedder.yaml).
//
// Same arithmetic as pairs_kernel<.., MFMA> of score_pairs.hip -- every output element is ONE
// sequential fmaf chain over the coordinate pairs c = 0..d/2-1 (first-half element, then
// second-half element), which is what v_mfma_f32_32x32x2_f32 computes (k = 0: first half,
// k = 1: second half) -- with a tile that feeds the matrix cores better:
//
//   tile      128 query rows x 128 target rows per 256-thread workgroup; each wave owns a 64 x 64
//             quadrant = 2 x 2 accumulators of 32 x 32, so one operand read serves two MFMAs
//             (the 64 x 64 tile of score_pairs.hip: one read per operand per MFMA, one accumulator)
//   K chunks  16 coordinate pairs, double-buffered in LDS ([half][pair][row], row fastest:
//             conflict-free ds_read_b32 of MFMA operands): the global loads of chunk c+2 fly and
//             the query build + LDS stores of chunk c+1 run while chunk c is multiplied; one
//             barrier per chunk
//   bound     MFMA f32: 2*n*d*m flops; 64 MFMAs (4,096 cycles) per wave and chunk
#include "common.hpp"

namespace kge {

constexpr int F3_BM = 128, F3_BN = 128, F3_KC = 16, F3_LD = 132;  // LD: row pitch of a [pair] line
typedef float f32x2 __attribute__((ext_vector_type(2)));

// RANK: the finished tile is not stored but counted against the rows' true scores (rank_tile_rows, common.hpp):
// the accumulators go through the operand buffers (free behind the last chunk: 128 x 132 floats, exactly their
// size) so that a thread sees 64 consecutive columns of ONE row = one filter word per filter set.
template <int SCORER, typename T, bool RANK>
__global__ __launch_bounds__(256) void pairs_f32_kernel(Operand A, Operand R, Operand TG, int dir, int d,
                                                        long long n, long long m, int round_q,
                                                        float* __restrict__ out, long long ldo, RankArgs rk) {
  // [buffer][q|t][half][pair][row]
  __shared__ __attribute__((aligned(16))) float lds[2][2][2][F3_KC][F3_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // RANK: this workgroup walks CT consecutive column tiles of its 128 rows and keeps the counts in registers
  const int CT = RANK ? rk.col_tiles : 1;
  const long long row0 = (long long)blockIdx.y * F3_BM;
  const int hh = d / 2;
  const int nchunk = (hh + F3_KC - 1) / F3_KC;
  RankAcc racc{};
  for (int ct = 0; ct < CT; ++ct) {
  const long long col0 = ((long long)blockIdx.x * CT + ct) * F3_BN;
  if (col0 >= m) break;
  if (ct > 0) __syncthreads();  // the previous tile's epilogue is done with the operand buffers

  // staging role: 2 query units and 2 target units per thread; unit u = (row sru, coordinates
  // 4*scq .. +3 of the chunk).  (Plain variables, no arrays: arrays indexed in the lambdas below
  // ended up in scratch memory.)
  const int scq = tid & 3;
  const int sr0 = tid >> 2, sr1 = (tid >> 2) + 64;
  auto qrow_of = [&](int sr) {
    long long qr = row0 + sr;
    return qr >= n ? n - 1 : qr;  // clamp: rows beyond n are computed but never stored
  };
  auto trow_of = [&](int sr) {
    long long tr = col0 + sr;
    return tr >= m ? m - 1 : tr;
  };
  const T* const arow0 = (const T*)A.base + index_at(A.idx, qrow_of(sr0)) * A.ld;
  const T* const arow1 = (const T*)A.base + index_at(A.idx, qrow_of(sr1)) * A.ld;
  const T* const rrow0 = (const T*)R.base + index_at(R.idx, qrow_of(sr0)) * R.ld;
  const T* const rrow1 = (const T*)R.base + index_at(R.idx, qrow_of(sr1)) * R.ld;
  const T* const trow0 = (const T*)TG.base + index_at(TG.idx, trow_of(sr0)) * TG.ld;
  const T* const trow1 = (const T*)TG.base + index_at(TG.idx, trow_of(sr1)) * TG.ld;

  struct Unit {
    f32x4 a0, a1, r0, r1, t0, t1;
  };
  Unit u0, u1;
  auto gload1 = [&](Unit& u, const T* arow, const T* rrow, const T* trow, int c) {
    if (c >= hh) {  // chunk tail beyond the row (hh % 4 == 0 on this path): zeros
      u.a0 = u.a1 = u.r0 = u.r1 = u.t0 = u.t1 = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      u.a0 = ld4<T>(arow + c);
      u.a1 = ld4<T>(arow + hh + c);
      u.r0 = ld4<T>(rrow + c);
      u.r1 = ld4<T>(rrow + hh + c);
      u.t0 = ld4<T>(trow + c);
      u.t1 = ld4<T>(trow + hh + c);
    }
  };
  auto gload = [&](int ch) {
    const int c = ch * F3_KC + scq * 4;
    gload1(u0, arow0, rrow0, trow0, c);
    gload1(u1, arow1, rrow1, trow1, c);
  };
  auto sstore1 = [&](const Unit& u, int sr, int buf) {
    f32x4 q0, q1;
    build_q4<SCORER>(dir, u.a0, u.a1, u.r0, u.r1, q0, q1);
    if (round_q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q0[i] = round_bf16(q0[i]);
        q1[i] = round_bf16(q1[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds[buf][0][0][scq * 4 + i][sr] = q0[i];
      lds[buf][0][1][scq * 4 + i][sr] = q1[i];
      lds[buf][1][0][scq * 4 + i][sr] = u.t0[i];
      lds[buf][1][1][scq * 4 + i][sr] = u.t1[i];
    }
  };
  auto sstore = [&](int buf) {
    sstore1(u0, sr0, buf);
    sstore1(u1, sr1, buf);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int mh = lane >> 5;                          // k of the 32x32x2 operand this lane holds
  const int qb = 64 * (wave >> 1) + (lane & 31);     // query row of the operand (block i adds 32)
  const int tb = 64 * (wave & 1) + (lane & 31);      // target row of the operand (block j adds 32)

  gload(0);
  sstore(0);
  if (nchunk > 1) gload(1);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    // chunk ch+1 (loaded during the previous iteration) -> the other buffer, free since the
    // barrier that ended iteration ch-1; then the loads of chunk ch+2 take off
    if (ch + 1 < nchunk) sstore(buf ^ 1);
    if (ch + 2 < nchunk) gload(ch + 2);
    // Operands of pair cc+1 are read right behind the first MFMA of pair cc (inline asm: hipcc
    // sinks compiler-visible reads behind the fourth MFMA and then waits for them with an empty
    // matrix pipe), into the other register pair; `lds` is the only __shared__ object, so LDS
    // byte addresses are plain offsets into it.
    const unsigned int qaddr = (unsigned int)((((buf * 2 + 0) * 2 + mh) * F3_KC * F3_LD + qb) * 4);
    const unsigned int taddr = (unsigned int)((((buf * 2 + 1) * 2 + mh) * F3_KC * F3_LD + tb) * 4);
    f32x2 qv[2], tv[2];
    auto oread = [&](f32x2& q2, f32x2& t2, int cc) {
      const unsigned int qa_ = qaddr + cc * (F3_LD * 4), ta_ = taddr + cc * (F3_LD * 4);
      asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(q2) : "v"(qa_) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(t2) : "v"(ta_) : "memory");
    };
    oread(qv[0], tv[0], 0);
#pragma unroll
    for (int cc = 0; cc < F3_KC; ++cc) {
      const int cur = cc & 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[cur][0], tv[cur][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (cc + 1 < F3_KC) oread(qv[cur ^ 1], tv[cur ^ 1], cc + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[cur][0], tv[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[cur][1], tv[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[cur][1], tv[cur][1], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // the next pair's wait stays behind these three
    }
    __syncthreads();
  }

  // D[i][j]: lane holds column j = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 mh
  if constexpr (RANK) {
    static_assert(sizeof(lds) >= F3_BM * F3_LD * 4, "the score tile fits the operand buffers");
    float* const tile = &lds[0][0][0][0][0];  // [128][F3_LD]; the loop ended with a barrier
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int lc = 64 * (wave & 1) + 32 * bj + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = 64 * (wave >> 1) + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * mh;
          tile[lr * F3_LD + lc] = acc[bi][bj][r];
        }
      }
    __syncthreads();
    rank_acc_add<F3_BM, F3_BN, F3_LD>(racc, tile, row0, col0, n, m, rk, tid);
    continue;
  }
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const long long ocol = col0 + 64 * (wave & 1) + 32 * bj + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long orow = row0 + 64 * (wave >> 1) + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * mh;
        if (orow < n && ocol < m) out[orow * ldo + ocol] = acc[bi][bj][r];
      }
    }
  }  // column tiles
  if constexpr (RANK) rank_acc_flush<F3_BM>(racc, row0, n, rk, tid);
}

template <int SCORER, typename T>
static int launch_pairs_f32(const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                            long long m, int round_q, float* out, long long ldo, hipStream_t st, const RankArgs* rk) {
  dim3 grid((unsigned)((m + F3_BN - 1) / F3_BN), (unsigned)((n + F3_BM - 1) / F3_BM));
  if (rk != nullptr) {
    // column tiles per workgroup: as many as still leave two workgroups per compute unit (at most 16) -- fewer
    // workgroups than that cost more than the saved atomics (one round of 228 workgroups walking two tiles each:
    // 0.35 -> 0.41 ms per evaluation batch at the FB15k-237 shape)
    RankArgs r2 = *rk;
    long long ct = (long long)grid.x * grid.y / 512;
    r2.col_tiles = (int)(ct < 1 ? 1 : (ct > 16 ? 16 : ct));
    grid.x = (grid.x + r2.col_tiles - 1) / r2.col_tiles;
    hipLaunchKernelGGL((pairs_f32_kernel<SCORER, T, true>), grid, dim3(256), 0, st, A, R, TG, dir, d, n, m, round_q,
                       out, ldo, r2);
  } else
    hipLaunchKernelGGL((pairs_f32_kernel<SCORER, T, false>), grid, dim3(256), 0, st, A, R, TG, dir, d, n, m, round_q,
                       out, ldo, RankArgs{});
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ComplEx / DistMult, vectorisable layout (d % 8 == 0, 16-byte aligned rows): the caller checked.  rk != NULL:
// counts instead of scores (`out` is not touched).
int run_pairs_f32(int scorer, int dtype, const Operand& A, const Operand& R, const Operand& TG, int dir, int d,
                  long long n, long long m, int round_q, float* out, long long ldo, hipStream_t st,
                  const RankArgs* rk) {
  if (n > 65535LL * F3_BM) return KGE_ERR_UNSUPPORTED;
  if (scorer == KGE_COMPLEX)
    return dtype == KGE_BF16
               ? launch_pairs_f32<KGE_COMPLEX, unsigned short>(A, R, TG, dir, d, n, m, round_q, out, ldo, st, rk)
               : launch_pairs_f32<KGE_COMPLEX, float>(A, R, TG, dir, d, n, m, round_q, out, ldo, st, rk);
  if (scorer == KGE_DISTMULT)
    return dtype == KGE_BF16
               ? launch_pairs_f32<KGE_DISTMULT, unsigned short>(A, R, TG, dir, d, n, m, round_q, out, ldo, st, rk)
               : launch_pairs_f32<KGE_DISTMULT, float>(A, R, TG, dir, d, n, m, round_q, out, ldo, st, rk);
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
