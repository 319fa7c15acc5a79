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

template <int SCORER, typename T>
__global__ __launch_bounds__(256) void pairs_f32_kernel(Operand A, Operand R, Operand TG, int dir, int d,
                                                        long long n, long long m, int round_q,
                                                        float* __restrict__ out, long long ldo) {
  // [buffer][q|t][half][pair][row]
  __shared__ __attribute__((aligned(16))) float lds[2][2][2][F3_KC][F3_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long col0 = (long long)blockIdx.x * F3_BN;
  const long long row0 = (long long)blockIdx.y * F3_BM;
  const int hh = d / 2;
  const int nchunk = (hh + F3_KC - 1) / F3_KC;

  // staging role: 2 query units and 2 target units per thread; unit u = (row sr[u], coordinates
  // 4*scq .. +3 of the chunk)
  const int scq = tid & 3;
  const T* arow[2];
  const T* rrow[2];
  const T* trow[2];
  int sr[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    sr[u] = (tid >> 2) + 64 * u;
    long long qr = row0 + sr[u];
    if (qr >= n) qr = n - 1;  // clamp: rows beyond n are computed but never stored
    long long tr = col0 + sr[u];
    if (tr >= m) tr = m - 1;
    arow[u] = (const T*)A.base + index_at(A.idx, qr) * A.ld;
    rrow[u] = (const T*)R.base + index_at(R.idx, qr) * R.ld;
    trow[u] = (const T*)TG.base + index_at(TG.idx, tr) * TG.ld;
  }

  f32x4 a0[2], a1[2], r0[2], r1[2], t0[2], t1[2];
  auto gload = [&](int ch) {
    const int c = ch * F3_KC + scq * 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (c >= hh) {  // chunk tail beyond the row (hh % 4 == 0 on this path): zeros
        a0[u] = a1[u] = r0[u] = r1[u] = t0[u] = t1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        a0[u] = ld4<T>(arow[u] + c);
        a1[u] = ld4<T>(arow[u] + hh + c);
        r0[u] = ld4<T>(rrow[u] + c);
        r1[u] = ld4<T>(rrow[u] + hh + c);
        t0[u] = ld4<T>(trow[u] + c);
        t1[u] = ld4<T>(trow[u] + hh + c);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 q0, q1;
      build_q4<SCORER>(dir, a0[u], a1[u], r0[u], r1[u], q0, q1);
      if (round_q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          q0[i] = round_bf16(q0[i]);
          q1[i] = round_bf16(q1[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lds[buf][0][0][scq * 4 + i][sr[u]] = q0[i];
        lds[buf][0][1][scq * 4 + i][sr[u]] = q1[i];
        lds[buf][1][0][scq * 4 + i][sr[u]] = t0[u][i];
        lds[buf][1][1][scq * 4 + i][sr[u]] = t1[u][i];
      }
    }
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
#pragma unroll
    for (int cc = 0; cc < F3_KC; ++cc) {
      const float qa = lds[buf][0][mh][cc][qb], qc = lds[buf][0][mh][cc][qb + 32];
      const float ta = lds[buf][1][mh][cc][tb], tc = lds[buf][1][mh][cc][tb + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, ta, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, tc, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc, ta, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc, tc, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // D[i][j]: lane holds column j = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 mh
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
}

template <int SCORER, typename T>
static int launch_pairs_f32(const Operand& A, const Operand& R, const Operand& TG, int dir, int d, long long n,
                            long long m, int round_q, float* out, long long ldo, hipStream_t st) {
  dim3 grid((unsigned)((m + F3_BN - 1) / F3_BN), (unsigned)((n + F3_BM - 1) / F3_BM));
  hipLaunchKernelGGL((pairs_f32_kernel<SCORER, T>), grid, dim3(256), 0, st, A, R, TG, dir, d, n, m, round_q, out,
                     ldo);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ComplEx / DistMult, vectorisable layout (d % 8 == 0, 16-byte aligned rows): the caller checked
int run_pairs_f32(int scorer, int dtype, const Operand& A, const Operand& R, const Operand& TG, int dir, int d,
                  long long n, long long m, int round_q, float* out, long long ldo, hipStream_t st) {
  if (n > 65535LL * F3_BM) return KGE_ERR_UNSUPPORTED;
  if (scorer == KGE_COMPLEX)
    return dtype == KGE_BF16
               ? launch_pairs_f32<KGE_COMPLEX, unsigned short>(A, R, TG, dir, d, n, m, round_q, out, ldo, st)
               : launch_pairs_f32<KGE_COMPLEX, float>(A, R, TG, dir, d, n, m, round_q, out, ldo, st);
  if (scorer == KGE_DISTMULT)
    return dtype == KGE_BF16
               ? launch_pairs_f32<KGE_DISTMULT, unsigned short>(A, R, TG, dir, d, n, m, round_q, out, ldo, st)
               : launch_pairs_f32<KGE_DISTMULT, float>(A, R, TG, dir, d, n, m, round_q, out, ldo, st);
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
