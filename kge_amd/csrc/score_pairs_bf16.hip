// score_pairs_bf16.hip -- ComplEx / DistMult sp_ and _po scoring on bf16 tables with the
// bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16): the BASELINE.json headline
// path ("FB15k-237 ComplEx d=512 1vsAll bf16").
//
//   score[i, j] = sum_k q_i[k] * T_j[k],   q_i = bf16( s_i (x) r_i )  (complex.py:30-39)
//
// One launch fuses: gather of the n query rows (s and r), the complex/Hadamard query
// build in f32, rounding of q to bf16 (a bf16 GEMM operand), the [n x d] x [d x m]
// contraction with f32 accumulation, and the f32 score store.  The f32 score matrix
// dominates HBM traffic (n*m*4 of n*m*4 + m*d*2 bytes), so the kernel is HBM-bound.
//
//   tile       128 query rows x 128 targets per 256-thread workgroup (2x2 waves, each
//              2x2 MFMA 32x32 tiles = 64 accumulator VGPRs)
//   K step     32 coordinate pairs = 64 k values: LDS rows of 128 B =
//              [first-half c0..c0+31 | second-half c0..c0+31]
//   LDS        16-byte slot s of row r lives at slot s ^ ((r >> 1) & 7): the four 16-lane
//              groups of ds_read_b128 each cover all 64 banks (conflict free)
//   grid       1-D, XCD-aware: workgroup b runs on XCD b % 8; each XCD gets a contiguous
//              range of target tiles, so every XCD streams 1/8 of the table through its own
//              4 MiB L2 and the table is read from HBM once
//
// Not bit-reproducible (the MFMA's internal summation order is unspecified): compared
// with the oracle under the reference tolerance; run_pairs_exact() is the exact twin.
#include "common.hpp"

namespace kge {

constexpr int BT_BM = 128, BT_BN = 128, BT_KC = 32;

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// q for two coordinates packed in one dword of each operand (elements 2i, 2i+1)
template <int SCORER>
__device__ __forceinline__ void q_pair(int dir, unsigned int a0, unsigned int a1,
                                       unsigned int r0, unsigned int r1, unsigned int& q0,
                                       unsigned int& q1) {
  float a0l = bf_lo(a0), a0h = bf_hi(a0), a1l = bf_lo(a1), a1h = bf_hi(a1);
  float r0l = bf_lo(r0), r0h = bf_hi(r0), r1l = bf_lo(r1), r1h = bf_hi(r1);
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
  q0 = pack_bf16x2(q0l, q0h);
  q1 = pack_bf16x2(q1l, q1h);
}

__device__ __forceinline__ int lds_off(int row, int slot) {
  return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

template <int SCORER>
__global__ __launch_bounds__(256) void pairs_bf16_kernel(Operand A, Operand R, Operand TG,
                                                         int dir, int d, long long n,
                                                         long long m, int ntm, int ntn,
                                                         float* __restrict__ out,
                                                         long long ldo) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BT_BM * 128];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BT_BM * 128;

  // XCD-aware tile id: consecutive ids of one XCD walk down the row tiles of a target tile
  const int total = ntm * ntn;
  const int per_xcd = (total + 7) / 8;
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || lin >= total) return;
  const int tile_n = lin / ntm, tile_m = lin - tile_n * ntm;
  const long long row0 = (long long)tile_m * BT_BM, col0 = (long long)tile_n * BT_BN;

  const int tid = threadIdx.x;
  const int hh = d >> 1;
  const int nchunk = hh / BT_KC;

  // staging role: row sr, coordinates 16*shc .. +15 of the chunk
  const int sr = tid >> 1, shc = tid & 1;
  long long qrow = row0 + sr;
  if (qrow >= n) qrow = n - 1;
  long long trow = col0 + sr;
  if (trow >= m) trow = m - 1;
  const unsigned short* arow = (const unsigned short*)A.base + index_at(A.idx, qrow) * A.ld;
  const unsigned short* rrow = (const unsigned short*)R.base + index_at(R.idx, qrow) * R.ld;
  const unsigned short* tgrow = (const unsigned short*)TG.base + index_at(TG.idx, trow) * TG.ld;

  u32x4 a0[2], a1[2], r0[2], r1[2], t0[2], t1[2];
  auto gload = [&](int ch) {
    const int c = ch * BT_KC + shc * 16;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      a0[v] = *reinterpret_cast<const u32x4*>(arow + c + 8 * v);
      a1[v] = *reinterpret_cast<const u32x4*>(arow + hh + c + 8 * v);
      r0[v] = *reinterpret_cast<const u32x4*>(rrow + c + 8 * v);
      r1[v] = *reinterpret_cast<const u32x4*>(rrow + hh + c + 8 * v);
      t0[v] = *reinterpret_cast<const u32x4*>(tgrow + c + 8 * v);
      t1[v] = *reinterpret_cast<const u32x4*>(tgrow + hh + c + 8 * v);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      u32x4 q0, q1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned int x0, x1;
        q_pair<SCORER>(dir, a0[v][i], a1[v][i], r0[v][i], r1[v][i], x0, x1);
        q0[i] = x0;
        q1[i] = x1;
      }
      *reinterpret_cast<u32x4*>(As + lds_off(sr, 2 * shc + v)) = q0;
      *reinterpret_cast<u32x4*>(As + lds_off(sr, 4 + 2 * shc + v)) = q1;
      *reinterpret_cast<u32x4*>(Bs + lds_off(sr, 2 * shc + v)) = t0[v];
      *reinterpret_cast<u32x4*>(Bs + lds_off(sr, 4 + 2 * shc + v)) = t1[v];
    }
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  gload(0);
  sstore();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk) gload(ch + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(As + lds_off(64 * wr + 32 * i + fr, 2 * kk + fh));
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + lds_off(64 * wc + 32 * i + fr, 2 * kk + fh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (ch + 1 < nchunk) sstore();
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long orow = row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (orow >= n) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long long ocol = col0 + 64 * wc + 32 * j + fr;
        if (ocol < m) out[orow * ldo + ocol] = acc[i][j][r];
      }
    }
}

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Can this request run on the bf16 MFMA kernel?
bool pairs_bf16_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                          const Operand& TG) {
  if (dtype != KGE_BF16) return false;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return false;
  if (d % (2 * BT_KC)) return false;
  if (!al16(A.base) || !al16(R.base) || !al16(TG.base)) return false;
  if ((A.ld % 8) || (R.ld % 8) || (TG.ld % 8)) return false;
  return true;
}

int run_pairs_bf16(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir,
                   int d, long long n, long long m, float* out, long long ldo,
                   hipStream_t st) {
  if (n == 0 || m == 0) return KGE_OK;
  const int ntm = (int)((n + BT_BM - 1) / BT_BM), ntn = (int)((m + BT_BN - 1) / BT_BN);
  const long long total = (long long)ntm * ntn;
  const long long per_xcd = (total + 7) / 8;
  dim3 grid((unsigned)(per_xcd * 8));
  if (scorer == KGE_COMPLEX)
    hipLaunchKernelGGL((pairs_bf16_kernel<KGE_COMPLEX>), grid, dim3(256), 0, st, A, R, TG, dir,
                       d, n, m, ntm, ntn, out, ldo);
  else
    hipLaunchKernelGGL((pairs_bf16_kernel<KGE_DISTMULT>), grid, dim3(256), 0, st, A, R, TG,
                       dir, d, n, m, ntm, ntn, out, ldo);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
