// bwd_gemm32.hip -- the two gradient contractions of the ComplEx / DistMult backward on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32), hand-written: what the reference gets from autograd's mm backward on float32 parameters
// (complex.py:30-39 / distmult.py:15-21 under loss.backward(), kge/job/train_1vsAll.py:70,81), and -- on widened
// bf16 operands -- the fallback / cross-check of the bf16 kernels of bwd_gemm16.hip for the shapes they decline.
//
//     C[M, N] = A * B     B = [K, N] row-major (N contiguous)
//     A_KCONT:  A = [M, K] row-major (K contiguous)        dQ = G * T       M = n, N = d, K = m (long: split-K)
//     !A_KCONT: A given as [K, M] row-major (M contiguous)  dT = G^T * Q     M = m, N = d, K = n
//
// Structure of pairs_f32_kernel (score_pairs_f32.hip), whose inner loop this shares: 128 x 128 tile per 256-thread
// workgroup, each wave a 64 x 64 quadrant = 2 x 2 accumulators of 32 x 32 (one operand read serves two MFMAs),
// K chunks of 32 double-buffered in LDS as [k & 1][k >> 1][row] (row fastest: conflict-free ds_read2_b32 of the
// MFMA operands), the global loads of chunk c + 2 in flight and the LDS stores of chunk c + 1 issued while chunk c
// is multiplied, one barrier per chunk.  A K-contiguous A is transposed on its way into LDS (scalar stores); the
// other operands go in as 16-byte stores.  Split-K over blockIdx.z into partial outputs (caller's scratch) summed
// by bwdg_reduce_kernel: the 512 x 512 x 14,541 dQ product is 16 output tiles.  Ragged M, N, K: guarded.
// Bound: MFMA f32, 2 M N K flops.
#include "common.hpp"

namespace kge {

constexpr int G32_BN = 128, G32_KC = 32;
typedef float f32x2g __attribute__((ext_vector_type(2)));

__global__ void bwdg_reduce_kernel(const float* __restrict__ part, long long cnt, int P, float* __restrict__ out,
                                   float* __restrict__ zero, long long zero_cnt);

// 4 consecutive elements from p (valid: how many of them exist), widened to f32
template <typename T>
__device__ __forceinline__ f32x4 g32_load4(const T* p, long long valid, bool vec) {
  if (valid >= 4 && vec) return ld4<T>(p);
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < valid) r[e] = ld1<T>(p + e);
  return r;
}

// BM = 256: 512 threads, eight waves (two per SIMD: one wave's LDS waits and barriers hide behind the other's MFMAs),
// one workgroup per compute unit -- 57 x 4 = 228 tiles for dT at the FB15k-237 shape, ONE round; BM = 128: 256 threads,
// for outputs with few rows.
template <typename T, bool A_KCONT, int BM>
__global__ __launch_bounds__(2 * BM) void gemm32_kernel(const T* __restrict__ A, long long lda, const T* __restrict__ B,
                                                        long long ldb, float* __restrict__ C, long long ldc, long long M,
                                                        long long N, long long K, long long kc, long long slot, int vec) {
  constexpr int THREADS = 2 * BM, LDA = BM + 4, LDB = G32_BN + 4, KH = G32_KC / 2;
  constexpr int BUF = 2 * KH * (LDA + LDB);  // floats per buffer: A as [k & 1][k >> 1][row], then B likewise
  constexpr int NB = G32_KC * G32_BN / 4 / THREADS;  // 16-byte pieces of the B chunk per thread
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long col0 = (long long)blockIdx.x * G32_BN, row0 = (long long)blockIdx.y * BM;
  const long long k_lo = (long long)blockIdx.z * kc;
  const long long k_hi = k_lo + kc < K ? k_lo + kc : K;
  const int nchunk = (int)((k_hi - k_lo + G32_KC - 1) / G32_KC);
  C += (long long)blockIdx.z * slot;
  auto a_at = [&](int buf, int k, int row) -> float* { return lds + buf * BUF + ((k & 1) * KH + (k >> 1)) * LDA + row; };
  auto b_at = [&](int buf, int k, int col) -> float* {
    return lds + buf * BUF + 2 * KH * LDA + ((k & 1) * KH + (k >> 1)) * LDB + col;
  };

  f32x4 ra[4], rb[NB];
  // B: thread t takes k rows (t >> 5) + (THREADS / 32) j and the columns 4 (t & 31) .. + 3
  const int skr = tid >> 5, sc4 = (tid & 31) * 4;
  // an M-contiguous A: k rows t / (BM / 4) + 8 j, tile rows 4 (t % (BM / 4)) .. + 3
  const int akr = tid / (BM / 4), ar4 = (tid % (BM / 4)) * 4;
  // a K-contiguous A: tile row t >> 1 and the k range 16 (t & 1) .. + 15
  const int sar = tid >> 1, sak = (tid & 1) * 16;
  // Interior chunks (the tile inside M x N, the chunk inside the K range, 16-byte-aligned rows: everything but the
  // ragged edges) load with plain vector loads; the guarded form -- per-element validity, lane-divergent, so the
  // compiler serialises the loads behind exec-mask branches -- only at the edges.  The branch is uniform.
  const bool inner = vec != 0 && row0 + BM <= M && col0 + G32_BN <= N;
  auto gload = [&](int ch) {
    const long long k0 = k_lo + (long long)ch * G32_KC;
    if (inner && k0 + G32_KC <= k_hi) {
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = ld4<T>(B + (k0 + skr + (THREADS / 32) * j) * ldb + col0 + sc4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (!A_KCONT) ra[j] = ld4<T>(A + (k0 + akr + 8 * j) * lda + row0 + ar4);
        else ra[j] = ld4<T>(A + (row0 + sar) * lda + k0 + sak + 4 * j);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const long long k = k0 + skr + (THREADS / 32) * j;
      const long long c = col0 + sc4;
      rb[j] = (k < k_hi && c < N) ? g32_load4<T>(B + k * ldb + c, N - c, vec != 0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (!A_KCONT) {
        const long long k = k0 + akr + 8 * j, r = row0 + ar4;
        ra[j] = (k < k_hi && r < M) ? g32_load4<T>(A + k * lda + r, M - r, vec != 0) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const long long r = row0 + sar, kk = k0 + sak + 4 * j;
        ra[j] = (r < M && kk < k_hi) ? g32_load4<T>(A + r * lda + kk, k_hi - kk, vec != 0) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(b_at(buf, skr + (THREADS / 32) * j, sc4)) = rb[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (!A_KCONT) {
        *reinterpret_cast<f32x4*>(a_at(buf, akr + 8 * j, ar4)) = ra[j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *a_at(buf, sak + 4 * j + e, sar) = ra[j][e];
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
  const int mh = lane >> 5;
  const int ab = 64 * (wave >> 1) + (lane & 31);  // A-operand row of this lane (block i adds 32)
  const int bb = 64 * (wave & 1) + (lane & 31);   // B-operand column (block j adds 32)

  if (nchunk > 0) {
    gload(0);
    sstore(0);
    if (nchunk > 1) gload(1);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) sstore(buf ^ 1);
    if (ch + 2 < nchunk) gload(ch + 2);
    // (`lds` is the only __shared__ object: LDS byte addresses are plain offsets into it)
    const unsigned int aaddr = (unsigned int)((buf * BUF + mh * KH * LDA + ab) * 4);
    const unsigned int baddr = (unsigned int)((buf * BUF + 2 * KH * LDA + mh * KH * LDB + bb) * 4);
    // operands of k-step cc + 2 are requested behind the first MFMA of step cc (three register pairs in rotation)
    f32x2g av[3], bv[3];
    auto oread = [&](f32x2g& a2, f32x2g& b2, int cc) {
      const unsigned int aa = aaddr + cc * (LDA * 4), ba = baddr + cc * (LDB * 4);
      asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(a2) : "v"(aa) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(b2) : "v"(ba) : "memory");
    };
    oread(av[0], bv[0], 0);
    oread(av[1], bv[1], 1);
#pragma unroll
    for (int cc = 0; cc < KH; ++cc) {
      const int cur = cc % 3, nxt = (cc + 2) % 3;
      // the reads of step cc have returned once at most those of step cc + 1 (two instructions) are outstanding
      if (cc + 1 < KH) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][0], bv[cur][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (cc + 2 < KH) oread(av[nxt], bv[nxt], cc + 2);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][0], bv[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][1], bv[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][1], bv[cur][1], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
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
        if (orow < M && ocol < N) C[orow * ldc + ocol] = acc[bi][bj][r];
      }
    }
}

// C[M, N] (row-major, ldc) = A * B (see the top).  in16: bf16 operands (widened: the products are exact).  Split-K
// needs `scratch` (P * M * N floats) and ldc == N; without it the reduction runs in one workgroup per tile.
bool run_gemm32(bool a_kcont, int in16, long long M, long long N, long long K, const void* A, long long lda,
                const void* B, long long ldb, float* C, long long ldc, float* scratch, size_t scratch_bytes,
                hipStream_t st) {
  if (M <= 0 || N <= 0) return true;
  if (M >= (1LL << 37) || N >= (1LL << 22)) return false;
  // 256-row tiles (eight waves) whenever the output has that many rows: at a tile count near the number of compute
  // units one workgroup per compute unit is one round (228 tiles for dT at the FB15k-237 shape)
  const int BM = M > 128 ? 256 : 128;
  const long long tm = (M + BM - 1) / BM, tn = (N + G32_BN - 1) / G32_BN;
  if (tm > 0x7fffffffLL || tn > 65535) return false;
  const int es = in16 ? 2 : 4, al = in16 ? 8 : 16;
  const int vec = (((uintptr_t)A % al) == 0 && ((uintptr_t)B % al) == 0 && (lda * es) % al == 0 && (ldb * es) % al == 0)
                      ? 1 : 0;
  long long P = 1, kc = K > 0 ? K : 1;
  if (K > 4 * G32_KC && tm * tn < 128 && scratch != nullptr && ldc == N && (M * N) % 4 == 0) {
    P = 256 / (tm * tn);  // one workgroup per compute unit
    const long long fit = (long long)(scratch_bytes / ((size_t)M * N * 4));
    if (P > fit) P = fit;
    if (P > K / (2 * G32_KC)) P = K / (2 * G32_KC);
    if (P < 2) P = 1;
    if (P > 1) {
      kc = ((K + P - 1) / P + G32_KC - 1) / G32_KC * G32_KC;  // chunk-aligned K ranges: aligned vector loads
      P = (K + kc - 1) / kc;
    }
  }
  float* out = P > 1 ? scratch : C;
  const long long oldc = P > 1 ? N : ldc;
  dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)P);
#define KGE_G32(TT, KC_, BM_)                                                                                      \
  hipLaunchKernelGGL((gemm32_kernel<TT, KC_, BM_>), grid, dim3(2 * BM_), 0, st, (const TT*)A, lda, (const TT*)B, ldb, \
                     out, oldc, M, N, K, kc, M * N, vec)
#define KGE_G32B(TT, KC_)                                    \
  do {                                                       \
    if (BM == 256) KGE_G32(TT, KC_, 256); else KGE_G32(TT, KC_, 128); \
  } while (0)
  if (in16) {
    if (a_kcont) KGE_G32B(unsigned short, true); else KGE_G32B(unsigned short, false);
  } else {
    if (a_kcont) KGE_G32B(float, true); else KGE_G32B(float, false);
  }
#undef KGE_G32B
#undef KGE_G32
  if (P > 1) {
    const long long cnt = M * N;
    hipLaunchKernelGGL(bwdg_reduce_kernel, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, st, scratch, cnt,
                       (int)P, C, (float*)nullptr, 0LL);
  }
  return hipGetLastError() == hipSuccess;
}

}  // namespace kge
