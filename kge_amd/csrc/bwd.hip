// bwd.hip -- backward twins of the scoring kernels, gfx950.
//
// The reference gets these from torch autograd through its scorer ops (complex.py:30-39,
// distmult.py:15-21, transe.py:18-34, rotate.py:30-64), triggered at
// kge/job/train_1vsAll.py:70,81, train_KvsAll.py:293-294, train_negative_sampling.py:161.
// Tolerance-level parity (autograd's summation order is unspecified); f32 only.
//
//   bwd_pairs_kernel<.., WHICH=0>  dQ[i,:] = sum_j g_ij * dscore_ij/dq_i, chained in the
//                                  epilogue to the gathered entity row (g_a) and relation
//                                  row (g_p) of query i
//   bwd_pairs_kernel<.., WHICH=1>  dT[j,:] = sum_i g_ij * dscore_ij/dt_j  (g_tgt)
//   bwd_spo_kernel                 row-wise gradients of score_spo
//
// Tile: 64 output rows x 32 coordinate pairs (both halves) per 256-thread workgroup, the
// reduction index streamed through LDS in chunks of 16; each thread owns 4 rows x 2
// coordinate pairs.
#include "common.hpp"

namespace kge {

constexpr int BW_TR = 64, BW_TC = 32, BW_KY = 16;

__device__ __forceinline__ float ldf(const float* row, int k, int limit) {
  return k < limit ? row[k] : 0.0f;
}

// weight of one distance component e (TransE) given the pair's distance
template <int NORM>
__device__ __forceinline__ float transe_w(float e, float dist, float p) {
  if (NORM == NORM_L1) return (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
  if (NORM == NORM_L2) return dist > 0.f ? e / dist : 0.f;
  if (dist <= 0.f || e == 0.f) return 0.f;
  float ae = __builtin_fabsf(e);
  return (e > 0.f ? 1.f : -1.f) * powf(ae, p - 1.f) / powf(dist, p - 1.f);
}

// weights (wre, wim) of one complex distance component (RotatE)
template <int NORM>
__device__ __forceinline__ void rotate_w(float dre, float dim_, float dist, float p, float& wre,
                                         float& wim) {
  float ab = sqrt_rn_fast(__builtin_fmaf(dim_, dim_, dre * dre));  // (correctly rounded: common.hpp)
  float f;
  if (NORM == NORM_L1) f = ab > 0.f ? 1.f / ab : 0.f;
  else if (NORM == NORM_L2) f = dist > 0.f ? 1.f / dist : 0.f;
  else f = (dist > 0.f && ab > 0.f) ? powf(ab, p - 2.f) / powf(dist, p - 1.f) : 0.f;
  wre = dre * f;
  wim = dim_ * f;
}

template <int SCORER, int NORM, int WHICH>
__global__ __launch_bounds__(256) void bwd_pairs_kernel(Operand A, Operand R, Operand TG, int dir,
                                                        int d, int dr, long long n, long long m,
                                                        float lp, const float* __restrict__ gout,
                                                        long long ldg,
                                                        const float* __restrict__ scores,
                                                        long long lds, float* __restrict__ g_a,
                                                        float* __restrict__ g_p,
                                                        float* __restrict__ g_tgt, long long ychunk) {
  // ychunk < Y (WHICH == 0 only): the reduction over the other side is split over blockIdx.z and
  // the (linear) chain-ruled partial sums are added atomically into pre-zeroed g_a / g_p -- with
  // 512 query rows the unsplit grid is 64 workgroups on a 256-CU chip
  constexpr bool DOT = (SCORER == KGE_COMPLEX || SCORER == KGE_DISTMULT);
  constexpr bool NEED_DIST = !DOT && NORM != NORM_L1;
  __shared__ float Gs[BW_KY][BW_TR + 4];
  __shared__ float Ds[BW_KY][BW_TR + 4];
  __shared__ float V0[BW_KY][BW_TC + 1];
  __shared__ float V1[BW_KY][BW_TC + 1];

  const int tid = threadIdx.x;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  const int c0 = blockIdx.x * BW_TC;
  const long long row0 = (long long)blockIdx.y * BW_TR;
  const long long X = WHICH == 0 ? n : m, Y = WHICH == 0 ? m : n;
  const int tx = tid & 15, ty = tid >> 4;

  // own-side values for this thread's 4 rows x 2 coordinates
  float own0[4][2], own1[4][2];
  float oa0[4][2], oa1[4][2], or0[4][2], or1[4][2];  // WHICH==0: gathered a / r values
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long x = row0 + ty * 4 + i;
    if (x >= X) x = X - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + tx * 2 + j;
      if (WHICH == 0) {
        const float* arow = (const float*)A.base + index_at(A.idx, x) * A.ld;
        const float* rrow = (const float*)R.base + index_at(R.idx, x) * R.ld;
        f32x4 a0v{ldf(arow, c, hh), 0, 0, 0}, a1v{ldf(arow + hh, c, lim1), 0, 0, 0};
        f32x4 r0v{ldf(rrow, c, rl0), 0, 0, 0}, r1v{0, 0, 0, 0};
        if (SCORER != KGE_ROTATE) r1v[0] = ldf(rrow + hh, c, rl1);
        f32x4 q0v, q1v;
        build_q4<SCORER>(dir, a0v, a1v, r0v, r1v, q0v, q1v);
        own0[i][j] = q0v[0];
        own1[i][j] = q1v[0];
        oa0[i][j] = a0v[0]; oa1[i][j] = a1v[0]; or0[i][j] = r0v[0]; or1[i][j] = r1v[0];
      } else {
        const float* trow = (const float*)TG.base + index_at(TG.idx, x) * TG.ld;
        own0[i][j] = ldf(trow, c, hh);
        own1[i][j] = ldf(trow + hh, c, lim1);
      }
    }
  }

  float acc0[4][2], acc1[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc0[i][j] = acc1[i][j] = 0.f;

  const bool split = ychunk < Y;
  const long long ybeg = split ? (long long)blockIdx.z * ychunk : 0;
  const long long yend = split ? (ybeg + ychunk < Y ? ybeg + ychunk : Y) : Y;
  for (long long y0 = ybeg; y0 < yend; y0 += BW_KY) {
    // ---- stage weights g (and distances) of this chunk
    if (WHICH == 0) {
      const int x = tid >> 2, yq = (tid & 3) * 4;
      const long long gx = row0 + x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long gy = y0 + yq + k;
        const bool ok = gx < X && gy < yend;
        Gs[yq + k][x] = ok ? gout[gx * ldg + gy] : 0.f;
        if (NEED_DIST) Ds[yq + k][x] = ok ? -scores[gx * lds + gy] : 0.f;
      }
    } else {
      const int y = tid >> 4, xq = (tid & 15) * 4;
      const long long gy = y0 + y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long gx = row0 + xq + k;
        const bool ok = gx < X && gy < yend;
        Gs[y][xq + k] = ok ? gout[gy * ldg + gx] : 0.f;
        if (NEED_DIST) Ds[y][xq + k] = ok ? -scores[gy * lds + gx] : 0.f;
      }
    }
    // ---- stage the other side's vectors: 16 rows x 32 coordinates x 2 halves
    {
      const int y = tid >> 4, cq = (tid & 15) * 2;
      long long gy = y0 + y;
      const bool ok = gy < yend;
      if (!ok) gy = Y - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = c0 + cq + j;
        float v0, v1;
        if (WHICH == 0) {
          const float* trow = (const float*)TG.base + index_at(TG.idx, gy) * TG.ld;
          v0 = ldf(trow, c, hh);
          v1 = ldf(trow + hh, c, lim1);
        } else {
          const float* arow = (const float*)A.base + index_at(A.idx, gy) * A.ld;
          const float* rrow = (const float*)R.base + index_at(R.idx, gy) * R.ld;
          f32x4 a0v{ldf(arow, c, hh), 0, 0, 0}, a1v{ldf(arow + hh, c, lim1), 0, 0, 0};
          f32x4 r0v{ldf(rrow, c, rl0), 0, 0, 0}, r1v{0, 0, 0, 0};
          if (SCORER != KGE_ROTATE) r1v[0] = ldf(rrow + hh, c, rl1);
          f32x4 q0v, q1v;
          build_q4<SCORER>(dir, a0v, a1v, r0v, r1v, q0v, q1v);
          v0 = q0v[0];
          v1 = q1v[0];
        }
        V0[y][cq + j] = ok ? v0 : 0.f;
        V1[y][cq + j] = ok ? v1 : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int y = 0; y < BW_KY; ++y) {
      float v0[2], v1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v0[j] = V0[y][tx * 2 + j];
        v1[j] = V1[y][tx * 2 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g = Gs[y][ty * 4 + i];
        const float dist = NEED_DIST ? Ds[y][ty * 4 + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (DOT) {
            acc0[i][j] = __builtin_fmaf(g, v0[j], acc0[i][j]);
            acc1[i][j] = __builtin_fmaf(g, v1[j], acc1[i][j]);
          } else {
            // e = q - t ; dscore/dq = -w(e), dscore/dt = +w(e)
            const float e0 = WHICH == 0 ? own0[i][j] - v0[j] : v0[j] - own0[i][j];
            const float e1 = WHICH == 0 ? own1[i][j] - v1[j] : v1[j] - own1[i][j];
            const float sg = WHICH == 0 ? -g : g;
            if (SCORER == KGE_TRANSE) {
              acc0[i][j] = __builtin_fmaf(sg, transe_w<NORM>(e0, dist, lp), acc0[i][j]);
              acc1[i][j] = __builtin_fmaf(sg, transe_w<NORM>(e1, dist, lp), acc1[i][j]);
            } else {
              float wre, wim;
              rotate_w<NORM>(e0, e1, dist, lp, wre, wim);
              acc0[i][j] = __builtin_fmaf(sg, wre, acc0[i][j]);
              acc1[i][j] = __builtin_fmaf(sg, wim, acc1[i][j]);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long x = row0 + ty * 4 + i;
    if (x >= X) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + tx * 2 + j;
      if (c >= hh) continue;
      const bool has1 = c < lim1;
      const float dq0 = acc0[i][j], dq1 = acc1[i][j];
      if (WHICH == 1) {
        g_tgt[x * d + c] = dq0;
        if (has1) g_tgt[x * d + hh + c] = dq1;
        continue;
      }
      const float a0 = oa0[i][j], a1 = oa1[i][j], r0 = or0[i][j], r1 = or1[i][j];
      float da0, da1, dr0, dr1 = 0.f;
      if (SCORER == KGE_DISTMULT) {
        da0 = dq0 * r0; da1 = dq1 * r1; dr0 = dq0 * a0; dr1 = dq1 * a1;
      } else if (SCORER == KGE_COMPLEX) {
        if (dir == KGE_SP_) {
          da0 = dq0 * r0 + dq1 * r1; da1 = dq1 * r0 - dq0 * r1;
          dr0 = dq0 * a0 + dq1 * a1; dr1 = dq1 * a0 - dq0 * a1;
        } else {
          da0 = dq0 * r0 - dq1 * r1; da1 = dq0 * r1 + dq1 * r0;
          dr0 = dq0 * a0 + dq1 * a1; dr1 = dq0 * a1 - dq1 * a0;
        }
      } else if (SCORER == KGE_TRANSE) {
        da0 = dq0; da1 = dq1;
        dr0 = dir == KGE_SP_ ? dq0 : -dq0;
        dr1 = dir == KGE_SP_ ? dq1 : -dq1;
      } else {  // ROTATE: r0 = phase
        float sn, cs;
        sincos_canon(r0, sn, cs);
        const float q0 = own0[i][j], q1 = own1[i][j];
        if (dir == KGE_SP_) {
          da0 = dq0 * cs + dq1 * sn; da1 = dq1 * cs - dq0 * sn;
          dr0 = dq1 * q0 - dq0 * q1;
        } else {
          da0 = dq0 * cs - dq1 * sn; da1 = dq0 * sn + dq1 * cs;
          dr0 = dq0 * q1 - dq1 * q0;
        }
      }
      if (split) {
        unsafeAtomicAdd(g_a + x * d + c, da0);
        if (has1) unsafeAtomicAdd(g_a + x * d + hh + c, da1);
        unsafeAtomicAdd(g_p + x * dr + c, dr0);
        if (SCORER != KGE_ROTATE && has1) unsafeAtomicAdd(g_p + x * dr + hh + c, dr1);
      } else {
        g_a[x * d + c] = da0;
        if (has1) g_a[x * d + hh + c] = da1;
        g_p[x * dr + c] = dr0;
        if (SCORER != KGE_ROTATE && has1) g_p[x * dr + hh + c] = dr1;
      }
    }
  }
}

// ---- score_spo backward ---------------------------------------------------------------------
// gradients of g * score(s, p, o) w.r.t. one coordinate pair (first-half element 0, second-half
// element 1) of the s, p and o rows
// (ROTPRE: RotatE with r0 = cos, r1 = sin of the phase, precomputed per relation -- rot_table_kernel)
template <int SCORER, int NORM, bool ROTPRE = false>
__device__ __forceinline__ void spo_pair_grads(float s0, float s1, float r0, float r1, float o0, float o1,
                                               bool has1, float g, float dist, float lp, float& ds0,
                                               float& ds1, float& dp0, float& dp1, float& do0, float& do1) {
  dp1 = 0.f;
  if (SCORER == KGE_DISTMULT) {
    ds0 = g * (r0 * o0); ds1 = g * (r1 * o1);
    dp0 = g * (s0 * o0); dp1 = g * (s1 * o1);
    do0 = g * (s0 * r0); do1 = g * (s1 * r1);
  } else if (SCORER == KGE_COMPLEX) {
    ds0 = g * (o0 * r0 + o1 * r1); ds1 = g * (o1 * r0 - o0 * r1);
    dp0 = g * (o0 * s0 + o1 * s1); dp1 = g * (o1 * s0 - o0 * s1);
    do0 = g * (s0 * r0 - s1 * r1); do1 = g * (s1 * r0 + s0 * r1);
  } else if (SCORER == KGE_TRANSE) {
    const float e0 = ((s0 + r0) - o0) + 1e-6f, e1 = ((s1 + r1) - o1) + 1e-6f;
    const float w0 = -g * transe_w<NORM>(e0, dist, lp);
    const float w1 = has1 ? -g * transe_w<NORM>(e1, dist, lp) : 0.f;
    ds0 = w0; ds1 = w1; dp0 = w0; dp1 = w1; do0 = -w0; do1 = -w1;
  } else {
    float sn, cs;
    if constexpr (ROTPRE) {
      cs = r0;
      sn = r1;
    } else {
      sincos_canon(r0, sn, cs);
    }
    const float q0 = s0 * cs - s1 * sn, q1 = s0 * sn + s1 * cs;
    float wre, wim;
    rotate_w<NORM>(q0 - o0, q1 - o1, dist, lp, wre, wim);
    const float dq0 = -g * wre, dq1 = -g * wim;
    ds0 = dq0 * cs + dq1 * sn; ds1 = dq1 * cs - dq0 * sn;
    dp0 = dq1 * q0 - dq0 * q1;
    do0 = -dq0; do1 = -dq1;
  }
}

// row-wise gradients: one wave per triple, lanes strided over the coordinate pairs
template <int SCORER, int NORM>
__global__ __launch_bounds__(256) void bwd_spo_kernel(Operand S, Operand R, Operand O, int d, int dr,
                                                      long long n, float lp,
                                                      const float* __restrict__ gout,
                                                      const float* __restrict__ scores,
                                                      float* __restrict__ g_s,
                                                      float* __restrict__ g_p,
                                                      float* __restrict__ g_o) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  const float* srow = (const float*)S.base + index_at(S.idx, i) * S.ld;
  const float* rrow = (const float*)R.base + index_at(R.idx, i) * R.ld;
  const float* orow = (const float*)O.base + index_at(O.idx, i) * O.ld;
  const float g = gout[i];
  const float dist = (SCORER == KGE_TRANSE || SCORER == KGE_ROTATE) && NORM != NORM_L1 ? -scores[i] : 0.f;
  for (int c = lane; c < hh; c += 64) {
    const bool has1 = c < lim1;
    const float s0 = srow[c], s1 = has1 ? srow[hh + c] : 0.f;
    const float o0 = orow[c], o1 = has1 ? orow[hh + c] : 0.f;
    const float r0 = c < rl0 ? rrow[c] : 0.f, r1 = c < rl1 ? rrow[hh + c] : 0.f;
    float ds0, ds1, dp0, dp1, do0, do1;
    spo_pair_grads<SCORER, NORM>(s0, s1, r0, r1, o0, o1, has1, g, dist, lp, ds0, ds1, dp0, dp1, do0, do1);
    g_s[i * d + c] = ds0;
    g_o[i * d + c] = do0;
    g_p[i * dr + c] = dp0;
    if (has1) {
      g_s[i * d + hh + c] = ds1;
      g_o[i * d + hh + c] = do1;
      if (SCORER != KGE_ROTATE) g_p[i * dr + hh + c] = dp1;
    }
  }
}

// The same gradients ACCUMULATED into the dense table gradients (what autograd's scatter-add of
// the gathered rows does afterwards: lookup_embedder.py:97 backward with sparse=False), without
// materialising three [n, d] row-gradient tensors: one wave walks a chunk of SPA_CH consecutive
// triples; the s-row and p-row gradients are summed in registers while the index stays the same
// (negative sampling scores n*K triples whose s and p repeat K times in a row, sampler.py:291-306:
// 1000 same-address atomics become one), the o-row gradient goes out as one float atomic per
// element.  d <= 1024 (8 coordinate pairs per lane).
constexpr int SPA_CH = 32, SPA_NC = 8;

template <int SCORER, int NORM>
__global__ __launch_bounds__(256) void bwd_spo_accum_kernel(Operand S, Operand R, Operand O, int d, int dr,
                                                            long long n, float lp,
                                                            const float* __restrict__ gout,
                                                            const float* __restrict__ scores,
                                                            float* __restrict__ ge, long long ge_ld,
                                                            float* __restrict__ gr, long long gr_ld, int ch) {
  // `ch` triples per wave: SPA_CH for the n*K triples of a negative-sampling batch (runs of equal s / p are
  // summed in registers), fewer when there are few triples -- a wave walks its chunk sequentially (index ->
  // rows -> atomics per triple), and the 512 positives of a batch in 16 chunks of 32 took 77 us of latency
  const long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long i0 = chunk * ch;
  if (i0 >= n) return;
  const long long i1 = i0 + ch < n ? i0 + ch : n;
  const int lane = threadIdx.x & 63;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  float as0[SPA_NC], as1[SPA_NC], ap0[SPA_NC], ap1[SPA_NC];
#pragma unroll
  for (int k = 0; k < SPA_NC; ++k) as0[k] = as1[k] = ap0[k] = ap1[k] = 0.f;
  long long run_s = -1, run_p = -1;
  auto flush = [&](float* tab, long long ld, long long row, float* a0, float* a1, int l0, int l1) {
    if (row < 0) return;
    float* dst = tab + row * ld;
#pragma unroll
    for (int k = 0; k < SPA_NC; ++k) {
      const int c = lane + 64 * k;
      if (c < l0) unsafeAtomicAdd(dst + c, a0[k]);
      if (c < l1) unsafeAtomicAdd(dst + hh + c, a1[k]);
      a0[k] = a1[k] = 0.f;
    }
  };
  for (long long i = i0; i < i1; ++i) {
    const long long si = index_at(S.idx, i), pi = index_at(R.idx, i), oi = index_at(O.idx, i);
    if (si != run_s) {
      flush(ge, ge_ld, run_s, as0, as1, hh, lim1);
      run_s = si;
    }
    if (pi != run_p) {
      flush(gr, gr_ld, run_p, ap0, ap1, rl0, rl1);
      run_p = pi;
    }
    const float* srow = (const float*)S.base + si * S.ld;
    const float* rrow = (const float*)R.base + pi * R.ld;
    const float* orow = (const float*)O.base + oi * O.ld;
    float* god = ge + oi * ge_ld;
    const float g = gout[i];
    const float dist = (SCORER == KGE_TRANSE || SCORER == KGE_ROTATE) && NORM != NORM_L1 ? -scores[i] : 0.f;
#pragma unroll
    for (int k = 0; k < SPA_NC; ++k) {
      const int c = lane + 64 * k;
      if (c >= hh) break;
      const bool has1 = c < lim1;
      const float s0 = srow[c], s1 = has1 ? srow[hh + c] : 0.f;
      const float o0 = orow[c], o1 = has1 ? orow[hh + c] : 0.f;
      const float r0 = c < rl0 ? rrow[c] : 0.f, r1 = c < rl1 ? rrow[hh + c] : 0.f;
      float ds0, ds1, dp0, dp1, do0, do1;
      spo_pair_grads<SCORER, NORM>(s0, s1, r0, r1, o0, o1, has1, g, dist, lp, ds0, ds1, dp0, dp1, do0, do1);
      as0[k] += ds0;
      as1[k] += ds1;
      ap0[k] += dp0;
      ap1[k] += dp1;
      unsafeAtomicAdd(god + c, do0);
      if (has1) unsafeAtomicAdd(god + hh + c, do1);
    }
  }
  flush(ge, ge_ld, run_s, as0, as1, hh, lim1);
  flush(gr, gr_ld, run_p, ap0, ap1, rl0, rl1);
}

// Backward of kge_score_neg (BatchNegativeSample.score, sampler.py:263-306, followed by autograd's
// scatter-add of the gathered rows): gradients of sum_{i,k} gout[i,k] * score(triple i with `slot`
// replaced by neg[i,k]) accumulated into the dense table gradients.  One wave per (positive, chunk
// of NGA_CH negatives): the two FIXED rows of the positive (relation + the uncorrupted entity) are
// loaded once and their gradients summed in registers over the chunk (one atomic per element and
// chunk instead of one per negative); only the corrupted rows stream, and only their gradients go
// out per negative.  No [n*K, 3] index tensor, no [n*K, d] row-gradient tensors.  d <= 1024.
constexpr int NGA_CH = 64;

template <int SCORER, int NORM, int SLOT>
__global__ __launch_bounds__(256) void bwd_neg_accum_kernel(
    Operand S, Operand R, Operand O, int d, int dr, long long n, const void* __restrict__ neg,
    int neg_itype, long long neg_ld, long long K, int chunks_per_row, int ch, float lp,
    const float* __restrict__ gout, long long ldg, const float* __restrict__ scores, long long lds,
    float* __restrict__ ge, long long ge_ld, float* __restrict__ gr, long long gr_ld) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long row = w / chunks_per_row;
  if (row >= n) return;
  const long long k0 = (w % chunks_per_row) * ch;
  const long long k1 = k0 + ch < K ? k0 + ch : K;
  const int lane = threadIdx.x & 63;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  const long long fi = SLOT == 0 ? index_at(O.idx, row) : index_at(S.idx, row);
  const long long pi = index_at(R.idx, row);
  const float* frow = (const float*)S.base + fi * S.ld;  // S.base == O.base: the entity table
  const float* rrow = (const float*)R.base + pi * R.ld;
  float f0[SPA_NC], f1[SPA_NC], r0[SPA_NC], r1[SPA_NC];
  float af0[SPA_NC], af1[SPA_NC], ap0[SPA_NC], ap1[SPA_NC];
#pragma unroll
  for (int k = 0; k < SPA_NC; ++k) {
    const int c = lane + 64 * k;
    f0[k] = c < hh ? frow[c] : 0.f;
    f1[k] = c < lim1 ? frow[hh + c] : 0.f;
    r0[k] = c < rl0 ? rrow[c] : 0.f;
    r1[k] = c < rl1 ? rrow[hh + c] : 0.f;
    af0[k] = af1[k] = ap0[k] = ap1[k] = 0.f;
  }
  for (long long kk = k0; kk < k1; ++kk) {
    const long long vi = neg_itype ? ((const long long*)neg)[row * neg_ld + kk]
                                   : (long long)((const int*)neg)[row * neg_ld + kk];
    const float* vrow = (const float*)S.base + vi * S.ld;
    float* gv = ge + vi * ge_ld;
    const float g = gout[row * ldg + kk];
    const float dist =
        (SCORER == KGE_TRANSE || SCORER == KGE_ROTATE) && NORM != NORM_L1 ? -scores[row * lds + kk] : 0.f;
#pragma unroll
    for (int k = 0; k < SPA_NC; ++k) {
      const int c = lane + 64 * k;
      if (c >= hh) break;
      const bool has1 = c < lim1;
      const float v0 = vrow[c], v1 = has1 ? vrow[hh + c] : 0.f;
      float ds0, ds1, dp0, dp1, do0, do1;
      if (SLOT == 0)
        spo_pair_grads<SCORER, NORM>(v0, v1, r0[k], r1[k], f0[k], f1[k], has1, g, dist, lp, ds0, ds1, dp0, dp1,
                                     do0, do1);
      else
        spo_pair_grads<SCORER, NORM>(f0[k], f1[k], r0[k], r1[k], v0, v1, has1, g, dist, lp, ds0, ds1, dp0, dp1,
                                     do0, do1);
      const float dv0 = SLOT == 0 ? ds0 : do0, dv1 = SLOT == 0 ? ds1 : do1;
      af0[k] += SLOT == 0 ? do0 : ds0;
      af1[k] += SLOT == 0 ? do1 : ds1;
      ap0[k] += dp0;
      ap1[k] += dp1;
      unsafeAtomicAdd(gv + c, dv0);
      if (has1) unsafeAtomicAdd(gv + hh + c, dv1);
    }
  }
  float* gf = ge + fi * ge_ld;
  float* gp = gr + pi * gr_ld;
#pragma unroll
  for (int k = 0; k < SPA_NC; ++k) {
    const int c = lane + 64 * k;
    if (c < hh) unsafeAtomicAdd(gf + c, af0[k]);
    if (c < lim1) unsafeAtomicAdd(gf + hh + c, af1[k]);
    if (c < rl0) unsafeAtomicAdd(gp + c, ap0[k]);
    if (c < rl1) unsafeAtomicAdd(gp + hh + c, ap1[k]);
  }
}

int run_neg_bwd_accum(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                      int dr, long long n, int slot, const void* neg, int neg_itype, long long neg_ld,
                      long long K, const float* gout, long long ldg, const float* scores, long long lds,
                      float* ge, long long ge_ld, float* gr, long long gr_ld, hipStream_t st) {
  if (n == 0 || K == 0) return KGE_OK;
  if ((d + 1) / 2 > 64 * SPA_NC) return KGE_ERR_UNSUPPORTED;
  const int norm = norm_mode(lp);
  const bool dot = scorer == KGE_COMPLEX || scorer == KGE_DISTMULT;
  if (!dot && norm != NORM_L1 && !scores) return KGE_ERR_INVALID_ARG;
  // negatives per wave: NGA_CH for large batches; fewer when n * K is small -- a wave walks its negatives one
  // after the other (index -> row -> atomics), and 1,024 waves of 64 left the chip latency-bound (90 us for
  // 512 x 100 negatives); ~8 k waves hide it
  long long chl = n * K / 8192;
  if (chl < 4) chl = 4;
  if (chl > NGA_CH) chl = NGA_CH;
  const int ch = (int)chl;
  const long long cpr = (K + ch - 1) / ch;
  const long long waves = n * cpr;
  if (cpr > (1LL << 30) || (waves + 3) / 4 > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define KGE_NA2(SC, NM, SL)                                                                         \
  hipLaunchKernelGGL((bwd_neg_accum_kernel<SC, NM, SL>), grid, dim3(256), 0, st, S, R, O, d, dr, n,  \
                     neg, neg_itype, neg_ld, K, (int)cpr, ch, lp, gout, ldg, scores, lds, ge, ge_ld, gr, \
                     gr_ld)
#define KGE_NA(SC, NM)                                                \
  {                                                                   \
    if (slot == 0) KGE_NA2(SC, NM, 0);                                \
    else KGE_NA2(SC, NM, 2);                                          \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; \
  }
  switch (scorer) {
    case KGE_COMPLEX: KGE_NA(KGE_COMPLEX, NORM_L1);
    case KGE_DISTMULT: KGE_NA(KGE_DISTMULT, NORM_L1);
    case KGE_TRANSE:
      if (norm == NORM_L1) KGE_NA(KGE_TRANSE, NORM_L1);
      if (norm == NORM_L2) KGE_NA(KGE_TRANSE, NORM_L2);
      KGE_NA(KGE_TRANSE, NORM_LP);
    case KGE_ROTATE:
      if (norm == NORM_L1) KGE_NA(KGE_ROTATE, NORM_L1);
      if (norm == NORM_L2) KGE_NA(KGE_ROTATE, NORM_L2);
      KGE_NA(KGE_ROTATE, NORM_LP);
  }
#undef KGE_NA
#undef KGE_NA2
  return KGE_ERR_INVALID_ARG;
}

// ---- kge_score_neg_bwd_accum_sorted: the backward of the negatives without one atomic per occurrence -------------
// bwd_neg_accum_kernel adds every (positive i, negative k)'s gradient row to grad_ent[neg[i, k]] with float atomics:
// n K d of them -- 262 M per slot at the WN18RR shape with 512 x 1000 negatives, 0.86-0.92 ms of a 2.4 ms training
// step (profiles/r5_ns_step_kernels.txt), the same for TransE and RotatE: not the arithmetic.  Two things hold it:
// the atomics, and a wave that walks its negatives ONE after the other (index -> row -> use: a dependent round trip
// per negative; without any atomic the same loop still takes 0.40 ms where the forward's gather of the same rows takes
// 0.15).  Here:
//   * `order` = the occurrences sorted by the entity they corrupt (kge_neg_order: a counting sort -- the caller's
//     histogram + prefix sum give every entity its range, one atomic cursor bump per occurrence places it);
//   * bwd_neg_sorted_kernel: a wave walks NGS_CH consecutive occurrences of that order, keeps the running sum of the
//     current entity's gradient row in registers and flushes it (one atomic per element) only where the entity changes
//     or the chunk ends: ~(E + n K) / NGS_CH row flushes instead of n K.  The fixed rows of an occurrence (relation +
//     uncorrupted entity of positive i: n distinct rows each) come from the L2, the rows of NGS_U occurrences requested
//     before the first is used;
//   * bwd_neg_fixed_kernel: the fixed rows' gradients, a positive's negatives summed in registers as before, the
//     rows of NGF_U negatives in flight at a time.
// Sums in another order than the atomics' (which have none): the same values up to float rounding.
constexpr int NGS_CH = 32, NGS_U = 4, NGF_U = 4;

template <int NC>
__device__ __forceinline__ void ng_load_row(const float* __restrict__ row, int lane, int hh, int lim1, float (&x0)[NC],
                                            float (&x1)[NC]) {
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = lane + 64 * k;
    x0[k] = c < hh ? row[c] : 0.f;
    x1[k] = c < lim1 ? row[hh + c] : 0.f;
  }
}

template <int SCORER, int NORM, int SLOT, int NC>
__global__ __launch_bounds__(256) void bwd_neg_fixed_kernel(
    Operand S, Operand R, Operand O, int d, int dr, long long n, const void* __restrict__ neg, int neg_itype,
    long long neg_ld, long long K, int chunks_per_row, int ch, float lp, const float* __restrict__ gout, long long ldg,
    const float* __restrict__ scores, long long lds, float* __restrict__ ge, long long ge_ld, float* __restrict__ gr,
    long long gr_ld) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long row = w / chunks_per_row;
  if (row >= n) return;
  const long long k0 = (w % chunks_per_row) * ch;
  const int cnt = (int)(k0 + ch < K ? ch : K - k0);  // <= 64
  const int lane = threadIdx.x & 63;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  const long long fi = SLOT == 0 ? index_at(O.idx, row) : index_at(S.idx, row);
  const long long pi = index_at(R.idx, row);
  const float* frow = (const float*)S.base + fi * S.ld;  // S.base == O.base: the entity table
  const float* rrow = (const float*)R.base + pi * R.ld;
  // the chunk's negatives, one per lane
  long long m_vi = 0;
  float m_g = 0.f, m_dist = 0.f;
  if (lane < cnt) {
    m_vi = neg_itype ? ((const long long*)neg)[row * neg_ld + k0 + lane] : (long long)((const int*)neg)[row * neg_ld + k0 + lane];
    m_g = gout[row * ldg + k0 + lane];
    m_dist = (SCORER == KGE_TRANSE || SCORER == KGE_ROTATE) && NORM != NORM_L1 ? -scores[row * lds + k0 + lane] : 0.f;
  }
  float f0[NC], f1[NC], r0[NC], r1[NC], af0[NC], af1[NC], ap0[NC], ap1[NC];
  ng_load_row<NC>(frow, lane, hh, lim1, f0, f1);
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = lane + 64 * k;
    r0[k] = c < rl0 ? rrow[c] : 0.f;
    r1[k] = c < rl1 ? rrow[hh + c] : 0.f;
    af0[k] = af1[k] = ap0[k] = ap1[k] = 0.f;
  }
  for (int j = 0; j < cnt; j += NGF_U) {
    float v0[NGF_U][NC], v1[NGF_U][NC];
#pragma unroll
    for (int u = 0; u < NGF_U; ++u) {  // (a negative beyond the chunk: lane j + u holds m_vi = 0, row 0 -- loaded, not used)
      const long long vi = __shfl(m_vi, (j + u) & 63, 64);
      ng_load_row<NC>((const float*)S.base + vi * S.ld, lane, hh, lim1, v0[u], v1[u]);
    }
#pragma unroll
    for (int u = 0; u < NGF_U; ++u) {
      if (j + u < cnt) {
        const float g = __shfl(m_g, (j + u) & 63, 64), dist = __shfl(m_dist, (j + u) & 63, 64);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int c = lane + 64 * k;
          if (c >= hh) break;
          const bool has1 = c < lim1;
          float ds0, ds1, dp0, dp1, do0, do1;
          if (SLOT == 0)
            spo_pair_grads<SCORER, NORM>(v0[u][k], v1[u][k], r0[k], r1[k], f0[k], f1[k], has1, g, dist, lp, ds0, ds1, dp0,
                                         dp1, do0, do1);
          else
            spo_pair_grads<SCORER, NORM>(f0[k], f1[k], r0[k], r1[k], v0[u][k], v1[u][k], has1, g, dist, lp, ds0, ds1, dp0,
                                         dp1, do0, do1);
          af0[k] += SLOT == 0 ? do0 : ds0;
          af1[k] += SLOT == 0 ? do1 : ds1;
          ap0[k] += dp0;
          ap1[k] += dp1;
        }
      }
    }
  }
  float* gf = ge + fi * ge_ld;
  float* gp = gr + pi * gr_ld;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = lane + 64 * k;
    if (c < hh) unsafeAtomicAdd(gf + c, af0[k]);
    if (c < lim1) unsafeAtomicAdd(gf + hh + c, af1[k]);
    if (c < rl0) unsafeAtomicAdd(gp + c, ap0[k]);
    if (c < rl1) unsafeAtomicAdd(gp + hh + c, ap1[k]);
  }
}

// cos / sin of every relation phase, once per call: table[r][c] = cos, table[r][dr + c] = sin (sincos_canon: the values
// the scoring and gradient kernels compute themselves) -- in bwd_neg_sorted_kernel the relation changes with every
// occurrence and RotatE's four sincos per lane and occurrence were half its time
__global__ __launch_bounds__(256) void rot_table_kernel(const float* __restrict__ rel, long long rel_ld, long long num_rel,
                                                        int dr, float* __restrict__ table) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= num_rel * dr) return;
  const long long r = i / dr;
  const int c = (int)(i - r * dr);
  float sn, cs;
  sincos_canon(rel[r * rel_ld + c], sn, cs);
  table[r * 2 * dr + c] = cs;
  table[r * 2 * dr + dr + c] = sn;
}

template <int SCORER, int NORM, int SLOT, int NC, bool ROTPRE>
__global__ __launch_bounds__(256) void bwd_neg_sorted_kernel(
    Operand S, Operand R, Operand O, int d, int dr, const void* __restrict__ neg, int neg_itype, long long neg_ld,
    long long K, const long long* __restrict__ order, long long total, float lp, const float* __restrict__ gout,
    long long ldg, const float* __restrict__ scores, long long lds, float* __restrict__ ge, long long ge_ld,
    const float* __restrict__ rot) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long j0 = w * NGS_CH;
  if (j0 >= total) return;
  const int cnt = (int)(total - j0 < NGS_CH ? total - j0 : NGS_CH);
  const int lane = threadIdx.x & 63;
  const int hh = (d + 1) / 2, lim1 = d - hh;
  const int rl0 = (SCORER == KGE_ROTATE) ? dr : hh;
  const int rl1 = (SCORER == KGE_ROTATE) ? 0 : lim1;
  // the chunk's metadata, one occurrence per lane (one round of dependent loads for the whole chunk)
  long long m_vi = -1, m_fi = 0, m_pi = 0;
  float m_g = 0.f, m_dist = 0.f;
  if (lane < cnt) {
    const long long pos = order[j0 + lane];
    const long long row = pos / K, kk = pos - row * K;
    m_vi = neg_itype ? ((const long long*)neg)[row * neg_ld + kk] : (long long)((const int*)neg)[row * neg_ld + kk];
    m_fi = SLOT == 0 ? index_at(O.idx, row) : index_at(S.idx, row);
    m_pi = index_at(R.idx, row);
    m_g = gout[row * ldg + kk];
    m_dist = (SCORER == KGE_TRANSE || SCORER == KGE_ROTATE) && NORM != NORM_L1 ? -scores[row * lds + kk] : 0.f;
  }
  float a0[NC], a1[NC], v0[NC], v1[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) a0[k] = a1[k] = v0[k] = v1[k] = 0.f;
  long long cur = -1;
  auto flush = [&]() {
    if (cur < 0) return;
    float* gv = ge + cur * ge_ld;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int c = lane + 64 * k;
      if (c < hh) unsafeAtomicAdd(gv + c, a0[k]);
      if (c < lim1) unsafeAtomicAdd(gv + hh + c, a1[k]);
      a0[k] = a1[k] = 0.f;
    }
  };
  for (int j = 0; j < cnt; j += NGS_U) {
    float f0[NGS_U][NC], f1[NGS_U][NC], r0[NGS_U][NC], r1[NGS_U][NC];
#pragma unroll
    for (int u = 0; u < NGS_U; ++u) {  // (an occurrence beyond the chunk: rows 0 -- loaded, not used)
      const long long fi = __shfl(m_fi, (j + u) & 63, 64), pi = __shfl(m_pi, (j + u) & 63, 64);
      ng_load_row<NC>((const float*)S.base + fi * S.ld, lane, hh, lim1, f0[u], f1[u]);
      if constexpr (ROTPRE) {  // (cos | sin) of relation pi
        const float* trow = rot + pi * 2 * dr;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int c = lane + 64 * k;
          r0[u][k] = c < dr ? trow[c] : 1.f;
          r1[u][k] = c < dr ? trow[dr + c] : 0.f;
        }
      } else {
        const float* rrow = (const float*)R.base + pi * R.ld;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int c = lane + 64 * k;
          r0[u][k] = c < rl0 ? rrow[c] : 0.f;
          r1[u][k] = c < rl1 ? rrow[hh + c] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NGS_U; ++u) {
      if (j + u < cnt) {
        const long long vi = __shfl(m_vi, (j + u) & 63, 64);
        const float g = __shfl(m_g, (j + u) & 63, 64), dist = __shfl(m_dist, (j + u) & 63, 64);
        if (vi != cur) {  // (wave-uniform: every lane holds the same vi)
          flush();
          cur = vi;
          ng_load_row<NC>((const float*)S.base + vi * S.ld, lane, hh, lim1, v0, v1);  // S.base == O.base
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int c = lane + 64 * k;
          if (c >= hh) break;
          const bool has1 = c < lim1;
          float ds0, ds1, dp0, dp1, do0, do1;
          if (SLOT == 0)
            spo_pair_grads<SCORER, NORM, ROTPRE>(v0[k], v1[k], r0[u][k], r1[u][k], f0[u][k], f1[u][k], has1, g, dist, lp, ds0,
                                                 ds1, dp0, dp1, do0, do1);
          else
            spo_pair_grads<SCORER, NORM, ROTPRE>(f0[u][k], f1[u][k], r0[u][k], r1[u][k], v0[k], v1[k], has1, g, dist, lp, ds0,
                                                 ds1, dp0, dp1, do0, do1);
          a0[k] += SLOT == 0 ? ds0 : do0;
          a1[k] += SLOT == 0 ? ds1 : do1;
        }
      }
    }
  }
  flush();
}

// order[cursor[id]++] = position, for every position i K + k of the [n, K] sample matrix: the scatter step of a counting
// sort by entity id (`cursor` = exclusive prefix sums of the ids' histogram on entry; clobbered).
__global__ __launch_bounds__(256) void neg_order_kernel(const void* __restrict__ neg, int neg_itype, long long neg_ld,
                                                        long long K, long long total, long long num_ent,
                                                        unsigned long long* __restrict__ cursor,
                                                        long long* __restrict__ order) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= total) return;
  const long long row = j / K, kk = j - row * K;
  long long id = neg_itype ? ((const long long*)neg)[row * neg_ld + kk] : (long long)((const int*)neg)[row * neg_ld + kk];
  if (id < 0) id = 0;
  if (id >= num_ent) id = num_ent - 1;  // (ids are the caller's contract; never write outside `order`)
  const unsigned long long pos = atomicAdd(cursor + id, 1ULL);
  if (pos < (unsigned long long)total) order[pos] = j;
}

// counts[id] += 1 for every sample (the histogram of the counting sort; `counts` zeroed by the caller)
__global__ __launch_bounds__(256) void neg_histogram_kernel(const void* __restrict__ neg, int neg_itype, long long neg_ld,
                                                            long long K, long long total, long long num_ent,
                                                            unsigned long long* __restrict__ counts) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= total) return;
  const long long row = j / K, kk = j - row * K;
  long long id = neg_itype ? ((const long long*)neg)[row * neg_ld + kk] : (long long)((const int*)neg)[row * neg_ld + kk];
  if (id < 0) id = 0;
  if (id >= num_ent) id = num_ent - 1;
  atomicAdd(counts + id, 1ULL);
}

int run_neg_order(const void* neg, int neg_itype, long long neg_ld, long long n, long long K, long long num_ent,
                  long long* cursor, long long* order, hipStream_t st) {
  const long long total = n * K;
  if (total == 0) return KGE_OK;
  if ((total + 255) / 256 > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  if (order == nullptr) {  // first step: the histogram
    hipLaunchKernelGGL(neg_histogram_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, neg, neg_itype, neg_ld,
                       K, total, num_ent, (unsigned long long*)cursor);
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(neg_order_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, neg, neg_itype, neg_ld, K,
                     total, num_ent, (unsigned long long*)cursor, order);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// kge_score_neg_bwd_accum with the occurrences sorted by corrupted entity.  `order`: int64 [n K] (device).
int run_neg_bwd_accum_sorted(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d, int dr,
                             long long n, int slot, const void* neg, int neg_itype, long long neg_ld, long long K,
                             const long long* order, const float* gout, long long ldg, const float* scores,
                             long long lds, float* ge, long long ge_ld, float* gr, long long gr_ld, float* rot,
                             long long num_rel, hipStream_t st) {
  if (n == 0 || K == 0) return KGE_OK;
  if ((d + 1) / 2 > 64 * SPA_NC) return KGE_ERR_UNSUPPORTED;
  const bool rotpre = scorer == KGE_ROTATE && rot != nullptr;
  if (rotpre) {
    const long long cells = num_rel * dr;
    hipLaunchKernelGGL(rot_table_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const float*)R.base, R.ld,
                       num_rel, dr, rot);
  }
  const int norm = norm_mode(lp);
  const bool dot = scorer == KGE_COMPLEX || scorer == KGE_DISTMULT;
  if (!dot && norm != NORM_L1 && !scores) return KGE_ERR_INVALID_ARG;
  long long chl = n * K / 8192;
  if (chl < 4) chl = 4;
  if (chl > NGA_CH) chl = NGA_CH;
  const int ch = (int)chl;
  const long long cpr = (K + ch - 1) / ch;
  const long long waves = n * cpr;
  const long long total = n * K, swaves = (total + NGS_CH - 1) / NGS_CH;
  if (cpr > (1LL << 30) || (waves + 3) / 4 > 0x7fffffffLL || (swaves + 3) / 4 > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((waves + 3) / 4)), sgrid((unsigned)((swaves + 3) / 4));
  const bool small = (d + 1) / 2 <= 64 * 4;  // coordinate pairs per lane: 4 (d <= 512) or 8
#define KGE_NS3(SC, NM, SL, NCV)                                                                                     \
  hipLaunchKernelGGL((bwd_neg_fixed_kernel<SC, NM, SL, NCV>), grid, dim3(256), 0, st, S, R, O, d, dr, n, neg,        \
                     neg_itype, neg_ld, K, (int)cpr, ch, lp, gout, ldg, scores, lds, ge, ge_ld, gr, gr_ld);          \
  if (rotpre && SC == KGE_ROTATE)                                                                                    \
    hipLaunchKernelGGL((bwd_neg_sorted_kernel<SC, NM, SL, NCV, SC == KGE_ROTATE>), sgrid, dim3(256), 0, st, S, R, O, d, dr, \
                       neg, neg_itype, neg_ld, K, order, total, lp, gout, ldg, scores, lds, ge, ge_ld, rot);         \
  else                                                                                                               \
    hipLaunchKernelGGL((bwd_neg_sorted_kernel<SC, NM, SL, NCV, false>), sgrid, dim3(256), 0, st, S, R, O, d, dr, neg, \
                       neg_itype, neg_ld, K, order, total, lp, gout, ldg, scores, lds, ge, ge_ld, rot)
#define KGE_NS2(SC, NM, SL)          \
  if (small) { KGE_NS3(SC, NM, SL, 4); } \
  else { KGE_NS3(SC, NM, SL, 8); }
#define KGE_NS(SC, NM)                                                \
  {                                                                   \
    if (slot == 0) { KGE_NS2(SC, NM, 0) }                             \
    else { KGE_NS2(SC, NM, 2) }                                       \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; \
  }
  switch (scorer) {
    case KGE_COMPLEX: KGE_NS(KGE_COMPLEX, NORM_L1);
    case KGE_DISTMULT: KGE_NS(KGE_DISTMULT, NORM_L1);
    case KGE_TRANSE:
      if (norm == NORM_L1) KGE_NS(KGE_TRANSE, NORM_L1);
      if (norm == NORM_L2) KGE_NS(KGE_TRANSE, NORM_L2);
      KGE_NS(KGE_TRANSE, NORM_LP);
    case KGE_ROTATE:
      if (norm == NORM_L1) KGE_NS(KGE_ROTATE, NORM_L1);
      if (norm == NORM_L2) KGE_NS(KGE_ROTATE, NORM_L2);
      KGE_NS(KGE_ROTATE, NORM_LP);
  }
#undef KGE_NS
#undef KGE_NS2
#undef KGE_NS3
  return KGE_ERR_INVALID_ARG;
}

template <int SCORER, int NORM>
static int launch_bwd_pairs(int dir, const Operand& A, const Operand& R, const Operand& TG, int d,
                            int dr, long long n, long long m, float lp, const float* gout,
                            long long ldg, const float* scores, long long lds, float* g_a,
                            float* g_p, float* g_tgt, hipStream_t st) {
  const int hh = (d + 1) / 2;
  const unsigned gc = (unsigned)((hh + BW_TC - 1) / BW_TC);
  const unsigned gr = (unsigned)((n + BW_TR - 1) / BW_TR);
  // query-side pass: split the reduction over the targets until ~1024 workgroups exist
  long long ys = 1024 / ((long long)gc * gr);
  if (ys > 32) ys = 32;
  long long ychunk = m;
  if (ys > 1) {
    ychunk = ((m + ys - 1) / ys + BW_KY - 1) / BW_KY * BW_KY;
    ys = (m + ychunk - 1) / ychunk;
  }
  if (ys > 1) {
    if (!fill_words_async(g_a, 0, (size_t)n * d * sizeof(float), st) ||
        !fill_words_async(g_p, 0, (size_t)n * dr * sizeof(float), st))
      return KGE_ERR_LAUNCH;
  } else {
    ys = 1;
    ychunk = m;
  }
  hipLaunchKernelGGL((bwd_pairs_kernel<SCORER, NORM, 0>), dim3(gc, gr, (unsigned)ys), dim3(256), 0, st, A, R, TG,
                     dir, d, dr, n, m, lp, gout, ldg, scores, lds, g_a, g_p, g_tgt, ychunk);
  hipLaunchKernelGGL((bwd_pairs_kernel<SCORER, NORM, 1>), dim3(gc, (unsigned)((m + BW_TR - 1) / BW_TR)),
                     dim3(256), 0, st, A, R, TG, dir, d, dr, n, m, lp, gout, ldg, scores, lds, g_a,
                     g_p, g_tgt, (long long)n);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_pairs_bwd_gemm(int scorer, int dir, const Operand& A, const Operand& R, const Operand& TG,
                       int d, int dr, long long n, long long m, const float* gout, long long ldg,
                       float* g_a, float* g_p, float* g_tgt, hipStream_t st);

int run_pairs_bwd(int scorer, float lp, int dir, const Operand& A, const Operand& R,
                  const Operand& TG, int d, int dr, long long n, long long m, const float* gout,
                  long long ldg, const float* scores, long long lds, float* g_a, float* g_p,
                  float* g_tgt, hipStream_t st, bool self_contained) {
  if (n == 0 || m == 0) return KGE_OK;
  if (!self_contained) {  // ComplEx / DistMult: two library GEMMs + elementwise kernels (bwd_gemm.hip)
    const int rc = run_pairs_bwd_gemm(scorer, dir, A, R, TG, d, dr, n, m, gout, ldg, g_a, g_p, g_tgt, st);
    if (rc != KGE_ERR_UNSUPPORTED) return rc;
  }
  const int norm = norm_mode(lp);
  const bool dot = scorer == KGE_COMPLEX || scorer == KGE_DISTMULT;
  if (!dot && norm != NORM_L1 && !scores) return KGE_ERR_INVALID_ARG;
#define KGE_B(SC, NM) \
  return launch_bwd_pairs<SC, NM>(dir, A, R, TG, d, dr, n, m, lp, gout, ldg, scores, lds, g_a, g_p, g_tgt, st)
  switch (scorer) {
    case KGE_COMPLEX: KGE_B(KGE_COMPLEX, NORM_L1);
    case KGE_DISTMULT: KGE_B(KGE_DISTMULT, NORM_L1);
    case KGE_TRANSE:
      if (norm == NORM_L1) KGE_B(KGE_TRANSE, NORM_L1);
      if (norm == NORM_L2) KGE_B(KGE_TRANSE, NORM_L2);
      KGE_B(KGE_TRANSE, NORM_LP);
    case KGE_ROTATE:
      if (norm == NORM_L1) KGE_B(KGE_ROTATE, NORM_L1);
      if (norm == NORM_L2) KGE_B(KGE_ROTATE, NORM_L2);
      KGE_B(KGE_ROTATE, NORM_LP);
  }
#undef KGE_B
  return KGE_ERR_INVALID_ARG;
}

int run_spo_bwd(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                int dr, long long n, const float* gout, const float* scores, float* g_s, float* g_p,
                float* g_o, hipStream_t st) {
  if (n == 0) return KGE_OK;
  const int norm = norm_mode(lp);
  const bool dot = scorer == KGE_COMPLEX || scorer == KGE_DISTMULT;
  if (!dot && norm != NORM_L1 && !scores) return KGE_ERR_INVALID_ARG;
  const dim3 grid((unsigned)((n + 3) / 4));
#define KGE_S(SC, NM)                                                                            \
  {                                                                                              \
    hipLaunchKernelGGL((bwd_spo_kernel<SC, NM>), grid, dim3(256), 0, st, S, R, O, d, dr, n, lp,   \
                       gout, scores, g_s, g_p, g_o);                                             \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                            \
  }
  switch (scorer) {
    case KGE_COMPLEX: KGE_S(KGE_COMPLEX, NORM_L1);
    case KGE_DISTMULT: KGE_S(KGE_DISTMULT, NORM_L1);
    case KGE_TRANSE:
      if (norm == NORM_L1) KGE_S(KGE_TRANSE, NORM_L1);
      if (norm == NORM_L2) KGE_S(KGE_TRANSE, NORM_L2);
      KGE_S(KGE_TRANSE, NORM_LP);
    case KGE_ROTATE:
      if (norm == NORM_L1) KGE_S(KGE_ROTATE, NORM_L1);
      if (norm == NORM_L2) KGE_S(KGE_ROTATE, NORM_L2);
      KGE_S(KGE_ROTATE, NORM_LP);
  }
#undef KGE_S
  return KGE_ERR_INVALID_ARG;
}

int run_spo_bwd_accum(int scorer, float lp, const Operand& S, const Operand& R, const Operand& O, int d,
                      int dr, long long n, const float* gout, const float* scores, float* ge, long long ge_ld,
                      float* gr, long long gr_ld, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if ((d + 1) / 2 > 64 * SPA_NC) return KGE_ERR_UNSUPPORTED;
  const int norm = norm_mode(lp);
  const bool dot = scorer == KGE_COMPLEX || scorer == KGE_DISTMULT;
  if (!dot && norm != NORM_L1 && !scores) return KGE_ERR_INVALID_ARG;
  long long chl = n / 4096;  // ~4 k waves before the chunks grow
  if (chl < 1) chl = 1;
  if (chl > SPA_CH) chl = SPA_CH;
  const int ch = (int)chl;
  const long long chunks = (n + ch - 1) / ch;
  const dim3 grid((unsigned)((chunks + 3) / 4));
#define KGE_SA(SC, NM)                                                                                \
  {                                                                                                   \
    hipLaunchKernelGGL((bwd_spo_accum_kernel<SC, NM>), grid, dim3(256), 0, st, S, R, O, d, dr, n, lp,  \
                       gout, scores, ge, ge_ld, gr, gr_ld, ch);                                       \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                 \
  }
  switch (scorer) {
    case KGE_COMPLEX: KGE_SA(KGE_COMPLEX, NORM_L1);
    case KGE_DISTMULT: KGE_SA(KGE_DISTMULT, NORM_L1);
    case KGE_TRANSE:
      if (norm == NORM_L1) KGE_SA(KGE_TRANSE, NORM_L1);
      if (norm == NORM_L2) KGE_SA(KGE_TRANSE, NORM_L2);
      KGE_SA(KGE_TRANSE, NORM_LP);
    case KGE_ROTATE:
      if (norm == NORM_L1) KGE_SA(KGE_ROTATE, NORM_L1);
      if (norm == NORM_L2) KGE_SA(KGE_ROTATE, NORM_L2);
      KGE_SA(KGE_ROTATE, NORM_LP);
  }
#undef KGE_SA
  return KGE_ERR_INVALID_ARG;
}

}  // namespace kge
