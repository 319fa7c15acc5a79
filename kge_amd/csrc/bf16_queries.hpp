// bf16_queries.hpp -- prepared query fragments of the bf16 matrix-core pair kernels (score_pairs_bf16_v4.hip,
// score_pairs_bf16_v6.hip): the description of a batch whose queries are built outside the launch that scores
// them (NextQ), the builder itself, and the compile-time loop both kernels unroll their pipelines with.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace kge {

template <int I, int N, class F>
__device__ __forceinline__ void v4_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    v4_static_for<I + 1, N>(f);
  }
}

// ---- prepared queries (kge_build_queries / kge_score_queries, include/kge_amd.h) -------------------------------
// The query vectors q_i = s_i (x) r_i of a batch in MFMA-fragment order, built OUTSIDE the scoring launch that
// consumes them: by query_build_kernel, or by the spare workgroups of the PREVIOUS batch's scoring launch
// (NextQ).  The scoring kernel then starts with the fragment loads and the tile DMA -- the five dependent round
// trips of the in-launch cooperative build (index -> rows -> write-through ack -> flag -> fragments, ~12 k cycles
// during which nothing is scored, profiles/r12_phase_timestamps.txt) are off its critical path.
//
// SPLIT (KGE_FLAG_SPLIT_QUERY): q is carried as q_hi + q_lo, q_hi = bf16(q), q_lo = bf16(q - q_hi), as two
// VIRTUAL query rows; a row group is 64 real rows = 128 virtual rows (32-row blocks 0, 1: q_hi of real rows
// 0-31 / 32-63, blocks 2, 3: q_lo), the consumer waves are unchanged, the store waves add the two partial
// scores.  Products of bf16 values are exact in f32, so score = fl(sum q_hi t) + fl(sum q_lo t) differs from
// f32 arithmetic on the same bf16 tables only by f32 summation order and the 2^-17 relative residue of
// q - q_hi - q_lo (exactly 0 for DistMult, whose q has 16 significant bits) -- SURVEY.md 8(c) gate 4.
struct NextQ {
  Operand A, A2, R;  // entity rows of the first side, of the second side (two-sided), relation rows
  int dir;           // combine of the first side (KGE_SP_ / KGE_PO_); a second side is always KGE_PO_
  long long n;       // rows per side
  int rgn, rgn1;     // row groups in all / of the first side
  u32x4* qf;         // destination; nullptr: nothing to build
  // who builds: mode 1 -- the launch's idle workgroups: column-group slots beyond ncg, which sit on compute units of
  // their own from the first cycle (nblocks of them, numbered rg * slots-per-row-group + slot); mode 2 (a geometry
  // without idle slots, e.g. 16 row groups x 16 column groups) -- the consumer waves of EVERY scoring workgroup,
  // behind their last tile, while the store waves drain (nblocks = scoring workgroups, 256 threads each)
  int mode, nblocks;
  // a GROUP of equally shaped batches (kge_score_queries_multi / kge_build_queries_multi): batch l = rows
  // [l n, (l + 1) n) of the index vectors, its fragments `qstride` 16-byte words behind batch l - 1's.  0 = 1 batch.
  int nbatch;
  long long qstride;
};

template <int SCORER>
__device__ __forceinline__ void v4_q_f32(int dir, unsigned int a0, unsigned int a1, unsigned int r0, unsigned int r1,
                                         f32x2q& Q0, f32x2q& Q1) {
  const f32x2q A0 = {__uint_as_float(a0 << 16), __uint_as_float(a0 & 0xffff0000u)};
  const f32x2q A1 = {__uint_as_float(a1 << 16), __uint_as_float(a1 & 0xffff0000u)};
  const f32x2q R0 = {__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u)};
  const f32x2q R1 = {__uint_as_float(r1 << 16), __uint_as_float(r1 & 0xffff0000u)};
  if (SCORER == KGE_DISTMULT) {
    Q0 = A0 * R0;
    Q1 = A1 * R1;
  } else if (dir == KGE_SP_) {  // (bf16 x bf16 products are exact in f32: the fma IS the two-rounding form)
    Q0 = __builtin_elementwise_fma(A0, R0, -(A1 * R1));
    Q1 = __builtin_elementwise_fma(A1, R0, A0 * R1);
  } else {
    Q0 = __builtin_elementwise_fma(R0, A0, R1 * A1);
    Q1 = __builtin_elementwise_fma(R0, A1, -(R1 * A0));
  }
}

// The row-major bf16 query matrix Q16 [rows, 2 HH] of the gradient products (bwd_gemm.hip: dT = G16^T Q16) written by
// the same pass that builds the fragments -- the same rounded values in another order (rows of the second side behind
// the n rows of the first) -- and a float buffer cleared on the way (the relation-gradient accumulator of
// kge_ce_sp_po_bwd_accum): what bwdg_build_q16_kernel did in a launch of its own (round 6: one launch less per step).
struct Q16Out {
  unsigned short* q16;   // NULL: none
  float* zero;           // NULL: none
  long long zero_cnt;
};

// items [item0, item0 + stride, ...) of the batch: one item = 8 coordinates of both halves of one query row
template <int SCORER, int HH, int SPLIT, bool WITH_Q16 = false>
__device__ __forceinline__ void v4_build_queries(const NextQ& nx, long long item0, long long stride,
                                                 const Q16Out* qo = nullptr) {
  if constexpr (WITH_Q16) {
    if (qo->zero != nullptr)
      for (long long k = item0; k < qo->zero_cnt; k += stride) qo->zero[k] = 0.0f;
  }
  constexpr int NKB = 2 * HH / 16, NKH = HH / 16, CGR = HH / 8;
  constexpr int RGR = SPLIT ? 64 : 128;  // real rows per row group
  const int nb = nx.nbatch > 1 ? nx.nbatch : 1;
  const long long items = (long long)nb * nx.rgn * RGR * CGR;
  for (long long it = item0; it < items; it += stride) {
    const int rgg = (int)(it / (RGR * CGR));
    const int lb = rgg / nx.rgn, rg = rgg - lb * nx.rgn;  // batch of the group, row group of the batch
    const int rr = (int)((it / CGR) % RGR);
    const int c8 = (int)(it % CGR);
    const bool second = rg >= nx.rgn1;
    const long long lrow = (long long)(second ? rg - nx.rgn1 : rg) * RGR + rr;
    const long long qrow = (lrow < nx.n ? lrow : nx.n - 1) + (long long)lb * nx.n;  // padded rows repeat row n-1
    const Operand& E = second ? nx.A2 : nx.A;
    const int dir = second ? KGE_PO_ : nx.dir;
    const unsigned short* a = (const unsigned short*)E.base + index_at(E.idx, qrow) * E.ld + c8 * 8;
    const unsigned short* r = (const unsigned short*)nx.R.base + index_at(nx.R.idx, qrow) * nx.R.ld + c8 * 8;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(a), a1 = *reinterpret_cast<const u32x4*>(a + HH);
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(r), r1 = *reinterpret_cast<const u32x4*>(r + HH);
    u32x4 q0, q1, l0, l1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (SPLIT) {
        f32x2q Q0, Q1;
        v4_q_f32<SCORER>(dir, a0[e], a1[e], r0[e], r1[e], Q0, Q1);
        q0[e] = bf16_pack_hw(Q0);
        q1[e] = bf16_pack_hw(Q1);
        const f32x2q H0 = {__uint_as_float(q0[e] << 16), __uint_as_float(q0[e] & 0xffff0000u)};
        const f32x2q H1 = {__uint_as_float(q1[e] << 16), __uint_as_float(q1[e] & 0xffff0000u)};
        l0[e] = bf16_pack_hw(Q0 - H0);  // exact differences (Sterbenz-like: |q - q_hi| <= ulp_bf16(q) / 2)
        l1[e] = bf16_pack_hw(Q1 - H1);
      } else {
        unsigned int x0, x1;
        bf16_qpair_fast<SCORER>(dir, a0[e], a1[e], r0[e], r1[e], x0, x1);
        q0[e] = x0;
        q1[e] = x1;
      }
    }
    // fragment-major (as the in-launch build below): K-block kb of 32-row block rb is 64 lanes x 16 B
    const long long row = (long long)rg * 128 + rr;  // virtual row (SPLIT: the q_hi row; q_lo 64 rows behind)
    u32x4* dst = nx.qf + (long long)lb * nx.qstride + ((row >> 5) * NKB) * 64 + (row & 31) + 32 * (c8 & 1);
    dst[(c8 >> 1) * 64] = q0;
    dst[(NKH + (c8 >> 1)) * 64] = q1;
    if constexpr (WITH_Q16) {
      if (qo->q16 != nullptr && lrow < nx.n) {
        unsigned short* qr = qo->q16 + ((second ? nx.n : 0) + lrow) * (2 * HH) + c8 * 8;
        *reinterpret_cast<u32x4*>(qr) = q0;
        *reinterpret_cast<u32x4*>(qr + HH) = q1;
      }
    }
    if constexpr (SPLIT) {
      u32x4* dl = dst + 2 * NKB * 64;  // two 32-row blocks further
      dl[(c8 >> 1) * 64] = l0;
      dl[(NKH + (c8 >> 1)) * 64] = l1;
    }
  }
}

template <int SCORER, int HH, int SPLIT>
__global__ __launch_bounds__(256) void query_build_kernel(NextQ nx) {
  v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
}

// fragments + Q16 + the cleared accumulator in one launch (single batch, plain queries): run_query_build_q16
template <int SCORER, int HH>
__global__ __launch_bounds__(256) void query_build_q16_kernel(NextQ nx, Q16Out qo) {
  v4_build_queries<SCORER, HH, 0, true>(nx, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, &qo);
}

// kge_eval_batch's first launch and its query build in one: blocks [0, build_blocks) build the fragments, the rest
// are eval_begin_kernel's (row block, list) pairs -- the filter lookup is a chain of dependent loads (~10 us of
// latency at 512 rows), the build overlaps it.
template <int SCORER, int HH, int SPLIT>
__global__ __launch_bounds__(256) void eval_begin_build_kernel(NextQ nx, EvalLists L, Index s, Index o, int build_blocks,
                                                               int row_blocks, long long n, long long m, long long rs,
                                                               long long us, long long* __restrict__ tgt) {
  if ((int)blockIdx.x < build_blocks) {
    v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)blockIdx.x * 256 + threadIdx.x, (long long)build_blocks * 256);
    return;
  }
  const int b = (int)blockIdx.x - build_blocks;
  const long long i = (long long)(b % row_blocks) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  eval_begin_row(L, s, o, n, m, rs, us, tgt, b / row_blocks, i, threadIdx.x & 63);
}

// The two launches in front of the counting kernel (kge_score_rank_sp_po) in one: blocks [0, build_blocks) build the
// query fragments, the rest set the filter bits of (row, list) -- independent work, one launch gap less.
template <int SCORER, int HH, int SPLIT>
__global__ __launch_bounds__(256) void query_build_bits_kernel(NextQ nx, RankBitLists B, int build_blocks, int row_blocks,
                                                               long long n, long long col_begin, long long m,
                                                               long long rs, long long us) {
  if ((int)blockIdx.x < build_blocks) {
    v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)blockIdx.x * 256 + threadIdx.x, (long long)build_blocks * 256);
    return;
  }
  const int b = (int)blockIdx.x - build_blocks;
  const int q = b / row_blocks;
  const long long i = (long long)(b % row_blocks) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  rank_bits_row(B, q, i, threadIdx.x & 63, col_begin, m, rs, us, 1);
}

}  // namespace kge
