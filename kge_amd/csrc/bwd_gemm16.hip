// bwd_gemm16.hip -- the two gradient contractions of the bf16 ComplEx / DistMult backward, hand-written
// for gfx950 (the autograd of complex.py:30-39 / distmult.py:15-21 that train_1vsAll.py:70,81
// and train_KvsAll.py:293-294 trigger).  With G16 = d loss / d score ([rows, mp] bf16, row pitch mp),
// Q16 the bf16 query matrix of the forward and T the bf16 entity table:
//
//     dQ [rows, d] = G16 * T          reduction over the m targets   (long K, small output: split-K)
//     dT [m, d]    = G16^T * Q16      reduction over the query rows  (short K, large output)
//
// Both are C[M, N] = sum_k A(m, k) * B(k, n) with B stored [K][N] (n contiguous).  A bf16 MFMA operand wants
// 8 CONSECUTIVE k per lane, so a [K][N] operand has to be transposed on the way: its tile is laid out in LDS
// as [k/4][n/16] blocks of [4 k][16 n] (128 B) and read with ds_read_b64_tr_b16, the CDNA4 transpose read --
// each 16-lane group reads one block and every lane receives the 4 k of its own column.  For dQ the A
// operand (G16, [M][K], k contiguous) is read with plain 16-byte LDS reads (XOR-swizzled rows); for dT the A
// operand is G16 again but as [K][M] (m contiguous): the same transpose read as B.
//
// One kernel, `gemm16_kernel<A_TR>`:
//   * 128 x 256 output tile per workgroup; the eight waves form two groups that split every 64-wide K stage
//     between them (waves 0-3: K slices 0, 1; waves 4-7: slices 2, 3), each wave 64 x 128 = 2 x 4 accumulators
//     of v_mfma_f32_32x32x16_bf16 (12 transpose reads per 8 MFMAs); at the end the groups exchange halves
//     through the dead ring, add, and store 16 bytes per lane from a wave-private row-major image;
//   * HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging), a ring of three 48 KiB stages
//     (A 16 KiB + B 32 KiB).  The DMA writes LDS lane-linearly (16 B per lane), so every layout above is
//     produced by choosing WHICH 16 bytes each lane fetches.  One barrier per stage, placed BEHIND the first
//     slice's 8 queued MFMAs, with the next stage's first fragments requested right after it; the six DMA
//     pieces a wave owes to stage t + 2 are issued one at a time between the following MFMAs (a DMA
//     instruction blocks its wave for ~90 cycles; in a burst all eight waves stalled together);
//   * ragged edges are clamped in the address (never predicated: constant VMEM counts); out-of-range k of the
//     last step are fetched from a zero page instead of A (G16's pad columns [m, mp) are zero by contract);
//   * XCD-aware block ids (block b runs on XCD b mod 8): dQ -- all output tiles of a K split on one XCD
//     (its slice of T and of G16 is fetched into one L2, once); dT -- the two column halves of a row
//     tile on one XCD (they share the G16 columns).
// dQ's split-K partials go to scratch and are summed by bwdg_reduce_kernel (bwd_gemm.hip): deterministic,
// no float atomics.  hipBLASLt stays in bwd_gemm.hip as the fallback for shapes outside this kernel
// (d not a multiple of 256) and as the checker (KGE_BWD_GEMM_LIB=1; tests/test_gpu_bwd_gemm16.py).
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

namespace kge {

namespace {

constexpr int G16_BM = 128, G16_BN = 256, G16_BK = 64;
constexpr int G16_A_BYTES = G16_BM * G16_BK * 2;   // 16 KiB
constexpr int G16_B_BYTES = G16_BK * G16_BN * 2;   // 32 KiB
constexpr int G16_STAGE = G16_A_BYTES + G16_B_BYTES;
constexpr int G16_NST = 3;
constexpr int G16_DMA_PER_WAVE = (G16_STAGE / 1024) / 8;  // 6

typedef __bf16 g16_bf4 __attribute__((ext_vector_type(4)));

// 16 readable zero bytes: the DMA source of every out-of-range k (see `issue`)
__device__ __attribute__((aligned(16))) unsigned int g16_zero_page[4] = {0u, 0u, 0u, 0u};

struct G16Args {
  const unsigned short* A;
  long long lda;
  int a_cols;  // readable columns of a row of A (multiple of 8)
  const unsigned short* B;
  long long ldb;
  float* C;
  long long ldc, c_split;
  int M, N, K;
  int mtiles, ntiles, splits, ksteps;
  unsigned long long* dbg;  // optional per-workgroup phase stamps (8 x u64 each; tools/gemm16_phases.py), else NULL
};

template <int I, int N, class F>
__device__ __forceinline__ void g16_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g16_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ u32x2 g16_tr_read(const unsigned char* p) {
  const g16_bf4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) g16_bf4*)p);
  return __builtin_bit_cast(u32x2, v);
}

template <bool A_TR>
__global__ __launch_bounds__(512, 1) void gemm16_kernel(G16Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G16_NST * G16_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  int dbg_i = 0;
  auto stamp = [&]() {
    if (g.dbg != nullptr && tid == 0) g.dbg[(long long)blockIdx.x * 8 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  stamp();  // 0: start
  int mt, nt, split;
  if (A_TR) {  // the column halves of a row tile next to each other on one XCD
    nt = idx % g.ntiles;
    mt = (idx / g.ntiles) * 8 + xcd;
    split = 0;
    if (mt >= g.mtiles) return;
  } else {  // all output tiles of a K split on one XCD
    const int per = g.mtiles * g.ntiles;
    const int tile = idx % per;
    split = (idx / per) * 8 + xcd;
    if (split >= g.splits) return;
    mt = tile / g.ntiles;
    nt = tile % g.ntiles;
  }
  const int m0 = mt * G16_BM, n0 = nt * G16_BN;
  const int st_lo = (int)((long long)g.ksteps * split / g.splits);
  const int st_hi = (int)((long long)g.ksteps * (split + 1) / g.splits);
  const int nsteps = st_hi - st_lo;
  const int K = g.K;
  const unsigned int lds0 = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  const unsigned short* zeros = (const unsigned short*)g16_zero_page;

  // ---- DMA: this wave's six 1-KiB pieces of a stage.  piece_src(step, u) is the general (clamping) form;
  // the loop keeps one running pointer per piece and advances it by a constant per stage, and only the
  // ragged last step goes through piece_src again.
  auto piece_src = [&](int step, int u) -> const unsigned short* {
    const int k0 = step * G16_BK;
    const int q = wave * G16_DMA_PER_WAVE + u;  // wave-uniform
    if (q < G16_A_BYTES / 1024) {
      if (A_TR) {  // piece q = the 4 k-rows kb4 = q, all 8 blocks of 16 m
        const int c = lane & 7;
        const int krow = k0 + q * 4 + (c >> 1);
        int col = m0 + (lane >> 3) * 16 + (c & 1) * 8;
        col = col <= g.a_cols - 8 ? col : g.a_cols - 8;
        return krow < K ? g.A + (long long)krow * g.lda + col : zeros;
      }
      // piece q = tile rows 8 q .. 8 q + 7, 128 bytes each, 16-byte slots XOR-swizzled
      const int mr = q * 8 + (lane >> 3);
      const int w = (lane & 7) ^ ((mr >> 1) & 7);
      int row = m0 + mr;
      row = row < g.M ? row : g.M - 1;
      const int col = k0 + w * 8;  // a chunk that straddles K ends in the zero columns [K, a_cols) of A
      return col < K ? g.A + (long long)row * g.lda + col : zeros;
    }
    // B piece qb: the 4 k-rows kb4 = qb / 2, blocks 8 (qb % 2) .. + 7 of 16 n
    const int qb = q - G16_A_BYTES / 1024;
    const int c = lane & 7;
    int krow = k0 + (qb >> 1) * 4 + (c >> 1);
    krow = krow < K ? krow : K - 1;
    const int col = n0 + ((qb & 1) * 8 + (lane >> 3)) * 16 + (c & 1) * 8;
    return g.B + (long long)krow * g.ldb + col;
  };
  const unsigned short* run[G16_DMA_PER_WAVE];  // source of piece u at the next stage to be issued
  long long inc[G16_DMA_PER_WAVE];              // elements per stage (wave-uniform)
#pragma unroll
  for (int u = 0; u < G16_DMA_PER_WAVE; ++u) {
    run[u] = piece_src(st_lo, u);
    const bool is_a = wave * G16_DMA_PER_WAVE + u < G16_A_BYTES / 1024;
    inc[u] = is_a ? (A_TR ? (long long)G16_BK * g.lda : (long long)G16_BK) : (long long)G16_BK * g.ldb;
  }
  const int full_steps = K / G16_BK;  // steps below this index need no clamping along k
  // Pieces are issued strictly in (stage, u) order, each exactly once: run[u] always points at the stage
  // being issued.
  auto issue = [&](int step, int buf, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    const unsigned short* src = run[u];
    if (step >= full_steps) src = piece_src(step, u);
    run[u] += inc[u];
    // inline asm, not __builtin_amdgcn_global_load_lds: the compiler would order every later LDS read behind
    // the DMA with vmcnt(0) (it cannot tell the ring's buffers apart) -- no latency hiding at all, 4.9 k
    // cycles per K step measured.  The waits are placed by hand below.
    const unsigned int dst = lds0 + (unsigned int)(buf * G16_STAGE + (wave * G16_DMA_PER_WAVE + u) * 1024);
    KGE_STALL(__LINE__ + 5000);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
  };
  auto issue_range = [&](int step, int buf, auto lo, auto hi) __attribute__((always_inline)) {
    g16_static_for<decltype(lo)::value, decltype(hi)::value>([&](auto uc) __attribute__((always_inline)) { issue(step, buf, uc); });
  };
  using I0 = std::integral_constant<int, 0>;
  using I3 = std::integral_constant<int, 3>;
  using I6 = std::integral_constant<int, 6>;

  if (nsteps > 0) issue_range(st_lo, 0, I0{}, I6{});
  if (nsteps > 1) issue_range(st_lo + 1, 1, I0{}, I6{});

  // ---- roles: waves 0-3 contract the K slices 0, 1 of every 64-wide stage, waves 4-7 the slices 2, 3 (an
  // intra-workgroup split of K: each wave then owns a 64 x 128 block and needs 12 transpose reads per 8 MFMAs
  // instead of 8 per 4); within a group
  // wave (wm, wn) owns rows 64 wm .., columns 128 wn .. of the tile = 2 x 4 accumulators
  const int kgrp = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
  const int h = lane >> 5, g1 = (lane >> 4) & 1, l16 = lane & 15, l32 = lane & 31;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // per-lane LDS byte offsets within a stage (everything but the stage base and the slice s = 0, 1 of the
  // wave's K pair, which are compile-time immediates of the reads): slice ks = 2 kgrp + s
  unsigned int a_off[2][2], b_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (A_TR) {
      a_off[i][0] = (unsigned int)((wm * 4 + i * 2 + g1) * 128 + l16 * 8 + (kgrp * 8 + 2 * h) * (8 * 128));
      a_off[i][1] = a_off[i][0] + 4 * (8 * 128);
    } else {
      const int mr = wm * 64 + i * 32 + l32;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        a_off[i][s2] = (unsigned int)(mr * 128 + ((((kgrp * 2 + s2) * 2 + h) ^ ((mr >> 1) & 7)) << 4));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    b_off[j] = (unsigned int)(G16_A_BYTES + (wn * 8 + j * 2 + g1) * 128 + l16 * 8 + (kgrp * 8 + 2 * h) * (16 * 128));

  struct Frags {
    bf16x8 a[2], b[4];
  };
  // fragments of K slice `s` (0, 1) of this wave's pair, from the stage buffer at byte offset `sboff`
  auto load = [&](auto sc, unsigned int sboff, Frags& f) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    const unsigned char* sb = smem + sboff;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (A_TR) {
        const u32x2 r0 = g16_tr_read(sb + a_off[i][s]);
        const u32x2 r1 = g16_tr_read(sb + a_off[i][s] + 8 * 128);
        u32x4 pk;
        pk[0] = r0[0]; pk[1] = r0[1]; pk[2] = r1[0]; pk[3] = r1[1];
        f.a[i] = __builtin_bit_cast(bf16x8, pk);
      } else {
        f.a[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i][s]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 r0 = g16_tr_read(sb + b_off[j] + s * 4 * (16 * 128));
      const u32x2 r1 = g16_tr_read(sb + b_off[j] + (s * 4 + 1) * (16 * 128));
      u32x4 pk;
      pk[0] = r0[0]; pk[1] = r0[1]; pk[2] = r1[0]; pk[3] = r1[1];
      f.b[j] = __builtin_bit_cast(bf16x8, pk);
    }
  };
  // The DMA of stage st + 2 may start once stage st is open (its buffer held stage st - 1).  A DMA instruction
  // blocks its wave until the texture addresser takes it (64 B/clk per CU: 768 cycles per stage against
  // 1,024 of MFMA), so the six pieces are not issued in a burst behind the barrier (every wave would stall
  // there together: measured, DMA and MFMA time simply added up) but ONE AT A TIME between the next sixteen
  // MFMAs: the addresser queue stays short and the matrix pipe keeps its other wave.
  int pend_step = -1, pend_buf = 0;
  auto mfmas = [&](const Frags& f, auto u0c) __attribute__((always_inline)) {
    constexpr int U0 = decltype(u0c)::value;  // first of the three pending DMA pieces issued in here
    __builtin_amdgcn_sched_barrier(0);
    g16_static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
      constexpr int idx = decltype(ic)::value, i = idx / 4, j = idx % 4;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
      if constexpr (idx == 1 || idx == 4 || idx == 6) {
        constexpr int u = U0 + (idx == 1 ? 0 : idx == 4 ? 1 : 2);
        __builtin_amdgcn_sched_barrier(0);
        if (pend_step >= 0) issue(pend_step, pend_buf, std::integral_constant<int, u>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  // stage `st` opens: its DMA has landed for every wave, nobody reads stage st - 1 any more
  auto open_stage = [&](int st) __attribute__((always_inline)) {
    if (st + 1 < nsteps) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G16_DMA_PER_WAVE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KGE_BARRIER();
    pend_step = st + 2 < nsteps ? st_lo + st + 2 : -1;
    pend_buf = (st + 2) % G16_NST;
  };
  // One stage: slice 1's fragments are requested before slice 0's MFMAs are issued, the next stage is opened
  // (barrier) and ITS slice-0 fragments are requested behind them, before slice 1's MFMAs: LDS latency and
  // barrier skew sit behind 8 queued MFMAs.  No masks: out-of-range k arrive as zeros.
  Frags f0, f1;
  unsigned int cur = 0;  // byte offset of the current stage's buffer in the ring
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using U0 = std::integral_constant<int, 0>;
  using U3 = std::integral_constant<int, 3>;
  open_stage(0);
  stamp();  // 1: first stage landed
  if (pend_step >= 0) issue_range(pend_step, pend_buf, I0{}, I3{});  // nothing to hide these three behind
  load(S0{}, 0u, f0);
  for (int t = 0; t < nsteps; ++t) {
    load(S1{}, cur, f1);
    mfmas(f0, U3{});  // pieces 3..5 of the DMA pending since the last open_stage
    if (t + 1 < nsteps) {
      open_stage(t + 1);
      cur = cur + G16_STAGE == G16_NST * G16_STAGE ? 0u : cur + G16_STAGE;
      load(S0{}, cur, f0);
    } else {
      pend_step = -1;
    }
    mfmas(f1, U0{});  // pieces 0..2 of the DMA that became possible with that barrier
  }

  // ---- the two K groups exchange halves through LDS (the ring is dead: 128 of its 144 KiB): wave w of
  // group 0 ends up with columns 0-63 of its 64 x 128 block (j = 0, 1), its partner w + 4 with columns 64-127
  stamp();  // 2: K loop done (MFMAs issued)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS
  KGE_BARRIER();                      // every wave has read its last fragments
  auto exchange = [&](auto kg) __attribute__((always_inline)) {  // kg = this wave's K group, compile-time:
    constexpr int KG = decltype(kg)::value;                       // no dynamic indexing of the accumulators
    unsigned char* give = smem + ((wq * 2 + KG) * 16) * 1024 + lane * 16;
    const unsigned char* take = smem + ((wq * 2 + (KG ^ 1)) * 16) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        constexpr int JOFF = KG ? 0 : 2;  // the half this wave gives away
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v[0] = acc[i][jj + JOFF][4 * q]; v[1] = acc[i][jj + JOFF][4 * q + 1];
          v[2] = acc[i][jj + JOFF][4 * q + 2]; v[3] = acc[i][jj + JOFF][4 * q + 3];
          *reinterpret_cast<f32x4*>(give + ((i * 2 + jj) * 4 + q) * 1024) = v;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KGE_BARRIER();
    stamp();  // 3: halves exchanged
    // the sums of this wave's 64 x 64 block, in place of the half it keeps
    constexpr int KOFF = KG ? 2 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(take + ((i * 2 + jj) * 4 + q) * 1024);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][jj + KOFF][4 * q + e] += v[e];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KGE_BARRIER();  // every wave has taken its partner's half: the exchange area is free
    // Store through a wave-private row-major image of the block (64 rows x 272 B): the accumulator layout
    // puts one column per lane (64 dword stores of 2 x 128 B each, 5.7 k cycles to issue); from the image every
    // store instruction writes 4 rows x 256 contiguous bytes, 16 bytes per lane.
    // Accumulator register r of a 32 x 32 block is row 8 (r / 4) + 4 h + r % 4, column lane % 32.
    constexpr int WP = 272;
    unsigned char* img = smem + (KG * 4 + wq) * (64 * WP);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r >> 2) * 8 + h * 4 + (r & 3);
          *reinterpret_cast<float*>(img + row * WP + (jj * 32 + l32) * 4) = acc[i][jj + KOFF][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private: no barrier)
    float* C = g.C + (long long)split * g.c_split;
    const int rsub = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + rsub;
      const f32x4 v = *reinterpret_cast<const f32x4*>(img + row * WP + c4 * 4);
      const int grow = m0 + wm * 64 + row;
      if (grow < g.M)
        *reinterpret_cast<f32x4*>(C + (long long)grow * g.ldc + n0 + wn * 128 + KOFF * 32 + c4) = v;
    }
  };
  if (kgrp) exchange(std::integral_constant<int, 1>{});
  else exchange(std::integral_constant<int, 0>{});
  stamp();  // 4: stores issued
  if (g.dbg != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();  // 5: stores acknowledged
  }
}

bool g16_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

static unsigned long long* g16_dbg = nullptr;  // set by kge_debug_gemm16_stamps
void set_g16_dbg(unsigned long long* p) { g16_dbg = p; }

bool bwd_gemm16_enabled() {
  return sw(SW_BWD_GEMM_LIB) != 1;
}

// bytes of split-K scratch with which run_gemm16_dq takes as many splits as it wants (0: it would not split)
long long gemm16_dq_scratch_bytes(int d, long long rows, long long m) {
  if ((d % G16_BN) || rows <= 0 || m <= 0) return 0;
  const long long per = ((rows + G16_BM - 1) / G16_BM) * (d / G16_BN);
  long long splits = 256 / per;
  const long long ksteps = (m + G16_BK - 1) / G16_BK;
  if (splits > ksteps) splits = ksteps;
  return splits < 2 ? 0 : splits * rows * (long long)d * (long long)sizeof(float);
}

// dQ[rows, d] = G16[rows, :m] * T[m, d].  Returns the number of split-K partials written to `scratch`
// ([splits][rows, d]; the caller sums them into C: bwdg_reduce_kernel), 1 if the product went straight
// to C (no split), 0 if the shape is not handled (the caller's library path).
int run_gemm16_dq(int d, long long rows, long long m, const unsigned short* T, long long ldt,
                  const unsigned short* G16, long long mp, float* C, float* scratch, size_t scratch_bytes,
                  hipStream_t st) {
  if (!bwd_gemm16_enabled() || (d % G16_BN) || rows <= 0 || m <= 0) return 0;
  if (rows >= (1LL << 30) || m >= (1LL << 30) || (mp & 7) || (ldt & 7) || !g16_al16(T) || !g16_al16(G16)) return 0;
  G16Args g{};
  g.A = G16; g.lda = mp; g.a_cols = (int)mp;
  g.B = T; g.ldb = ldt;
  g.ldc = d; g.c_split = rows * d;
  g.M = (int)rows; g.N = d; g.K = (int)m;
  g.mtiles = (int)((rows + G16_BM - 1) / G16_BM);
  g.ntiles = d / G16_BN;
  g.ksteps = (int)((m + G16_BK - 1) / G16_BK);
  const long long per = (long long)g.mtiles * g.ntiles;
  long long splits = 256 / per;  // ~one workgroup per CU
  const long long fit = scratch ? (long long)(scratch_bytes / ((size_t)rows * d * sizeof(float))) : 0;
  if (splits > fit) splits = fit;
  if (splits > g.ksteps) splits = g.ksteps;
  if (splits < 2) splits = 1;
  g.dbg = g16_dbg;
  g.splits = (int)splits;
  g.C = splits > 1 ? scratch : C;
  const long long grid = 8 * per * ((splits + 7) / 8);
  if (grid > 0x7fffffffLL) return 0;
  hipLaunchKernelGGL((gemm16_kernel<false>), dim3((unsigned)grid), dim3(512), 0, st, g);
  return hipGetLastError() == hipSuccess ? g.splits : 0;
}

// dT[m, d] = G16[rows, :m]^T * Q16[rows, d]
bool run_gemm16_dt(int d, long long rows, long long m, const unsigned short* Q16, const unsigned short* G16,
                   long long mp, float* dT, hipStream_t st) {
  if (!bwd_gemm16_enabled() || (d % G16_BN) || rows <= 0 || m <= 0) return false;
  if (rows >= (1LL << 30) || m >= (1LL << 30) || (mp & 7) || !g16_al16(Q16) || !g16_al16(G16)) return false;
  G16Args g{};
  g.A = G16; g.lda = mp; g.a_cols = (int)mp;
  g.B = Q16; g.ldb = d;
  g.C = dT; g.ldc = d; g.c_split = 0;
  g.M = (int)m; g.N = d; g.K = (int)rows;
  g.mtiles = (int)((m + G16_BM - 1) / G16_BM);
  g.ntiles = d / G16_BN;
  g.ksteps = (int)((rows + G16_BK - 1) / G16_BK);
  g.splits = 1;
  g.dbg = g16_dbg;
  const long long grid = 8LL * g.ntiles * ((g.mtiles + 7) / 8);
  if (grid > 0x7fffffffLL) return false;
  hipLaunchKernelGGL((gemm16_kernel<true>), dim3((unsigned)grid), dim3(512), 0, st, g);
  return hipGetLastError() == hipSuccess;
}

}  // namespace kge
