// score_pairs_bf16_v3.hip -- row-persistent ComplEx / DistMult sp_/_po kernel for bf16 tables,
// d in {128, 256, 512}, with 64-target tiles: the BASELINE.json headline path on gfx950.
//
// Same structure as v2 (score_pairs_bf16_v2.hip: query fragments built once per workgroup in
// MFMA operand registers, target tiles streamed through LDS by LDS-DMA, counted lgkmcnt B
// pipeline, LDS-transposed 16-byte stores interleaved into the next tile's MFMA chain), with
// the per-tile fixed cost (barrier, waits, address set-up: ~1,000 of v2's 2,250 cycles per
// 32-target tile) amortised over twice the work:
//
//   * a tile is 64 targets (64 KiB at d=512): 64 MFMAs per wave per tile, two accumulators
//     (targets 0-31 / 32-63); the two halves share each query fragment;
//   * LDS: ring of TWO tile buffers (the DMA of tile t+1 is issued during tile t, into the
//     buffer tile t-1 was read from), one 64 KiB prologue staging slot overlapping buffer 1,
//     one C-tile transpose buffer per wave reused by both halves;
//   * VMEM order per step: the DMA pieces of tile t+1 in the first half of the MFMA chain, the
//     stores of tile t-1 in the second half, so the next step's counted vmcnt(8) waits for the
//     tile and leaves the stores in flight.
#include "common.hpp"
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <type_traits>

namespace kge {

constexpr int V3_ROWS = 128, V3_TN = 64;

typedef float f32x4v3u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ long long v3_shfl64(long long v, int src) {
  int lo = __shfl((int)(v & 0xffffffffLL), src, 64);
  int hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// compile-time loop: the body gets std::integral_constant<int, I> (asm immediates need constants)
template <int I, int N, class F>
__device__ __forceinline__ void v3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    v3_static_for<I + 1, N>(f);
  }
}

// element of the tile's two accumulators at tile-relative column `rel` (minus 4 fh; -1: none):
// accumulator element r of half hf is column 32 hf + 8 (r >> 2) + 4 fh + (r & 3)
__device__ __forceinline__ float v3_pick(f32x16 h0, f32x16 h1, int rel, float cur) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int off = 8 * (r >> 2) + (r & 3);
    cur = rel == off ? h0[r] : cur;
    cur = rel == 32 + off ? h1[r] : cur;
  }
  return cur;
}

// COOP (cooperative query build, needs a workspace): the query rows of a row group are needed
// by all `ncg` workgroups of that row group.  Instead of every workgroup rebuilding all 128
// rows (ncg-fold redundant: gather 256 KiB + ~8,000 VALU cycles per wave), the first `nbuild`
// workgroups of the row group each build a share of the rows ONCE, write them fragment-major
// to the workspace (`qf`, agent-scope write-through stores) and publish a per-workgroup flag
// (= this launch's epoch); every workgroup then polls the flags (agent-scope loads) and loads
// its fragments with 32 plain 16-byte loads per lane.  All workgroups are co-resident (grid <= number of CUs, one
// workgroup per CU -- checked by the launcher), so the spin-wait cannot deadlock.
template <int SCORER, int HH, int TGMODE, bool COOP, int EPI = V3_STORE>
__global__ __launch_bounds__(256, 1) void pairs_bf16_v3_kernel(
    Operand A, Operand R, Operand TG, int dir, long long n, long long m, int rgn, int ncg,
    int tiles_per_cg, int ntiles, float* __restrict__ out, long long ldo,
    unsigned long long* __restrict__ dbg, u32x4* __restrict__ qf,
    unsigned long long* __restrict__ flags, unsigned long long epoch, int nbuild, CeArgs ce) {
  constexpr int NKB = 2 * HH / 16;        // K-blocks of 16
  constexpr int NKH = HH / 16;            // K-blocks per half
  constexpr int ROWB = 4 * HH;            // bytes per table row (2*HH bf16)
  constexpr int SPR = HH / 4;             // 16-byte slots per row
  constexpr int TILEB = V3_TN * ROWB;     // bytes per target tile
  constexpr int NL = TILEB / 16 / 256;    // 1-KiB DMA pieces per wave per tile
  constexpr int PASSES = HH / 64;         // prologue passes of 64 coordinates
  constexpr int STAGE = 4 * 16384;        // prologue staging slot: 16 KiB per wave
  constexpr int STG0 = TILEB;             // behind ring buffer 0 (overlaps buffer 1)
  constexpr int CST0 = (2 * TILEB > STG0 + STAGE) ? 2 * TILEB : STG0 + STAGE;
  constexpr int SMEM = CST0 + 4 * 32 * 144;
  constexpr int NQ = 2 * NKB;             // MFMAs per tile (two 32-target halves)
  constexpr bool IS_DS = EPI == V3_DS || EPI == V3_DSIG;      // writes bf16 gradients of the scores
  constexpr int NST = EPI == V3_STORE ? 8 : (IS_DS ? 4 : 0);  // vector stores per tile per lane
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  // ---- which rows / target tiles
  const int b = blockIdx.x;
  const int q8 = b >> 3;
  const int rg = q8 % rgn;
  const int cg = (q8 / rgn) * 8 + (b & 7);
  if (cg >= ncg) return;
  const int tile_lo = cg * tiles_per_cg;
  int ntl = ntiles - tile_lo;
  if (ntl > tiles_per_cg) ntl = tiles_per_cg;
  if (ntl <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
  const int fi = lane & 31, fh = lane >> 5;
  // Two-sided fused-loss launch (kge_ce_sp_po_*): row groups [0, rgn1) are the n (s, p, ?) queries,
  // row groups [rgn1, rgn) the n (?, p, o) queries; per-row arrays of the second side start at
  // row ce.side2_off.  One-sided: ce.rgn1 == 0.
  int rgl = rg;              // row group within its side
  long long roff = 0;        // offset of this side's rows in lse / g_rows / part / true_score / G16
  Index lab_ix = ce.label;
  if constexpr (EPI != V3_STORE) {
    if (ce.rgn1 > 0 && rg >= ce.rgn1) {
      rgl = rg - ce.rgn1;
      roff = ce.side2_off;
      A = ce.a2;
      dir = KGE_PO_;
      lab_ix = ce.label2;
    }
  }
  const long long row0 = (long long)rgl * V3_ROWS + 32 * wave;  // first query row (within its side)
  const unsigned short* tgb = (const unsigned short*)TG.base;

  int dbg_i = 0;
  auto stamp = [&]() {  // optional per-phase timestamps (tools/v2_phases.py); dbg == NULL in production
    if (dbg != nullptr && tid == 0 && dbg_i < 64)
      dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  stamp();  // 0: kernel start

  // ---- target tile DMA (HBM -> LDS, no registers), one 1-KiB piece per call.  The LDS image
  // of a tile is lane-linear ([row][16-B slot]); the XOR swizzle (slot ^ (row & 15)) is applied
  // on the SOURCE address.
  auto dma_piece = [&](int tt, int buf, int k) {
    if (tt >= ntl) tt = ntl - 1;  // constant VMEM op count per step
    const long long trow0 = (long long)(tile_lo + tt) * V3_TN;
    unsigned char* dst = smem + buf * TILEB + (wave * NL + k) * 1024;  // wave-uniform
    const int L = (wave * NL + k) * 64 + lane;
    const int row = L / SPR, slot = L % SPR;
    long long tr = trow0 + row;
    if (tr >= m) tr = m - 1;
    const unsigned short* src =
        tgb + index_mode<TGMODE>(TG.idx, tr) * TG.ld + ((slot ^ (row & 15)) << 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto tile_dma = [&](int tt, int buf) {
#pragma unroll
    for (int k = 0; k < NL; ++k) dma_piece(tt, buf, k);
  };

  // ---- prologue: build the query fragments of this wave's 32 rows in registers: the s / r
  // rows are gathered in passes of 64 coordinates as 128-byte segments (full cache lines) ->
  // registers -> wave-private LDS staging -> re-read in fragment shape; the loads of pass p+1
  // fly while pass p is built.
  bf16x8 afr[NKB];
  if constexpr (COOP) {
    // one item = 8 coordinates of both halves of one query row, written fragment-major (K-block kb of 32-row block rb
    // is 64 lanes x 16 B, contiguous) with agent-scope (sc1) write-through stores: visible to the other XCDs' L2s
    // once acknowledged, without the whole-L2 write-back of a release fence.  A 128-byte line (8 rows x 16 B) is
    // written by one workgroup only (or by several with the same bytes).
    auto build_item = [&](int rr, int c8) __attribute__((always_inline)) {
      const long long row = (long long)rg * V3_ROWS + rr;    // row of the fragment workspace
      const long long lrow = (long long)rgl * V3_ROWS + rr;  // query row within its side
      const long long qrow = lrow < n ? lrow : n - 1;        // padded rows repeat row n-1
      const unsigned short* a = (const unsigned short*)A.base + index_at(A.idx, qrow) * A.ld + c8 * 8;
      const unsigned short* r = (const unsigned short*)R.base + index_at(R.idx, qrow) * R.ld + c8 * 8;
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(a), a1 = *reinterpret_cast<const u32x4*>(a + HH);
      const u32x4 r0 = *reinterpret_cast<const u32x4*>(r), r1 = *reinterpret_cast<const u32x4*>(r + HH);
      u32x4 q0, q1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned int x0, x1;
        bf16_qpair<SCORER>(dir, a0[e], a1[e], r0[e], r1[e], x0, x1);
        q0[e] = x0;
        q1[e] = x1;
      }
      u32x4* dst = qf + ((row >> 5) * NKB) * 64 + (row & 31) + 32 * (c8 & 1);
      // s_nop: the two wait states a >64-bit VMEM store needs before its data registers may be overwritten
      // (the hazard recogniser cannot see into inline asm)
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (c8 >> 1) * 64), "v"(q0) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (NKH + (c8 >> 1)) * 64), "v"(q1)
                   : "memory");
    };
    constexpr int CGR = HH / 8;  // groups of 8 coordinates per row
    if (cg < nbuild) {  // this workgroup's share of the row group's query rows
      for (int it = cg * 256 + tid; it < V3_ROWS * CGR; it += nbuild * 256) build_item(it / CGR, it % CGR);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every thread: its stores are acknowledged ...
      __syncthreads();                                  // ... before thread 0 publishes
      if (tid == 0)
        __hip_atomic_store(flags + rg * 16 + cg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the workspace's degraded word (shared with the loader/consumer kernel, score_pairs_bf16_v4.hip): a COUNTDOWN of
    // launches during which the hand-off is not tried; one thread per launch takes one off
    unsigned long long* const degraded = flags + 512 * 8;
    if (blockIdx.x == 0 && tid == 0) {
      const unsigned long long v = __hip_atomic_load(degraded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0ull) __hip_atomic_store(degraded, v - 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp();  // 1: share built and published
    tile_dma(0, 0);
    tile_dma(1, 1);
    stamp();  // 2: tiles 0, 1 issued
    {
      // A consumer never trusts a builder blindly: the wait is bounded (a builder workgroup that is not running --
      // its CU busy with another stream's kernel, CU masking, a second process -- must neither hang this wave nor
      // kill the process).  On a time-out the wave builds the fragments of its own 32 rows itself, through the
      // workspace with the builders' code (same bytes, so it does not matter who else writes them), and marks the
      // workspace degraded: the next launches skip the hand-off at once.  Never a trap.
      const unsigned long long* f = flags + rg * 16;
      bool ok = nbuild > 0 && __hip_atomic_load(degraded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
      for (int spin = 0; ok; ++spin) {
        const unsigned long long v =
            lane < nbuild ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all(v == epoch)) break;
        if (spin == (1 << 16)) {  // ~0.1 s: far beyond any launch skew
          ok = false;
          if (lane == 0 && nbuild > 0)
            __hip_atomic_store(degraded, 4096ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) {  // slow (32 items per lane) and rare
        for (int it = lane; it < 32 * CGR; it += 64) build_item(32 * wave + it / CGR, it % CGR);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // No acquire fence (it would invalidate this CU's L1 and cost ~1.7 us): the builders
      // stored with sc1, the fragments are read with sc1 loads below (served by L2, never by L1).
      asm volatile("" ::: "memory");
    }
    stamp();  // 3: all shares of this row group published
    const unsigned char* sb =
        (const unsigned char*)(qf + ((long long)(rg * (V3_ROWS / 32) + wave) * NKB) * 64);  // uniform
    // sc1 buffer loads through the compiler's builtin (not inline asm): the register allocator then
    // knows the values arrive asynchronously.  With asm loads + a later asm s_waitcnt it is free to
    // move a fragment register (say into an AGPR, under register pressure) between the two -- copying
    // garbage and letting the late data land in a register it has reused (seen as memory faults in
    // the 450-VGPR softplus variant at d = 512).
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)sb, 0, NKB * 1024, 0x00020000);
    v3_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      afr[kb] = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, (unsigned int)(lane * 16 + kb * 1024), 0, 16 /* sc1 */));
    });
    // not waited for here: the compiler waits for fragment kb in front of its first MFMA (tile 0)
    stamp();  // 4: fragments loaded
  } else {
    const unsigned short* ab = (const unsigned short*)A.base;
    const unsigned short* rb = (const unsigned short*)R.base;
    long long qrow = row0 + fi;
    if (qrow >= n) qrow = n - 1;
    const long long aoff = index_at(A.idx, qrow) * A.ld;  // element offsets of row `fi`
    const long long roff = index_at(R.idx, qrow) * R.ld;
    // source pointer of gather load k: row rr = 2k + fh, LDS slot (lane & 31) holds the
    // logical slot p5 = (lane & 31) ^ (rr & 15) = array (p5 >> 3), 16-B chunk (p5 & 7)
    const unsigned short* gsrc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int rr = 2 * k + fh;
      const int p5 = (lane & 31) ^ (rr & 15);
      const long long ao = v3_shfl64(aoff, rr), ro = v3_shfl64(roff, rr);
      gsrc[k] = ((p5 < 16) ? ab + ao : rb + ro) + ((p5 >> 3) & 1) * HH + (p5 & 7) * 8;
    }
    stamp();  // 1: indices loaded, source pointers built
    u32x4 g[16];
    auto gather = [&](int p) {
#pragma unroll
      for (int k = 0; k < 16; ++k) g[k] = *reinterpret_cast<const u32x4*>(gsrc[k] + 64 * p);
    };
    unsigned char* const stg = smem + STG0 + wave * 16384;
    auto stage_write = [&]() {
#pragma unroll
      for (int k = 0; k < 16; ++k) *reinterpret_cast<u32x4*>(stg + lane * 16 + k * 1024) = g[k];
    };
    auto wave_fence = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto build = [&](int p) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        u32x4 v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int p5 = a * 8 + 2 * jj + fh;
          v[a] = *reinterpret_cast<const u32x4*>(stg + fi * 512 + ((p5 ^ (fi & 15)) << 4));
        }
        u32x4 q0, q1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned int x0, x1;
          bf16_qpair<SCORER>(dir, v[0][e], v[1][e], v[2][e], v[3][e], x0, x1);
          q0[e] = x0;
          q1[e] = x1;
        }
        afr[4 * p + jj] = __builtin_bit_cast(bf16x8, q0);
        afr[NKH + 4 * p + jj] = __builtin_bit_cast(bf16x8, q1);
      }
    };
    gather(0);
    tile_dma(0, 0);  // behind the gather in the in-order VMEM queue
    stamp();  // 2: gather 0 and tile 0 issued
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      stage_write();  // waits for the loads of pass p (compiler-tracked); slot is free (fence below)
      wave_fence();
      stamp();  // 3+2p: pass p landed and staged
      if (p + 1 < PASSES) gather(p + 1);  // flies while pass p is built
      build(p);
      wave_fence();  // this wave's reads of the slot are done before the next stage_write
      stamp();  // 4+2p: pass p built
    }
    // staging overlaps ring buffer 1: everyone must be done before tile 1 streams in
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    KGE_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    tile_dma(1, 1);
    stamp();  // 3+2*PASSES: prologue done, all waves synchronised
  }

  // ---- main loop over this workgroup's target tiles
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(((2 * t + fh) ^ (fi & 15)) << 4);
  long long orow = row0 + fi;
  if (orow >= n) orow = n - 1;
  float* const orow_ptr = out + orow * ldo;

  // ---- fused loss epilogues: per-lane state of query row `fi` (both lanes fh = 0 / 1 of a row
  // hold it; padded rows repeat row n-1 like their query fragments do)
  long long lab = -1;             // label column of this lane's row
  float rmax = -__builtin_inff(); // V3_LSE: running max over this lane's columns so far
  float rsum = 0.0f;              //         running sum of exp(score - rmax)
  float tsc = 0.0f;               //         score(row, label) if one of this lane's columns
  bool tfound = false;
  float lse_i = 0.0f, g_i = 0.0f, gb_i = 0.0f; // V3_DS (gb_i = g_i * row_bias[i])
  if constexpr (EPI != V3_STORE) {
    if (lab_ix.ptr != nullptr) lab = index_at(lab_ix, orow);
    if constexpr (IS_DS) {
      if constexpr (EPI == V3_DS) lse_i = ce.lse[orow + roff];
      g_i = ce_row_gradient(ce, orow + roff);
      if (ce.rowptr != nullptr && ce.rowptr[orow + 1] == ce.rowptr[orow]) g_i = 0.0f;
      if (EPI == V3_DS && ce.row_bias != nullptr) gb_i = g_i * ce.row_bias[orow + roff];
    }
  }
  // accumulator element r of half hf is column  col0(tile) + 32 hf + 8 (r >> 2) + 4 fh + (r & 3)
  auto lse_pick = [&](int tt, const f32x16& h0, const f32x16& h1) __attribute__((always_inline)) {
    const long long rel = lab - ((long long)(tile_lo + tt) * V3_TN + 4 * fh);
    const bool hit = rel >= 0 && rel < V3_TN && (rel & 7) < 4 && lab < m;  // not a padded column
    if (__any(hit)) {  // rare: a row's label lies in exactly one tile of the table
      tsc = v3_pick(h0, h1, hit ? (int)rel : -1, tsc);
      tfound = tfound || hit;
    }
  };
  auto ds_value = [&](float sc, long long rel, int off) {
    float pv;
    if constexpr (EPI == V3_DSIG)  // sigmoid(score + offset)
      pv = g_i / (1.0f + __builtin_amdgcn_exp2f(-(sc + ce.offset) * V3_LOG2E));
    else  // softmax
      pv = __builtin_amdgcn_exp2f((sc - lse_i) * V3_LOG2E) * g_i - gb_i;
    if (rel == off) pv -= g_i;
    return pv;
  };
  // four d-loss/d-score values (group g of half hf) -> bf16 -> transpose buffer; a row of the
  // buffer is the tile's 64 columns x 2 B = 128 B
  unsigned char* const cst16 = smem + CST0 + wave * (32 * 144);
  auto ds_write = [&](int tt, const f32x16& acc, int hf, int g) {
    const long long rel = lab - ((long long)(tile_lo + tt) * V3_TN + 4 * fh);
    const int off = 32 * hf + 8 * g;
    float p0 = ds_value(acc[4 * g], rel, off), p1 = ds_value(acc[4 * g + 1], rel, off + 1);
    float p2 = ds_value(acc[4 * g + 2], rel, off + 2), p3 = ds_value(acc[4 * g + 3], rel, off + 3);
    // The pad columns [m, pitch) of the last tile are ZERO: the gradient products (bwd_gemm16.hip) read
    // whole 16-byte chunks of a row and rely on it.  (Wave-uniform branch: only the ragged last tile.)
    const long long colb = (long long)(tile_lo + tt) * V3_TN;
    if (colb + V3_TN > m) {
      const long long lim = m - (colb + 4 * fh + off);  // columns of this group that exist
      p0 = lim > 0 ? p0 : 0.0f;
      p1 = lim > 1 ? p1 : 0.0f;
      p2 = lim > 2 ? p2 : 0.0f;
      p3 = lim > 3 ? p3 : 0.0f;
    }
    u32x2 v = {bf16_pack(p0, p1), bf16_pack(p2, p3)};
    *reinterpret_cast<u32x2*>(cst16 + fi * 144 + (off + 4 * fh) * 2) = v;
  };
  auto ds_store = [&](int tt, int i, const f32x4& v) {
    long long orow_i = row0 + 8 * i + (lane >> 3);
    if (orow_i >= n) orow_i = n - 1;
    *reinterpret_cast<f32x4*>(ce.g16 + (orow_i + roff) * ce.ld16 + (long long)(tile_lo + tt) * V3_TN +
                              8 * (lane & 7)) = v;
  };

  // acc[4g + e] = score(query fi, target col0 + 8g + 4fh + e) for one 32-target half.  The
  // finished half goes through a wave-private LDS transpose so that every store instruction
  // writes 8 rows x 128 contiguous bytes; its instructions are spread between the MFMAs of the
  // NEXT tile (two accumulator sets per half, ping-pong).
  unsigned char* const cst = smem + CST0 + wave * (32 * 144);
  auto ep_write = [&](const f32x16& acc, int g) {
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    *reinterpret_cast<f32x4*>(cst + fi * 144 + (8 * g + 4 * fh) * 4) = v;
  };
  auto ep_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto ep_read = [&](int i) {
    return *reinterpret_cast<const f32x4*>(cst + (8 * i + (lane >> 3)) * 144 + (lane & 7) * 16);
  };
  auto ep_store = [&](int tt, int half, int i, const f32x4& v) {
    long long orow_i = row0 + 8 * i + (lane >> 3);
    if (orow_i >= n) orow_i = n - 1;  // clamped rows rewrite the bits of row n-1
    *reinterpret_cast<f32x4v3u*>(out + orow_i * ldo + (long long)(tile_lo + tt) * V3_TN + 32 * half +
                                 4 * (lane & 7)) = v;
  };
  auto half_full = [&](int tt, int half) {
    return (long long)(tile_lo + tt) * V3_TN + 32 * half + 32 <= m;
  };
  auto store_half_now = [&](int tt, int half, const f32x16& acc) {  // after the loop
    const long long col0 = (long long)(tile_lo + tt) * V3_TN + 32 * half;
    if (col0 >= m) return;
    if (half_full(tt, half)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) ep_write(acc, g);
      ep_fence();
      f32x4 cv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cv[i] = ep_read(i);
      ep_fence();
#pragma unroll
      for (int i = 0; i < 4; ++i) ep_store(tt, half, i, cv[i]);
    } else {  // ragged end of the table: scalar stores, clamped to column m-1
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        long long oc = col0 + 8 * (r >> 2) + 4 * fh + (r & 3);
        if (oc >= m) oc = m - 1;
        orow_ptr[oc] = acc[r];
      }
    }
  };

  // One tile (64 targets): wait + barrier, then a chain of NQ = 2*NKB MFMAs (q = 2*kb + half);
  // between them: the B-fragment reads (inline asm, PF in flight, counted lgkmcnt), the DMA
  // pieces of tile tt+1 and the epilogue of tile tt-1 (both halves, one after the other through
  // the same transpose buffer).  Only the last tile of the table can be ragged; it is stored
  // after the loop, so every in-loop epilogue is a full 2 x 4 vector stores.
  auto tile_body = [&](int tt, f32x16& acc0, f32x16& acc1, const f32x16& accp0, const f32x16& accp1,
                       bool store_prev) {
    // in-order VMEM queue at this point: step 0: [tile 0][tile 1]; step 1: [tile 1]; later:
    // [DMA of tile tt (NL pieces)][NST stores of tile tt-2] -- the stores may stay in flight
    if (tt == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NL) : "memory");
    else if (tt == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(NST) : "memory");
    KGE_BARRIER();  // tile tt visible to all; everyone finished reading tile tt-1
    __builtin_amdgcn_sched_barrier(0);
    stamp();  // tile tt released
    // B fragment (K-block kb, half hf) of target row 32*hf + fi: 16-B slot s = s0(kb) + fh, stored
    // at slot s ^ (fi & 15): with s = 16*a + b the swizzle only touches b -> 8 address registers
    // plus immediates a*256 + hf*32*ROWB.
    const unsigned int bt = (unsigned int)((tt & 1) * TILEB + fi * ROWB);
    unsigned int bp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bp[t] = bt + boff[t];
    constexpr int PF = 8;
    constexpr int U = NQ / 16;  // epilogue schedule unit
    bf16x8 bq[PF];
    f32x4 cv0[4], cv1[4];
    float tmax = 0.0f, tsum = 0.0f;  // V3_LSE: max / partial sum of the tile being folded in
    auto bread = [&](bf16x8& dst, auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int kb = q >> 1, hf = q & 1;
      constexpr int s0 = (kb < NKH) ? (2 * kb) : (HH / 8 + 2 * (kb - NKH));
      const unsigned int addr = bp[(s0 & 15) >> 1];
      asm volatile("ds_read_b128 %0, %1 offset:%2"
                   : "=v"(dst)
                   : "v"(addr), "i"((s0 >> 4) * 256 + hf * 32 * ROWB)
                   : "memory");
    };
    v3_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    v3_static_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      // reads newer than read q: min(PF - 1, NQ - 1 - q)
      if (NQ - 1 - q >= PF - 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(PF - 1) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NQ - 1 - q) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % PF], afr[q >> 1], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % PF], afr[q >> 1], acc0, 0, 0, 0);
      if constexpr (q + PF < NQ) bread(bq[q % PF], std::integral_constant<int, q + PF>{});
      // first half of the chain: DMA of tile tt+1 into the buffer tile tt-1 was read from (tile 1
      // was issued by the prologue); second half: the stores of tile tt-1 (behind the DMA in
      // the VMEM queue, so the next step's counted wait leaves them in flight)
      if (tt >= 1 && (q & 1) == 0 && (q >> 1) < NL) dma_piece(tt + 1, (tt + 1) & 1, q >> 1);
      if constexpr (EPI == V3_LSE) {
        if (store_prev) {  // online softmax over tile tt-1: 32 columns of this lane's row
          if (q == 0) {
            float mx = rmax;
            v3_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
              constexpr int r = decltype(rc)::value;
              mx = __builtin_fmaxf(mx, __builtin_fmaxf(accp0[r], accp1[r]));
            });
            tmax = mx;
            tsum = 0.0f;
          }
          if constexpr (q % U == 0 && q / U >= 1 && q / U <= 8) {
            constexpr int k = q / U - 1;
            tsum += (__builtin_amdgcn_exp2f((accp0[2 * k] - tmax) * V3_LOG2E) +
                     __builtin_amdgcn_exp2f((accp0[2 * k + 1] - tmax) * V3_LOG2E)) +
                    (__builtin_amdgcn_exp2f((accp1[2 * k] - tmax) * V3_LOG2E) +
                     __builtin_amdgcn_exp2f((accp1[2 * k + 1] - tmax) * V3_LOG2E));
          }
          if (q == 9 * U) {
            rsum = rsum * __builtin_amdgcn_exp2f((rmax - tmax) * V3_LOG2E) + tsum;
            rmax = tmax;
          }
          if (q == 10 * U) lse_pick(tt - 1, accp0, accp1);
        }
      } else if constexpr (EPI == V3_SPLUS) {
        if (store_prev) {  // running sum of softplus over tile tt-1: 32 columns of this lane's row
          if constexpr (q % U == 0 && q / U >= 1 && q / U <= 8) {
            constexpr int k = q / U - 1;
            rsum += (v3_softplus(accp0[2 * k] + ce.offset) + v3_softplus(accp0[2 * k + 1] + ce.offset)) +
                    (v3_softplus(accp1[2 * k] + ce.offset) + v3_softplus(accp1[2 * k + 1] + ce.offset));
          }
          if (q == 10 * U) lse_pick(tt - 1, accp0, accp1);
        }
      } else if constexpr (IS_DS) {
        if (store_prev) {  // d loss / d score of tile tt-1 -> bf16 -> transpose -> 4 x 16-byte stores
          if constexpr (q % U == 0 && q / U < 8) {
            constexpr int k = q / U;
            if constexpr (k < 4) ds_write(tt - 1, accp0, 0, k);
            else ds_write(tt - 1, accp1, 1, k - 4);
          }
          if (q == 8 * U) ep_fence();
          if (q == 9 * U) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cv0[i] = ep_read(i);
          }
          if (q == 10 * U) ep_fence();
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (q == (11 + i) * U) ds_store(tt - 1, i, cv0[i]);
        }
      } else if (store_prev) {  // epilogue of tile tt-1: half 0 then half 1 through the same buffer
        if (q == 0) { ep_write(accp0, 0); ep_write(accp0, 1); }
        if (q == U) { ep_write(accp0, 2); ep_write(accp0, 3); }
        if (q == 2 * U) ep_fence();
        if (q == 3 * U) {
#pragma unroll
          for (int i = 0; i < 4; ++i) cv0[i] = ep_read(i);
        }
        if (q == 4 * U) ep_fence();
        if (q == 5 * U) { ep_write(accp1, 0); ep_write(accp1, 1); }
        if (q == 6 * U) { ep_write(accp1, 2); ep_write(accp1, 3); }
        if (q == 7 * U) ep_fence();
        if (q == 8 * U) {
#pragma unroll
          for (int i = 0; i < 4; ++i) cv1[i] = ep_read(i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (q == (8 + i) * U) ep_store(tt - 1, 0, i, cv0[i]);
          if (q == (12 + i) * U) ep_store(tt - 1, 1, i, cv1[i]);
        }
      }
    });
    stamp();  // tile tt: MFMA chain issued
  };

  f32x16 a0 = {}, a1 = {}, b0 = {}, b1 = {};
  tile_body(0, a0, a1, b0, b1, false);
  int tt = 1;
  for (; tt + 1 < ntl; tt += 2) {
    tile_body(tt, b0, b1, a0, a1, true);
    tile_body(tt + 1, a0, a1, b0, b1, true);
  }
  if (tt < ntl) {  // odd tail: the last tile accumulated in b0 / b1
    tile_body(tt, b0, b1, a0, a1, true);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a0[r] = b0[r];
      a1[r] = b1[r];
    }
  }
  // the workgroup's last tile (ntl - 1) is in a0 / a1
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (EPI == V3_LSE) {
    // only this tile can reach beyond column m (its rows then repeat row m-1): mask
    const long long c0 = (long long)(tile_lo + ntl - 1) * V3_TN + 4 * fh;
    f32x16 m0, m1;  // masked copies
    float mx = rmax;
    v3_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      const long long c = c0 + 8 * (r >> 2) + (r & 3);
      m0[r] = c < m ? a0[r] : -__builtin_inff();
      m1[r] = c + 32 < m ? a1[r] : -__builtin_inff();
      mx = __builtin_fmaxf(mx, __builtin_fmaxf(m0[r], m1[r]));
    });
    // A lane whose columns of this tile are ALL padding, in a column group that holds nothing but this tile
    // (m % 64 <= 4 and one tile per group), still has mx = -inf: (-inf) - (-inf) would make its (0, 0) a NaN
    // and the row's loss with it.  Exponents are taken against a finite reference then (same bits otherwise).
    const float rf = mx == -__builtin_inff() ? 0.0f : mx;
    float sm = 0.0f;
    v3_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      sm += __builtin_amdgcn_exp2f((m0[r] - rf) * V3_LOG2E) + __builtin_amdgcn_exp2f((m1[r] - rf) * V3_LOG2E);
    });
    rsum = rsum * __builtin_amdgcn_exp2f((rmax - rf) * V3_LOG2E) + sm;
    rmax = mx;
    lse_pick(ntl - 1, a0, a1);
    // the two lanes of a row -> one (max, sum exp) per row and column group
    const float omax = __shfl_xor(rmax, 32, 64), osum = __shfl_xor(rsum, 32, 64);
    const float M = __builtin_fmaxf(rmax, omax);
    const float L = rsum * __builtin_amdgcn_exp2f((rmax - M) * V3_LOG2E) +
                    osum * __builtin_amdgcn_exp2f((omax - M) * V3_LOG2E);
    const long long row = row0 + fi;
    if (row < n) {
      if (fh == 0) {
        float* pp = ce.part + ((row + roff) * ncg + cg) * 2;
        pp[0] = M;
        pp[1] = L;
      }
      if (tfound) ce.true_score[row + roff] = tsc;  // (never with ce.label.ptr == NULL: lab stays -1)
    }
  } else if constexpr (EPI == V3_SPLUS) {
    const long long c0 = (long long)(tile_lo + ntl - 1) * V3_TN + 4 * fh;
    float sm = 0.0f;  // columns beyond m contribute nothing
    v3_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      const long long c = c0 + 8 * (r >> 2) + (r & 3);
      sm += (c < m ? v3_softplus(a0[r] + ce.offset) : 0.0f) + (c + 32 < m ? v3_softplus(a1[r] + ce.offset) : 0.0f);
    });
    rsum += sm;
    lse_pick(ntl - 1, a0, a1);
    const float S = rsum + __shfl_xor(rsum, 32, 64);  // the two lanes of a row
    const long long row = row0 + fi;
    if (row < n) {
      if (fh == 0) {
        float* pp = ce.part + ((row + roff) * ncg + cg) * 2;
        pp[0] = S;
        pp[1] = 0.0f;
      }
      if (tfound) ce.true_score[row + roff] = tsc;
    }
  } else if constexpr (IS_DS) {
#pragma unroll
    for (int g = 0; g < 4; ++g) ds_write(ntl - 1, a0, 0, g);
#pragma unroll
    for (int g = 0; g < 4; ++g) ds_write(ntl - 1, a1, 1, g);
    ep_fence();
    f32x4 cv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cv[i] = ep_read(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) ds_store(ntl - 1, i, cv[i]);
  } else {
    store_half_now(ntl - 1, 0, a0);
    ep_fence();
    store_half_now(ntl - 1, 1, a1);
  }
}

static inline bool v3_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool pairs_bf16_v3_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                             const Operand& TG) {
  if (dtype != KGE_BF16) return false;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return false;
  if (d != 128 && d != 256 && d != 512) return false;
  if (!v3_al16(A.base) || !v3_al16(R.base) || !v3_al16(TG.base)) return false;
  if ((A.ld % 8) || (R.ld % 8) || (TG.ld % 8)) return false;
  return true;
}

// epoch of the cooperative launches of this process: unique per launch, so flags left in a
// workspace by earlier launches (or arbitrary initial contents) never match
static std::atomic<unsigned long long> g_v3_epoch{0};

static int v3_cu_count() {
  static int cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
  }();
  return cus;
}

long long pairs_bf16_v3_workspace_bytes(int d, long long n) {
  // [control block: the builders' flag lines (512 x 64 B) + the "degraded" word of the v4 hand-off]
  // [query fragments of whole 128-row groups, both sides of a score_sp_po call].
  // The control block comes FIRST, at an offset that does not depend on n: calls with different row counts
  // share one workspace (per stream), and behind the fragments the flag lines and the degraded word of a
  // small call lay inside the fragment area of a larger one -- whose fragment data then read as "degraded"
  // to every later small call (own query build in every workgroup: 17 -> 24 us at C2, found in the bench
  // line's one-sided figure).
  const long long rgn = 2 * ((n + V3_ROWS - 1) / V3_ROWS);
  return PAIRS_WS_CTRL_BYTES + rgn * V3_ROWS * (long long)d * 2;
}

// one workgroup per CU (256 CUs): the target tiles are split into `ncg` column groups of `tpc` tiles
static void v3_geometry(long long n, long long m, int& rgn, int& ntiles, int& ncg, int& tpc) {
  rgn = (int)((n + V3_ROWS - 1) / V3_ROWS);
  ntiles = (int)((m + V3_TN - 1) / V3_TN);
  ncg = 256 / rgn;
  if (ncg < 1) ncg = 1;
  tpc = (ntiles + ncg - 1) / ncg;
  if (tpc < 1) tpc = 1;
  ncg = (ntiles + tpc - 1) / tpc;
}

int pairs_bf16_v3_column_groups(long long n, long long m) {
  int rgn, ntiles, ncg, tpc;
  v3_geometry(n, m, rgn, ntiles, ncg, tpc);
  return ncg;
}

template <int SCORER, int HH, int EPI = V3_STORE>
static int launch_v3(const Operand& A, const Operand& R, const Operand& TG, int dir, long long n,
                     long long m, float* out, long long ldo, hipStream_t st,
                     unsigned long long* dbg, void* ws, long long ws_bytes, const CeArgs& ce_in = CeArgs{}) {
  int rgn, ntiles, ncg, tpc;
  // two-sided fused-loss launch (ce_in.rgn1 != 0): both sides padded to whole row groups
  const bool two = EPI != V3_STORE && ce_in.rgn1 != 0;
  v3_geometry(two ? 2 * ((n + V3_ROWS - 1) / V3_ROWS) * V3_ROWS : n, m, rgn, ntiles, ncg, tpc);
  CeArgs ce = ce_in;
  ce.rgn1 = two ? rgn / 2 : 0;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  const int tgmode = TG.idx.ptr == nullptr ? 0 : (TG.idx.itype ? 2 : 1);
  // cooperative query build: needs the workspace, more than one column group, every workgroup
  // resident at once (spin-wait) and a fresh epoch per launch (so: not under graph capture,
  // where the kernel arguments are frozen)
  const long long qf_bytes = (long long)rgn * V3_ROWS * HH * 4;
  bool coop = ws != nullptr && v3_al16(ws) && ncg > 1 && rgn <= 256 &&
              ws_bytes >= qf_bytes + PAIRS_WS_CTRL_BYTES && grid <= v3_cu_count();
  if (coop) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) coop = false;
  }
  // control block FIRST (see pairs_bf16_v3_workspace_bytes), fragments behind it
  unsigned long long* flags = (unsigned long long*)ws;
  u32x4* qf = (u32x4*)((char*)ws + PAIRS_WS_CTRL_BYTES);
  unsigned long long epoch = 0;
  int nbuild = 0;
  if (coop) {
    static const unsigned long long seed =
        (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() << 20;
    epoch = seed + ++g_v3_epoch;
    nbuild = ncg < 16 ? ncg : 16;
    const int items = V3_ROWS * (HH / 8);  // at least one item per builder thread
    while (nbuild > 1 && nbuild * 256 > items) --nbuild;
    // KGE_V4_OWN_BUILD=1 (tests): nobody builds for anybody -- every wave takes the path it otherwise only takes
    // after a time-out or on a degraded workspace
    if (sw(SW_V4_OWN_BUILD) == 1) nbuild = 0;
  }
#define KGE_V3L(MODE)                                                                                 \
  if (coop)                                                                                           \
    hipLaunchKernelGGL((pairs_bf16_v3_kernel<SCORER, HH, MODE, true, EPI>), dim3(grid), dim3(256), 0,  \
                       st, A, R, TG, dir, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg, qf, flags,      \
                       epoch, nbuild, ce);                                                            \
  else                                                                                                \
    hipLaunchKernelGGL((pairs_bf16_v3_kernel<SCORER, HH, MODE, false, EPI>), dim3(grid), dim3(256), 0, \
                       st, A, R, TG, dir, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg, qf, flags,      \
                       epoch, nbuild, ce)
  if constexpr (EPI != V3_STORE) {  // the fused-loss launches score against all entities only
    if (tgmode != 0) return KGE_ERR_UNSUPPORTED;
    KGE_V3L(0);
  } else {
    if (tgmode == 0) { KGE_V3L(0); }
    else if (tgmode == 1) { KGE_V3L(1); }
    else { KGE_V3L(2); }
  }
#undef KGE_V3L
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Fused 1vsAll loss launches (ce_loss.hip): the same kernel with the V3_LSE / V3_DS epilogue.
// `ws` = fragment + flag scratch of pairs_bf16_v3_workspace_bytes (cooperative query build), or NULL.
int run_pairs_bf16_v3_ce(int scorer, int epi, const Operand& A, const Operand& R, const Operand& TG, int dir,
                         int d, long long n, long long m, hipStream_t st, void* ws, long long ws_bytes,
                         const CeArgs& ce, unsigned long long* dbg) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V3C(SC, EP)                                                                                       \
  switch (d) {                                                                                                \
    case 128: return launch_v3<SC, 64, EP>(A, R, TG, dir, n, m, nullptr, 0, st, dbg, ws, ws_bytes, ce);        \
    case 256: return launch_v3<SC, 128, EP>(A, R, TG, dir, n, m, nullptr, 0, st, dbg, ws, ws_bytes, ce);       \
    case 512: return launch_v3<SC, 256, EP>(A, R, TG, dir, n, m, nullptr, 0, st, dbg, ws, ws_bytes, ce);       \
  }
  if (scorer == KGE_COMPLEX && epi == V3_LSE) { KGE_V3C(KGE_COMPLEX, V3_LSE) }
  else if (scorer == KGE_COMPLEX && epi == V3_DS) { KGE_V3C(KGE_COMPLEX, V3_DS) }
  else if (scorer == KGE_COMPLEX && epi == V3_SPLUS) { KGE_V3C(KGE_COMPLEX, V3_SPLUS) }
  else if (scorer == KGE_COMPLEX && epi == V3_DSIG) { KGE_V3C(KGE_COMPLEX, V3_DSIG) }
  else if (scorer == KGE_DISTMULT && epi == V3_LSE) { KGE_V3C(KGE_DISTMULT, V3_LSE) }
  else if (scorer == KGE_DISTMULT && epi == V3_DS) { KGE_V3C(KGE_DISTMULT, V3_DS) }
  else if (scorer == KGE_DISTMULT && epi == V3_SPLUS) { KGE_V3C(KGE_DISTMULT, V3_SPLUS) }
  else if (scorer == KGE_DISTMULT && epi == V3_DSIG) { KGE_V3C(KGE_DISTMULT, V3_DSIG) }
#undef KGE_V3C
  return KGE_ERR_UNSUPPORTED;
}

int run_pairs_bf16_v3(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir,
                      int d, long long n, long long m, float* out, long long ldo, hipStream_t st,
                      unsigned long long* dbg, void* ws, long long ws_bytes) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V3(SC)                                                                              \
  switch (d) {                                                                                  \
    case 128: return launch_v3<SC, 64>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes);    \
    case 256: return launch_v3<SC, 128>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes);   \
    case 512: return launch_v3<SC, 256>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes);   \
  }
  if (scorer == KGE_COMPLEX) { KGE_V3(KGE_COMPLEX) } else { KGE_V3(KGE_DISTMULT) }
#undef KGE_V3
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
