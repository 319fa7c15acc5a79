// score_pairs_bf16_v4.hip -- row-persistent ComplEx / DistMult sp_/_po kernel for bf16 tables,
// d in {256, 512}, workspace given: the BASELINE.json headline path on gfx950.
//
// One workgroup per CU = 128 query rows x a contiguous range of 64-target tiles, 8 waves in two
// roles (one of each per SIMD):
//
//   * consumer waves 0-3 (32 query rows each): hold their query fragments in MFMA operand
//     registers for the whole kernel and do nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16
//     (64 per tile, two accumulators); the finished accumulators go to an LDS staging buffer
//     during the first MFMAs of the next tile;
//   * loader waves: 4 and 5 stream the target tiles HBM -> LDS with LDS-DMA (ring of two 64 KiB
//     buffers, a whole tile time ahead; each the pieces of two quarters of a tile), 6 and 7 move
//     the staged scores LDS -> registers -> HBM with 16-byte stores (4 rows x 256 contiguous bytes
//     per instruction; each the blocks of two consumers).  One kind of traffic per wave: the
//     score stores of tile t-2 are issued while the other two waves issue the DMA of tile t+1
//     (with all four waves doing both in turn the tile period was 3.8 k cycles, the consumers
//     waiting ~1 k of it at B1; now the 2.7 k-cycle MFMA chain sets it).
//
// Why two roles: a vector-memory instruction blocks its wave until the texture addresser takes
// it (64 B/clk per CU: a tile is 1,024 cycles of loads + 512 of stores against 2,100 cycles of
// MFMA), and with one wave per SIMD that wave's MFMA pipe drains meanwhile (v3: 3,950 cycles
// per tile).  Here the waves that block have nothing else to do.
//
// Query vectors: built ONCE per row group, cooperatively (first `nbuild` workgroups of the row
// group build a share each, publish through the workspace with agent-scope write-through
// stores + a per-workgroup flag = this launch's epoch; everybody polls the flags and loads its
// fragments).  The launcher asks for one workgroup per CU (grid <= number of CUs), so builders and
// consumers normally run side by side; a plain launch does not GUARANTEE that (other streams'
// kernels, a second process, CU masking), so the wait is bounded and a consumer whose builders
// do not show up builds its own fragments instead (same bits) -- never a hang, never a trap.
//
// Synchronisation per tile: two workgroup barriers.  B1(t): tile t has landed (the DMA waves
// waited for their pieces) / everybody is done with tile t-1's buffer and the staging buffer has
// been drained into the store waves' registers.  B2(t), three quarters into the MFMA chain: the
// scores of tile t-1 are in staging; the store waves read them and issue the global stores after
// B1(t+1).
#include "common.hpp"
#include "bf16_queries.hpp"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <type_traits>

namespace kge {

constexpr int V4_ROWS = 128, V4_TN = 64;
typedef float f32x4v4u __attribute__((ext_vector_type(4), aligned(4)));

// EPI (common.hpp): V3_STORE writes the score tiles; V3_LSE folds them into the per-row running
// (max, sum exp) of the 1vsAll cross entropy and picks out the label's score instead -- the consumer
// waves do that on the accumulators right after a tile's MFMA chain, the DMA waves keep streaming,
// the store waves have nothing to do (kge_ce_fwd / kge_ce_sp_po_fwd: the [n, E] matrix is never
// written; per row and column group 8 bytes leave the kernel, merged by ce_combine_kernel).
constexpr int V4_DEGRADED_LAUNCHES = 4096;  // launches a timed-out hand-off is skipped for before it is tried again

// nbuild < 0: PREPARED queries -- qf already holds this launch's fragments (no builders, no flags, no polling, no
// co-residency requirement).  nx.qf != NULL: idle workgroups of the launch (NextQ::mode) build the NEXT batch's
// fragments into nx.qf instead of scoring (they take the compute units the launch geometry leaves idle).
template <int SCORER, int HH, int TGMODE, int EPI, int SPLIT = 0>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v4_kernel(
    Operand A, Operand A2, Operand R, Operand TG, int dir, long long n, long long m, int rgn,
    int rgn1, long long out2_off, int ncg, int tiles_per_cg, int ntiles, float* __restrict__ out,
    long long ldo, unsigned long long* __restrict__ dbg, u32x4* __restrict__ qf,
    unsigned long long* __restrict__ flags, unsigned long long epoch, int nbuild, CeArgs ce, NextQ nx,
    int st_sc1) {
  static_assert(!SPLIT || EPI == V3_STORE, "split queries: score store path only");
  constexpr int RGR = SPLIT ? 64 : V4_ROWS;  // real query rows per row group
  constexpr int NKB = 2 * HH / 16;       // K-blocks of 16
  constexpr int NKH = HH / 16;           // K-blocks per half
  constexpr int ROWB = 4 * HH;           // bytes per table row (2*HH bf16)
  constexpr int SPR = HH / 4;            // 16-byte slots per row
  constexpr int TILEB = V4_TN * ROWB;    // bytes per target tile
  constexpr int NL = TILEB / 1024 / 4;   // 1-KiB DMA pieces per loader wave per tile
  constexpr int RPP = 64 / SPR;          // target rows per piece
  constexpr bool IS_DS = EPI == V3_DS || EPI == V3_DSIG;  // writes bf16 gradients of the scores (G16)
  constexpr bool STAGED = EPI == V3_STORE || IS_DS;       // tiles go through the staging buffer to the store waves
  // target tiles in flight: the DMA of a tile is a full L2 round trip (~3 k cycles under load) -- with two buffers
  // (d = 512: 2 x 64 KiB + staging = the whole LDS) one tile streams in while one is consumed and the loop period
  // cannot drop below that round trip; d = 256 has room for four 32 KiB buffers: three tiles in flight
  constexpr int NBUF = HH == 128 ? 4 : 2;
  constexpr int CST0 = NBUF * TILEB;     // score staging: 4 x [32 rows][64 cols] f32
  constexpr int CSTW = 32 * V4_TN * 4;
  constexpr int SMEM = CST0 + 4 * CSTW;
  constexpr int NQ = 2 * NKB;            // MFMAs per tile (two 32-target halves)
  constexpr int QB2 = 3 * NQ / 4;         // MFMA slot of barrier B2
  static_assert(NL * RPP == 16 && SPR >= 32, "d in {256, 512}");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  const bool prebuilt = nbuild < 0;  // (uniform)

  // ---- which rows / target tiles
  const int b = blockIdx.x;
  const int q8 = b >> 3;
  const int rg = q8 % rgn;
  const int cg = (q8 / rgn) * 8 + (b & 7);
  if (cg >= ncg) {
    if (nx.qf != nullptr && nx.mode == 1) {  // an idle workgroup: the next batch's query fragments
      const int spg = ((ncg + 7) & ~7) - ncg;  // idle column-group slots per row group
      v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)(rg * spg + (cg - ncg)) * 512 + threadIdx.x,
                                          (long long)nx.nblocks * 512);
    }
    return;
  }
  // Which 64-target tiles: tiles_per_cg > 0: the contiguous range [cg * tiles_per_cg, ...);
  // tiles_per_cg == 0: every ncg-th tile (cg, cg + ncg, ...) -- the workgroups then write
  // NEIGHBOURING 256-byte segments of the same score rows at about the same time, which is what the
  // memory controllers need once the score matrix no longer fits the Infinity Cache (a 574,311-column
  // shard: 2.35 GB per call; with contiguous ranges 32 k write streams 0.5 MB apart thrash the DRAM pages)
  const int tile_lo = tiles_per_cg > 0 ? cg * tiles_per_cg : cg;
  const int tile_st = tiles_per_cg > 0 ? 1 : ncg;
  int ntl = tiles_per_cg > 0 ? ntiles - tile_lo : (ntiles - cg + ncg - 1) / ncg;
  if (tiles_per_cg > 0 && ntl > tiles_per_cg) ntl = tiles_per_cg;
  if (ntl <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
  const int w4 = wave & 3;  // consumer w4 works on query rows 32*w4 .. 32*w4+31 of the row group
  // Two-sided launch (score_sp_po, EntityRankingJob's call): row groups [0, rgn1) are the n
  // (s, p, ?) queries, row groups [rgn1, rgn) the n (?, p, o) queries, scored into the column
  // block behind the first one.  One-sided: rgn1 == rgn.
  const bool second = rg >= rgn1;
  const int rgl = second ? rg - rgn1 : rg;
  if (second) {
    A = A2;
    dir = KGE_PO_;
    out += out2_off;
  }

  int dbg_i = 0;
  auto stamp = [&]() {  // optional per-phase timestamps (tools/v2_phases.py); dbg == NULL in production
    if (dbg != nullptr && tid == 0 && dbg_i < 64)
      dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  stamp();  // 0: kernel start
  auto stamp_at = [&](int slot) {  // loader-side stamps (slots 32..): lane 0 of the calling wave
    if (dbg != nullptr && lane == 0) dbg[(long long)blockIdx.x * 64 + slot] = __builtin_readcyclecounter();
  };

  // ---- in-launch query build: one item = 8 coordinates of both halves of one query row, written fragment-major
  // (K-block kb of 32-row block rb is 64 lanes x 16 B, contiguous) with agent-scope (sc1) write-through stores:
  // visible to the other XCDs' L2s once acknowledged, without the whole-L2 write-back of a release fence.  A
  // 128-byte line (8 rows x 16 B) is written by one workgroup only (or by several with the same bytes).
  auto build_item = [&](int rr, int c8) __attribute__((always_inline)) {
    const long long row = (long long)rg * V4_ROWS + rr;    // row of the fragment workspace
    const long long lrow = (long long)rgl * V4_ROWS + rr;  // query row within its side
    const long long qrow = lrow < n ? lrow : n - 1;        // padded rows repeat row n-1
    const unsigned short* a = (const unsigned short*)A.base + index_at(A.idx, qrow) * A.ld + c8 * 8;
    const unsigned short* r = (const unsigned short*)R.base + index_at(R.idx, qrow) * R.ld + c8 * 8;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(a), a1 = *reinterpret_cast<const u32x4*>(a + HH);
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(r), r1 = *reinterpret_cast<const u32x4*>(r + HH);
    u32x4 q0, q1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned int x0, x1;
      bf16_qpair_fast<SCORER>(dir, a0[e], a1[e], r0[e], r1[e], x0, x1);
      q0[e] = x0;
      q1[e] = x1;
    }
    u32x4* dst = qf + ((row >> 5) * NKB) * 64 + (row & 31) + 32 * (c8 & 1);
    // s_nop 1: the two wait states a >64-bit VMEM store needs before its data registers may be
    // overwritten -- the hazard recogniser cannot see into inline asm (found the hard way: an
    // unrolled variant of this loop reused q0's registers for the next address and stored garbage)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (c8 >> 1) * 64), "v"(q0) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (NKH + (c8 >> 1)) * 64), "v"(q1)
                 : "memory");
  };
  // cooperative: this workgroup's share of the row group's rows
  if (cg < nbuild) {
    constexpr int CGR = HH / 8;  // groups of 8 coordinates per row
    for (int it = cg * 512 + tid; it < V4_ROWS * CGR; it += nbuild * 512) build_item(it / CGR, it % CGR);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every thread: its stores are acknowledged ...
    __syncthreads();                                  // ... before wave 0 publishes
    // one flag per (consumer workgroup, builder): every consumer polls a 64-byte line of its own
    // (64 pollers on one line serialise at its memory channel)
    if (wave == 0)
      for (int c = lane; c < ncg; c += 64)
        __hip_atomic_store(flags + ((long long)rg * ncg + c) * 8 + cg, epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  }
  // The degraded word is a COUNTDOWN of launches, not a latch: one thread per launch takes one off, so that a
  // workspace degraded by a single hiccup (a builder held up for > ~60 ms once) goes back to the cooperative
  // build after V4_DEGRADED_LAUNCHES launches instead of paying the own-build path for the life of the process.
  if (!prebuilt && blockIdx.x == 0 && tid == 0) {
    unsigned long long* const dgw = flags + 512 * 8;
    const unsigned long long v = __hip_atomic_load(dgw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != 0ull) __hip_atomic_store(dgw, v - 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamp();  // 1: share built and published

  if (wave >= 4) {
    // ============ loader waves, split duties: waves 4, 5 stream the target tiles in (each the
    // pieces of two quarters), waves 6, 7 move the staged scores out (each two consumers' blocks).
    // A VMEM instruction blocks its wave until the memory pipeline takes it, and the stores are
    // taken at HBM write speed: with one kind of traffic per wave the score stores of tile t-2 are
    // issued WHILE the other waves issue the DMA of tile t+1, instead of one after the other.
    const unsigned short* tgb = (const unsigned short*)TG.base;
    const long long tld2 = TG.ld * 2;
    const int nfull = (int)(m / V4_TN);
    const int lr = lane / SPR, slot = lane % SPR;
    const int j2 = (wave & 1) * 2;  // this wave's quarters: j2, j2 + 1
    // (the tile DMA is shared by both kinds of loader waves during the ring fill, see below)
    unsigned int dvoff[NL], dsw[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        dsw[k] = (unsigned int)(((slot ^ lr) << 4) ^ ((RPP * k) << 4));
        dvoff[k] = (unsigned int)(lr * (int)tld2) + dsw[k];
      }
      auto load_rows = [&](int tt, int w) -> long long {
        const int tc = tt < ntl ? tt : ntl - 1;
        long long tr = (long long)(tile_lo + tc * tile_st) * V4_TN + w * 16 + (lane & 15);
        if (tr >= m) tr = m - 1;
        return index_mode<TGMODE>(TG.idx, tr);
      };
      auto bcast_row = [&](long long rows, int l) -> long long {
        const int lo = __builtin_amdgcn_readlane((int)(rows & 0xffffffffLL), l);
        const int hi = __builtin_amdgcn_readlane((int)(rows >> 32), l);
        return ((long long)hi << 32) | (unsigned int)lo;
      };
      auto tile_dma = [&](int tt, long long rows, int w) {
        const int tc = tt < ntl ? tt : ntl - 1;
        const long long trow0 = (long long)(tile_lo + tc * tile_st) * V4_TN;
        unsigned int d = (unsigned int)((tt % NBUF) * TILEB + w * NL * 1024);
        if (TGMODE == 0 && tile_lo + tc * tile_st < nfull && tld2 < (1LL << 28)) {
          const unsigned char* p = (const unsigned char*)tgb + (trow0 + w * 16) * tld2;
          v4_static_for<0, NL>([&](auto kc) __attribute__((always_inline)) {
            const unsigned int vo = dvoff[decltype(kc)::value];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(d), "v"(vo), "s"(p)
                         : "memory", "m0");
            p += RPP * tld2;
            d += 1024;
          });
        } else {
#pragma unroll
          for (int k = 0; k < NL; ++k) {
            long long r = bcast_row(rows, RPP * k);
            if (RPP == 2) {
              const long long r1 = bcast_row(rows, RPP * k + 1);
              r = lr ? r1 : r;
            }
            const unsigned char* src = (const unsigned char*)tgb + r * tld2 + dsw[k];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + d + k * 1024),
                                             16, 0, 0);
          }
        }
      };
    // ---- score path of the loader waves: staging buffer -> registers -> HBM.  Waves 6, 7 carry it tile by tile;
    // waves 4, 5 (DMA) take over the LAST tile's stores, when they have nothing left to stream (see the end of
    // their loop): each of 4 / 6 handles the blocks of consumers 0, 1, each of 5 / 7 those of consumers 2, 3.
    const int cl = lane & 15, rq = lane >> 4;
    const int z = cl ^ rq;
    // SPLIT: this wave owns ONE real 32-row block (wave & 1); its q_hi partial scores are consumer (wave & 1)'s
    // staging block, the q_lo partials consumer (wave & 1) + 2's: read both, store their sum (NU = 1 output block)
    constexpr int NU = SPLIT ? 1 : 2;
    unsigned int crd[2], svoff[2][8];
    unsigned char* out_rb[2];
    if constexpr (SPLIT) crd[1] = (unsigned int)(CST0 + ((wave & 1) + 2) * CSTW + rq * 256);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int w = SPLIT ? (wave & 1) : j2 + u;
      const long long r0 = (long long)rgl * RGR + 32 * w;
      const long long rb = r0 < n ? r0 : n - 1;
      crd[u] = (unsigned int)(CST0 + w * CSTW + rq * 256);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        long long r = r0 + 4 * i + rq;
        if (r >= n) r = n - 1;
        svoff[u][i] = (unsigned int)((r - rb) * ldo * 4) + (unsigned int)(cl * 16);
      }
      if constexpr (IS_DS) {
        // G16: [rows][ld16] bf16, the second side's rows behind ce.side2_off; 8 bytes (4 columns) per lane
        const long long roff_s = second ? ce.side2_off : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          long long r = r0 + 4 * i + rq;
          if (r >= n) r = n - 1;
          svoff[u][i] = (unsigned int)((r - rb) * ce.ld16 * 2) + (unsigned int)(cl * 8);
        }
        out_rb[u] = (unsigned char*)(ce.g16 + (rb + roff_s) * ce.ld16);
      } else {
        out_rb[u] = (unsigned char*)(out + rb * ldo);
      }
    }
    f32x4 cv[2][8];
    auto read_staging = [&]() {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          cv[u][i] = *reinterpret_cast<const f32x4*>(smem + crd[u] + i * 1024 + ((z ^ ((4 * i) & 15)) << 4));
    };
    auto store_tile = [&](int tt) {
      const long long col0 = (long long)(tile_lo + tt * tile_st) * V4_TN;
      if constexpr (IS_DS) {  // the pitch of G16 covers whole tiles (pad columns are zeros): no ragged path
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            u32x2 pk = {bf16_pack(cv[u][i][0], cv[u][i][1]), bf16_pack(cv[u][i][2], cv[u][i][3])};
            *reinterpret_cast<u32x2*>(out_rb[u] + col0 * 2 + svoff[u][i]) = pk;
          }
        return;
      }
      if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cv[0][i] = cv[0][i] + cv[1][i];  // score = (sum q_hi t) + (sum q_lo t)
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (col0 + V4_TN <= m) {
          unsigned char* sbase = out_rb[u] + col0 * 4;
          if (st_sc1) {
            // agent-scope write-through stores: the scores leave the L2 as they are written instead of sitting
            // there dirty until the end-of-kernel write-back (a 30 MB score block fits the 32 MB of L2)
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, 0x7fffffff, 0x00020000);
#define KGE_V4_ST(AUX)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cv[u][i]), srs, svoff[u][i], 0, AUX)
            if (st_sc1 == 1) { KGE_V4_ST(16); }        // sc1
            else if (st_sc1 == 2) { KGE_V4_ST(17); }   // sc0 sc1
            else if (st_sc1 == 3) { KGE_V4_ST(18); }   // sc1 nt
            else { KGE_V4_ST(2); }                     // nt
#undef KGE_V4_ST
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4v4u*>(sbase + svoff[u][i]) = cv[u][i];
          }
        } else {  // ragged end of the table (always this workgroup's last tile)
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col0 + 4 * cl + e < m)
                *reinterpret_cast<float*>(out_rb[u] + (col0 + e) * 4 + svoff[u][i]) = cv[u][i][e];
        }
      }
    };
    // ---- ring fill.  A DMA piece costs its wave 60-180 cycles to issue (more while the consumers'
    // fragment loads are in the same queue), nothing can be scored before tile 0 has landed, and with prepared
    // queries the fill IS the start-up of the launch.  So: (i) only the first 64 KiB (FUP tiles) go out before the
    // loop, the rest of the ring right behind B1(0) -- a whole MFMA chain ahead of its use --; (ii) tiles 0 and 1 are
    // shared by all four loader waves (a DMA wave takes quarter j2, its store-wave partner, wave + 2, quarter
    // j2 + 1: 16 instead of 32 pieces per wave in front of tile 0).  Measured (profiles/r3b): two waves issuing two
    // tiles released tile 0 at 6.2 k cycles.
    constexpr int FUP = NBUF / 2;
    if (wave < 6) {
      // ------------------------------- DMA waves -------------------------------
      long long ra[NBUF], rb[NBUF];
#pragma unroll
      for (int k = 0; k < NBUF; ++k) {
        ra[k] = load_rows(k, j2);
        rb[k] = k >= 2 ? load_rows(k, j2 + 1) : 0;
      }
#pragma unroll
      for (int k = 0; k < FUP; ++k)
        if (k < ntl) tile_dma(k, ra[k], j2);
      long long rna = load_rows(NBUF, j2), rnb = load_rows(NBUF, j2 + 1);
      if (wave == 4) stamp_at(32);  // first tiles issued
      if (!prebuilt) KGE_BARRIER();  // B0
      for (int tt = 0; tt <= ntl; ++tt) {
        // VMEM queue of this wave: tile pieces, in order (index loads of the gathered-target modes only make a wait
        // longer): NL per tile for tiles 0 and 1 (the partner issued the other quarter), 2 NL from tile 2 on.  Tile
        // tt has landed once at most the pieces of the tiles issued behind it are outstanding: at tt = 0 the rest of
        // the first FUP tiles, later min(ntl - tt - 1, NBUF - 2) tiles of 2 NL.
        const int behind = tt == 0 ? 0 : (ntl - tt - 1 < NBUF - 2 ? ntl - tt - 1 : NBUF - 2);
        bool waited = false;
        if (tt == 0 && FUP > 1 && ntl > 1) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
          waited = true;
        }
        if constexpr (NBUF > 2) {  // (6 NL = 48 <= the counter's 63 at d = 256)
          if (!waited && behind >= 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(6 * NL) : "memory");
            waited = true;
          } else if (!waited && behind == 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * NL) : "memory");
            waited = true;
          }
        }
        if (!waited) {
          if (behind >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NL) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tt == 0 && wave == 4) stamp_at(37);  // this wave's pieces of tile 0 have landed
        KGE_BARRIER();  // B1(tt): tile tt landed; the buffer of tile tt - 1 is free
        if (tt == 0) {  // the rest of the ring
#pragma unroll
          for (int k = FUP; k < NBUF; ++k) {
            if (k < ntl) {
              tile_dma(k, ra[k], j2);
              if (k >= 2) tile_dma(k, rb[k], j2 + 1);
            }
          }
        } else if (tt - 1 + NBUF < ntl) {
          tile_dma(tt - 1 + NBUF, rna, j2);
          tile_dma(tt - 1 + NBUF, rnb, j2 + 1);
          rna = load_rows(tt + NBUF, j2);
          rnb = load_rows(tt + NBUF, j2 + 1);
        }
        KGE_BARRIER();  // B2(tt)
      }
      // B2(ntl) is behind: the last tile's scores are staged and this wave has nothing left to stream -- it stores
      // them, while the store waves are still issuing the stores of the tile before (4.8 k -> 3.4 k cycles of tail)
      if constexpr (STAGED) {
        read_staging();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_tile(ntl - 1);
      }
      if (wave == 4) stamp_at(36);  // last tile's stores issued
      return;
    }
    // ------------------------------- store waves -------------------------------
    {  // ring fill, this wave's share: quarter j2 + 1 of the tiles < 2 among the first FUP
      const long long h0 = load_rows(0, j2 + 1);
      tile_dma(0, h0, j2 + 1);
      if (FUP > 1 && ntl > 1) tile_dma(1, load_rows(1, j2 + 1), j2 + 1);
    }
    const long long h1 = load_rows(1, j2 + 1);
    if (wave == 6) stamp_at(38);  // its share of the first tiles issued
    if (!prebuilt) KGE_BARRIER();  // B0
    for (int tt = 0; tt <= ntl; ++tt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging reads of the previous step are in registers
      // this wave's pieces of tiles 0 / 1 have landed (its first score stores are issued behind B1(2))
      if (tt == 0) {
        if (FUP > 1 && ntl > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (tt == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (tt == 0 && wave == 6) stamp_at(39);  // landed
      KGE_BARRIER();  // B1(tt): the consumers may overwrite the staging buffer
      if (tt == 0 && FUP == 1 && ntl > 1) tile_dma(1, h1, j2 + 1);  // its quarter of tile 1, with the rest of the ring
      if (tt == ntl) {
        // last round: nothing is staged behind this barrier pair for THIS wave (the DMA waves store the last tile)
        KGE_BARRIER();  // B2(ntl)
        if constexpr (STAGED)
          if (tt >= 2) store_tile(tt - 2);
        break;
      }
      if constexpr (STAGED)
        if (tt >= 2) store_tile(tt - 2);  // from registers, while the DMA waves issue tile tt+1
      KGE_BARRIER();  // B2(tt): scores of tile tt-1 are staged
      if constexpr (STAGED)
        if (tt >= 1) read_staging();
    }
    if (wave == 6) stamp_at(34);  // last store issued
    if (wave == 6 && dbg != nullptr) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp_at(35);  // last store acknowledged
    }
    return;
  }

  // =================================== consumer waves ===================================
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 afr[NKB];
  // A consumer never trusts a builder blindly.  The wait is bounded (a builder workgroup that is not
  // running -- its CU busy with another stream's kernel, CU masking, a second process -- must neither
  // hang this workgroup nor kill the process): on a time-out this workgroup builds its own fragments in
  // registers (same bits) and marks the workspace DEGRADED; the next V4_DEGRADED_LAUNCHES launches on that workspace
  // skip the hand-off at once (a flag published late cannot be told from a fresh one by a replay of a
  // captured launch, whose epoch is frozen).  KGE_V4_OWN_BUILD=1 (tests) takes that path on purpose.
  unsigned long long* const degraded = flags + 512 * 8;
  int* const sb_flag = reinterpret_cast<int*>(smem + CST0);  // the staging buffer is idle until tile 0 is done
  // The builders stored with sc1 (write-through); sc1 loads (served by L2, never by this CU's
  // L1) complete the hand-off without an acquire fence.  The 32 loads are NOT waited for here:
  // the MFMA chain of the first tile waits for fragment kb right before it needs it.
  // Compiler-visible buffer loads (not inline asm): the register allocator then knows the values
  // arrive asynchronously and places the vmcnt waits itself, in front of the first MFMA of tile 0
  // that needs each fragment (asm loads + asm waits are safe only while nothing
  // is moved between them).  Prepared queries (a previous launch wrote them) take the same loads.
  // In two parts: the first FR0 K-blocks at once, the rest behind B1(0) -- with prepared queries every byte that is
  // in the vector-memory queue in front of tile 0's last piece delays the first MFMA (128 KiB of fragments + 64 KiB
  // of tile through a 64 B/clk path: tile 0 released at 5.7 k cycles with all 32 loads up front, profiles/r3c);
  // FR0 K-blocks cover the first 2 FR0 MFMA slots, behind which the rest arrives.
  constexpr int FR0 = 8;
  const unsigned char* const frag_base =
      (const unsigned char*)(qf + ((long long)(rg * (V4_ROWS / 32) + w4) * NKB) * 64);  // uniform
  auto load_fragments = [&](auto lo, auto hi) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)frag_base, 0, NKB * 1024, 0x00020000);
    v4_static_for<decltype(lo)::value, decltype(hi)::value>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      afr[kb] = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, (unsigned int)(lane * 16 + kb * 1024), 0, 16 /* sc1 */));
    });
  };
  using FrLo = std::integral_constant<int, 0>;
  using FrMid = std::integral_constant<int, FR0>;
  if (prebuilt) {
    // prepared queries: the fragment loads go out at once, next to the DMA of tiles 0 and 1
    load_fragments(FrLo{}, FrMid{});  // (no B0: nobody publishes anything)
    stamp();  // 2
  } else {
    if (wave == 0) {
      // ONE wave per workgroup polls (255 pollers already cost chip bandwidth): one flag per builder
      unsigned long long* f = flags + ((long long)rg * ncg + cg) * 8;
      bool ok = nbuild > 0 && __hip_atomic_load(degraded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
      for (int spin = 0; ok; ++spin) {
        const unsigned long long v =
            lane < nbuild ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all(v == epoch)) break;
        if (spin == (1 << 16)) {  // ~0.1 s: far beyond any launch skew
          ok = false;
          if (lane == 0 && nbuild > 0)
            __hip_atomic_store(degraded, (unsigned long long)V4_DEGRADED_LAUNCHES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_sleep(1);
      }
      // This line belongs to this workgroup alone (the builders write it, nobody else reads it):
      // clear it, so that a replay of this very launch (hipGraph: the kernel arguments, epoch
      // included, are frozen at capture) starts from "not published" again.
      if (ok && lane < nbuild) __hip_atomic_store(f + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) *sb_flag = ok ? 1 : 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS write has landed before the barrier
    }
    KGE_BARRIER();  // B0: the shares of this row group are published (or given up on)
    stamp();  // 2: all shares of this row group published
    const bool coop = *reinterpret_cast<volatile int*>(sb_flag) != 0;
    if (!coop) {
      // The builders did not show up (time-out, degraded workspace, KGE_V4_OWN_BUILD): this wave builds the
      // fragments of its own 32 rows itself -- through the workspace, with the builders' code (same bytes, so it
      // does not matter who else writes them) --, waits for the acknowledgement of its own stores and loads them
      // like everybody else.  No cross-wave dependency, hence no barrier (the loader waves are parked at B1(0)).
      // Slow (32 items per lane) and rare.  [Until round 3 this was a register-resident build: 64 live
      // registers more than the MFMA loop needs, i.e. spills in the hot path.]
      if constexpr (!SPLIT) {
        constexpr int CGR = HH / 8;
        for (int it = lane; it < 32 * CGR; it += 64) build_item(32 * w4 + it / CGR, it % CGR);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    load_fragments(FrLo{}, FrMid{});
  }
  // B fragment (K-block kb, half hf) of target row 32*hf + fi: 16-B slot s = s0(kb) + fh, stored
  // at slot s ^ (fi & 15): with s = 16*a + b the swizzle only touches b -> 8 address registers
  // plus immediates a*256 + hf*32*ROWB.
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  // staging: acc[4g + e] = score(query fi, target 32*hf + 8g + 4fh + e) -> chunk (8hf + 2g) | fh of
  // row fi, stored at chunk ^ (fi & 15) = (8hf + 2g) ^ y
  const unsigned int cwr = (unsigned int)(CST0 + w4 * CSTW + fi * 256);
  const int y = fh ^ (fi & 15);
  auto c_write = [&](const f32x16& acc, int hf, int g) {
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    *reinterpret_cast<f32x4*>(smem + cwr + (((8 * hf + 2 * g) ^ y) << 4)) = v;
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
  stamp();  // 3: fragment loads issued

  // ---- fused 1vsAll loss (EPI == V3_LSE): per-lane state of query row fi (both lanes fh = 0 / 1 of
  // a row hold it; padded rows repeat row n-1 like their query fragments do).  Accumulator element r
  // of half hf is column  col0(tile) + 32 hf + 8 (r >> 2) + 4 fh + (r & 3).
  long long lab = -1;              // label column of this lane's row
  float rmax = -__builtin_inff();  // running max over this lane's columns so far
  float rsum = 0.0f;               // running sum of exp(score - rmax)
  float tsc = 0.0f;                // score(row, label) if one of this lane's columns
  bool tfound = false;
  const long long roff = (EPI != V3_STORE && second) ? ce.side2_off : 0;
  if constexpr (EPI == V3_LSE) {
    const Index& lix = second ? ce.label2 : ce.label;
    long long orow = (long long)rgl * V4_ROWS + 32 * w4 + fi;
    if (orow >= n) orow = n - 1;
    if (lix.ptr != nullptr) lab = index_at(lix, orow);
  }
  // ---- fused rank counts (EPI == V3_RANK): the lane's row against its true score rk_t, and per filter set the
  // counts over the columns whose bit is set (a filtered column scores -inf: rank.hip's sparse correction)
  const int rk_side = second ? 1 : 0;
  float rk_t = 0.0f, rk_al = 0.0f;  // true score (NaN -> -inf), allowed = atol + |rtol * t|
  bool rk_slow = false;             // some row of the wave has an infinite true score / tolerance: generic arithmetic
  int rk_g = 0, rk_c = 0;           // raw: greater-and-not-close, close
  int rk_fg[2] = {0, 0}, rk_fc[2] = {0, 0}, rk_fn[2] = {0, 0};  // per filter set: filtered & greater, & close, filtered
  unsigned long long rk_w[2] = {0, 0};  // the row's filter bits of the NEXT tile (prefetched one tile ahead)
  const unsigned int* rk_bp[2] = {nullptr, nullptr};
  auto rk_fetch = [&](int tt) __attribute__((always_inline)) {
    const long long tl = tile_lo + (long long)tt * tile_st;  // tile of 64 columns = two 32-bit words of the row
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < ce.rk_nfilt) {
        const unsigned int* wp = rk_bp[k] + 2 * tl * ce.rk_bits_us;
        rk_w[k] = tt < ntl ? (unsigned long long)wp[0] | ((unsigned long long)wp[ce.rk_bits_us] << 32) : 0ull;
      }
  };
  if constexpr (EPI == V3_RANK) {
    long long orow = (long long)rgl * V4_ROWS + 32 * w4 + fi;
    if (orow >= n) orow = n - 1;
    rk_t = ce.rk_true[rk_side][orow * ce.rk_true_stride];
    if (rk_t != rk_t) rk_t = -__builtin_inff();
    rk_al = ce.rk_atol + __builtin_fabsf(ce.rk_rtol * rk_t);
    rk_slow = __any(!(__builtin_isfinite(rk_t) && rk_al >= 0.0f && __builtin_isfinite(rk_al))) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (k < ce.rk_nfilt) rk_bp[k] = ce.rk_bits[rk_side][k] + orow * ce.rk_bits_rs;
    rk_fetch(0);
  }
  // Bit layout of the per-lane masks = the tile's 64 columns: element r of half hf is bit 32 hf + 8 (r >> 2) + 4 fh + (r & 3).
  // 16 dense bits (element r = 4 q + b at bit r) -> the tile layout before the 4 fh shift (bit 8 q + b)
  auto rk_spread = [](unsigned int d) __attribute__((always_inline)) -> unsigned int {
    return (d & 0xfu) | ((d & 0xf0u) << 4) | ((d & 0xf00u) << 8) | ((d & 0xf000u) << 12);
  };
  // rank_masks: the two result masks of one tile from its accumulators (before the 4 fh shift).
  auto rank_masks = [&](const f32x16& a0, const f32x16& a1, unsigned int (&g)[2], unsigned int (&c)[2], bool fast)
      __attribute__((always_inline)) {
    g[0] = g[1] = c[0] = c[1] = 0u;
    if (fast) {
      // finite true score, finite tolerance >= 0:  close <=> |x - t| <= allowed,  greater-and-not-close <=>
      // x - t > allowed  (NaN and -inf scores fail both, +inf is greater: what count_one gives).  As sign bits, so
      // that no comparison result travels through a scalar register: x' = max(x, -inf) (NaN -> -inf), e = x' - t,
      // sign(allowed - e) = greater, sign(allowed - |e|) = NOT close; one v_alignbit shifts each into its mask.
      unsigned int ng[2] = {0u, 0u}, nc[2] = {0u, 0u};
#pragma unroll
      for (int r = 15; r >= 0; --r) {  // element r ends up at bit r
        const float e0 = __builtin_fmaxf(a0[r], -__builtin_inff()) - rk_t;
        const float e1 = __builtin_fmaxf(a1[r], -__builtin_inff()) - rk_t;
        ng[0] = __builtin_amdgcn_alignbit(ng[0], __builtin_bit_cast(unsigned int, rk_al - e0), 31);
        nc[0] = __builtin_amdgcn_alignbit(nc[0], __builtin_bit_cast(unsigned int, rk_al - __builtin_fabsf(e0)), 31);
        ng[1] = __builtin_amdgcn_alignbit(ng[1], __builtin_bit_cast(unsigned int, rk_al - e1), 31);
        nc[1] = __builtin_amdgcn_alignbit(nc[1], __builtin_bit_cast(unsigned int, rk_al - __builtin_fabsf(e1)), 31);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        g[h] = rk_spread(ng[h]);
        c[h] = rk_spread(~nc[h] & 0xffffu);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned int bit = 1u << (8 * (r >> 2) + (r & 3));
        int g0 = 0, c0 = 0, g1 = 0, c1 = 0;
        count_one(a0[r], rk_t, ce.rk_atol, ce.rk_rtol, g0, c0);
        count_one(a1[r], rk_t, ce.rk_atol, ce.rk_rtol, g1, c1);
        g[0] |= g0 ? bit : 0u;
        c[0] |= c0 ? bit : 0u;
        g[1] |= g1 ? bit : 0u;
        c[1] |= c1 ? bit : 0u;
      }
    }
  };
  // rank_finish: the masks of tile tt into the counters (its filter words are in rk_w); fetches the next tile's words
  auto rank_finish = [&](int tt, unsigned int (&g)[2], unsigned int (&c)[2]) __attribute__((always_inline)) {
    const long long c0t = (long long)(tile_lo + tt * tile_st) * V4_TN;
    // this lane's columns of the tile that exist (the ragged last tile of the slice ends at m)
    unsigned long long mine = 0x0f0f0f0f0f0f0f0full << (4 * fh);
    const long long rem = m - c0t;
    if (rem < V4_TN) mine &= (1ull << rem) - 1ull;  // (rem >= 1: the tile exists)
    const unsigned int m0 = (unsigned int)mine, m1 = (unsigned int)(mine >> 32);
    g[0] = (g[0] << (4 * fh)) & m0;
    c[0] = (c[0] << (4 * fh)) & m0;
    g[1] = (g[1] << (4 * fh)) & m1;
    c[1] = (c[1] << (4 * fh)) & m1;
    rk_g += __builtin_popcount(g[0]) + __builtin_popcount(g[1]);
    rk_c += __builtin_popcount(c[0]) + __builtin_popcount(c[1]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < ce.rk_nfilt) {
        const unsigned int w0 = (unsigned int)rk_w[k] & m0, w1 = (unsigned int)(rk_w[k] >> 32) & m1;
        rk_fg[k] += __builtin_popcount(g[0] & w0) + __builtin_popcount(g[1] & w1);
        rk_fc[k] += __builtin_popcount(c[0] & w0) + __builtin_popcount(c[1] & w1);
        rk_fn[k] += __builtin_popcount(w0) + __builtin_popcount(w1);
      }
    }
    rk_fetch(tt + 1);
  };
  auto rank_tile = [&](int tt) __attribute__((always_inline)) {
    unsigned int g[2], c[2];
    rank_masks(acc0, acc1, g, c, !rk_slow);
    rank_finish(tt, g, c);
  };
  // d = 256 (registers to spare): the comparisons of tile tt - 1 are issued between the MFMAs of tile tt, on a copy
  // of its accumulators -- one element per MFMA slot -- instead of between two chains with the matrix pipe idle.
  // Measured: 3 % (one wave issues MFMA, fragment read and comparisons in order, so little of it
  // really overlaps).  A wave with an infinite true score (rk_slow) does the generic arithmetic on the copy after
  // the chain instead.
  constexpr bool RK_PIPE = EPI == V3_RANK && HH == 128;
  f32x16 pv0, pv1;               // RK_PIPE: accumulators of the previous tile
  unsigned int pg[2] = {0u, 0u}, pc[2] = {0u, 0u};
  if constexpr (RK_PIPE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pv0[r] = pv1[r] = 0.0f;
  }
  auto rank_step = [&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    constexpr int hf = q & 1, r = 15 - ((q >> 1) & 15);  // descending: element r ends up at bit r (see rank_masks)
    const float e = __builtin_fmaxf(hf ? pv1[r] : pv0[r], -__builtin_inff()) - rk_t;
    pg[hf] = __builtin_amdgcn_alignbit(pg[hf], __builtin_bit_cast(unsigned int, rk_al - e), 31);
    pc[hf] = __builtin_amdgcn_alignbit(pc[hf], __builtin_bit_cast(unsigned int, rk_al - __builtin_fabsf(e)), 31);
  };

  float lse_i = 0.0f, g_i = 0.0f, gb_i = 0.0f;  // V3_DS / V3_DSIG: the row's logsumexp, upstream gradient, g_i * row_bias[i]
  if constexpr (IS_DS) {
    const Index& lix = second ? ce.label2 : ce.label;
    long long orow = (long long)rgl * V4_ROWS + 32 * w4 + fi;
    if (orow >= n) orow = n - 1;
    if (lix.ptr != nullptr) lab = index_at(lix, orow);
    if constexpr (EPI == V3_DS) lse_i = ce.lse[orow + roff];
    g_i = ce_row_gradient(ce, orow + roff);
    if (ce.rowptr != nullptr && ce.rowptr[orow + 1] == ce.rowptr[orow]) g_i = 0.0f;  // (one-sided multi-label loss)
    if (EPI == V3_DS && ce.row_bias != nullptr) gb_i = g_i * ce.row_bias[orow + roff];
  }
  // d loss / d score of a finished tile, in place of the scores (same chain, same bits as the forward):
  // g_i * (softmax - [label]) or g_i * sigmoid(score + offset); the pad columns of the ragged last tile are
  // zeros (the gradient products read whole 16-byte chunks of a G16 row and rely on it).
  auto ds_tile = [&](int tt) __attribute__((always_inline)) {
    const long long c0 = (long long)(tile_lo + tt * tile_st) * V4_TN + 4 * fh;
    const long long rel = lab - c0;
    const bool ragged = c0 - 4 * fh + V4_TN > m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = 8 * (r >> 2) + (r & 3);
      float p0, p1;
      if constexpr (EPI == V3_DSIG) {
        p0 = g_i / (1.0f + __builtin_amdgcn_exp2f(-(acc0[r] + ce.offset) * V3_LOG2E));
        p1 = g_i / (1.0f + __builtin_amdgcn_exp2f(-(acc1[r] + ce.offset) * V3_LOG2E));
      } else {
        p0 = __builtin_amdgcn_exp2f((acc0[r] - lse_i) * V3_LOG2E) * g_i - gb_i;
        p1 = __builtin_amdgcn_exp2f((acc1[r] - lse_i) * V3_LOG2E) * g_i - gb_i;
      }
      if (rel == off) p0 -= g_i;
      if (rel == 32 + off) p1 -= g_i;
      if (ragged) {
        p0 = c0 + off < m ? p0 : 0.0f;
        p1 = c0 + 32 + off < m ? p1 : 0.0f;
      }
      acc0[r] = p0;
      acc1[r] = p1;
    }
  };
  auto lse_tile = [&](int tt) __attribute__((always_inline)) {
    const long long c0 = (long long)(tile_lo + tt * tile_st) * V4_TN + 4 * fh;
    if (c0 - 4 * fh + V4_TN > m) {  // the ragged last tile of the table: columns beyond m do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long c = c0 + 8 * (r >> 2) + (r & 3);
        acc0[r] = c < m ? acc0[r] : -__builtin_inff();
        acc1[r] = c + 32 < m ? acc1[r] : -__builtin_inff();
      }
    }
    float mx = rmax;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, __builtin_fmaxf(acc0[r], acc1[r]));
    // all of this lane's columns padded and nothing before them (a column group of just the ragged last tile,
    // m % 64 <= 4): mx is still -inf; a finite reference keeps (-inf) - (-inf) out of the exponents
    const float rf = mx == -__builtin_inff() ? 0.0f : mx;
    float sm = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      sm += __builtin_amdgcn_exp2f((acc0[r] - rf) * V3_LOG2E) + __builtin_amdgcn_exp2f((acc1[r] - rf) * V3_LOG2E);
    rsum = rsum * __builtin_amdgcn_exp2f((rmax - rf) * V3_LOG2E) + sm;
    rmax = mx;
    const long long rel = lab - c0;
    const bool hit = rel >= 0 && rel < V4_TN && (rel & 7) < 4 && lab < m;  // not a padded column
    if (__any(hit)) {  // rare: a row's label lies in exactly one tile of the table
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = 8 * (r >> 2) + (r & 3);
        tsc = (hit && rel == off) ? acc0[r] : tsc;
        tsc = (hit && rel == 32 + off) ? acc1[r] : tsc;
      }
      tfound = tfound || hit;
    }
  };

  constexpr int PF = 8;
  // tile 0 is peeled off the loop: in straight-line code the compiler waits for fragment kb right in
  // front of its first MFMA (inside a loop it waits for all of them at the loop entry)
  auto tile = [&](int tt, auto first) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KGE_BARRIER();  // B1(tt): tile tt landed; staging drained
    __builtin_amdgcn_sched_barrier(0);
    stamp();  // tile tt released
    const unsigned int bt = (unsigned int)((tt % NBUF) * TILEB);
    unsigned int bp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bp[t] = bt + boff[t];
    bf16x8 bq[PF];
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
    v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
    // the PREVIOUS tile's scores -> staging, behind the first reads of this tile in the LDS queue
    // (tile 0 stages zeros that nobody reads: one schedule for every tile)
    if constexpr (STAGED) {
#pragma unroll
      for (int g = 0; g < 4; ++g) c_write(acc0, 0, g);
#pragma unroll
      for (int g = 0; g < 4; ++g) c_write(acc1, 1, g);
    }
    // LDS ops younger than read q when slot q waits: q < 8: the rest of the prefetch, the 8
    // writes (V3_STORE) and the reads of slots 0..q-1 = 15 (7 without the writes); q >= 8: min(7, NQ-1-q) reads
    v4_static_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int younger = q < 8 ? (STAGED ? 15 : 7) : ((NQ - 1 - q >= PF - 1) ? PF - 1 : NQ - 1 - q);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(younger) : "memory");
      // first tile: fragment kb has arrived (in-order returns: at most NKB-1-kb younger loads
      // outstanding); a no-op afterwards
      if constexpr (q == QB2) KGE_BARRIER();  // B2(tt): the scores of tile tt-1 are staged
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (q == 0) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], afr[0], zero, 0, 0, 0);
      } else if constexpr (q == 1) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[1], afr[0], zero, 0, 0, 0);
      } else if constexpr (q & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % PF], afr[q >> 1], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % PF], afr[q >> 1], acc0, 0, 0, 0);
      }
      if constexpr (q + PF < NQ) bread(bq[q % PF], std::integral_constant<int, q + PF>{});
      // first tile: the K-blocks FR0.. of the query fragments are requested from inside the chain, one per MFMA
      // slot (K-block FR0 + q in slot q, 2 (FR0 + q) - q >= 2 FR0 slots = 700 cycles ahead of its MFMA): issued in
      // front of the chain they cost 1.7 k cycles of blocked issue (a vector-memory instruction holds its wave
      // ~70 cycles while four waves share the 64 B/clk path), here each hides behind an MFMA (profiles/r3e)
      if constexpr (decltype(first)::value && q < NKB - FR0)
        load_fragments(std::integral_constant<int, FR0 + q>{}, std::integral_constant<int, FR0 + q + 1>{});
      if constexpr (RK_PIPE) rank_step(qc);
    });
    stamp();  // tile tt: MFMA chain issued
  };
  tile(0, std::true_type{});
  if constexpr (EPI == V3_LSE) lse_tile(0);
  if constexpr (IS_DS) ds_tile(0);
  if constexpr (EPI == V3_RANK && !RK_PIPE) rank_tile(0);
  for (int tt = 1; tt < ntl; ++tt) {
    if constexpr (RK_PIPE) {
      pv0 = acc0;
      pv1 = acc1;
      pg[0] = pg[1] = pc[0] = pc[1] = 0u;
    }
    tile(tt, std::false_type{});
    if constexpr (EPI == V3_LSE) lse_tile(tt);
    if constexpr (IS_DS) ds_tile(tt);
    if constexpr (EPI == V3_RANK && !RK_PIPE) rank_tile(tt);
    if constexpr (RK_PIPE) {  // tile tt - 1: compared during tile tt's chain
      if (rk_slow) {
        rank_masks(pv0, pv1, pg, pc, false);
      } else {  // the slots left dense (greater, NOT close) bits
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          pg[h] = rk_spread(pg[h] & 0xffffu);
          pc[h] = rk_spread(~pc[h] & 0xffffu);
        }
      }
      rank_finish(tt - 1, pg, pc);
    }
  }
  if constexpr (RK_PIPE) rank_tile(ntl - 1);  // the last tile has no chain to hide behind
  // the last tile's scores: stage them for the loaders' final pass
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KGE_BARRIER();  // B1(ntl)
  if constexpr (STAGED) {
#pragma unroll
    for (int g = 0; g < 4; ++g) c_write(acc0, 0, g);
#pragma unroll
    for (int g = 0; g < 4; ++g) c_write(acc1, 1, g);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KGE_BARRIER();  // B2(ntl)
  if (nx.qf != nullptr && nx.mode == 2)  // no idle workgroups in this geometry: a slice of the next batch's queries
    v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)(rg * ncg + cg) * 256 + tid, (long long)nx.nblocks * 256);
  if constexpr (EPI == V3_LSE) {
    // the two lanes of a row -> one (max, sum exp) per row and column group
    const float omax = __shfl_xor(rmax, 32, 64), osum = __shfl_xor(rsum, 32, 64);
    const float M = __builtin_fmaxf(rmax, omax);
    const float L = rsum * __builtin_amdgcn_exp2f((rmax - M) * V3_LOG2E) +
                    osum * __builtin_amdgcn_exp2f((omax - M) * V3_LOG2E);
    const long long row = (long long)rgl * V4_ROWS + 32 * w4 + fi;
    if (row < n) {
      if (fh == 0) {
        float* pp = ce.part + ((row + roff) * ncg + cg) * 2;
        pp[0] = M;
        pp[1] = L;
      }
      if (tfound) ce.true_score[row + roff] = tsc;  // (never with a NULL label vector: lab stays -1)
    }
  }
  if constexpr (EPI == V3_RANK) {
    // the two lanes of a row -> one contribution per row, column group and ranking
    const int G = rk_g + __shfl_xor(rk_g, 32, 64), C = rk_c + __shfl_xor(rk_c, 32, 64);
    const long long row = (long long)rgl * V4_ROWS + 32 * w4 + fi;
    const bool wr = row < n && fh == 0;
    unsigned long long* rank = ce.rk_rank[rk_side] + row;
    unsigned long long* ties = ce.rk_ties[rk_side] + row;
    if (wr && G != 0) atomicAdd(rank, (unsigned long long)G);
    if (wr && C != 0) atomicAdd(ties, (unsigned long long)C);
    const int fc = rk_t == -__builtin_inff() ? 1 : 0;  // is -inf (a filtered column's score) close to the true score
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < ce.rk_nfilt) {
        const int FG = rk_fg[k] + __shfl_xor(rk_fg[k], 32, 64), FC = rk_fc[k] + __shfl_xor(rk_fc[k], 32, 64);
        const int FN = rk_fn[k] + __shfl_xor(rk_fn[k], 32, 64);
        const int Gk = G - FG, Ck = C - FC + fc * FN;
        if (wr && Gk != 0) atomicAdd(rank + (k + 1) * ce.rk_ld, (unsigned long long)Gk);
        if (wr && Ck != 0) atomicAdd(ties + (k + 1) * ce.rk_ld, (unsigned long long)Ck);
      }
    }
  }
}

int run_pairs_bf16_v6(int scorer, bool split, const Operand& TG, bool two_sided, int d, long long n, long long m,
                      float* out, long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                      const void* qf, const NextQ& nx, int reserve_cus);
int run_pairs_bf16_v8(int scorer, bool split, const Operand& TG, bool two_sided, int d, long long n, long long m,
                      int nbatch, const void* qf, long long q_stride_bytes, float* out, long long out_stride,
                      long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg, const NextQ& nx,
                      int reserve_cus);

static inline bool v4_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static std::atomic<unsigned long long> g_v4_epoch{0};

// compute units of the CURRENT device (one process may drive several: the count is looked up per device, once)
static int v4_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

// Prepared queries of a launch (kge_build_queries / kge_score_queries; KGE_FLAG_SPLIT_QUERY):
//   ready        this launch's fragments, already built (by query_build_kernel or by a previous launch's spare
//                workgroups); NULL: the in-launch cooperative build through the workspace
//   build_first  ready == NULL: build them into the workspace by a separate launch on the same stream, then score
//                prepared (what the one-call entry points do for split queries)
//   next         the NEXT batch, built by this launch's spare workgroups (qf == NULL: none)
struct V4Prep {
  const void* ready = nullptr;
  bool build_first = false;
  NextQ next = NextQ{};
};

template <int SCORER, int HH, int SPLIT>
static NextQ v4_nextq(const Operand& A, const Operand* A2, const Operand& R, int dir, long long n, void* qf) {
  constexpr int RGR = SPLIT ? 64 : V4_ROWS;
  NextQ q{};
  q.A = A;
  q.A2 = A2 ? *A2 : A;
  q.R = R;
  q.dir = dir;
  q.n = n;
  q.rgn1 = (int)((n + RGR - 1) / RGR);
  q.rgn = A2 ? 2 * q.rgn1 : q.rgn1;
  q.qf = (u32x4*)qf;
  return q;
}

// bytes of the fragments of one batch: rgn row groups of 128 (virtual) rows x 2 HH bf16
long long pairs_bf16_v4_query_bytes(int d, long long n, bool two_sided, bool split) {
  const long long rgr = split ? 64 : V4_ROWS;
  const long long rgn = (two_sided ? 2 : 1) * ((n + rgr - 1) / rgr);
  return rgn * V4_ROWS * (long long)d * 2;
}

template <int SCORER, int HH, int SPLIT>
static int launch_query_build(const NextQ& q, hipStream_t st) {
  constexpr int RGR = SPLIT ? 64 : V4_ROWS;
  const long long items = (long long)(q.nbatch > 1 ? q.nbatch : 1) * q.rgn * RGR * (HH / 8);
  long long blocks = (items + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((query_build_kernel<SCORER, HH, SPLIT>), dim3((unsigned)blocks), dim3(256), 0, st, q);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// A2 != nullptr: two-sided launch (A = subjects scored sp_, A2 = objects scored _po into the
// column block `out2_off` floats behind).
template <int SCORER, int HH, int EPI = V3_STORE, int SPLIT = 0>
static int launch_v4(const Operand& A, const Operand* A2, const Operand& R, const Operand& TG, int dir,
                     long long n, long long m, float* out, long long ldo, long long out2_off,
                     hipStream_t st, unsigned long long* dbg, void* ws, long long ws_bytes,
                     int reserve_cus, const CeArgs& ce = CeArgs{}, const V4Prep& pp = V4Prep{}) {
  constexpr int RGR = SPLIT ? 64 : V4_ROWS;
  const int rgn1 = (int)((n + RGR - 1) / RGR);
  const int rgn = A2 ? 2 * rgn1 : rgn1;
  const int ntiles = (int)((m + V4_TN - 1) / V4_TN);
  const bool prepared = pp.ready != nullptr || pp.build_first;
  // one workgroup per CU (minus the CUs the caller keeps free for concurrent work, e.g. the
  // RCCL kernels of an overlapped exchange): split the target tiles into column groups
  const int cu_all = v4_cu_count();
  int cus = cu_all - reserve_cus;
  if (cus > 256) cus = 256;
  if (cus < 8) cus = 8;
  // fused loss: the caller sized its partial-result scratch for the geometry of
  // pairs_bf16_v3_column_groups (256 workgroup slots); fewer CUs: the launch check below declines
  if (EPI != V3_STORE) cus = 256;
  int ncg = cus / rgn;
  // Workgroup b runs on XCD b % 8 and the launch maps its low three bits to the low three bits of the column
  // group: the grid is 8 * rgn * ceil(ncg / 8) workgroups, one per CU.  Whole groups of eight column groups that
  // fit the CUs (rgn = 3: 80 instead of 85 -> 240 workgroups, not 264 and a declined launch; a two-sided batch of
  // 1,151 rows: 8 instead of 14).  The fused-loss epilogues keep the geometry their scratch was sized for.
  if ((EPI == V3_STORE || EPI == V3_RANK) && ncg > 8) ncg = 8 * (cus / 8 / rgn > 0 ? cus / 8 / rgn : 1);
  if (ncg < 1) ncg = 1;
  int tpc = (ntiles + ncg - 1) / ncg;
  if (tpc < 1) tpc = 1;
  ncg = (ntiles + tpc - 1) / tpc;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  const int tgmode = TG.idx.ptr == nullptr ? 0 : (TG.idx.itype ? 2 : 1);
  const long long qf_bytes = (long long)rgn * V4_ROWS * HH * 4;
  if (ldo >= (1LL << 24)) return KGE_ERR_UNSUPPORTED;
  unsigned long long* flags = nullptr;
  u32x4* qf = nullptr;
  unsigned long long epoch = 0;
  int nbuild = -1;
  if (pp.ready != nullptr) {
    if (!v4_al16(pp.ready)) return KGE_ERR_INVALID_ARG;
    qf = (u32x4*)pp.ready;  // no flags, no polling: workgroups need not be co-resident, any grid goes
  } else {
    // the in-launch build needs the workspace and every workgroup resident at once (spin-wait on the builders'
    // flags).  Fine under hipGraph capture: the epoch is frozen then, but every consumer clears its own
    // flag line after reading it, so a replay never sees the previous run's flags.
    if (ws == nullptr || !v4_al16(ws) || ws_bytes < qf_bytes + PAIRS_WS_CTRL_BYTES) return KGE_ERR_UNSUPPORTED;
    // control block first, at an offset independent of n (pairs_bf16_v3_workspace_bytes)
    flags = (unsigned long long*)ws;
    qf = (u32x4*)((char*)ws + PAIRS_WS_CTRL_BYTES);
    if (pp.build_first) {
      const int rc = launch_query_build<SCORER, HH, SPLIT>(v4_nextq<SCORER, HH, SPLIT>(A, A2, R, dir, n, qf), st);
      if (rc != KGE_OK) return rc;
    } else {
      if (SPLIT || (long long)rgn * ncg > 512 || grid > cu_all) return KGE_ERR_UNSUPPORTED;
      static const unsigned long long seed =
          ((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() << 20) | (1ull << 63);
      epoch = seed + ++g_v4_epoch;  // never 0 (= "cleared")
      nbuild = ncg < 8 ? ncg : 8;
      const int items = V4_ROWS * (HH / 8);  // at least one item per builder thread
      while (nbuild > 1 && nbuild * 512 > items) --nbuild;
      // KGE_V4_OWN_BUILD=1 (tests): no cooperative build, every consumer wave builds its own fragments --
      // the path a consumer otherwise only takes after a time-out or on a degraded workspace
      if (sw(SW_V4_OWN_BUILD) == 1) nbuild = 0;
    }
  }
  // d = 512, prepared queries, score store, all entities (or a contiguous slice): the unit-pipelined kernel
  // (score_pairs_bf16_v6.hip) -- same bits, first store ~4 k instead of ~12 k cycles into the launch
  if constexpr (EPI == V3_STORE && HH == 256) {
    if (prepared && tgmode == 0) {
      if (pp.next.qf != nullptr && !v4_al16(pp.next.qf)) return KGE_ERR_INVALID_ARG;
      // the persistent kernel with two consumer waves per SIMD (score_pairs_bf16_v8.hip); KGE_V8=0: v6 / v7
      const int rc8 = run_pairs_bf16_v8(SCORER, SPLIT != 0, TG, A2 != nullptr, 2 * HH, n, m, 1, qf, 0, out, 0, ldo,
                                        out2_off, st, dbg, pp.next, reserve_cus);
      if (rc8 != KGE_ERR_UNSUPPORTED) return rc8;
      const int rc = run_pairs_bf16_v6(SCORER, SPLIT != 0, TG, A2 != nullptr, 2 * HH, n, m, out, ldo, out2_off, st, dbg,
                                       qf, pp.next, reserve_cus);
      if (rc != KGE_ERR_UNSUPPORTED) return rc;
    }
  }
  // the next batch's queries: built on the compute units the
  // geometry leaves idle (the column groups rarely fill all of them: 228 of 256 at the FB15k-237 shape).
  NextQ nx = pp.next;
  if (nx.qf != nullptr) {
    if (!prepared || !v4_al16(nx.qf)) return KGE_ERR_INVALID_ARG;
    // the idle column-group slots of the grid (8 * ceil(ncg / 8) - ncg per row group) are resident on compute
    // units of their own from the first cycle: they build.  Fewer than four of them: every scoring workgroup's
    // consumer waves take a slice behind their last tile instead.
    const int spare = rgn * ((((ncg + 7) / 8) * 8) - ncg);
    if (spare >= 4) {
      nx.mode = 1;
      nx.nblocks = spare;
    } else {
      nx.mode = 2;
      nx.nblocks = rgn * ncg;
    }
  }
  const Operand& AA2 = A2 ? *A2 : A;
  // interleaved tiles (tiles_per_cg = 0) once a launch's score block outgrows the Infinity Cache
  const long long il = sw(SW_V4_INTERLEAVE);
  // AND the output pitch is sector-aligned (padded pitch: 433 -> 417 us on a 574,311-column shard; with an
  // unpadded pitch the partial sectors at tile edges then come from two XCDs' L2s: 9.9 -> 10.5 ms, so no)
  const bool interleave =
      il >= 0 ? il == 1 : ((double)n * (double)m * 4.0 * (A2 ? 2 : 1) > 192e6 && (ldo & 7) == 0);
  const int tpc_arg = interleave ? 0 : tpc;
  // Score stores: plain (the lines stay dirty in the 32 MB of L2 until they are evicted or the end-of-kernel
  // write-back flushes them: ~3 us behind a launch whose 30 MB score block fits) or agent-scope write-through (sc1:
  // the bytes go to memory as they are stored, nothing is left to flush -- but a partial 32-byte sector is then a
  // read-modify-write at the memory instead of a merge in L2).  Measured (tools/ab_probe.py, tools/big_m_probe.py,
  // round 3): write-through wins whenever the rows are sector-aligned (FB15k-237 shape one-sided 13.9 -> 12.8 us,
  // a 574,311-column Wikidata5M shard 394 -> 353 us) and for blocks that fit the L2 even when they are not (14.5 ->
  // 14.0 us); it loses 7 % on a 1.2 GB slab with an unaligned pitch.  KGE_V4_STORE_SC1=0/1 forces either.
  const long long sc1e = sw(SW_V4_STORE_SC1);
  const bool st_aligned = (ldo & 7) == 0 && (out2_off & 7) == 0 && ((uintptr_t)out & 31) == 0;
  const bool st_small = (double)n * (double)m * 4.0 * (A2 ? 2 : 1) <= 48e6;
  const int st_sc1 = sc1e >= 0 ? (int)sc1e : ((st_aligned || st_small) ? 1 : 0);
#define KGE_V4L(MODE)                                                                                  \
  hipLaunchKernelGGL((pairs_bf16_v4_kernel<SCORER, HH, MODE, EPI, SPLIT>), dim3(grid), dim3(512), 0, st, A, \
                     AA2, R, TG, dir, n, m, rgn, rgn1, out2_off, ncg, tpc_arg, ntiles, out, ldo, dbg, qf, flags, \
                     epoch, nbuild, ce, nx, st_sc1)
  if (tgmode == 0) KGE_V4L(0);
  else if (tgmode == 1) KGE_V4L(1);
  else KGE_V4L(2);
#undef KGE_V4L
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

bool pairs_bf16_v4_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                             const Operand& TG) {
  if (dtype != KGE_BF16) return false;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return false;
  if (d != 256 && d != 512) return false;
  if (!v4_al16(A.base) || !v4_al16(R.base) || !v4_al16(TG.base)) return false;
  if ((A.ld % 8) || (R.ld % 8) || (TG.ld % 8)) return false;
  return true;
}

// KGE_ERR_UNSUPPORTED: the caller falls back to the single-role kernel (v3).  A2: see launch_v4.
int run_pairs_bf16_v4(int scorer, const Operand& A, const Operand* A2, const Operand& R,
                      const Operand& TG, int dir, int d, long long n, long long m, float* out,
                      long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                      void* ws, long long ws_bytes, int reserve_cus) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V4(SC)                                                                                 \
  switch (d) {                                                                                     \
    case 256:                                                                                      \
      return launch_v4<SC, 128>(A, A2, R, TG, dir, n, m, out, ldo, out2_off, st, dbg, ws, ws_bytes, \
                                reserve_cus);                                                      \
    case 512:                                                                                      \
      return launch_v4<SC, 256>(A, A2, R, TG, dir, n, m, out, ldo, out2_off, st, dbg, ws, ws_bytes, \
                                reserve_cus);                                                      \
  }
  if (scorer == KGE_COMPLEX) { KGE_V4(KGE_COMPLEX) } else { KGE_V4(KGE_DISTMULT) }
#undef KGE_V4
  return KGE_ERR_UNSUPPORTED;
}

// ---- prepared queries ------------------------------------------------------------------------------------------
// run_query_build: the fragments of one batch into `qf` (pairs_bf16_v4_query_bytes).  A2: two-sided.
int run_query_build(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                    long long n, void* qf, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (!v4_al16(qf)) return KGE_ERR_INVALID_ARG;
#define KGE_QB(SC, HHV, SP) return launch_query_build<SC, HHV, SP>(v4_nextq<SC, HHV, SP>(A, A2, R, dir, n, qf), st)
#define KGE_QB2(SC)                                    \
  if (d == 256) {                                      \
    if (split) KGE_QB(SC, 128, 1); else KGE_QB(SC, 128, 0); \
  } else if (d == 512) {                               \
    if (split) KGE_QB(SC, 256, 1); else KGE_QB(SC, 256, 0); \
  }
  if (scorer == KGE_COMPLEX) { KGE_QB2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_QB2(KGE_DISTMULT) }
#undef KGE_QB2
#undef KGE_QB
  return KGE_ERR_UNSUPPORTED;
}

// run_query_build + the row-major query matrix Q16 [rows of side 1, rows of side 2][d] of the gradient products + a
// cleared float buffer, in ONE launch (the backward of the fused losses: ce_loss.hip).  d in {256, 512}.
int run_query_build_q16(int scorer, const Operand& A, const Operand* A2, const Operand& R, int dir, int d, long long n,
                        void* qf, unsigned short* q16, float* zero, long long zero_cnt, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (!v4_al16(qf) || !v4_al16(q16)) return KGE_ERR_INVALID_ARG;
  Q16Out qo{q16, zero, zero_cnt};
#define KGE_QB16(SC, HHV)                                                                                  \
  {                                                                                                        \
    const NextQ q = v4_nextq<SC, HHV, 0>(A, A2, R, dir, n, qf);                                            \
    long long blocks = ((long long)q.rgn * V4_ROWS * (HHV / 8) + 255) / 256;                               \
    if (blocks > 1024) blocks = 1024;                                                                      \
    hipLaunchKernelGGL((query_build_q16_kernel<SC, HHV>), dim3((unsigned)blocks), dim3(256), 0, st, q, qo); \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                      \
  }
#define KGE_QB16B(SC) \
  if (d == 256) KGE_QB16(SC, 128) else if (d == 512) KGE_QB16(SC, 256)
  if (scorer == KGE_COMPLEX) { KGE_QB16B(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_QB16B(KGE_DISTMULT) }
#undef KGE_QB16B
#undef KGE_QB16
  return KGE_ERR_UNSUPPORTED;
}

// run_query_build + run_eval_begin (kge_eval_batch) as ONE launch (eval_begin_build_kernel)
int run_eval_begin_build(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                         long long n, void* qf, const EvalLists& L, const Index& s, const Index& o, long long m,
                         long long rs, long long us, long long* tgt, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (!v4_al16(qf)) return KGE_ERR_INVALID_ARG;
  const int row_blocks = (int)((n + 3) / 4);
#define KGE_EBB(SC, HHV, SP)                                                                                   \
  {                                                                                                            \
    const NextQ q = v4_nextq<SC, HHV, SP>(A, A2, R, dir, n, qf);                                               \
    constexpr int RGR = SP ? 64 : V4_ROWS;                                                                     \
    long long blocks = ((long long)q.rgn * RGR * (HHV / 8) + 255) / 256;                                       \
    if (blocks > 1024) blocks = 1024;                                                                          \
    if (blocks < 1) blocks = 1;                                                                                \
    hipLaunchKernelGGL((eval_begin_build_kernel<SC, HHV, SP>),                                                 \
                       dim3((unsigned)(blocks + (long long)(L.nq + 1) * row_blocks)), dim3(256), 0, st, q, L, s, o, \
                       (int)blocks, row_blocks, n, m, rs, us, tgt);                                            \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                          \
  }
#define KGE_EBB2(SC)                                          \
  if (d == 256) {                                             \
    if (split) KGE_EBB(SC, 128, 1) else KGE_EBB(SC, 128, 0)   \
  } else if (d == 512) {                                      \
    if (split) KGE_EBB(SC, 256, 1) else KGE_EBB(SC, 256, 0)   \
  }
  if (scorer == KGE_COMPLEX) { KGE_EBB2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_EBB2(KGE_DISTMULT) }
#undef KGE_EBB2
#undef KGE_EBB
  return KGE_ERR_UNSUPPORTED;
}

// run_query_build + the filter-bit set launch of kge_score_rank_sp_po as ONE launch (query_build_bits_kernel)
int run_query_build_bits(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                         long long n, void* qf, const RankBitLists& B, int lists, long long col_begin, long long m,
                         long long rs, long long us, hipStream_t st) {
  if (n == 0) return KGE_OK;
  if (!v4_al16(qf) || lists < 1 || lists > 4) return KGE_ERR_INVALID_ARG;
  const int row_blocks = (int)((n + 3) / 4);
#define KGE_QBB(SC, HHV, SP)                                                                                   \
  {                                                                                                            \
    const NextQ q = v4_nextq<SC, HHV, SP>(A, A2, R, dir, n, qf);                                               \
    constexpr int RGR = SP ? 64 : V4_ROWS;                                                                     \
    long long blocks = ((long long)q.rgn * RGR * (HHV / 8) + 255) / 256;                                       \
    if (blocks > 1024) blocks = 1024;                                                                          \
    if (blocks < 1) blocks = 1;                                                                                \
    hipLaunchKernelGGL((query_build_bits_kernel<SC, HHV, SP>), dim3((unsigned)(blocks + (long long)lists * row_blocks)), \
                       dim3(256), 0, st, q, B, (int)blocks, row_blocks, n, col_begin, m, rs, us);              \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                          \
  }
#define KGE_QBB2(SC)                                          \
  if (d == 256) {                                             \
    if (split) KGE_QBB(SC, 128, 1) else KGE_QBB(SC, 128, 0)   \
  } else if (d == 512) {                                      \
    if (split) KGE_QBB(SC, 256, 1) else KGE_QBB(SC, 256, 0)   \
  }
  if (scorer == KGE_COMPLEX) { KGE_QBB2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_QBB2(KGE_DISTMULT) }
#undef KGE_QBB2
#undef KGE_QBB
  return KGE_ERR_UNSUPPORTED;
}

// A group of `nbatch` equally shaped batches (kge_build_queries_multi / kge_score_queries_multi): batch l = rows
// [l n, (l + 1) n) of the index vectors, its fragments at qf + l * qstride_bytes.
NextQ pairs_bf16_nextq(bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, long long n,
                       int nbatch, void* qf, long long qstride_bytes) {
  NextQ q = split ? v4_nextq<KGE_COMPLEX, 256, 1>(A, A2, R, dir, n, qf) : v4_nextq<KGE_COMPLEX, 256, 0>(A, A2, R, dir, n, qf);
  q.nbatch = nbatch;
  q.qstride = qstride_bytes / 16;
  return q;
}

int run_query_build_multi(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R, int dir, int d,
                          long long n, int nbatch, void* qf, long long qstride_bytes, hipStream_t st) {
  if (n == 0 || nbatch == 0) return KGE_OK;
  if (!v4_al16(qf) || (qstride_bytes & 15)) return KGE_ERR_INVALID_ARG;
  const NextQ q = pairs_bf16_nextq(split, A, A2, R, dir, n, nbatch, qf, qstride_bytes);
#define KGE_QB(SC, HHV, SP) return launch_query_build<SC, HHV, SP>(q, st)
#define KGE_QB2(SC)                                    \
  if (d == 256) {                                      \
    if (split) KGE_QB(SC, 128, 1); else KGE_QB(SC, 128, 0); \
  } else if (d == 512) {                               \
    if (split) KGE_QB(SC, 256, 1); else KGE_QB(SC, 256, 0); \
  }
  if (scorer == KGE_COMPLEX) { KGE_QB2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_QB2(KGE_DISTMULT) }
#undef KGE_QB2
#undef KGE_QB
  return KGE_ERR_UNSUPPORTED;
}

// run_pairs_bf16_v4_prepared: score with prepared queries.
//   ready != NULL            the batch's fragments (A / A2 / R are not read)
//   ready == NULL            split / build-first: fragments are built into `ws` by a separate launch, then scored
//   nA / nA2 / nR, nn, nqf   nqf != NULL: the next batch, built by spare workgroups of this launch
int run_pairs_bf16_v4_prepared(int scorer, bool split, const Operand& A, const Operand* A2, const Operand& R,
                               const Operand& TG, int dir, int d, long long n, long long m, float* out,
                               long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                               const void* ready, void* ws, long long ws_bytes, int reserve_cus, const Operand* nA,
                               const Operand* nA2, const Operand* nR, long long nn, void* nqf) {
  if (n == 0 || m == 0) return KGE_OK;
  V4Prep pp;
  pp.ready = ready;
  pp.build_first = ready == nullptr;
#define KGE_V4P(SC, HHV, SP)                                                                                    \
  do {                                                                                                          \
    if (nqf != nullptr && nn > 0) pp.next = v4_nextq<SC, HHV, SP>(*nA, nA2, *nR, dir, nn, nqf);                   \
    return launch_v4<SC, HHV, V3_STORE, SP>(A, A2, R, TG, dir, n, m, out, ldo, out2_off, st, dbg, ws, ws_bytes, \
                                            reserve_cus, CeArgs{}, pp);                                         \
  } while (0)
#define KGE_V4P2(SC)                                         \
  if (d == 256) {                                            \
    if (split) KGE_V4P(SC, 128, 1); else KGE_V4P(SC, 128, 0); \
  } else if (d == 512) {                                     \
    if (split) KGE_V4P(SC, 256, 1); else KGE_V4P(SC, 256, 0); \
  }
  if (scorer == KGE_COMPLEX) { KGE_V4P2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_V4P2(KGE_DISTMULT) }
#undef KGE_V4P2
#undef KGE_V4P
  return KGE_ERR_UNSUPPORTED;
}

// Fused 1vsAll loss forward (ce_loss.hip) on the loader/consumer kernel: V3_LSE epilogue.  A2 != NULL:
// both directions of a batch in one launch (ce.label / ce.label2, ce.side2_off).  `ws` = the fragment
// + flag block of the cooperative build (layout of pairs_bf16_v3_workspace_bytes).  KGE_ERR_UNSUPPORTED:
// the caller uses the single-role kernel.
int run_pairs_bf16_v4_epi(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R,
                          const Operand& TG, int dir, int d, long long n, long long m, hipStream_t st, void* ws,
                          long long ws_bytes, const CeArgs& ce, unsigned long long* dbg);

int run_pairs_bf16_v4_lse(int scorer, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG,
                          int dir, int d, long long n, long long m, hipStream_t st, void* ws, long long ws_bytes,
                          const CeArgs& ce, unsigned long long* dbg) {
  return run_pairs_bf16_v4_epi(scorer, V3_LSE, A, A2, R, TG, dir, d, n, m, st, ws, ws_bytes, ce, dbg);
}

// Would a two-sided counting launch (V3_RANK, in-launch cooperative build) of n rows per side against m targets be
// taken?  The launch conditions of launch_v4 that depend on the shape: kge_score_rank_* checks every row block BEFORE
// the first one counts anything (a decline halfway through would leave the counters of the earlier blocks behind).
bool pairs_bf16_v4_rank_launchable(int d, long long n, long long m, long long ws_bytes) {
  const int HH = d / 2;
  const int rgn = 2 * (int)((n + V4_ROWS - 1) / V4_ROWS);
  const int ntiles = (int)((m + V4_TN - 1) / V4_TN);
  const int cus = 256;
  int ncg = cus / rgn;
  if (ncg > 8) ncg = 8 * (cus / 8 / rgn > 0 ? cus / 8 / rgn : 1);
  if (ncg < 1) ncg = 1;
  int tpc = (ntiles + ncg - 1) / ncg;
  if (tpc < 1) tpc = 1;
  ncg = (ntiles + tpc - 1) / tpc;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  const long long qf_bytes = (long long)rgn * V4_ROWS * HH * 4;
  return ws_bytes >= qf_bytes + PAIRS_WS_CTRL_BYTES && (long long)rgn * ncg <= 512 && grid <= v4_cu_count();
}

// Any fused-loss epilogue on the loader/consumer kernel: V3_LSE (forward), V3_DS / V3_DSIG (the G16 pass of
// the backward: the consumers turn a finished tile into d loss / d score in place, the store waves round to
// bf16 and write 8 bytes per lane).  V3_SPLUS stays on the single-role kernel.
int run_pairs_bf16_v4_epi(int scorer, int epi, const Operand& A, const Operand* A2, const Operand& R,
                          const Operand& TG, int dir, int d, long long n, long long m, hipStream_t st, void* ws,
                          long long ws_bytes, const CeArgs& ce, unsigned long long* dbg) {
  if (n == 0 || m == 0) return KGE_OK;
  if (TG.idx.ptr != nullptr) return KGE_ERR_UNSUPPORTED;
#define KGE_V4E(SC, EP)                                                                                    \
  switch (d) {                                                                                            \
    case 256:                                                                                             \
      return launch_v4<SC, 128, EP>(A, A2, R, TG, dir, n, m, nullptr, 1, 0, st, dbg, ws, ws_bytes, 0, ce); \
    case 512:                                                                                             \
      return launch_v4<SC, 256, EP>(A, A2, R, TG, dir, n, m, nullptr, 1, 0, st, dbg, ws, ws_bytes, 0, ce); \
  }
#define KGE_V4S(EP)                                                                                              \
  if (scorer == KGE_COMPLEX) { KGE_V4E(KGE_COMPLEX, EP) } else if (scorer == KGE_DISTMULT) { KGE_V4E(KGE_DISTMULT, EP) }
  if (epi == V3_LSE) { KGE_V4S(V3_LSE) } else if (epi == V3_DS) { KGE_V4S(V3_DS) } else if (epi == V3_DSIG) { KGE_V4S(V3_DSIG) }
  else if (epi == V3_RANK) { KGE_V4S(V3_RANK) }
#undef KGE_V4S
#undef KGE_V4E
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
