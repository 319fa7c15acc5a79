// score_pairs_bf16_v5.hip -- ComplEx / DistMult sp_/_po kernel for bf16 tables, d in {256, 512}, WITHOUT any
// hand-off between workgroups: every workgroup builds the query vectors of its own 64 rows.
//
// Why a second design next to score_pairs_bf16_v4.hip (an experiment that became the fallback).  In v4 a row group's 128 query vectors are built
// once by a few workgroups and handed to the other ~56 through memory: index load -> row gather ->
// write-through + acknowledgement -> flag -> poll -> fragment fetch, six dependent round trips of ~2 k
// cycles each = the 12 k-cycle start-up that is 40 % of a launch at n = 512.  Building
// them locally costs vector-memory bandwidth (2 KiB of table rows per query row through 64 B/clk) and
// VALU work (~4 ops per element): too much for 128 rows per workgroup, affordable for 64 -- which in
// turn doubles the tile stream per CU.  Here:
//
//   * one workgroup = 64 query rows x a range of 64-target tiles, 8 waves;
//   * start-up (all 8 waves, two dependent round trips): indices -> coalesced row gathers (thread per
//     (row, 8 coordinates)) -> q = bf16(s (x) r) -> fragment-major into the SECOND tile buffer (XOR
//     swizzle: conflict-free for writers and readers) -> barrier -> every consumer wave reads its 32 rows'
//     fragments into MFMA operand registers.  Tile 0 is streaming into the first buffer meanwhile;
//   * tile loop as in v4 (roles, two barriers per tile, LDS-DMA ring, staged 16-byte stores), with four
//     consumer waves = 2 row blocks x 2 target halves: one 32x32 accumulator, NKB MFMAs per tile each;
//   * no workspace, no flags, no spin-wait, no co-residency requirement, any n.
//
// Arithmetic: the K order of every score (K-blocks 0 .. NKB-1 into one accumulator) and the query
// rounding are those of v4 / v3: the scores are bit-identical, whichever kernel a call lands on.
//
// Measured (C2, MI355X, round 2): one-sided 18.2-18.5 us, two-sided 27.1 us against 17.9 / 24.2 us for
// v4 -- the shorter start-up (7.9 k instead of 12 k cycles) is paid back by the tile phase (a 64-row
// workgroup moves twice the table bytes through LDS per score).  So v4 stays the default where it can
// run; this kernel takes the calls v4 cannot: no workspace, more than 32 row groups (n > 4096), fewer
// CUs than workgroups -- where it replaces the single-role kernel's 20 k-cycle own build (21.4 us).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace kge {

constexpr int V5_ROWS = 64, V5_TN = 64;
typedef float f32x4v5u __attribute__((ext_vector_type(4), aligned(4)));

template <int I, int N, class F>
__device__ __forceinline__ void v5_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    v5_static_for<I + 1, N>(f);
  }
}

// idx[i] without control flow (index_at branches on the index type: the loads of several items could then
// not be in flight together).  Ids are non-negative and < 2^31: the low dword of an int64 is the value.
// NULL = identity; `safe` is any readable address.
__device__ __forceinline__ long long v5_index_nb(const Index& ix, long long i, const void* safe) {
  const int sh = ix.itype ? 3 : 2;
  const char* p = ix.ptr ? (const char*)ix.ptr + ((i * ix.stride) << sh) : (const char*)safe;
  const unsigned int v = *reinterpret_cast<const unsigned int*>(p);
  return ix.ptr ? (long long)v : i;
}

template <int SCORER, int HH, int TGMODE>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v5_kernel(
    Operand A, Operand A2, Operand R, Operand TG, int dir, long long n, long long m, int rgn, int rgn1,
    long long out2_off, int ncg, int tiles_per_cg, int ntiles, float* __restrict__ out, long long ldo,
    unsigned long long* __restrict__ dbg) {
  constexpr int NKB = 2 * HH / 16;       // K-blocks of 16
  constexpr int NKH = HH / 16;           // K-blocks per half
  constexpr int ROWB = 4 * HH;           // bytes per table row (2*HH bf16)
  constexpr int SPR = HH / 4;            // 16-byte slots per row
  constexpr int TILEB = V5_TN * ROWB;    // bytes per target tile
  constexpr int NL = TILEB / 1024 / 4;   // 1-KiB DMA pieces per quarter of a tile
  constexpr int RPP = 64 / SPR;          // target rows per piece
  constexpr int CST0 = 2 * TILEB;        // score staging: [64 rows][64 cols] f32
  constexpr int SMEM = CST0 + V5_ROWS * V5_TN * 4;
  constexpr int XB = TILEB;              // fragment exchange area = the second tile buffer (64 rows x ROWB)
  constexpr int QB2 = 3 * NKB / 4;       // MFMA slot of barrier B2
  constexpr int CGR = HH / 8;            // items (groups of 8 coordinates) per query row
  constexpr int IPT = V5_ROWS * CGR / 512;  // items per thread
  static_assert(NL * RPP == 16 && SPR >= 32 && IPT >= 1, "d in {256, 512}");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  // ---- which rows / target tiles (XCD-aware as in v4: block id mod 8 = XCD, all row groups of a
  // target range share an L2)
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // two-sided launch (score_sp_po): row groups [0, rgn1) are the (s, p, ?) queries, [rgn1, rgn) the
  // (?, p, o) queries, scored into the column block behind the first one
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

  // =========================== DMA machinery (waves 4, 5; used before and after the build) ==========
  const unsigned short* tgb = (const unsigned short*)TG.base;
  const long long tld2 = TG.ld * 2;
  const int nfull = (int)(m / V5_TN);
  const int lr = lane / SPR, slot = lane % SPR;
  const int j2 = (wave & 1) * 2;  // a DMA wave's quarters: j2, j2 + 1
  unsigned int dvoff[NL], dsw[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    dsw[k] = (unsigned int)(((slot ^ lr) << 4) ^ ((RPP * k) << 4));
    dvoff[k] = (unsigned int)(lr * (int)tld2) + dsw[k];
  }
  auto load_rows = [&](int tt, int w) -> long long {
    const int tc = tt < ntl ? tt : ntl - 1;
    long long tr = (long long)(tile_lo + tc) * V5_TN + w * 16 + (lane & 15);
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
    const long long trow0 = (long long)(tile_lo + tc) * V5_TN;
    unsigned int d = (unsigned int)((tt & 1) * TILEB + w * NL * 1024);
    if (TGMODE == 0 && tile_lo + tc < nfull && tld2 < (1LL << 28)) {
      const unsigned char* p = (const unsigned char*)tgb + (trow0 + w * 16) * tld2;
      v5_static_for<0, NL>([&](auto kc) __attribute__((always_inline)) {
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
                                         (__attribute__((address_space(3))) void*)(smem + d + k * 1024), 16, 0, 0);
      }
    }
  };
  long long rna = 0, rnb = 0;
  if (wave == 4 || wave == 5) {  // tile 0 streams into buffer 0 while everybody builds
    const long long ra0 = load_rows(0, j2), rb0 = load_rows(0, j2 + 1);
    tile_dma(0, ra0, j2);
    tile_dma(0, rb0, j2 + 1);
    rna = load_rows(1, j2);
    rnb = load_rows(1, j2 + 1);
  }

  // =========================== query build: all 512 threads ==========================================
  // item = (query row, 8 coordinates): four 16-byte loads (first / second half of the entity and of the
  // relation row: coalesced, 32 consecutive threads share a row), 16 outputs, two 16-byte LDS writes.
  // All items of a thread go through the two round trips (indices, rows) TOGETHER.
  {
    const unsigned short *ap[IPT], *rp[IPT];
    int rowl[IPT], c8[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int it = tid + 512 * u;
      rowl[u] = it / CGR;
      c8[u] = it % CGR;
      const long long lrow = (long long)rgl * V5_ROWS + rowl[u];  // query row within its side
      const long long qrow = lrow < n ? lrow : n - 1;             // padded rows repeat row n-1
      ap[u] = (const unsigned short*)A.base + v5_index_nb(A.idx, qrow, TG.base) * A.ld + c8[u] * 8;
      rp[u] = (const unsigned short*)R.base + v5_index_nb(R.idx, qrow, TG.base) * R.ld + c8[u] * 8;
    }
    u32x4 a0[IPT], a1[IPT], r0[IPT], r1[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      a0[u] = *reinterpret_cast<const u32x4*>(ap[u]);
      a1[u] = *reinterpret_cast<const u32x4*>(ap[u] + HH);
      r0[u] = *reinterpret_cast<const u32x4*>(rp[u]);
      r1[u] = *reinterpret_cast<const u32x4*>(rp[u] + HH);
    }
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      u32x4 q0, q1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned int x0, x1;
        bf16_qpair_fast<SCORER>(dir, a0[u][e], a1[u][e], r0[u][e], r1[u][e], x0, x1);
        q0[e] = x0;
        q1[e] = x1;
      }
      // fragment-major: K-block kb of 32-row block rb is 64 positions x 16 B; position = (row & 31) +
      // 32 * (c8 & 1), stored at position ^ (2 * (kb & 7) + (c8 & 1)): the 16 lanes of a write group hit
      // 16 different bank quads, and so do the 16 lanes of a read group below
      const int rb = rowl[u] >> 5, fhh = c8[u] & 1, kbr = c8[u] >> 1;
      const int pos = (rowl[u] & 31) + 32 * fhh;
      const int sw = ((kbr & 7) << 1) | fhh;  // (NKH % 8 == 0: the same for the imaginary K-block)
      unsigned char* base = smem + XB + (rb * NKB) * 1024 + ((pos ^ sw) << 4);
      *reinterpret_cast<u32x4*>(base + kbr * 1024) = q0;
      *reinterpret_cast<u32x4*>(base + (NKH + kbr) * 1024) = q1;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // BX: all fragments of the 64 rows are in the exchange area
  stamp();  // 1: query vectors built

  if (wave >= 4) {
    if (wave < 6) {
      // ------------------------------- DMA waves -------------------------------
      for (int tt = 0; tt <= ntl; ++tt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile tt have landed
        __builtin_amdgcn_s_barrier();  // B1(tt) (tt == 0: the consumers have read their fragments)
        if (tt + 1 < ntl) {
          tile_dma(tt + 1, rna, j2);
          tile_dma(tt + 1, rnb, j2 + 1);
          rna = load_rows(tt + 2, j2);
          rnb = load_rows(tt + 2, j2 + 1);
        }
        __builtin_amdgcn_s_barrier();  // B2(tt)
      }
      return;
    }
    // ------------------------------- store waves: rows 32 s .. 32 s + 31 of the staging block --------
    const int sw_ = wave & 1;
    const int cl = lane & 15, rq = lane >> 4;
    const int z = cl ^ rq;
    unsigned int svoff[8];
    const long long r0 = (long long)rgl * V5_ROWS + 32 * sw_;
    const long long rbase = r0 < n ? r0 : n - 1;
    const unsigned int crd = (unsigned int)(CST0 + sw_ * 32 * 256 + rq * 256);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      long long r = r0 + 4 * i + rq;
      if (r >= n) r = n - 1;
      svoff[i] = (unsigned int)((r - rbase) * ldo * 4) + (unsigned int)(cl * 16);
    }
    unsigned char* const out_rb = (unsigned char*)(out + rbase * ldo);
    f32x4 cv[8];
    auto read_staging = [&]() {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        cv[i] = *reinterpret_cast<const f32x4*>(smem + crd + i * 1024 + ((z ^ ((4 * i) & 15)) << 4));
    };
    auto store_tile = [&](int tt) {
      const long long col0 = (long long)(tile_lo + tt) * V5_TN;
      if (col0 + V5_TN <= m) {
        unsigned char* sbase = out_rb + col0 * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4v5u*>(sbase + svoff[i]) = cv[i];
      } else {  // ragged end of the table (always this workgroup's last tile)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col0 + 4 * cl + e < m) *reinterpret_cast<float*>(out_rb + (col0 + e) * 4 + svoff[i]) = cv[i][e];
      }
    };
    for (int tt = 0; tt <= ntl; ++tt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging reads of the previous step are in registers
      __builtin_amdgcn_s_barrier();  // B1(tt): the consumers may overwrite the staging buffer
      if (tt >= 2) store_tile(tt - 2);  // from registers, while the DMA waves issue tile tt+1
      __builtin_amdgcn_s_barrier();  // B2(tt): scores of tile tt-1 are staged
      if (tt >= 1) read_staging();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_tile(ntl - 1);
    return;
  }

  // =================================== consumer waves ===================================
  // wave w: query rows 32 rb .. + 31 (rb = w & 1) x targets 32 th .. + 31 of every tile (th = w >> 1)
  const int rb = wave & 1, th = wave >> 1;
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 afr[NKB];
  v5_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value;
    constexpr int sw = ((kb & 7) << 1);
    afr[kb] = *reinterpret_cast<const bf16x8*>(smem + XB + (rb * NKB + kb) * 1024 + ((lane ^ (sw | fh)) << 4));
  });
  // B fragment (K-block kb) of target row 32 th + fi: 16-B slot s = s0(kb) + fh, stored at slot
  // s ^ (fi & 15): with s = 16 a + b the swizzle only touches b -> 8 address registers + immediates a * 256
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
    boff[t] = (unsigned int)(th * 32 * ROWB + fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  // staging: acc[4 g + e] = score(row 32 rb + fi, target 32 th + 8 g + 4 fh + e) -> 16-B chunk 8 th + 2 g + fh
  // of staging row 32 rb + fi, stored at chunk ^ (fi & 15)
  const unsigned int cwr = (unsigned int)(CST0 + (32 * rb + fi) * 256);
  const int y = fh ^ (fi & 15);
  auto c_write = [&](const f32x16& acc, int g) {
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    *reinterpret_cast<f32x4*>(smem + cwr + (((8 * th + 2 * g) ^ y) << 4)) = v;
  };

  // ONE accumulation chain per score, K-blocks in increasing order: the bits of v4 / v3.  (Two chains --
  // even / odd K-blocks -- were measured too: the same 48 cycles per MFMA; what paces the chain is the
  // LDS, which has to deliver a 1 KiB target fragment per MFMA and consumer wave next to the tile ring's
  // DMA writes, not the dependency on the previous result.)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  stamp();  // 2: fragments in registers

  constexpr int PF = 8;
  auto tile = [&](int tt) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // B1(tt): tile tt landed; staging drained (tt == 0: fragments read)
    __builtin_amdgcn_sched_barrier(0);
    stamp();  // tile tt released
    const unsigned int bt = (unsigned int)((tt & 1) * TILEB);
    unsigned int bp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bp[t] = bt + boff[t];
    bf16x8 bq[PF];
    auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      constexpr int s0 = (kb < NKH) ? (2 * kb) : (HH / 8 + 2 * (kb - NKH));
      const unsigned int addr = bp[(s0 & 15) >> 1];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((s0 >> 4) * 256) : "memory");
    };
    v5_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
    // the PREVIOUS tile's scores -> staging, behind the first reads of this tile in the LDS queue
    // (tile 0 stages zeros that nobody reads: one schedule for every tile)
#pragma unroll
    for (int g = 0; g < 4; ++g) c_write(acc, g);
    // LDS ops younger than read kb when slot kb waits: kb < PF: the rest of the prefetch, the 4 writes and
    // the reads issued by slots 0 .. kb-1 = PF + 3; kb >= PF: min(PF - 1, NKB - 1 - kb) reads
    v5_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      constexpr int younger = kb < PF ? PF + 3 : ((NKB - 1 - kb >= PF - 1) ? PF - 1 : NKB - 1 - kb);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(younger) : "memory");
      if constexpr (kb == QB2) __builtin_amdgcn_s_barrier();  // B2(tt): the scores of tile tt-1 are staged
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kb == 0) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], afr[0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % PF], afr[kb], acc, 0, 0, 0);
      }
      if constexpr (kb + PF < NKB) bread(bq[kb % PF], std::integral_constant<int, kb + PF>{});
    });
    stamp();  // tile tt: MFMA chain issued
  };
  for (int tt = 0; tt < ntl; ++tt) tile(tt);
  // the last tile's scores: stage them for the store waves' final pass
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // B1(ntl)
#pragma unroll
  for (int g = 0; g < 4; ++g) c_write(acc, g);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // B2(ntl)
}

static inline bool v5_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// A2 != nullptr: two-sided launch (A = subjects scored sp_, A2 = objects scored _po into the column block
// `out2_off` floats behind).
template <int SCORER, int HH>
static int launch_v5(const Operand& A, const Operand* A2, const Operand& R, const Operand& TG, int dir, long long n,
                     long long m, float* out, long long ldo, long long out2_off, hipStream_t st,
                     unsigned long long* dbg) {
  const long long rgn1l = (n + V5_ROWS - 1) / V5_ROWS;
  const long long rgnl = A2 ? 2 * rgn1l : rgn1l;
  const int ntiles = (int)((m + V5_TN - 1) / V5_TN);
  if (rgnl > (1 << 20) || ldo >= (1LL << 24)) return KGE_ERR_UNSUPPORTED;
  const int rgn1 = (int)rgn1l, rgn = (int)rgnl;
  // ~one workgroup per CU slot: split the target tiles into column groups (one group when there are more
  // row groups than CUs: the launch then runs in waves)
  int ncg = 256 / rgn;
  if (ncg < 1) ncg = 1;
  int tpc = (ntiles + ncg - 1) / ncg;
  if (tpc < 1) tpc = 1;
  ncg = (ntiles + tpc - 1) / tpc;
  const long long gridl = 8LL * rgn * ((ncg + 7) / 8);
  if (gridl > 0x7fffffffLL) return KGE_ERR_UNSUPPORTED;
  const int tgmode = TG.idx.ptr == nullptr ? 0 : (TG.idx.itype ? 2 : 1);
  const Operand& AA2 = A2 ? *A2 : A;
#define KGE_V5L(MODE)                                                                                          \
  hipLaunchKernelGGL((pairs_bf16_v5_kernel<SCORER, HH, MODE>), dim3((unsigned)gridl), dim3(512), 0, st, A, AA2, \
                     R, TG, dir, n, m, rgn, rgn1, out2_off, ncg, tpc, ntiles, out, ldo, dbg)
  if (tgmode == 0) KGE_V5L(0);
  else if (tgmode == 1) KGE_V5L(1);
  else KGE_V5L(2);
#undef KGE_V5L
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

bool pairs_bf16_v5_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R, const Operand& TG) {
  if (dtype != KGE_BF16) return false;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return false;
  if (d != 256 && d != 512) return false;
  if (!v5_al16(A.base) || !v5_al16(R.base) || !v5_al16(TG.base)) return false;
  if ((A.ld % 8) || (R.ld % 8) || (TG.ld % 8)) return false;
  return true;
}

// KGE_ERR_UNSUPPORTED: the caller falls back to the cooperative kernel (v4) / the single-role kernel (v3).
int run_pairs_bf16_v5(int scorer, const Operand& A, const Operand* A2, const Operand& R, const Operand& TG, int dir,
                      int d, long long n, long long m, float* out, long long ldo, long long out2_off, hipStream_t st,
                      unsigned long long* dbg) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V5(SC)                                                                              \
  switch (d) {                                                                                  \
    case 256: return launch_v5<SC, 128>(A, A2, R, TG, dir, n, m, out, ldo, out2_off, st, dbg);  \
    case 512: return launch_v5<SC, 256>(A, A2, R, TG, dir, n, m, out, ldo, out2_off, st, dbg);  \
  }
  if (scorer == KGE_COMPLEX) { KGE_V5(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_V5(KGE_DISTMULT) }
#undef KGE_V5
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
