// score_pairs_bf16_v6.hip -- ComplEx / DistMult sp_/_po scores of bf16 tables at d = 512 from PREPARED query
// fragments (kge_score_queries, and the one-call entry points behind a query_build_kernel launch): the store
// path of the BASELINE.json headline configuration.
//
// Why a second kernel next to pairs_bf16_v4_kernel.  With prepared queries the v4 launch at the FB15k-237 shape is
// bounded by its WRITE stream, not by its matrix pipe (profiles/r3_phase_timestamps.txt): the 29.8 MB score block
// of a one-sided batch leaves the chip at ~4.7 TB/s (16 stores of 1 KiB per store wave and 64-target tile take
// ~3.4 k cycles to issue, whatever they wait for; torch.fill_ of the same block: 4.5 TB/s), and the launch takes
//     launch gap + (time until the FIRST store is issued) + bytes written / write rate + acknowledgement.
// In v4 the first store goes out 12.3 k cycles after the workgroup starts -- a 64-target tile of LDS-DMA
// (3.8 k), the first MFMA chain with the query-fragment loads in it (4.5 k), then the staging of tile t at the
// start of chain t + 1, barrier B2 three quarters into that chain and the store after B1(t + 2): a lag of 1 1/4
// chains -- i.e. nothing is written during half of the kernel.  This kernel is built around that number:
//
//   * the unit of work is 32 targets x 128 query rows (32 KiB of table, 16 KiB of scores): the first unit lands
//     after 32 DMA pieces instead of 64 and is scored by 32 MFMAs per consumer wave instead of 64;
//   * the LDS holds a ring of FOUR units (128 KiB) and TWO staging buffers (2 x 16 KiB): a consumer wave writes the
//     unit it has just finished to staging[u & 1] at once, and ONE workgroup barrier R(u + 1) hands it to the store
//     waves, which read it and issue its stores right behind that barrier -- chain end to first store ~0.4 k
//     cycles -- while the consumers score unit u + 1 and stage it into the OTHER buffer;
//   * one barrier per unit (v4: two per tile = the same rate); what R(k) guarantees:
//       (a) unit k has landed in ring buffer k % 4           (the DMA waves waited for their pieces),
//       (b) the scores of unit k - 1 are in staging[(k - 1) & 1]   (the consumers wrote them and waited),
//       (c) the store waves have read staging[k & 1] (unit k - 2)  (they did so right after R(k - 1)),
//       (d) the consumers are done with ring buffer (k - 1) % 4    (the DMA waves refill it with unit k + 3).
//
// Roles as in v4 (one consumer and one loader wave per SIMD): consumer waves 0-3 keep the query fragments of their
// 32 rows in 128 operand registers and issue ds_read_b128 + v_mfma_f32_32x32x16_bf16, one accumulation chain per
// score in K order -- the bits of v4 / v3 / v5 and of the oracle's bf16 mode; DMA waves 4, 5 stream the table with
// LDS-DMA (XOR swizzle on the source address); store waves 6, 7 move staged scores to HBM, 8 rows x 128 bytes per
// instruction.  The ring fill (units 0, 1) and the last unit's stores are shared by all four loader waves.
//
// Not handled here (the launcher falls back to v4): target index lists, d = 256, the fused-loss / counting
// epilogues, the in-launch cooperative query build.
#include "common.hpp"
#include "bf16_queries.hpp"
#include <atomic>
#include <cstdlib>

namespace kge {

constexpr int V6_ROWS = 128, V6_UT = 32;
typedef float f32x4v6u __attribute__((ext_vector_type(4), aligned(4)));

// one LDS-DMA piece: 64 lanes x 16 B from row pointer P (SGPR pair) + per-lane offset VO to LDS address D (m0)
#define KGE_V6_DMA(D, VO, P) \
  KGE_STALL(__LINE__ + 6000); \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(D), "v"(VO), "s"(P) : "memory", "m0")

// LDS operations younger than the fragment read of slot kb when slot kb waits for it: the reads of the next seven
// slots (the read for slot j is issued in slot j - 8, across unit boundaries; behind the last unit the reads go to
// the next ring buffer all the same and nobody uses what they return: one schedule for every chain) and the staging
// writes of the previous unit (one each in slots V6_W0 .. V6_W0 + 3, behind that slot's read).
constexpr int V6_W0 = 2;   // two MFMA slots behind the previous chain's last MFMA: its result is there, no wait states
constexpr int V6_PB = 14;  // slot of barrier P(u): every staging write is older than the reads still allowed in flight
constexpr int v6_younger(int kb, bool writes) {
  int y = 7;
  if (writes)
    for (int j = V6_W0; j < V6_W0 + 4; ++j)
      if (j >= kb - 8 && j <= kb - 1) ++y;
  return y;
}

template <int SCORER, int SPLIT>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v6_kernel(
    Operand TG, long long n, long long m, int rgn, int rgn1, long long out2_off, int ncg, int units_per_cg,
    int nunits, float* __restrict__ out, long long ldo, unsigned long long* __restrict__ dbg,
    const u32x4* __restrict__ qf, NextQ nx, int st_sc1) {
  constexpr int HH = 256;
  constexpr int RGR = SPLIT ? 64 : V6_ROWS;  // real query rows per row group
  constexpr int NKB = 2 * HH / 16;           // 32 K-blocks of 16
  constexpr int ROWB = 4 * HH;               // 1 KiB per table row = one DMA piece
  constexpr int UNITB = V6_UT * ROWB;        // 32 KiB
  constexpr int NBUF = 4;
  constexpr int STG0 = NBUF * UNITB;         // staging: 2 x 4 blocks x [32 rows][32 cols] f32
  constexpr int STGW = 32 * V6_UT * 4;       // one consumer's block: 4 KiB
  constexpr int STGB = 4 * STGW;
  constexpr int SMEM = STG0 + 2 * STGB;      // 160 KiB
  constexpr int FR0 = 16;                    // query K-blocks requested before the first chain
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  const int b = blockIdx.x;
  const int q8 = b >> 3;
  const int rg = q8 % rgn;
  const int cg = (q8 / rgn) * 8 + (b & 7);  // workgroup b runs on XCD b % 8: a column range stays in one L2
  if (cg >= ncg) {
    if (nx.qf != nullptr && nx.mode == 1) {  // an idle workgroup: the next batch's query fragments
      const int spg = ((ncg + 7) & ~7) - ncg;
      v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)(rg * spg + (cg - ncg)) * 512 + threadIdx.x,
                                          (long long)nx.nblocks * 512);
    }
    return;
  }
  // units_per_cg > 0: the contiguous range [cg * units_per_cg, ...); 0: every ncg-th unit (score blocks beyond the
  // Infinity Cache with a sector-aligned pitch, see launch_v4)
  const int unit_lo = units_per_cg > 0 ? cg * units_per_cg : cg;
  const int unit_st = units_per_cg > 0 ? 1 : ncg;
  int NU = units_per_cg > 0 ? nunits - unit_lo : (nunits - cg + ncg - 1) / ncg;
  if (units_per_cg > 0 && NU > units_per_cg) NU = units_per_cg;
  if (NU <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool second = rg >= rgn1;  // two-sided launch: row groups [rgn1, rgn) are the (?, p, o) queries
  const int rgl = second ? rg - rgn1 : rg;
  if (second) out += out2_off;

  int dbg_i = 0;
  auto stamp = [&]() {  // optional per-phase timestamps (tools/prep_probe.py); dbg == NULL in production
    if (dbg != nullptr && tid == 0 && dbg_i < 32) dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  auto stamp_at = [&](int slot) {
    if (dbg != nullptr && lane == 0) dbg[(long long)blockIdx.x * 64 + slot] = __builtin_readcyclecounter();
  };
  stamp();  // 0: start

  // Barriers (all eight waves): R0; P(u), u = 0 .. NU - 1, which the consumers pass in slot V6_PB of chain u; F.
  //   R0    unit 0 has landed.
  //   P(u)  (a) unit u + 1 has landed -- the consumers start reading it in slot 24 of chain u, so that no read latency
  //             is exposed between two chains;
  //         (b) the scores of unit u - 1 are in staging[(u - 1) & 1] (written in slots V6_W0.. of chain u);
  //         (c) the store waves have read staging[u & 1] (unit u - 2; they did so behind P(u - 1));
  //         (d) the consumers are done with ring buffer (u - 1) % 4: the DMA waves refill it with unit u + 3.
  //   F     the last unit is staged.
  if (wave >= 4) {
    // =============================== loader waves ===============================
    const unsigned char* const tgb = (const unsigned char*)TG.base;
    const long long tld2 = TG.ld * 2;
    const unsigned int lane16 = (unsigned int)lane << 4;
    // rows [r0, r0 + CNT) of unit uu -> ring buffer uu % 4: one 1-KiB piece per row, lane l fetching the row's 16-byte
    // slot l ^ (row & 15) into slot l (the consumers read slot s of row fi at s ^ (fi & 15): conflict-free).  Rows
    // beyond the table repeat its last row (their scores are never stored).
    auto dma_rows = [&](int uu, int r0, auto cnt) __attribute__((always_inline)) {
      constexpr int CNT = decltype(cnt)::value;
      const long long row0 = (long long)(unit_lo + uu * unit_st) * V6_UT + r0;
      const unsigned int d0 = (unsigned int)((uu & (NBUF - 1)) * UNITB + r0 * ROWB);
      const unsigned int x0 = (unsigned int)(r0 & 15) << 4;
      // (scalar work per piece matters: the issuing wave is a pacemaker of the unit loop -- one pointer add and one
      // m0 value per piece in the common case; the clamped form only for the unit that reaches the end of the table)
      if (row0 + CNT <= m) {
        const unsigned char* p = tgb + row0 * tld2;
        v4_static_for<0, CNT>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          const unsigned int vo = lane16 ^ (x0 + (k << 4));
          const unsigned int dk = d0 + k * ROWB;
          const unsigned char* pk = p + k * tld2;
          KGE_V6_DMA(dk, vo, pk);
        });
      } else {
        v4_static_for<0, CNT>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          long long r = row0 + k;
          if (r >= m) r = m - 1;
          const unsigned char* pk = tgb + r * tld2;
          const unsigned int vo = lane16 ^ (x0 + (k << 4));
          const unsigned int dk = d0 + k * ROWB;
          KGE_V6_DMA(dk, vo, pk);
        });
      }
    };
    using C8 = std::integral_constant<int, 8>;
    using C16 = std::integral_constant<int, 16>;

    // ---- score path: staging -> registers -> HBM.  Lane (rq = lane >> 3, cl = lane & 7) moves columns 4 cl .. 4 cl + 3
    // of rows 8 i + rq of a consumer's [32][32] block: an instruction writes 8 rows x 128 contiguous bytes.
    const int cl = lane & 7, rq = lane >> 3;
    f32x4 cv[2][4];
    unsigned int svoff[2][4];
    unsigned char* out_rb[2];
    // which consumers' blocks this wave stores, per unit (store waves: two each; SPLIT: one REAL block = the sum of
    // the q_hi block (wave & 1) and the q_lo block (wave & 1) + 2) and for the last unit (all four loader waves: one)
    auto setup_blocks = [&](int w0, int nb) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < nb) {
          const int w = w0 + u;
          const long long r0 = (long long)rgl * RGR + 32 * w;
          const long long rb = r0 < n ? r0 : n - 1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            long long r = r0 + 8 * i + rq;
            if (r >= n) r = n - 1;  // padded rows repeat row n - 1 (as their query fragments do): same bytes
            svoff[u][i] = (unsigned int)((r - rb) * ldo * 4) + (unsigned int)(cl * 16);
          }
          out_rb[u] = (unsigned char*)(out + rb * ldo);
        }
      }
    };
    auto read_block = [&](int sb, int w, f32x4 (&dst)[4]) __attribute__((always_inline)) {
      const unsigned int a = (unsigned int)(STG0 + sb * STGB + w * STGW + rq * 128 + ((cl ^ rq) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const f32x4*>(smem + a + i * 1024);
    };
    auto store_block = [&](int uu, int u) __attribute__((always_inline)) {
      const long long col0 = (long long)(unit_lo + uu * unit_st) * V6_UT;
      if (col0 + V6_UT <= m) {
        unsigned char* sbase = out_rb[u] + col0 * 4;
        if (st_sc1) {
          // agent-scope write-through: the scores leave the L2 as they are written (launch_v4's measurements)
          const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, 0x7fffffff, 0x00020000);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cv[u][i]), srs, svoff[u][i], 0, 16);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4v6u*>(sbase + svoff[u][i]) = cv[u][i];
        }
      } else {  // ragged end of the table (the last unit of the last column group)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col0 + 4 * cl + e < m)
              *reinterpret_cast<float*>(out_rb[u] + (col0 + e) * 4 + svoff[u][i]) = cv[u][i][e];
      }
    };
    // the last unit: one block per loader wave (SPLIT: the two real blocks go to waves 6, 7)
    auto last_unit = [&]() __attribute__((always_inline)) {
      const int sb = (NU - 1) & 1;
      if constexpr (SPLIT) {
        if (wave < 6) return;
        setup_blocks(wave & 1, 1);
        read_block(sb, wave & 1, cv[0]);
        read_block(sb, (wave & 1) + 2, cv[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) cv[0][i] = cv[0][i] + cv[1][i];
        store_block(NU - 1, 0);
      } else {
        setup_blocks(wave - 4, 1);
        read_block(sb, wave - 4, cv[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_block(NU - 1, 0);
      }
    };

    const int l4 = wave - 4;  // ring fill: this wave's eight rows of units 0 and 1
    dma_rows(0, 8 * l4, C8{});
    if (NU > 1) dma_rows(1, 8 * l4, C8{});

    if (wave < 6) {
      // ------------------------------- DMA waves -------------------------------
      // VMEM queue of this wave, in order (LDS-DMA only, in-order returns): u0 (8), u1 (8) | behind P(u): u(u + 3)
      // (16).  A unit has landed once only the pieces issued behind it are outstanding.
      const int r16 = 16 * (wave & 1);
      if (wave == 4) stamp_at(32);  // first units issued
      if (NU > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (wave == 4) stamp_at(37);  // unit 0: this wave's pieces have landed
      KGE_BARRIER();  // R0
      // (unit 2 is the store waves', behind P(0): they have nothing to move before P(1), and whoever issues 16 more
      // pieces before P(0) holds the consumers there -- the pieces queue behind the query fragments in the
      // vector-memory path: P(0) passed at 5.1 k cycles instead of 3.7 k)
      for (int u = 0; u < NU; ++u) {
        // unit u + 1 (if any) has landed: behind it in this wave's queue only unit u + 2 (from u = 1 on)
        if (u >= 1 && u + 2 < NU) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 4 && u < 8) stamp_at(40 + u);  // arrival at P(u)
        KGE_BARRIER();  // P(u)
        if (u + 3 < NU) dma_rows(u + 3, r16, C16{});  // into the buffer of unit u - 1
      }
      KGE_BARRIER();  // F
      last_unit();
      if (wave == 4) stamp_at(36);  // last unit's stores issued
      return;
    }
    // ------------------------------- store waves -------------------------------
    if (wave == 6) stamp_at(38);
    setup_blocks(SPLIT ? (wave & 1) : 2 * (wave & 1), SPLIT ? 1 : 2);
    if (NU > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 6) stamp_at(39);
    KGE_BARRIER();  // R0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // its rows of unit 1
    if (wave == 6) stamp_at(48);
    KGE_BARRIER();  // P(0)
    if (NU > 2) dma_rows(2, 16 * (wave & 1), C16{});  // unit 2: due at P(1), ~30 MFMA slots away
    for (int u = 1; u < NU; ++u) {
      if (u == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // unit 2 (no store has been issued yet)
      if (wave == 6 && u < 8) stamp_at(48 + u);  // arrival at P(u)
      KGE_BARRIER();  // P(u): unit u - 1 is staged in staging[(u - 1) & 1]
      const int sb = (u - 1) & 1;
      if constexpr (SPLIT) {
        read_block(sb, wave & 1, cv[0]);
        read_block(sb, (wave & 1) + 2, cv[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) cv[0][i] = cv[0][i] + cv[1][i];  // score = (sum q_hi t) + (sum q_lo t)
        store_block(u - 1, 0);
      } else {
        read_block(sb, 2 * (wave & 1), cv[0]);
        read_block(sb, 2 * (wave & 1) + 1, cv[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        store_block(u - 1, 0);
        store_block(u - 1, 1);
      }
      if (u == 1 && wave == 6) stamp_at(33);  // first stores issued
    }
    KGE_BARRIER();  // F
    last_unit();
    if (wave == 6) stamp_at(34);  // last store issued
    if (wave == 6 && dbg != nullptr) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp_at(35);  // acknowledged
    }
    return;
  }

  // =================================== consumer waves ===================================
  const int w4 = wave;
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 afr[NKB];
  // the query fragments of this wave's 32 rows: K-block kb = 64 lanes x 16 B, written by a previous launch (or by
  // query_build_kernel): agent-scope loads (served by L2).  Compiler-visible, so that the vmcnt waits are placed
  // in front of the first MFMA that needs each fragment (straight-line code: the first unit is peeled).
  const unsigned char* const frag_base = (const unsigned char*)(qf + ((long long)(rg * (V6_ROWS / 32) + w4) * NKB) * 64);
  auto load_fragments = [&](auto lo, auto hi) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)frag_base, 0, NKB * 1024, 0x00020000);
    v4_static_for<decltype(lo)::value, decltype(hi)::value>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      afr[kb] = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, (unsigned int)(lane * 16 + kb * 1024), 0, 16 /* sc1 */));
    });
  };
  // (holding these loads back until the loader waves have issued unit 0 -- 0.25 k to 1 k cycles of s_sleep -- moves
  // nothing: unit 0 lands at ~3.3 k cycles either way; the cold first access, not the queue, sets that time)
  load_fragments(std::integral_constant<int, 0>{}, std::integral_constant<int, FR0>{});
  stamp();  // 1: first fragments requested
  // target fragment kb of row fi: the 16-byte slot 2 kb + fh, stored at slot ^ (fi & 15)
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  // staging: acc[4 g + e] = score(query fi, target 8 g + 4 fh + e) -> 16-byte chunk 2 g | fh of row fi, at chunk ^ (fi & 7)
  const unsigned int cwr = (unsigned int)(STG0 + w4 * STGW + fi * 128);
  const int y = fh ^ (fi & 7);
  auto c_write = [&](const f32x16& a, int sb, int g) __attribute__((always_inline)) {
    const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    *reinterpret_cast<f32x4*>(smem + cwr + (unsigned int)(sb * STGB) + (((2 * g) ^ y) << 4)) = v;
  };

  constexpr int PF = 8;
  bf16x8 bq[PF];  // fragment ring, continuous across units: the read for slot j of unit u lands in bq[j % 8]
  unsigned int bp[8];  // boff + the base of the ring buffer the next read goes to (moved on in slot 24 of a chain)
#pragma unroll
  for (int t = 0; t < 8; ++t) bp[t] = boff[t];
  auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256) : "memory");
  };
  f32x16 acc0, acc1;
  // chain u into `acc`; `prev` = the finished accumulators of unit u - 1 (FIRST: none), staged in slots V6_W0..;
  // the first eight fragment reads of unit u + 1 are issued in slots 24..31
  auto chain = [&](int u, f32x16& acc, const f32x16& prev, auto first) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first)::value;
    const unsigned int bn = (unsigned int)(((u + 1) & (NBUF - 1)) * UNITB);
    const int sbp = (u - 1) & 1;
    v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(v6_younger(kb, !FIRST)) : "memory");
      if constexpr (kb == V6_PB) KGE_BARRIER();  // P(u)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kb == 0) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], afr[0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % PF], afr[kb], acc, 0, 0, 0);
      }
      if constexpr (kb + PF == NKB) {  // this unit's reads are all issued: on to the next ring buffer
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("v_add_u32 %0, %1, %2" : "=v"(bp[t]) : "s"(bn), "v"(boff[t]));
      }
      bread(bq[kb % PF], std::integral_constant<int, (kb + PF) % NKB>{});
      if constexpr (!FIRST && kb >= V6_W0 && kb < V6_W0 + 4) c_write(prev, sbp, kb - V6_W0);
      // first unit: the query K-blocks FR0.. are requested from inside the chain, one per MFMA slot, 16 slots ahead
      // of their use (issued in front of the chain they would sit in the vector-memory queue before unit 0's pieces)
      if constexpr (FIRST && kb < NKB - FR0)
        load_fragments(std::integral_constant<int, FR0 + kb>{}, std::integral_constant<int, FR0 + kb + 1>{});
    });
    stamp();  // unit u: chain issued
  };
  using T = std::true_type;
  using Fz = std::false_type;
  KGE_BARRIER();  // R0: unit 0 landed
  stamp();  // 2
  v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
  chain(0, acc0, acc1, T{});
  {
    int u = 1;
    for (; u + 1 < NU; u += 2) {
      chain(u, acc1, acc0, Fz{});
      chain(u + 1, acc0, acc1, Fz{});
    }
    if (u < NU) chain(u, acc1, acc0, Fz{});
  }
  // the last unit -> staging, for the loaders' final pass
  {
    const int sb = (NU - 1) & 1;
    if ((NU - 1) & 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) c_write(acc1, sb, g);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) c_write(acc0, sb, g);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // (the look-ahead reads issued by the last chain land in bq[] until this wait: the registers stay "in use" up to
  // here, or the compiler hands them out as temporaries while a read is still on its way -- see pairs_bf16_v7_kernel)
  asm volatile("" : : "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]), "v"(bq[4]), "v"(bq[5]), "v"(bq[6]), "v"(bq[7]));
  KGE_BARRIER();  // F
  if (nx.qf != nullptr && nx.mode == 2)  // no idle workgroups in this geometry: a slice of the next batch's queries
    v4_build_queries<SCORER, HH, SPLIT>(nx, (long long)(rg * ncg + cg) * 256 + tid, (long long)nx.nblocks * 256);
}

// ---------------------------------------------------------------------------------------------------------------
// pairs_bf16_v7_kernel: the same launch with the scores stored STRAIGHT from the accumulators (one rounded query
// vector per row; split queries keep the staged path above: their two partial blocks live in different waves).
//
// Why.  In the kernel above the two DMA waves are the pacemaker of the unit loop: 16 LDS-DMA pieces per unit and wave
// at ~90 cycles each = 1.44 k cycles, against 1.26 k for the consumers' own loop (tools/ubench/unit_loop.hip) --
// the pieces share the vector-memory path with the store waves' 1-KiB stores.  Measured alternatives
// (profiles/r3_unit_loop_ubench.txt, r3_wide_loop_ubench.txt): a dword store per MFMA slot costs the consumer
// 3 cycles per MFMA (36.6 against 33.6 bare; the staging writes + their barrier discipline cost 5.8), two 128-byte
// row segments per instruction.  So: no staging buffers, no store waves -- consumer wave w stores element r of
// unit u - 1 (row 8 (r >> 2) + 4 fh + (r & 3) of its 32, column = lane & 31) in slot 2 r of chain u, and all FOUR
// loader waves stream the table, 8 pieces per unit each (their queues hold loads only: counted waits stay valid).
// Rows beyond n are dropped by the buffer descriptor's range, columns beyond m (ragged last unit) by a per-lane
// out-of-range offset.  Bits: the same accumulation chains -- identical to v6 / v4.
//   barriers (all eight waves): R0 (unit 0 landed); P(u) in slot V6_PB of chain u: unit u + 1 has landed and the
//   consumers are done with ring buffer (u - 1) % 4, which the loaders refill with unit u + 3.
template <int SCORER, int SC1, int PROBE = 0>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v7_kernel(
    Operand TG, long long n, long long m, int rgn, int rgn1, long long out2_off, int ncg, int units_per_cg,
    int nunits, float* __restrict__ out, long long ldo, unsigned long long* __restrict__ dbg,
    const u32x4* __restrict__ qf, NextQ nx, int probe) {
  constexpr int HH = 256;
  constexpr int RGR = V6_ROWS;
  constexpr int NKB = 2 * HH / 16;
  constexpr int ROWB = 4 * HH;
  constexpr int UNITB = V6_UT * ROWB;
  constexpr int NBUF = 4;
  constexpr int SMEM = NBUF * UNITB;  // 128 KiB
  constexpr int FR0 = 16;
  // the ring: LDS bytes [0, 128 KiB), addressed by the DMA pieces (m0) and the fragment reads (ds_read_b128 in asm)
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  if (n < 0) smem[threadIdx.x] = 0;  // (never: keeps the allocation -- nothing else names the array)

  const int b = blockIdx.x;
  const int q8 = b >> 3;
  const int rg = q8 % rgn;
  const int cg = (q8 / rgn) * 8 + (b & 7);
  if (cg >= ncg) {
    if (nx.qf != nullptr && nx.mode == 1) {
      const int spg = ((ncg + 7) & ~7) - ncg;
      v4_build_queries<SCORER, HH, 0>(nx, (long long)(rg * spg + (cg - ncg)) * 512 + threadIdx.x,
                                      (long long)nx.nblocks * 512);
    }
    return;
  }
  const int unit_lo = units_per_cg > 0 ? cg * units_per_cg : cg;
  const int unit_st = units_per_cg > 0 ? 1 : ncg;
  int NU = units_per_cg > 0 ? nunits - unit_lo : (nunits - cg + ncg - 1) / ncg;
  if (units_per_cg > 0 && NU > units_per_cg) NU = units_per_cg;
  if (NU <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool second = rg >= rgn1;
  const int rgl = second ? rg - rgn1 : rg;
  if (second) out += out2_off;

  int dbg_i = 0;
  auto stamp = [&]() {
    if (dbg != nullptr && tid == 0 && dbg_i < 32) dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  auto stamp_at = [&](int slot) {
    if (dbg != nullptr && lane == 0) dbg[(long long)blockIdx.x * 64 + slot] = __builtin_readcyclecounter();
  };
  stamp();  // 0: start

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const unsigned char* const tgb = (const unsigned char*)TG.base;
    const long long tld2 = TG.ld * 2;
    const unsigned int lane16 = (unsigned int)lane << 4;
    const int r8 = 8 * (wave - 4);  // this wave's eight rows of every unit
    auto dma8 = [&](int uu) __attribute__((always_inline)) {
      const long long row0 = (long long)(unit_lo + uu * unit_st) * V6_UT + r8;
      const unsigned int d0 = (unsigned int)((uu & (NBUF - 1)) * UNITB + r8 * ROWB);
      const unsigned int x0 = (unsigned int)(r8 & 15) << 4;
      if (row0 + 8 <= m) {
        const unsigned char* p = tgb + row0 * tld2;
        v4_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          const unsigned int vo = lane16 ^ (x0 + (k << 4));
          const unsigned int dk = d0 + k * ROWB;
          const unsigned char* pk = p + k * tld2;
          KGE_V6_DMA(dk, vo, pk);
        });
      } else {  // rows beyond the table repeat its last row (their scores are never stored)
        v4_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          long long r = row0 + k;
          if (r >= m) r = m - 1;
          const unsigned char* pk = tgb + r * tld2;
          const unsigned int vo = lane16 ^ (x0 + (k << 4));
          const unsigned int dk = d0 + k * ROWB;
          KGE_V6_DMA(dk, vo, pk);
        });
      }
    };
    // this wave's queue (loads only, in-order returns): u0 (8), u1 (8) | P(0) | u2, u3 | P(1) | u4 | P(2) | u5 ...
    dma8(0);
    if (NU > 1) dma8(1);
    if (wave == 4) stamp_at(32);  // first units issued
    if (NU > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 4) stamp_at(37);  // unit 0: this wave's pieces have landed
    KGE_BARRIER();  // R0
    for (int u = 0; u < NU; ++u) {
      // unit u + 1 has landed: behind it in this wave's queue only unit u + 2 (from u = 1 on, while it exists)
      if (u >= 1 && u + 2 < NU) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (wave == 4 && u < 8) stamp_at(40 + u);  // arrival at P(u)
      if (wave == 6 && u < 8) stamp_at(48 + u);
      if constexpr (!(PROBE & 8)) KGE_BARRIER();  // P(u)
      if (u == 0 && NU > 2) dma8(2);
      if constexpr (!(PROBE & 2))
        if (u + 3 < NU) dma8(u + 3);  // into the buffer of unit u - 1
    }
    return;
  }

  // =================================== consumer waves ===================================
  const int w4 = wave;
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 afr[NKB];
  const unsigned char* const frag_base = (const unsigned char*)(qf + ((long long)(rg * (V6_ROWS / 32) + w4) * NKB) * 64);
  auto load_fragments = [&](auto lo, auto hi) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)frag_base, 0, NKB * 1024, 0x00020000);
    v4_static_for<decltype(lo)::value, decltype(hi)::value>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      afr[kb] = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, (unsigned int)(lane * 16 + kb * 1024), 0, 16 /* sc1 */));
    });
  };
  load_fragments(std::integral_constant<int, 0>{}, std::integral_constant<int, FR0>{});
  stamp();  // 1: first fragments requested
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  // ---- stores.  The MFMA operands are swapped against v6 (queries first, targets second: the same products summed
  // in the same K order, the tile transposed), so that a lane holds ONE target and sixteen query rows:
  // acc[r] = score(row 8 (r >> 2) + 4 fh + (r & 3) of this wave's 32, target fi of the unit).  One
  // instruction = element r of all lanes = two rows x 128 contiguous bytes.  Descriptor over this wave's rows that
  // exist: a store to a padded row (>= n) falls outside its range and is dropped by the hardware (the range check
  // sees the per-lane offset = row and target; the unit's column offset travels in the scalar offset).
  const long long rb = (long long)rgl * RGR + 32 * w4;
  long long rows_here = rb < n ? (n - rb < 32 ? n - rb : 32) : 0;
  if (probe & 1) rows_here = 0;  // tools/r4_diag.py (KGE_V7_NOSTORE=1): every store falls outside the descriptor
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(out + (rows_here > 0 ? rb : 0) * ldo), 0, (int)(rows_here * ldo * 4), 0x00020000);
  const unsigned int svo = (unsigned int)(((long long)(4 * fh) * ldo + fi) * 4);
  const unsigned int ldo4 = (unsigned int)(ldo * 4);
  auto store_elem = [&](const f32x16& a, auto rc, unsigned int vo, unsigned int colb) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    // the row in the per-lane offset (one v_add per store): gfx9 bounds-checks the VGPR offset only, the scalar
    // offset (the unit's column) is added unchecked
    const unsigned int vr = vo + (unsigned int)(8 * (r >> 2) + (r & 3)) * ldo4;
    const float v = a[r];  // (a copy first: __builtin_bit_cast straight on the vector element stored element 0 every time)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), srs, vr, colb, SC1 ? 16 : 0);
  };
  // per-lane offset for unit uu: out of range for the columns >= m of the ragged last unit
  auto unit_vo = [&](int uu, unsigned int& colb) __attribute__((always_inline)) -> unsigned int {
    const long long col0 = (long long)(unit_lo + uu * unit_st) * V6_UT;
    colb = (unsigned int)(col0 * 4);
    return (col0 + V6_UT <= m || col0 + fi < m) ? svo : 0x80000000u;
  };

  constexpr int PF = 8;
  bf16x8 bq[PF];
  unsigned int bp[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bp[t] = boff[t];
  auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256) : "memory");
  };
  f32x16 acc0, acc1;
  auto chain = [&](int u, f32x16& acc, const f32x16& prev, auto first) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first)::value;
    const unsigned int bn = (unsigned int)(((u + 1) & (NBUF - 1)) * UNITB);
    unsigned int colb = 0;
    const unsigned int vo = FIRST ? 0u : unit_vo(u - 1, colb);
    v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      if constexpr (kb == V6_PB && !(PROBE & 8)) KGE_BARRIER();  // P(u)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kb == 0) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bq[0], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kb], bq[kb % PF], acc, 0, 0, 0);
      }
      if constexpr (kb + PF == NKB) {
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("v_add_u32 %0, %1, %2" : "=v"(bp[t]) : "s"(bn), "v"(boff[t]));
      }
      if constexpr (!(PROBE & 4)) bread(bq[kb % PF], std::integral_constant<int, (kb + PF) % NKB>{});
      if constexpr (!FIRST && (kb & 1) == 0 && !(PROBE & 1)) store_elem(prev, std::integral_constant<int, kb / 2>{}, vo, colb);
      if constexpr (FIRST && kb < NKB - FR0)
        load_fragments(std::integral_constant<int, FR0 + kb>{}, std::integral_constant<int, FR0 + kb + 1>{});
    });
    stamp();  // unit u: chain issued
  };
  using T = std::true_type;
  using Fz = std::false_type;
  KGE_BARRIER();  // R0: unit 0 landed
  stamp();  // 2
  v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
  chain(0, acc0, acc1, T{});
  {
    int u = 1;
    for (; u + 1 < NU; u += 2) {
      chain(u, acc1, acc0, Fz{});
      chain(u + 1, acc0, acc1, Fz{});
    }
    if (u < NU) chain(u, acc1, acc0, Fz{});
  }
  {  // the last unit, straight away
    unsigned int colb = 0;
    const unsigned int vo = unit_vo(NU - 1, colb);
    if ((NU - 1) & 1) {
      v4_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) { store_elem(acc1, rc, vo, colb); });
    } else {
      v4_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) { store_elem(acc0, rc, vo, colb); });
    }
  }
  // The look-ahead reads behind the last unit return into bq[] whenever the LDS gets to them.  Nobody uses what they
  // return -- which is exactly why the registers must be kept: to the compiler an asm output is there at once, a dead
  // one is free at once, and it put the address of the slot-30 store into such a register (the read then landed on
  // top of it: a store dropped or misdirected now and then, only with several launches in flight).  The empty asm
  // below "uses" the eight registers AFTER the wait that drains the reads.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("" : : "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]), "v"(bq[4]), "v"(bq[5]), "v"(bq[6]), "v"(bq[7]));
  if (w4 == 0) stamp_at(34);  // last store issued
  if (nx.qf != nullptr && nx.mode == 2)
    v4_build_queries<SCORER, HH, 0>(nx, (long long)(rg * ncg + cg) * 256 + tid, (long long)nx.nblocks * 256);
}

static std::atomic<unsigned long long*> g_v6_stamps{nullptr};
void v6_set_stamps(unsigned long long* p) { g_v6_stamps.store(p); }
unsigned long long* v6_get_stamps() { return g_v6_stamps.load(std::memory_order_relaxed); }

static int v6_cu_count() {
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

template <int SCORER, int SPLIT>
static int launch_v6(const Operand& TG, bool two_sided, long long n, long long m, float* out, long long ldo,
                     long long out2_off, hipStream_t st, unsigned long long* dbg, const void* qf, NextQ nx,
                     int reserve_cus) {
  constexpr int RGR = SPLIT ? 64 : V6_ROWS;
  const int rgn1 = (int)((n + RGR - 1) / RGR);
  const int rgn = two_sided ? 2 * rgn1 : rgn1;
  const int nunits = (int)((m + V6_UT - 1) / V6_UT);
  int cus = v6_cu_count() - reserve_cus;
  if (cus > 256) cus = 256;
  if (cus < 8) cus = 8;
  // the geometry of launch_v4: workgroup b runs on XCD b % 8 = the low bits of its column group; whole groups of
  // eight column groups that fit the compute units (prepared queries: no co-residency requirement, any grid goes)
  int ncg = cus / rgn;
  if (ncg > 8) ncg = 8 * (cus / 8 / rgn > 0 ? cus / 8 / rgn : 1);
  if (ncg < 1) ncg = 1;
  int upc = (nunits + ncg - 1) / ncg;
  if (upc < 1) upc = 1;
  ncg = (nunits + upc - 1) / upc;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  if (ldo >= (1LL << 24)) return KGE_ERR_UNSUPPORTED;
  if (dbg == nullptr) dbg = g_v6_stamps.load(std::memory_order_relaxed);
  if (nx.qf != nullptr) {
    const int spare = rgn * ((((ncg + 7) / 8) * 8) - ncg);
    if (spare >= 4) {
      nx.mode = 1;
      nx.nblocks = spare;
    } else {
      nx.mode = 2;
      nx.nblocks = rgn * ncg;
    }
  }
  const long long il = sw(SW_V4_INTERLEAVE);
  const bool interleave =
      il >= 0 ? il == 1 : ((double)n * (double)m * 4.0 * (two_sided ? 2 : 1) > 192e6 && (ldo & 7) == 0);
  const long long sc1e = sw(SW_V4_STORE_SC1);
  const bool st_aligned = (ldo & 7) == 0 && (out2_off & 7) == 0 && ((uintptr_t)out & 31) == 0;
  const bool st_small = (double)n * (double)m * 4.0 * (two_sided ? 2 : 1) <= 48e6;
  // write-through only for sector-aligned rows: this kernel's 128-byte row segments straddle a sector at each end
  // otherwise, and a partial sector written through is a read-modify-write at the memory (FB15k-237 shape, contiguous
  // pitch, two-sided: 27.2 us written through, 21.1 us through the L2's write-back; aligned: 20.3 / 20.6)
  const int st_sc1 = sc1e >= 0 ? (sc1e != 0) : (st_aligned ? 1 : 0);
  (void)st_small;
  if constexpr (!SPLIT) {
    // one rounded query vector per row: scores stored straight from the accumulators (pairs_bf16_v7_kernel);
    // KGE_V7=0: the staged kernel (A/B measurements)
    // (one-sided launches into rows that are not sector-aligned stay with the staged kernel: 12.4 against 13.3 us at
    // the FB15k-237 shape -- dword stores of unaligned 128-byte segments; KGE_V7=1 takes v7 there too)
    const long long e7 = sw(SW_V7);
    const int probe = sw(SW_V7_NOSTORE) == 1 ? 1 : 0;
    if (e7 != 0 && (st_aligned || two_sided || e7 == 1)) {
#ifdef KGE_V7_PROBES  // make CXXEXTRA=-DKGE_V7_PROBES: compile-time timing variants (tools/r4_diag2.py); wrong scores
      const int prb = sw(SW_V7_PROBE) > 0 ? (int)sw(SW_V7_PROBE) : 0;
#define KGE_V7P(PB)                                                                                               \
  if (prb == PB) {                                                                                               \
    hipLaunchKernelGGL((pairs_bf16_v7_kernel<SCORER, 0, PB>), dim3(grid), dim3(512), 0, st, TG, n, m, rgn, rgn1,  \
                       out2_off, ncg, interleave ? 0 : upc, nunits, out, ldo, dbg, (const u32x4*)qf, nx, probe);  \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                            \
  }
      KGE_V7P(1) KGE_V7P(2) KGE_V7P(3) KGE_V7P(4) KGE_V7P(5) KGE_V7P(7) KGE_V7P(8) KGE_V7P(9) KGE_V7P(15)
#undef KGE_V7P
#endif
      if (st_sc1)
        hipLaunchKernelGGL((pairs_bf16_v7_kernel<SCORER, 1>), dim3(grid), dim3(512), 0, st, TG, n, m, rgn, rgn1,
                           out2_off, ncg, interleave ? 0 : upc, nunits, out, ldo, dbg, (const u32x4*)qf, nx, probe);
      else
        hipLaunchKernelGGL((pairs_bf16_v7_kernel<SCORER, 0>), dim3(grid), dim3(512), 0, st, TG, n, m, rgn, rgn1,
                           out2_off, ncg, interleave ? 0 : upc, nunits, out, ldo, dbg, (const u32x4*)qf, nx, probe);
      return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL((pairs_bf16_v6_kernel<SCORER, SPLIT>), dim3(grid), dim3(512), 0, st, TG, n, m, rgn, rgn1,
                     out2_off, ncg, interleave ? 0 : upc, nunits, out, ldo, dbg, (const u32x4*)qf, nx, st_sc1);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Scores of prepared queries `qf` (fragment order of v4_build_queries) against the identity-indexed bf16 table TG, d = 512.
// KGE_ERR_UNSUPPORTED: not this kernel's case (the caller takes pairs_bf16_v4_kernel).  KGE_V6=0 declines everything
// (A/B measurements).
int run_pairs_bf16_v6(int scorer, bool split, const Operand& TG, bool two_sided, int d, long long n, long long m,
                      float* out, long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg,
                      const void* qf, const NextQ& nx, int reserve_cus) {
  if (d != 512 || TG.idx.ptr != nullptr || qf == nullptr) return KGE_ERR_UNSUPPORTED;
  if (sw(SW_V6) == 0) return KGE_ERR_UNSUPPORTED;
  if (TG.ld * 2 >= (1LL << 28)) return KGE_ERR_UNSUPPORTED;
#define KGE_V6L(SC)                                                                                            \
  return split ? launch_v6<SC, 1>(TG, two_sided, n, m, out, ldo, out2_off, st, dbg, qf, nx, reserve_cus)       \
               : launch_v6<SC, 0>(TG, two_sided, n, m, out, ldo, out2_off, st, dbg, qf, nx, reserve_cus)
  if (scorer == KGE_COMPLEX) { KGE_V6L(KGE_COMPLEX); }
  if (scorer == KGE_DISTMULT) { KGE_V6L(KGE_DISTMULT); }
#undef KGE_V6L
  return KGE_ERR_UNSUPPORTED;
}

#undef KGE_V6_DMA
}  // namespace kge
