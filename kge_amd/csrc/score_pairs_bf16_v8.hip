// score_pairs_bf16_v8.hip -- ComplEx / DistMult sp_/_po scores of bf16 tables at d = 512 from PREPARED query
// fragments: the store path of the BASELINE.json headline configuration, round 4.  One launch scores a GROUP of
// equally shaped batches (kge_score_queries_multi; a group of one = kge_score_queries).
//
// What round 4 measured on pairs_bf16_v7_kernel (tools/r4_diag.py, r4_diag2.py; profiles/r4_diag*.txt):
//   * its unit period (32 targets x 128 query rows) is 1.44 k cycles with the stores issued -- also when the hardware
//     DROPS every one of them, also without the table DMA -- and 1.15 k cycles (36 per MFMA, the matrix pipe's own
//     rate) once the store INSTRUCTIONS are compiled out: a buffer_store_dword holds its wave ~45 cycles at the
//     vector-memory port, the consumer wave issues in order, and with one consumer wave per SIMD every cycle it
//     stands there the matrix pipe of that SIMD idles;
//   * per launch ~5.4 k cycles pass before the first accumulation chain, 28 of 256 compute units get no workgroup,
//     and a dependent launch starts ~1.5 us after its predecessor.
//
// This kernel therefore
//   (1) puts TWO consumer waves on every SIMD: eight waves of 32 query rows (a workgroup = 256 rows), each doing ALL
//       three jobs for its rows -- ds_read_b128 + v_mfma_f32_32x32x16_bf16 (queries in 128 operand registers), the
//       direct dword stores of the unit before (v7's transposed tile: a lane holds one target and sixteen rows, one
//       instruction = two rows x 128 contiguous bytes) and four of a unit's 32 LDS-DMA pieces.  While one wave of a
//       SIMD stands at the memory port the other one feeds the matrix pipe; a table unit is streamed into LDS once
//       per 256 rows instead of once per 128;
//   (2) is PERSISTENT: 8 x (CUs / 8) workgroups; XCD x (= blockIdx % 8, a placement the hardware is observed to keep
//       and nothing depends on) owns the column slice x of the table -- 1/8 of the units, ~1.9 MB at the FB15k-237
//       shape: it stays in that XCD's L2 --, the (batch, side, 256-row chunk) pairs x the slice's units form one
//       list per XCD, and workgroup j of the XCD walks a contiguous 1/32 of it, streaming units across pair
//       boundaries (at a boundary the waves flush their last unit and load the next pair's fragments; the table
//       stream and the LDS ring never stop).  Cold start, launch gap and idle units are paid once per group.
//
// Synchronisation: one workgroup barrier P(k) per unit, in MFMA slot V8_PB of chain k.  Behind it: unit k + 1 has
// landed in ring buffer (k + 1) % 4 (every wave waited for ITS pieces: s_waitcnt vmcnt(N), N = the vector-memory
// operations it has issued since -- stores and younger pieces; gfx9 retires a wave's VMEM operations in issue
// order) and everybody is done reading ring buffer (k - 1) % 4, which the pieces of unit k + 3 then overwrite
// (slots 17, 21, 25, 29).  The instruction stream of a chain is the same for every unit -- a chain without a unit
// before it stores through an out-of-range offset, a chain without a unit three ahead re-requests the list's last
// unit into the free buffer -- so that N is one constant.
//
// Bits: one accumulation chain per score in K order = pairs_bf16_v4/v6/v7 and the oracle's bf16 mode.  Split
// queries (KGE_FLAG_SPLIT_QUERY): the q_hi and q_lo rows of 16 real rows are the 32 operand rows of a wave, arranged
// so that both partial scores of a (row, target) sit in ONE lane (elements r and r + 4): score = hi + lo, one add,
// eight stores per unit -- the bits of pairs_bf16_v6_kernel<SPLIT>.  Fragment buffers: the layout of
// v4_build_queries (bf16_queries.hpp), unchanged.
#include "common.hpp"
#include "bf16_queries.hpp"
#include <atomic>
#include <cstdlib>

namespace kge {

constexpr int V8_UT = 32;   // targets per unit
constexpr int V8_BAND_CAP = 255;  // pairs a wave can list per (side, 256-row chunk): a list = 1 + 255 records of 16 bytes
constexpr int V8_PB = 14;   // MFMA slot of the barrier
constexpr unsigned int V8_DROP = 0x80000000u;  // per-lane store offset beyond every descriptor range

struct V8Args {
  Operand TG;
  long long n, m;          // rows per batch and side, targets
  int rgn1;                // 128-row fragment groups per side (v4_build_queries' layout)
  int sides;               // 1 / 2 (two-sided: the second side's fragment groups and score block follow the first's)
  int chunks;              // workgroup chunks (two fragment groups) per side
  int nbatch;              // batches in the group
  long long q_stride;      // 16-byte words between the fragments of two batches
  long long out_stride;    // floats between the score blocks of two batches
  long long out2_off, ldo;
  int nunits, su, wpx;     // units in all, per XCD slice; workgroups per XCD
  float* out;
  const u32x4* qf;
  unsigned long long* dbg;
  NextQ nx;
};

#define KGE_V8_DMA(D, VO, P) \
  KGE_STALL(__LINE__ + 7000); \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(D), "v"(VO), "s"(P) : "memory", "m0")

template <int SCORER, int SPLIT, int AUX, int VAR = 0>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v8_kernel(V8Args a) {
  constexpr int HH = 256;
  constexpr int NKB = 2 * HH / 16;        // 32 K-blocks of 16
  constexpr int ROWB = 4 * HH;            // 1 KiB per table row = one DMA piece
  constexpr int UNITB = V8_UT * ROWB;     // 32 KiB
  constexpr int NBUF = 4;
  constexpr int SMEM = NBUF * UNITB;      // 128 KiB
  constexpr int RW = SPLIT ? 16 : 32;     // real query rows per wave
  constexpr int NST = SPLIT ? 8 : 16;     // stores per unit and wave, in slots 0, 2, ..
  // LDS reads in flight per wave.  Split queries: 4 -- the 16 registers eight cost were six spilled VGPRs in this
  // instantiation (scratch reloads, each behind a compiler-made s_waitcnt vmcnt(0), at every pair boundary of a kernel
  // whose table stream is ordered by counted waits); four slots of >= 32 cycles still cover the LDS latency, as in
  // pairs_bf16_v8_rank_kernel.  (32 % PF == 0: a read issued in slot kb lands in bq[kb % PF] for slot kb + PF.)
  constexpr int PF = SPLIT ? 4 : 8;
  static_assert(NKB % PF == 0 && PF <= 8, "look-ahead depth");
  constexpr int FR0 = 12;  // query K-blocks requested before the first chain (cold start)
  // vector-memory operations a wave issues between the last piece of unit k + 1 (slot 29 of chain k - 2) and the wait
  // in slot V8_PB of chain k: the stores behind slot 29, chain k - 1 (stores + 4 pieces), the stores of slots < V8_PB
  // (steady state: 28 / 19 plain / split).  The chain behind the cold one waits for unit 2, requested in slots 1 - 13 of
  // the cold chain, behind which 9 / 1 stores + 6 fragment loads + 4 pieces + 7 stores = 26 / 18 operations follow: two
  // less than the steady count serves every chain (the two extra operations waited for are a unit old)
  constexpr int VMN_STEADY = (NST > 15 ? 1 : 0) + NST + 4 + V8_PB / 2;
  constexpr int VMN = VMN_STEADY - 2;
  static_assert(VMN <= (NST > 15 ? 9 : 1) + (NKB - FR0 - 14) + 4 + V8_PB / 2, "the chain behind the cold chain");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  if (a.n < 0) smem[threadIdx.x] = 0;  // (never: keeps the allocation -- only asm names the array)

  const int b = blockIdx.x;
  const int x = b & 7, j = b >> 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int u_lo = x * a.su;
  int sux = a.nunits - u_lo;
  if (sux > a.su) sux = a.su;
  const int P = a.nbatch * a.sides * a.chunks;
  // This workgroup's list: `nseg` segments of `sux` units each -- segment k = pair pair0 + k * pstep over the units
  // [u_lo, u_lo + sux) -- walked as positions g in [0, nseg * sux): pair pair0 + (g / sux) * pstep, unit u_lo + g % sux.
  //   P >= workgroups per XCD: workgroup j takes the pairs j, j + wpx, ... over the XCD's whole slice;
  //   fewer pairs: the slice is cut into wpx / P sub-ranges and workgroup j takes pair j % P over sub-range j / P.
  // Either way the workgroups of an XCD that run side by side stream the SAME table units at about the same time: a
  // unit comes from HBM once per XCD and P - 1 times from its L2 (a Wikidata5M shard does not fit the Infinity Cache).
  const int g0 = 0;
  int g1 = 0, pair0 = 0, pstep = a.wpx;
  if (sux > 0) {
    if (P >= a.wpx) {
      pair0 = j;
      g1 = j < P ? ((P - j + a.wpx - 1) / a.wpx) * sux : 0;
    } else {
      const int nsub = a.wpx / P;
      if (j < P * nsub) {
        pair0 = j % P;
        const int sub = j / P;
        const int lo = (int)((long long)sux * sub / nsub), hi = (int)((long long)sux * (sub + 1) / nsub);
        u_lo += lo;
        sux = hi - lo;
        g1 = sux;  // one segment
      }
    }
  }

  int dbg_i = 0;
  auto stamp = [&]() {  // optional per-phase timestamps (tools/r4_diag.py); dbg == NULL in production
    if (a.dbg != nullptr && tid == 0 && dbg_i < 32) a.dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  auto stamp_at = [&](int slot) {
    if (a.dbg != nullptr && tid == 0) a.dbg[(long long)blockIdx.x * 64 + slot] = __builtin_readcyclecounter();
  };

  if (g1 > g0) {
    stamp();  // 0: start
    // ---------------- the table stream: this wave's four rows of every unit ----------------
    const unsigned char* const tgb = (const unsigned char*)a.TG.base;
    const long long tld2 = a.TG.ld * 2;
    const long long m = a.m;
    const unsigned int lane16 = (unsigned int)lane << 4;
    const int r4 = 4 * wave;
    // lane l fetches the row's 16-byte slot l ^ (row & 15) into slot l (the consumers read slot s of row fi at
    // s ^ (fi & 15): conflict-free); rows beyond the table repeat its last row (their scores are never stored)
    const unsigned int dx0 = (unsigned int)((r4 & 15) << 4);
    int dq = g0;             // list position of the next unit to request
    int du = g0 % sux;       // ... its unit within the slice
    const int ulast = (g1 - 1) % sux;
    // unit `un` of the slice -> ring buffer `ks & 3`, piece `kk`
    auto dma_piece = [&](int un, int ks, auto kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(kc)::value;
      long long r = (long long)(u_lo + un) * V8_UT + r4 + kk;
      if (r >= m) r = m - 1;
      const unsigned char* pk = tgb + r * tld2;
      const unsigned int dk = (unsigned int)((ks & (NBUF - 1)) * UNITB + (r4 + kk) * ROWB);
      const unsigned int vo = lane16 ^ (dx0 + (unsigned int)(kk << 4));
      KGE_V8_DMA(dk, vo, pk);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
      ++dq;
      if (++du == sux) du = 0;
    };
    // ring fill: units g0, g0 + 1, g0 + 2 (positions beyond the range: the last unit again, into buffers nobody
    // reads), interleaved with the first pair's fragment loads below: unit 0 | K-blocks 0-15 | unit 1 | 16-31 | unit 2
    auto fill_unit = [&](int k) __attribute__((always_inline)) {
      const int un = dq < g1 ? du : ulast;
      v4_static_for<0, 4>([&](auto kc) __attribute__((always_inline)) { dma_piece(un, k, kc); });
      dma_advance();
    };

    // ---------------- the consumer ----------------
    const int fi = lane & 31, fh = lane >> 5;
    bf16x8 afr[NKB];
    unsigned int bp[8];  // read addresses of the current ring buffer; moved on by one buffer per chain
    // target fragment kb of row fi: the 16-byte slot 2 kb + fh, stored at slot ^ (fi & 15)
#pragma unroll
    for (int t = 0; t < 8; ++t) bp[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
    bf16x8 bq[PF];
    auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      const unsigned int addr = bp[kb & 7];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256) : "memory");
    };
    // ---- stores.  MFMA(queries, targets): acc[r] = score(operand row 8 (r >> 2) + 4 fh + (r & 3), target fi).
    //   plain queries: operand row = query row of this wave's 32;
    //   split queries: operand row 16 a + 8 part + jj = real row 8 a + jj of this wave's 16, part 0 = q_hi, 1 = q_lo:
    //     elements r (r & 4 == 0) and r + 4 are the two partial scores of real row 8 (r >> 3) + 4 fh + (r & 3).
    // The row part of a store's address travels in the per-lane offset (gfx9 bounds-checks the VGPR offset only): rows
    // >= n fall outside the descriptor and are dropped by the hardware, the unit's column goes in the scalar offset.
    const unsigned int ldo4 = (unsigned int)(a.ldo * 4);
    const unsigned int svo = (unsigned int)(((long long)(4 * fh) * a.ldo + fi) * 4);
    __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);
    auto store_q = [&](const f32x16& acc, auto qc, unsigned int vo, unsigned int colb) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;  // q-th store of the unit
      constexpr int r = SPLIT ? (q & 3) + 8 * (q >> 2) : q;
      constexpr int rowc = SPLIT ? 8 * (r >> 3) + (r & 3) : 8 * (r >> 2) + (r & 3);
      const unsigned int vr = vo + (unsigned int)rowc * ldo4;
      float v = acc[r];  // (a copy first: __builtin_bit_cast straight on the vector element stored element 0 every time)
      if constexpr (SPLIT) {
        const float lo = acc[r + 4];
        v = v + lo;  // score = (sum q_hi t) + (sum q_lo t)
      }
      if constexpr (VAR & 1) {  // PROBE: the score computed, the store instruction left out
        asm volatile("" : : "v"(v), "v"(vr));
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), srs, vr, colb, AUX);
      }
    };
    // per-lane offset of the unit at slice position `cu`: out of range for the columns >= m of the ragged last unit
    auto unit_vo = [&](int cu, unsigned int& colb) __attribute__((always_inline)) -> unsigned int {
      const long long col0 = (long long)(u_lo + cu) * V8_UT;
      colb = (unsigned int)(col0 * 4);
      return (col0 + V8_UT <= m || col0 + fi < m) ? svo : V8_DROP;
    };

    f32x16 acc0, acc1;
    // chain of the unit in ring buffer ks & 3 into `acc`; `prev` (offset vo, column colb) is stored on the way.
    // `cold` (the workgroup's very first chain): the ring holds units 0 and 1 only and the fragments K-blocks < FR0 --
    // unit 2 is requested in slots 1, 5, 9, 13, K-block FR0 + j in slot j (FR0 slots ahead of its MFMA); `frag` loads it
    auto chain = [&](int ks, f32x16& acc, const f32x16& prev, unsigned int vo, unsigned int colb, auto cold, auto&& frag)
        __attribute__((always_inline)) {
      constexpr bool COLD = decltype(cold)::value;
      // the next ring buffer: + one unit, or back to the first one (unsigned wrap-around)
      const unsigned int bdelta = ((ks + 1) & (NBUF - 1)) ? (unsigned int)UNITB : (unsigned int)(-(NBUF - 1) * UNITB);
      int un = dq < g1 ? du : ulast;  // requested by this chain: the unit three ahead (cold: first the unit two ahead)
      v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kc)::value;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(PF - 1) : "memory");
        if constexpr (kb == V8_PB) {
          // this wave's pieces of unit ks + 1 have landed (cold: requested right behind R0; since then 7 stores, 14
          // fragment loads and unit 2's pieces of slots 1 - 13)
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(COLD ? 25 : VMN) : "memory");
          KGE_BARRIER();  // P(ks)
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kb == 0) {
          const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bq[0], zero, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kb], bq[kb % PF], acc, 0, 0, 0);
        }
        if constexpr (kb + PF == NKB) {  // this unit's reads are all issued: on to the next ring buffer
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("v_add_u32 %0, %1, %0" : "+v"(bp[t]) : "s"(bdelta));
        }
        bread(bq[kb % PF], std::integral_constant<int, (kb + PF) % NKB>{});
        if constexpr ((kb & 1) == 0 && kb / 2 < NST) store_q(prev, std::integral_constant<int, kb / 2>{}, vo, colb);
        if constexpr (COLD && kb < NKB - FR0) frag(std::integral_constant<int, FR0 + kb>{});
        if constexpr (COLD && kb < 14 && (kb & 3) == 1) dma_piece(un, ks + 2, std::integral_constant<int, kb / 4>{});
        if constexpr (COLD && kb == 14) {
          dma_advance();
          un = dq < g1 ? du : ulast;
        }
        if constexpr (kb >= 17 && ((kb - 17) & 3) == 0) dma_piece(un, ks + 3, std::integral_constant<int, (kb - 17) / 4>{});
      });
      dma_advance();
      stamp();  // chain issued
    };
    using Cold = std::true_type;
    using Warm = std::false_type;
    auto nofrag = [](auto) {};

    int g = g0, ks = 0;
    int pair = pair0;
    const int cu = 0;
    bool first = true;
    while (g < g1) {
      const int cnt = sux;
      // ---- the pair: batch lb, side, chunk of 8 x RW rows -> this wave's rows and fragments
      const int per = a.sides * a.chunks;
      const int lb = pair / per, rem = pair - lb * per;
      const int side = rem / a.chunks, ch = rem - side * a.chunks;
      const long long rb = (long long)ch * (8 * RW) + RW * wave;  // first row (of the side) of this wave
      long long rows_here = rb < a.n ? (a.n - rb < RW ? a.n - rb : RW) : 0;
      float* const ob = a.out + (long long)lb * a.out_stride + (side ? a.out2_off : 0) + (rows_here > 0 ? rb : 0) * a.ldo;
      srs = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, (int)(rows_here * a.ldo * 4), 0x00020000);
      // fragments: groups of 128 operand rows (4 blocks of 32 rows x 32 K-blocks x 1 KiB); a chunk = two groups
      int grp = 2 * ch + (wave >> 2);
      if (grp >= a.rgn1) grp = a.rgn1 - 1;  // (a side with an odd number of groups: rows_here == 0 there)
      grp += side * a.rgn1;
      const unsigned char* const gbase = (const unsigned char*)(a.qf + (long long)lb * a.q_stride + (long long)grp * 4 * NKB * 64);
      // (compiler-visible loads: the waits land in front of the first MFMA that needs each fragment -- the first chain
      // of a pair is straight-line code behind them)
      unsigned int flo;
      const unsigned char* fb;
      int frange;
      if constexpr (SPLIT) {
        // operand row fi = 16 a + 8 part + jj of wave w: real row 16 (w & 3) + 8 a + jj of the group's 64, whose q_hi
        // sits in block (row >> 5), its q_lo in block 2 + (row >> 5)
        const int part = (fi >> 3) & 1, rr = 16 * (wave & 3) + 8 * (fi >> 4) + (fi & 7);
        flo = (unsigned int)((((2 * part + (rr >> 5)) * NKB) * 64 + (rr & 31) + 32 * fh) * 16);
        fb = gbase;
        frange = 4 * NKB * 1024;
      } else {
        flo = (unsigned int)(lane * 16);
        fb = gbase + (wave & 3) * (NKB * 1024);
        frange = NKB * 1024;
      }
      const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)fb, 0, frange, 0x00020000);
      auto load_fragments = [&](auto lo, auto hi) __attribute__((always_inline)) {
        v4_static_for<decltype(lo)::value, decltype(hi)::value>([&](auto kc) __attribute__((always_inline)) {
          constexpr int kb = decltype(kc)::value;
          afr[kb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, flo + kb * 1024, 0, 16 /* sc1 */));
        });
      };
      using C0 = std::integral_constant<int, 0>;
      using CF = std::integral_constant<int, FR0>;
      using C32 = std::integral_constant<int, NKB>;
      unsigned int pvo = V8_DROP, pcolb = 0;  // the unit before: none yet in this pair
      // the pair's first chain is peeled: straight-line code behind the fragment loads
      if (first) {
        // cold start: what is in the vector-memory queue in front of the first chain delays it (8 waves x 1 KiB per
        // instruction through one 64 B/clk port: ~16 cycles each) -- unit 0 and FR0 K-blocks only; R0; unit 1; the rest
        // of the fragments and unit 2 from inside the first chain
        fill_unit(0);
        load_fragments(C0{}, CF{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(FR0) : "memory");  // unit 0's four pieces: this wave's oldest operations
        KGE_BARRIER();  // R0: unit g0 has landed
        stamp();  // 1
        fill_unit(1);
        v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
        first = false;
        chain(ks, acc0, acc1, pvo, pcolb, Cold{}, [&](auto kc) __attribute__((always_inline)) {
          load_fragments(kc, std::integral_constant<int, decltype(kc)::value + 1>{});
        });
      } else {
        load_fragments(C0{}, C32{});
        chain(ks, acc0, acc1, pvo, pcolb, Warm{}, nofrag);
      }
      pvo = unit_vo(cu, pcolb);
      ++ks;
      int i = 1;
      for (; i + 1 < cnt; i += 2) {
        chain(ks, acc1, acc0, pvo, pcolb, Warm{}, nofrag);
        pvo = unit_vo(cu + i, pcolb);
        chain(ks + 1, acc0, acc1, pvo, pcolb, Warm{}, nofrag);
        pvo = unit_vo(cu + i + 1, pcolb);
        ks += 2;
      }
      if (i < cnt) {
        chain(ks, acc1, acc0, pvo, pcolb, Warm{}, nofrag);
        pvo = unit_vo(cu + i, pcolb);
        ++ks;
        v4_static_for<0, NST>([&](auto qc) __attribute__((always_inline)) { store_q(acc1, qc, pvo, pcolb); });
      } else {
        v4_static_for<0, NST>([&](auto qc) __attribute__((always_inline)) { store_q(acc0, qc, pvo, pcolb); });
      }
      g += cnt;
      pair += pstep;
    }
    // The look-ahead reads behind the last unit return into bq[] whenever the LDS gets to them.  Nobody uses what they
    // return -- which is exactly why the registers must be kept: to the compiler an asm output is there at once and a
    // dead one is free at once (pairs_bf16_v7_kernel lost stores that way).  The empty asm "uses" them AFTER the wait.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (PF == 8)
      asm volatile("" : : "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]), "v"(bq[4]), "v"(bq[5]), "v"(bq[6]), "v"(bq[7]));
    else
      asm volatile("" : : "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]));
    stamp_at(34);  // last store issued
  }
  // the NEXT group's query fragments: a slice per workgroup, behind its last unit
  if (a.nx.qf != nullptr)
    v4_build_queries<SCORER, HH, SPLIT>(a.nx, (long long)blockIdx.x * 512 + threadIdx.x, (long long)gridDim.x * 512);
}

void v6_set_stamps(unsigned long long* p);
unsigned long long* v6_get_stamps();

int run_pairs_bf16_v8_store256(const Operand& TG, bool split, bool two_sided, long long n, long long m, int nbatch,
                               const void* qf, long long q_stride_bytes, float* out, long long out_stride, long long ldo,
                               long long out2_off, int sc1, int reserve_cus, hipStream_t st);

static int v8_cu_count() {
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

// launches issued by this process (kge_debug_launch_count): 0 = pairs_bf16_v8_kernel, 1 = pairs_bf16_v8_rank_kernel
static int g_v8_launches[2] = {0, 0};
int v8_launch_count(int which) { return (which == 0 || which == 1) ? __atomic_load_n(&g_v8_launches[which], __ATOMIC_RELAXED) : -1; }

template <int SCORER, int SPLIT>
static int launch_v8(V8Args& a, int sc1, hipStream_t st) {
  int cus = v8_cu_count();
  (void)cus;
  __atomic_fetch_add(&g_v8_launches[0], 1, __ATOMIC_RELAXED);
  const dim3 grid(8 * a.wpx), block(512);
  // cache policy of the score stores: 0 plain (write-back: the lines stay in the XCD's L2), 16 sc1 (write-through, the
  // line leaves the L2), 2 nt, 18 sc1 nt
#ifdef KGE_V8_PROBES
  // PROBE build only (make CXXEXTRA=-DKGE_V8_PROBES; tools/split_var_probe.py): KGE_V8_VAR=1 = the split-query kernel with
  // its store INSTRUCTIONS compiled out (profiles/r5_split_store_probe.txt: 222 -> 194 us per group of eight two-sided
  // batches -- the stores cost 13 %, the rest of the distance to the bare matrix pipe is not theirs)
  if constexpr (SCORER == KGE_COMPLEX && SPLIT == 1) {
    if (sw(SW_V8_VAR) == 1) {
      hipLaunchKernelGGL((pairs_bf16_v8_kernel<SCORER, SPLIT, 2, 1>), grid, block, 0, st, a);
      return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
    }
  }
#endif
  if (sc1 == 1)
    hipLaunchKernelGGL((pairs_bf16_v8_kernel<SCORER, SPLIT, 16>), grid, block, 0, st, a);
  else if (sc1 == 2)
    hipLaunchKernelGGL((pairs_bf16_v8_kernel<SCORER, SPLIT, 2>), grid, block, 0, st, a);
  else if (sc1 == 3)
    hipLaunchKernelGGL((pairs_bf16_v8_kernel<SCORER, SPLIT, 18>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((pairs_bf16_v8_kernel<SCORER, SPLIT, 0>), grid, block, 0, st, a);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Scores of `nbatch` prepared batches (fragments of batch l at qf + l * q_stride_bytes, n rows per side each) against
// the identity-indexed bf16 table TG (d = 512, all rows or a contiguous slice) into out + l * out_stride (floats).
// nx.qf != NULL: the launch also builds the fragments nx describes (the next group).  KGE_ERR_UNSUPPORTED: not this
// kernel's case (the caller takes pairs_bf16_v6 / v4).  KGE_V8=0 declines everything (A/B measurements).
int run_pairs_bf16_v8(int scorer, bool split, const Operand& TG, bool two_sided, int d, long long n, long long m,
                      int nbatch, const void* qf, long long q_stride_bytes, float* out, long long out_stride,
                      long long ldo, long long out2_off, hipStream_t st, unsigned long long* dbg, const NextQ& nx,
                      int reserve_cus) {
  if ((d != 512 && d != 256) || TG.idx.ptr != nullptr || qf == nullptr || nbatch < 1) return KGE_ERR_UNSUPPORTED;
  // A single batch stays with pairs_bf16_v7 / v6 (a workgroup of this kernel loads the fragments of 256 rows before
  // its first chain -- 9-12 k cycles against 5 k there -- and a single batch's list gives it 7 units to amortise them
  // over: FB15k-237 shape two-sided 22.0 us against 19.2, profiles/r4_v8_probe.txt); KGE_V8=1 takes it here too.
  const long long e = sw(SW_V8);
  if (e == 0) return KGE_ERR_UNSUPPORTED;
  if (nbatch == 1 && e != 1) return KGE_ERR_UNSUPPORTED;
  if (TG.ld * 2 >= (1LL << 28) || ldo >= (1LL << 24)) return KGE_ERR_UNSUPPORTED;
  if ((q_stride_bytes & 15) || ((uintptr_t)qf & 15)) return KGE_ERR_INVALID_ARG;
  const long long rgr = split ? 64 : 128;  // real rows per fragment group
  const long long rgn1 = (n + rgr - 1) / rgr;
  const long long pairs = (long long)nbatch * (two_sided ? 2 : 1) * ((rgn1 + 1) / 2);
  const long long nunits = (m + V8_UT - 1) / V8_UT;
  if (rgn1 > (1 << 20) || pairs * nunits >= (1LL << 30)) return KGE_ERR_UNSUPPORTED;
  int cus = v8_cu_count() - reserve_cus;
  if (cus > 256) cus = 256;
  if (cus < 8) cus = 8;
  V8Args a{};
  a.TG = TG;
  a.n = n;
  a.m = m;
  a.rgn1 = (int)rgn1;
  a.sides = two_sided ? 2 : 1;
  a.chunks = (int)((rgn1 + 1) / 2);
  a.nbatch = nbatch;
  a.q_stride = q_stride_bytes / 16;
  a.out_stride = out_stride;
  a.out2_off = out2_off;
  a.ldo = ldo;
  a.nunits = (int)nunits;
  a.su = (int)((nunits + 7) / 8);
  a.wpx = cus / 8;
  a.out = out;
  a.qf = (const u32x4*)qf;
  a.dbg = dbg != nullptr ? dbg : v6_get_stamps();
  a.nx = nx;
  // write-through (sc1) stores only for sector-aligned rows and score blocks that fit the Infinity Cache comfortably:
  // tools/r4_diag.py -- a rotation of buffers beyond it runs faster through the L2's write-back
  const long long sc1e = sw(SW_V4_STORE_SC1);
  const bool st_aligned = (ldo & 7) == 0 && (out2_off & 7) == 0 && (out_stride & 7) == 0 && ((uintptr_t)out & 31) == 0;
  const double bytes = (double)nbatch * (double)n * (double)m * 4.0 * (two_sided ? 2 : 1);
  // (tools/v8_policy_probe.py, profiles/r4_v8_policy.txt: a group's blocks beyond ~160 MB stream to HBM -- `nt` keeps
  // them from evicting the table slice from the L2: 22.0 -> 14.7 us per two-sided batch in a group of eight; up to
  // ~48 MB write-through wins by the end-of-kernel write-back it saves; in between it makes no difference)
  // Rows that do not start on a 32-byte sector (a contiguous [n, E] block with E = 14,541: the one-call entries) leave
  // a partial sector at both ends of every 128-byte row segment a store instruction writes; the neighbouring unit's
  // store completes it one chain later -- in the L2, if the line is still there: plain write-back stores.  `nt` and
  // write-through push the partial sector out and the memory side reads, merges and writes it (n = 2048 two-sided,
  // 238 MB: 226 us with nt against 112 for the single-batch kernel's plain stores, tools/one_call_v8_probe.py).
  const int sc1 = sc1e >= 0 ? (int)sc1e : (!st_aligned ? 0 : (bytes > 160e6 ? 2 : (bytes <= 48e6 ? 1 : 0)));
  if (d == 256) {
    // d = 256 (single-pass and split queries): the parametric structure of ce_pairs_v8.hip with a score-store epilogue (the
    // kernel below is scheduled by hand for d = 512's 32-slot chains).  No in-launch build of the next group there.
    if (nx.qf != nullptr) return KGE_ERR_INVALID_ARG;
    // cache policy of its stores (tools/d256_store_probe.py, profiles/r6_d256_store_policies.txt): plain write-back up to
    // ~1 GB of scores per launch (FB15k-237 shape, 8 one-sided batches = 238 MB: 5.4 us per batch against 5.9 write-
    // through and 6.1-7.0 nt), write-through beyond (a Wikidata5M shard, 2.3 GB: 291 us per batch against 315)
    const int pol = sc1e >= 0 ? (int)(sc1e & 3) : (st_aligned && bytes > 1e9 ? 1 : 0);
    const int rc = run_pairs_bf16_v8_store256(TG, split, two_sided, n, m, nbatch, qf, q_stride_bytes, out, out_stride, ldo,
                                              out2_off, pol, reserve_cus, st);
    if (rc == KGE_OK) __atomic_fetch_add(&g_v8_launches[0], 1, __ATOMIC_RELAXED);
    return rc;
  }
#define KGE_V8L(SC) return split ? launch_v8<SC, 1>(a, sc1, st) : launch_v8<SC, 0>(a, sc1, st)
  if (scorer == KGE_COMPLEX) { KGE_V8L(KGE_COMPLEX); }
  if (scorer == KGE_DISTMULT) { KGE_V8L(KGE_DISTMULT); }
#undef KGE_V8L
  return KGE_ERR_UNSUPPORTED;
}


// ---------------------------------------------------------------------------------------------------------------
// pairs_bf16_v8_rank_kernel: the SAME launch shape with the scores COUNTED instead of stored -- the counts of
// EntityRankingJob._filter_and_rank / _get_ranks_and_num_ties (kge/job/eval_entity_ranking.py:533-596) for the entity
// slice, without the [n, 2 m] score matrix (kge_score_rank_sp_po, kge_eval_batch).  Round 3's counting epilogue sat in
// pairs_bf16_v4_kernel behind ONE consumer wave per SIMD, which alternated a chain with ~1 k cycles of comparisons while
// the matrix pipe idled (0.17 / 0.25 of the bf16 peak at the FB15k-237 / Wikidata5M-shard shapes, pipe 32 % busy).  Here
// every SIMD has two consumer waves: one wave's comparisons of unit k - 1 -- issued between the MFMAs of its own chain k,
// one element per slot (d = 256) / per two slots (d = 512) -- run under the other wave's MFMAs.
//
// Orientation: MFMA(targets, queries) as in pairs_bf16_v4: a lane owns ONE query row (operand row fi) and the unit's
// targets 8 (r >> 2) + 4 fh + (r & 3), so that the row's true score, tolerance, filter words and counters are per-lane
// state (v8's store orientation would need a cross-lane reduction per element).  Same products, same K order: the
// counted scores are the bits the store kernels write.
//
// Split queries (KGE_FLAG_SPLIT_QUERY: the parity-compliant evaluation mode): operand row fi = 16 a + 8 part + jj holds
// q_hi (part 0) / q_lo (part 1) of real row 8 a + jj of the wave's 16; the full score is acc(fi) + acc(fi ^ 8) = one
// DPP add (row_ror:8) per element, (sum q_hi t) + (sum q_lo t) as pairs_bf16_v8_kernel<SPLIT> stores it; the part-1
// lanes compute the same values and count nothing.
//
// Vector-memory operations of a wave per chain, ALL in inline asm so that the counted waits are exact: two filter-word
// loads in slot 1 (this unit's words, used behind the NEXT chain; a launch with fewer than two filter sets loads a
// dummy word) and NP table pieces behind the barrier.  No stores.
struct V8RankArgs {
  Operand TG;
  long long n, m;
  int rgn1, sides, chunks;
  int nunits, su, wpx;
  const u32x4* qf;
  unsigned long long* dbg;
  CeArgs ce;
};

// PROBE (builds with -DKGE_V8_PROBES only; KGE_V8R_PROBE picks one): timing variants that leave work out -- bit 0 the
// comparisons, 1 the table pieces of the steady state, 2 the unit barrier, 3 the filter-word loads, 4 the LDS reads
//
// BAND (round 6): band-and-rescore, first launch -- the parity-compliant counts at close to the
// single-pass price.  The fragments are the SPLIT set's (groups of 64 real rows [hi | hi | lo | lo]); the chains run on
// the q_hi blocks only: the single-pass score x_hi, the hi half of the split score x = fl(x_hi + x_lo) bit for bit.
// |x - x_hi| <= |x_lo| + an ulp and |x_lo| <= ||q_lo_i|| max_j ||t_j|| (Cauchy-Schwarz; ||q_lo_i|| from the row's lo
// block at the pair's start, the table's largest row norm from kge_table_max_row_norm), so with the row's tolerance
// widened by that bound every score OUTSIDE the widened band compares with the true score under x_hi as it does under
// x: greater ones are counted here, smaller ones ignored.  A score INSIDE the band (a trained model's true score sits
// in the far tail of its row: ~2e-5 of the pairs, profiles/r5_band_fraction.txt) is not counted: the lane appends the
// PAIR -- (row within the wave's 32, column, the row's filter bits of that column) -- to a list that belongs to this
// wave and this (side, 256-row chunk) alone: a running count in a scalar register, one 16-byte store per pair, no
// atomic, nothing to wait for (the workgroup's other seven waves stand at the next unit's barrier meanwhile: two
// forms that rescored the tile on the spot lost 200 us on a Wikidata5M shard to exactly that --
// profiles/r6_rank_band_inline_*_form.txt).  pairs_bf16_rescore_kernel gathers the listed columns' table rows 32 at a
// time and counts them with both chains.
template <int SCORER, int HH, int SPLIT, int PROBE = 0, int BAND = 0>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v8_rank_kernel(V8RankArgs a) {
  static_assert(!BAND || !SPLIT, "the band launch runs the q_hi chains only");
  // A unit = NACC sub-units of 32 table rows, one accumulator each: one at d = 512, TWO at d = 256 (64 rows x 512
  // bytes: the same 32 KiB).  With two accumulators consecutive MFMAs are independent (A0 B0 A1 B1 ...): what stands
  // between them -- LDS reads, table pieces, word loads -- no longer breaks a back-to-back dependent issue (measured
  // on the one-accumulator d = 256 form, tools/rank8_stamps.py with a -DKGE_V8_PROBES build: 2.5 k cycles per 32
  // columns against 0.6 k of bare MFMAs), and barrier, waits and pieces are paid once per 64 columns.
  // Split queries (q = q_hi + q_lo) come in two forms.  d = 256: TWO fragment sets and two accumulators per lane
  // (PARTS = 2: 128 operand registers, what d = 512 needs for one set): one LDS read feeds the q_hi and the q_lo MFMA,
  // the table streams once per 256 real rows, the comparisons run once per score, and the two MFMAs of a K-block are
  // independent.  d = 512 has no registers for a second set: the q_hi and q_lo rows of 16 real rows are the 32
  // operand rows of a wave and the partial scores meet by a DPP add (DUP).  Both give fl(sum q_hi t) + fl(sum q_lo t).
  constexpr bool DUP = SPLIT && HH == 256;
  constexpr int PARTS = (SPLIT && HH == 128) ? 2 : 1;
  constexpr int FSETS = PARTS;            // fragment sets a lane holds
  constexpr int NT = (HH == 128 && !SPLIT) ? 2 : 1;  // 32-row sub-units of a unit
  constexpr int NACC = NT * PARTS;                   // accumulators of a chain
  constexpr int UT = V8_UT * NT;          // table rows per unit: 32 / 64
  constexpr int NKB = 2 * HH / 16;        // 32 / 16 K-blocks
  constexpr int ROWB = 4 * HH;            // bytes per table row
  constexpr int SPR = ROWB / 16;          // 16-byte slots per row: 64 / 32
  constexpr int RPP = 64 / SPR;           // table rows per 1-KiB piece: 1 / 2
  constexpr int UNITB = UT * ROWB;        // 32 KiB
  constexpr int SUBB = V8_UT * ROWB;      // bytes of a sub-unit
  constexpr int NBUF = 4;
  constexpr int SMEM = NBUF * UNITB;
  constexpr int NP = UNITB / 1024 / 8;    // pieces per unit and wave: 4
  constexpr int RW = DUP ? 16 : 32;       // real query rows per wave
  constexpr int PF = 4;                   // K-blocks read ahead (8 at d = 512 cost 16 registers the kernel spilled)
  constexpr int PB = NKB == 32 ? 14 : 6;  // K-block of the barrier
  // rank_unit_raw for sub-units without a filtered column: at d = 256, where the comparisons weigh half as much as
  // the matrix work (at d = 512 the second form costs registers the kernel does not have: scratch, +3 us per launch)
  constexpr bool RAWFAST = HH == 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  if (a.n < 0) smem[threadIdx.x] = 0;

  const int b = blockIdx.x;
  const int x = b & 7, j = b >> 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int u_lo = x * a.su;
  int sux = a.nunits - u_lo;
  if (sux > a.su) sux = a.su;
  const int P = a.sides * a.chunks;
  // the workgroup's list of (pair, unit) positions: see pairs_bf16_v8_kernel
  const int g0 = 0;
  int g1 = 0, pair0 = 0, pstep = a.wpx;
  if (sux > 0) {
    if (P >= a.wpx) {
      pair0 = j;
      g1 = j < P ? ((P - j + a.wpx - 1) / a.wpx) * sux : 0;
    } else {
      const int nsub = a.wpx / P;
      if (j < P * nsub) {
        pair0 = j % P;
        const int sub = j / P;
        const int lo = (int)((long long)sux * sub / nsub), hi = (int)((long long)sux * (sub + 1) / nsub);
        u_lo += lo;
        sux = hi - lo;
        g1 = sux;
      }
    }
  }
  if (g1 <= g0) return;
  int dbg_i = 0;
  auto stamp = [&]() {
    if (a.dbg != nullptr && tid == 0 && dbg_i < 32) a.dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  stamp();
  if (a.dbg != nullptr && tid == 0) a.dbg[(long long)blockIdx.x * 64 + 42] = wall_clock64();  // 100 MHz, real time

  // ---------------- the table stream ----------------
  const unsigned char* const tgb = (const unsigned char*)a.TG.base;
  const long long tld2 = a.TG.ld * 2;
  const long long m = a.m;
  const int lr = lane / SPR, slot = lane % SPR;
  const int rp0 = wave * NP * RPP;  // this wave's first row of every unit
  int dq = g0, du = g0 % sux;
  const int ulast = (g1 - 1) % sux;
  auto dma_piece = [&](int un, int ks, auto kc) __attribute__((always_inline)) {
    constexpr int kk = decltype(kc)::value;
    const int ru = rp0 + kk * RPP;  // the piece's first row within the unit
    const long long r0 = (long long)(u_lo + un) * UT + ru;
    const unsigned int dk = (unsigned int)((ks & (NBUF - 1)) * UNITB + ru * ROWB);
    // lane (lr, slot) fetches the 16-byte slot `slot ^ (row & 15)` of row ru + lr into slot `slot` of its LDS row
    const unsigned int sw = (unsigned int)((slot ^ ((ru + lr) & 15)) << 4);
    // rows beyond the table repeat its last row (never counted): the scalar base clamped to row m - 1, the lane's
    // row within the piece clamped against what is left behind the base
    const long long rb = r0 < m ? r0 : m - 1;
    const unsigned char* pk = tgb + rb * tld2;
    unsigned int vo = sw;
    if constexpr (RPP > 1) {
      const int left = (int)(m - 1 - rb < RPP - 1 ? m - 1 - rb : RPP - 1);
      vo += (unsigned int)((lr < left ? lr : left) * (int)tld2);
    }
    KGE_V8_DMA(dk, vo, pk);
  };
  auto dma_advance = [&]() __attribute__((always_inline)) {
    ++dq;
    if (++du == sux) du = 0;
  };
#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k) {
    const int un = dq < g1 ? du : ulast;
    v4_static_for<0, NP>([&](auto kc) __attribute__((always_inline)) { dma_piece(un, k, kc); });
    dma_advance();
  }

  // ---------------- the consumer ----------------
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 afr[FSETS][NKB];
  unsigned int bp[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bp[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  bf16x8 bq[NT][PF];
  auto bread = [&](bf16x8& dst, auto kc, auto ac) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value, sub = decltype(ac)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256 + sub * SUBB) : "memory");
  };

  // ---- per-row state of the counts (rank.hip's arithmetic; see pairs_bf16_v4_kernel<V3_RANK>)
  const CeArgs& ce = a.ce;
  const bool counts_here = DUP ? ((fi >> 3) & 1) == 0 : true;  // DUP: the q_lo lanes duplicate their q_hi lane
  float rk_t = 0.0f, rk_al = 0.0f;
  float rk_hi = 0.0f, rk_lo = 0.0f;  // exact thresholds of the raw counts: see rank_unit_raw
  int bl_cnt = 0;                    // BAND: pairs this wave has listed for the current (side, chunk)
  u32x4* bl_list = nullptr;          // ... and its list (header + V8_BAND_CAP entries)
  bool rk_slow = false, rk_noraw = false;
  int rk_g = 0, rk_c = 0, rk_fg[2] = {0, 0}, rk_fc[2] = {0, 0}, rk_fn[2] = {0, 0};
  // filter words: scalar base per filter set (this side's bits; dummy: any readable word) + the row's byte offset
  const unsigned char* rk_base = (const unsigned char*)a.qf;
  unsigned int rk_off = 0;
  int orow_cur = 0;
  int side_cur = 0;
  // bits of a lane's 16 elements within the unit's 32 columns: element r at bit 8 (r >> 2) + 4 fh + (r & 3)
  auto rk_spread = [](unsigned int d) __attribute__((always_inline)) -> unsigned int {
    return (d & 0xfu) | ((d & 0xf0u) << 4) | ((d & 0xf00u) << 8) | ((d & 0xf000u) << 12);
  };
  auto full_score = [&](float v) __attribute__((always_inline)) -> float {
    if constexpr (DUP) {
      const float o = __builtin_bit_cast(
          float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
      return v + o;  // (sum q_hi t) + (sum q_lo t): both lanes of the pair hold it (f32 addition commutes)
    } else {
      return v;
    }
  };
  // the finished sub-unit `sub` of the unit at slice position `cu` into the counters; w0 / w1 = the row's filter words
  // -> the lane's "close" columns of the sub-unit (BAND: the columns inside the widened band, which are NOT counted)
  auto rank_unit = [&](const f32x16& pv, int cu, int sub, unsigned int w0, unsigned int w1, bool count_g = true)
      __attribute__((always_inline)) -> unsigned int {
    unsigned int g, c;
    if (BAND && rk_slow) {  // no finite band for some row of the wave: every pair of the tile goes to the second launch
      g = 0u;
      c = 0x0f0f0f0fu;
    } else if (rk_slow) {  // some row of the wave has an infinite true score / tolerance: the generic arithmetic
      g = c = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int g0_ = 0, c0_ = 0;
        count_one(full_score(pv[r]), rk_t, ce.rk_atol, ce.rk_rtol, g0_, c0_);
        const unsigned int bit = 1u << (8 * (r >> 2) + (r & 3));
        g |= g0_ ? bit : 0u;
        c |= c0_ ? bit : 0u;
      }
    } else {
      // finite true score, finite tolerance >= 0:  close <=> |x - t| <= allowed,  greater-and-not-close <=>
      // x - t > allowed  (NaN and -inf scores fail both, +inf is greater: what count_one gives).  As sign bits:
      // x' = max(x, -inf) (NaN -> -inf), e = x' - t, sign(allowed - e) = greater, sign(allowed - |e|) = NOT close
      unsigned int ng = 0u, nc = 0u;
#pragma unroll
      for (int r = 15; r >= 0; --r) {  // element r ends up at bit r
        // max(x, -inf): NaN -> -inf.  One v_max_f32 (fmaxf compiles to two: it first quiets a signalling NaN, which
        // an accumulator never holds)
        float xq;
        asm("v_max_f32 %0, 0xff800000, %1" : "=v"(xq) : "v"(full_score(pv[r])));
        const float e = xq - rk_t;
        ng = __builtin_amdgcn_alignbit(ng, __builtin_bit_cast(unsigned int, rk_al - e), 31);
        nc = __builtin_amdgcn_alignbit(nc, __builtin_bit_cast(unsigned int, rk_al - __builtin_fabsf(e)), 31);
      }
      g = rk_spread(ng & 0xffffu);
      c = rk_spread(~nc & 0xffffu);
    }
    const long long c0t = ((long long)(u_lo + cu) * NT + sub) * V8_UT;
    unsigned int mine = counts_here ? (0x0f0f0f0fu << (4 * fh)) : 0u;
    const long long rem = m - c0t;
    if (rem < V8_UT) mine &= rem > 0 ? (1u << rem) - 1u : 0u;  // (the unit exists; its second sub-unit may not)
    g = (g << (4 * fh)) & mine;
    c = (c << (4 * fh)) & mine;
    if (!count_g) return c;  // (BAND, behind rank_unit_raw: the greater ones are counted, the band's columns wanted)
    rk_g += __builtin_popcount(g);
    if constexpr (!BAND) rk_c += __builtin_popcount(c);
    const unsigned int ww[2] = {w0 & mine, w1 & mine};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < ce.rk_nfilt) {
        rk_fg[k] += __builtin_popcount(g & ww[k]);
        if constexpr (!BAND) rk_fc[k] += __builtin_popcount(c & ww[k]);
        rk_fn[k] += __builtin_popcount(ww[k]);
      }
    }
    return c;
  };
  // The raw counts alone, for a sub-unit without a filtered column in any row of the wave (all but ~1 % of them on a
  // Wikidata5M shard): two compares and two carry-adds per score instead of seven operations.  x - t rounds
  // monotonically in x, so  x - t > allowed  <=>  x > rk_hi  and  |x - t| <= allowed  <=>  rk_lo <= x <= rk_hi  for
  // rk_hi = the largest float with fl(rk_hi - t) <= allowed, rk_lo = the smallest with fl(rk_lo - t) >= -allowed
  // (found per row at the pair's start and CHECKED there -- a row whose thresholds do not verify sends its wave down
  // the generic path); NaN fails both compares like the -inf it stands for, +inf is greater, -inf nothing.
  auto rank_unit_raw = [&](const f32x16& pv) __attribute__((always_inline)) -> int {
    int g = 0, c2 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = full_score(pv[r]);
      g += x > rk_hi ? 1 : 0;
      c2 += x >= rk_lo ? 1 : 0;
    }
    rk_g += g;
    if constexpr (!BAND) rk_c += c2 - g;
    return c2 - g;  // (BAND: the lane's scores inside the widened band)
  };
  // the row's counters out (the two lanes fh = 0 / 1 of a row first), then zeroed
  auto rank_flush = [&]() __attribute__((always_inline)) {
    const int G = rk_g + __shfl_xor(rk_g, 32, 64), C = rk_c + __shfl_xor(rk_c, 32, 64);
    const bool wr = counts_here && orow_cur < a.n && fh == 0;
    unsigned long long* rank = ce.rk_rank[side_cur] + orow_cur;
    unsigned long long* ties = ce.rk_ties[side_cur] + orow_cur;
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
      rk_fg[k] = rk_fc[k] = rk_fn[k] = 0;
    }
    rk_g = rk_c = 0;
  };
  auto load_words = [&](int cu, u32x2 (&w)[NT]) __attribute__((always_inline)) {
    // the row's filter words of the unit at slice position cu, BOTH filter sets in one 8-byte load: scalar base (the
    // side's bits + the sub-unit's word column) + the row's offset (api.hip rank_bits_layout: word-major, the sets
    // of a side interleaved).  One set: the second word is the next row's (never looked at); none: 8 bytes of the
    // fragments.
#pragma unroll
    for (int sub = 0; sub < NT; ++sub) {
      const unsigned char* wb = rk_base + (ce.rk_nfilt > 0 ? ((long long)(u_lo + cu) * NT + sub) * ce.rk_bits_us * 4 : 0);
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(w[sub]) : "v"(rk_off), "s"(wb) : "memory");
    }
  };

  // The consumer loop, in two phase-shifted forms.  Per unit a wave runs its MFMA chain and then a BURST of ~140 VALU
  // operations (the comparisons: interleaved into the chain they cost more than on their own -- an instruction
  // between two MFMAs on the same accumulator breaks their back-to-back issue: measured 3.05 k cycles per unit against
  // 2.15 k of matrix time at d = 512).  The burst of one wave is meant to run under the chain of the OTHER wave of its
  // SIMD (waves w and w + 4), so the two halves of the workgroup stand half a period apart: waves 0-3 pass the unit's
  // barrier in the middle of their chain, waves 4-7 at the start of theirs.  The barrier's promises hold for both
  // (see pairs_bf16_v8_kernel): whoever passes P(k) is done reading unit k - 1, and every wave has waited for its own
  // pieces of unit k + 1 -- which nobody reads before slot NKB - PF of chain k.
  auto run = [&](auto half) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half)::value;
    constexpr int PBH = HALF ? 0 : PB;              // slot of the barrier
    // vector-memory operations in order, per chain: its NP pieces (behind the barrier / in K-blocks 1 .. NP), then --
    // behind the last MFMA -- the word loads of the NEXT unit.  Behind the pieces of unit k + 1 (chain k - NBUF + 2)
    // until P(k): that chain's word loads and the NBUF - 3 whole chains between (NP + WV each);  behind the words of
    // unit k (end of chain k - 1) until the end of chain k: its NP pieces.
    // The words travel wn -> wc by a register copy BEHIND the wait for wn (a copy of a register whose load is still
    // in flight reads the old value: nothing interlocks; with the copy at the end of the burst before -- a chain
    // after the request -- a table from HBM made a hub row's filtered counts differ now and then).
    constexpr int NPV = (PROBE & 2) ? 0 : NP, WV = (PROBE & 8) ? 0 : NT;  // (probes: what is really issued)
    constexpr int VMB = (NBUF - 3) * (NPV + WV) + WV;
    static_assert(VMB + NP + NT < 64, "vmcnt is a 6-bit counter");
    constexpr int VMF = NPV;
    f32x16 acc[NACC];
    u32x2 wc[NT] = {}, wn[NT] = {};  // filter words (set 0, set 1) of the current / the next unit
    auto chain = [&](int ks, int cu, int cu_next) __attribute__((always_inline)) {
      const unsigned int bdelta = ((ks + 1) & (NBUF - 1)) ? (unsigned int)UNITB : (unsigned int)(-(NBUF - 1) * UNITB);
      const int un = dq < g1 ? du : ulast;
      v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kc)::value;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"((PF - 1) * NT) : "memory");
        if constexpr (kb == PBH) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(VMB) : "memory");  // this wave's pieces of unit ks + 1 have landed
          if constexpr (!(PROBE & 4)) KGE_BARRIER();   // P(ks)
        }
        __builtin_amdgcn_sched_barrier(0);
        v4_static_for<0, NACC>([&](auto ac) __attribute__((always_inline)) {
          constexpr int ai = decltype(ac)::value, sub = ai / PARTS, part = ai % PARTS;
          if constexpr (kb == 0) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[sub][0], afr[part][0], zero, 0, 0, 0);
          } else {
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[sub][kb % PF], afr[part][kb], acc[ai], 0, 0, 0);
          }
        });
        if constexpr (kb + PF == NKB) {
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("v_add_u32 %0, %1, %0" : "+v"(bp[t]) : "s"(bdelta));
        }
        if constexpr (!(PROBE & 16))
          v4_static_for<0, NT>([&](auto ac) __attribute__((always_inline)) {
            bread(bq[decltype(ac)::value][kb % PF], std::integral_constant<int, (kb + PF) % NKB>{}, ac);
          });
        if constexpr (kb > PBH && kb <= PBH + NP && !(PROBE & 2))
          dma_piece(un, ks + NBUF - 1, std::integral_constant<int, kb - PBH - 1>{});
      });
      dma_advance();
      // this unit's words (requested behind the chain before) have landed: into wc, then the next unit's request
      if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wn[0]), "+v"(wn[1]) : "i"(VMF) : "memory");
      else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wn[0]) : "i"(VMF) : "memory");
#pragma unroll
      for (int sub = 0; sub < NT; ++sub) wc[sub] = wn[sub];
      // (the copies are instructions in front of the loads: the compiler must not fold wc into wn)
      if constexpr (NT == 2) asm volatile("" : "+v"(wc[0]), "+v"(wc[1]) : : "memory");
      else asm volatile("" : "+v"(wc[0]) : : "memory");
      if constexpr (!(PROBE & 8)) load_words(cu_next, wn);
      // the burst: this unit's comparisons -- the raw counts alone where no row of the wave has a filtered column
      // in the sub-unit and every one of its columns exists
#pragma unroll
      for (int sub = 0; sub < NT; ++sub) {
        if constexpr (PROBE & 1) {
          asm volatile("" : : "v"(acc[sub * PARTS][0]), "v"(acc[sub * PARTS + PARTS - 1][15]));
        } else {
          f32x16 sc = acc[sub * PARTS];
          if constexpr (PARTS == 2) {  // (sum q_hi t) + (sum q_lo t): one add per score
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = acc[sub * PARTS][r] + acc[sub * PARTS + 1][r];
          }
          const unsigned int wany = (ce.rk_nfilt > 0 ? wc[sub][0] : 0u) | (ce.rk_nfilt > 1 ? wc[sub][1] : 0u);
          const bool whole = ((long long)(u_lo + cu) * NT + sub + 1) * V8_UT <= m;
          const bool anyw = __any(wany != 0u) != 0;
          if constexpr (BAND) {
            unsigned int cm;
            if (!RAWFAST || rk_slow || rk_noraw || !whole || anyw) {
              cm = rank_unit(sc, cu, sub, wc[sub][0], wc[sub][1]);
            } else {
              const int nb = rank_unit_raw(sc);
              cm = 0u;
              if (__any(nb != 0) != 0) cm = rank_unit(sc, cu, sub, 0u, 0u, false);
            }
            if (orow_cur >= a.n) cm = 0u;  // (padded rows)
            if (__any(cm != 0u) != 0) {
              // ---- RARE: some row of the wave has a score of this sub-unit inside its widened band: the pairs onto
              // the wave's list.  One pass per bit position a lane still holds (one, nearly always): the lanes that
              // hold a pair take consecutive entries behind the wave's running count.
              const long long c0t = ((long long)(u_lo + cu) * NT + sub) * V8_UT;
              unsigned int mm = cm;
              while (true) {
                const unsigned long long bal = __ballot(mm != 0u);
                if (bal == 0ull) break;
                if (mm != 0u) {
                  const int bit = __builtin_ctz(mm);
                  const int slot = bl_cnt + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                  if (slot < V8_BAND_CAP) {
                    const unsigned int fw = (ce.rk_nfilt > 0 ? (wc[sub][0] >> bit) & 1u : 0u) |
                                            (ce.rk_nfilt > 1 ? ((wc[sub][1] >> bit) & 1u) << 1 : 0u);
                    const u32x4 rec = {(unsigned int)fi, (unsigned int)(c0t + bit), fw, 0u};
                    bl_list[1 + slot] = rec;
                  }
                  mm &= mm - 1u;
                }
                bl_cnt += __builtin_popcountll(bal);
              }
            }
          } else if (!RAWFAST || rk_slow || rk_noraw || !whole || anyw) rank_unit(sc, cu, sub, wc[sub][0], wc[sub][1]);
          else rank_unit_raw(sc);
          // This word is read by nobody else (every (row, sub-unit) of the batch belongs to one lane pair of one
          // workgroup): clear it here instead of in a launch behind the kernel.  Rare -- a few filtered columns per
          // row --, a plain store: one more vector-memory operation in flight only makes the counted waits wait longer.
          if (anyw && ce.rk_clear_bits) {
            const bool mine_row = counts_here && fh == 0 && orow_cur < a.n;
            unsigned int* wp = (unsigned int*)(rk_base + ((long long)(u_lo + cu) * NT + sub) * ce.rk_bits_us * 4 + rk_off);
            if (mine_row && wc[sub][0] != 0u && ce.rk_nfilt > 0) wp[0] = 0u;
            if (mine_row && wc[sub][1] != 0u && ce.rk_nfilt > 1) wp[1] = 0u;
          }
        }
      }
      if (HALF == 0 && dbg_i < 32) stamp();  // (a stamp is a store: the counted waits of this wave wait for more)
    };

    int g = g0, ks = 0;
    int pair = pair0;
    bool first = true;
    while (g < g1) {
      const int cnt = sux;
      const int side = pair / a.chunks, ch = pair - side * a.chunks;
      // ---- this lane's row of the pair
      long long lrow;
      if constexpr (DUP) lrow = (long long)ch * (8 * RW) + RW * wave + 8 * (fi >> 4) + (fi & 7);
      else lrow = (long long)ch * (8 * RW) + RW * wave + fi;
      const long long orow = lrow < a.n ? lrow : a.n - 1;  // padded rows repeat row n - 1 (and never write)
      orow_cur = (int)lrow;
      side_cur = side;
      rk_t = ce.rk_true[side][orow * ce.rk_true_stride];
      if (rk_t != rk_t) rk_t = -__builtin_inff();
      rk_al = ce.rk_atol + __builtin_fabsf(ce.rk_rtol * rk_t);
      // fragment group of this wave (see below)
      int grp = (PARTS == 2 || BAND) ? 4 * ch + (wave >> 1) : 2 * ch + (wave >> 2);
      if (grp >= a.rgn1) grp = a.rgn1 - 1;
      grp += side * a.rgn1;
      const unsigned char* const gbase = (const unsigned char*)(a.qf + (long long)grp * 4 * NKB * 64);
      auto set_thresholds = [&]() __attribute__((always_inline)) {
        bool fine = __builtin_isfinite(rk_t) && rk_al >= 0.0f && __builtin_isfinite(rk_al);
        if constexpr (RAWFAST) {  // the thresholds of rank_unit_raw: a start one rounding off at most, walked to the exact floats, then checked
          auto f2u = [](float x) { return __builtin_bit_cast(unsigned int, x); };
          auto u2f = [](unsigned int x) { return __builtin_bit_cast(float, x); };
          auto up = [&](float x) { return x == 0.0f ? u2f(1u) : u2f(x > 0.0f ? f2u(x) + 1u : f2u(x) - 1u); };
          auto down = [&](float x) { return x == 0.0f ? u2f(0x80000001u) : u2f(x > 0.0f ? f2u(x) - 1u : f2u(x) + 1u); };
          float hi = rk_t + rk_al, lo = rk_t - rk_al;
  #pragma unroll
          for (int it = 0; it < 3; ++it) {
            if (hi - rk_t > rk_al) hi = down(hi);
            if (lo - rk_t < -rk_al) lo = up(lo);
          }
  #pragma unroll
          for (int it = 0; it < 3; ++it) {
            const float u = up(hi), dn = down(lo);
            if (u - rk_t <= rk_al) hi = u;
            if (dn - rk_t >= -rk_al) lo = dn;
          }
          // A row whose thresholds do not verify keeps its wave off the raw counts (rank_unit's sign-bit arithmetic on
          // x - t needs no thresholds), it does not make it `slow`: with a tolerance that is not tiny against |t| --
          // the band launch's widened one, true scores near zero -- t - allowed has a smaller exponent than t and the
          // smallest float with fl(lo - t) >= -allowed lies many of ITS ulps from the start of the walk.
          const bool verified = __builtin_isfinite(hi) && __builtin_isfinite(lo) && hi - rk_t <= rk_al &&
                                up(hi) - rk_t > rk_al && lo - rk_t >= -rk_al && down(lo) - rk_t < -rk_al;
          rk_noraw = __any(fine && !verified) != 0;
          rk_hi = hi;
          rk_lo = lo;
        }
        rk_slow = __any(!fine) != 0;
      };
      if constexpr (BAND) {
        // ||q_lo|| of this lane's row from its lo block (read once, not kept), then the widened tolerance: 1.001 covers
        // the rounding of the norms and of the lo chain (K <= 512 products), 2^-21 (|t| + allowed + band) the
        // rounding of x_hi + x_lo and of x - t near the band.  A band that is not finite (a NaN / infinite table entry
        // or true score) makes the wave list every pair: rk_slow.
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(gbase + (2 + (wave & 1)) * (NKB * 1024)), 0, NKB * 1024, 0x00020000);
        float n2 = 0.0f;
#pragma unroll 4
        for (int kb = 0; kb < NKB; ++kb) {
          const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, lane * 16 + kb * 1024, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f0 = __uint_as_float(v[e] << 16), f1 = __uint_as_float(v[e] & 0xffff0000u);
            n2 += f0 * f0;
            n2 += f1 * f1;
          }
        }
        n2 += __shfl_xor(n2, 32, 64);
        float band = 1.001f * (__builtin_sqrtf(n2) * ce.rk_tmax[0]);
        band = band + 4.76837158203125e-7f * (__builtin_fabsf(rk_t) + rk_al + band);
        if (lrow >= a.n) band = 0.0f;  // (a padded row: its fragments are whatever the workspace held; it lists nothing)
        rk_al = rk_al + band;
        // this wave's list of the pair: the k-th pair of workgroup b, wave w (header {pairs, side, first row, 0})
        bl_cnt = 0;
        bl_list = ce.rk_list + (((long long)((pair - pair0) / pstep) * gridDim.x + blockIdx.x) * 8 + wave) * (V8_BAND_CAP + 1);
      }
      set_thresholds();
      // (the sets of a side lie interleaved: one base.  No set at all: the loads read the fragments)
      if (ce.rk_nfilt > 0) rk_base = (const unsigned char*)ce.rk_bits[side][0];
      rk_off = ce.rk_nfilt > 0 ? (unsigned int)(orow * ce.rk_bits_rs * 4) : 0u;
      // ---- fragments: groups of 128 operand rows = four blocks of 32 (bf16_queries.hpp).  Single-pass: a group = 128
      // real rows, a chunk = two groups, wave w takes block w & 3 of group w >> 2.  Split: a group = 64 real rows as
      // [hi 0-31 | hi 32-63 | lo 0-31 | lo 32-63]; DUP (d = 512): a chunk = two groups, a wave's 32 operand rows = the
      // q_hi and q_lo rows of 16 real rows; PARTS (d = 256): a chunk = FOUR groups, wave w takes the q_hi block w & 1
      // and the q_lo block 2 + (w & 1) of group w >> 1.
      // BAND: the split layout, of which this launch reads the q_hi block w & 1 of group w >> 1 (and the lo block's norm).
      v4_static_for<0, FSETS>([&](auto pc) __attribute__((always_inline)) {
        constexpr int part = decltype(pc)::value;
        unsigned int flo;
        const unsigned char* fb;
        int frange;
        if constexpr (DUP) {
          const int pt = (fi >> 3) & 1, rr = 16 * (wave & 3) + 8 * (fi >> 4) + (fi & 7);
          flo = (unsigned int)((((2 * pt + (rr >> 5)) * NKB) * 64 + (rr & 31) + 32 * fh) * 16);
          fb = gbase;
          frange = 4 * NKB * 1024;
        } else {
          flo = (unsigned int)(lane * 16);
          fb = gbase + ((PARTS == 2 || BAND) ? 2 * part + (wave & 1) : (wave & 3)) * (NKB * 1024);
          frange = NKB * 1024;
        }
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)fb, 0, frange, 0x00020000);
        v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
          constexpr int kb = decltype(kc)::value;
          afr[part][kb] =
              __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, flo + kb * 1024, 0, 16 /* sc1 */));
        });
      });
      load_words(0, wn);  // the pair's first unit
      // fragments, words, pieces of the ring fill, the atomics of the pair before: everything of this wave has landed
      if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wn[0]), "+v"(wn[1]) : : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wn[0]) : : "memory");
      // The compiler does not read the wait above: without a use of the fragments HERE it puts its own
      // s_waitcnt vmcnt(NKB - 1) ... vmcnt(0) in front of their first uses -- inside the chain loop, where they ran in
      // every chain and drained the table pieces requested for the units ahead (seen in the ISA; the ring's depth
      // was one chain instead of three, the price of a table that comes from HBM).
#pragma unroll
      for (int part = 0; part < FSETS; ++part)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(afr[part][kb]));
      if (first) {
        KGE_BARRIER();  // R0: units 0 .. 2 of the list have landed
        if (HALF == 0) stamp();
        v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) {
          v4_static_for<0, NT>([&](auto ac) __attribute__((always_inline)) {
            bread(bq[decltype(ac)::value][decltype(jc)::value], jc, ac);
          });
        });
        first = false;
      }
      for (int i = 0; i < cnt; ++i) {
        chain(ks, i, i + 1 < cnt ? i + 1 : i);
        ++ks;
      }
      rank_flush();
      if constexpr (BAND) {
        if (lane == 0) {
          const int kept = bl_cnt < V8_BAND_CAP ? bl_cnt : V8_BAND_CAP;
          const u32x4 h = {(unsigned int)kept, (unsigned int)side_cur, (unsigned int)(orow_cur - fi), 0u};
          bl_list[0] = h;
          if (ce.rk_status != nullptr) {
            if (bl_cnt != 0) atomicAdd(ce.rk_status, (unsigned int)bl_cnt);                        // pairs listed
            if (bl_cnt > V8_BAND_CAP) atomicAdd(ce.rk_status + 1, (unsigned int)(bl_cnt - V8_BAND_CAP));  // ... dropped
          }
        }
      }
      g += cnt;
      pair += pstep;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int sub = 0; sub < NT; ++sub)
#pragma unroll
      for (int jj = 0; jj < PF; ++jj) asm volatile("" : : "v"(bq[sub][jj]));
    if (HALF == 0 && a.dbg != nullptr && tid == 0) {  // the end of the workgroup's list and its length in units
      a.dbg[(long long)blockIdx.x * 64 + 40] = __builtin_readcyclecounter();
      a.dbg[(long long)blockIdx.x * 64 + 41] = (unsigned long long)(g1 - g0);
      a.dbg[(long long)blockIdx.x * 64 + 43] = wall_clock64();
    }
  };
  if (wave < 4) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// Counts of a two-sided batch of `n` rows per side (prepared query fragments `qf`, the layout of v4_build_queries)
// against the identity-indexed bf16 table TG (d in {256, 512}): ce carries the rk_* arguments.  KGE_ERR_UNSUPPORTED:
// not this kernel's case (KGE_V8_RANK=0 declines everything: pairs_bf16_v4_kernel<V3_RANK>).
int run_pairs_bf16_v8_rank(int scorer, bool split, const Operand& TG, int d, long long n, long long m, const void* qf,
                           const CeArgs& ce, hipStream_t st, unsigned long long* dbg, int reserve_cus, bool band,
                           long long* band_lists) {
  // band: `qf` holds the SPLIT fragments, the launch runs their q_hi chains (single-pass geometry) and rescores the
  // sub-units that hold a score inside a row's widened band (ce.rk_tmax; ce.rk_status counts them)
  if ((d != 512 && d != 256) || TG.idx.ptr != nullptr || qf == nullptr) return KGE_ERR_UNSUPPORTED;
  if (band) {
    if (!split || !ce.rk_tmax || !ce.rk_list) return KGE_ERR_INVALID_ARG;
    split = false;  // the geometry of the single-pass launch ...
  }
  if (sw(SW_V8_RANK) == 0) return KGE_ERR_UNSUPPORTED;
  if (TG.ld * 2 >= (1LL << 28) || ((uintptr_t)qf & 15)) return KGE_ERR_UNSUPPORTED;
  const long long rgr = (split || band) ? 64 : 128;  // ... on the split set's 64-row fragment groups
  const long long rgn1 = (n + rgr - 1) / rgr;
  const long long ut = (d == 256 && !split) ? 2 * V8_UT : V8_UT;  // table rows per unit (the kernel's NT sub-units of 32)
  const long long nunits = (m + ut - 1) / ut;
  if (rgn1 > (1 << 20) || (rgn1 + 1) * nunits >= (1LL << 30)) return KGE_ERR_UNSUPPORTED;
  if (ce.rk_nfilt > 0 && n * ce.rk_bits_rs * 4 >= (1LL << 32)) return KGE_ERR_UNSUPPORTED;  // (the row's 32-bit offset)
  int cus = v8_cu_count() - reserve_cus;
  if (cus > 256) cus = 256;
  if (cus < 8) cus = 8;
  V8RankArgs a{};
  a.TG = TG;
  a.n = n;
  a.m = m;
  a.rgn1 = (int)rgn1;
  a.sides = 2;
  a.chunks = (int)((split && d == 256) || band ? (rgn1 + 3) / 4 : (rgn1 + 1) / 2);  // 256 rows (d = 512 split: 128) per chunk
  a.nunits = (int)nunits;
  a.su = (int)((nunits + 7) / 8);
  a.wpx = cus / 8;
  a.qf = (const u32x4*)qf;
  a.dbg = dbg != nullptr ? dbg : v6_get_stamps();
  a.ce = ce;
  const dim3 grid(8 * a.wpx), block(512);
  if (band) {  // the lists: one per (pair index of a workgroup, workgroup, wave)
    const long long P = 2LL * a.chunks, kmax = P >= a.wpx ? (P + a.wpx - 1) / a.wpx : 1;
    if (kmax * grid.x * 8 * (V8_BAND_CAP + 1) * 16 > ce.rk_list_bytes) return KGE_ERR_WORKSPACE;
    if (band_lists != nullptr) *band_lists = kmax * grid.x * 8;
  }
#ifdef KGE_V8_PROBES
  if (sw(SW_V8R_PROBE) > 0) {
    const int pr = (int)sw(SW_V8R_PROBE);
    if (pr > 0 && scorer == KGE_COMPLEX && !split && d == 256) {
#define KGE_V8RP(PR)                                                                                          \
  if (pr == PR) {                                                                                             \
    hipLaunchKernelGGL((pairs_bf16_v8_rank_kernel<KGE_COMPLEX, 128, 0, PR>), grid, block, 0, st, a);          \
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;                                         \
  }
      KGE_V8RP(1) KGE_V8RP(2) KGE_V8RP(3) KGE_V8RP(8) KGE_V8RP(9) KGE_V8RP(11) KGE_V8RP(15) KGE_V8RP(31)
#undef KGE_V8RP
    }
  }
#endif
#define KGE_V8R(SC, HHV, SP) hipLaunchKernelGGL((pairs_bf16_v8_rank_kernel<SC, HHV, SP>), grid, block, 0, st, a)
#define KGE_V8RB(SC, HHV) hipLaunchKernelGGL((pairs_bf16_v8_rank_kernel<SC, HHV, 0, 0, 1>), grid, block, 0, st, a)
#define KGE_V8R2(SC)                                                \
  if (band) { if (d == 512) KGE_V8RB(SC, 256); else KGE_V8RB(SC, 128); } \
  else if (d == 512) { if (split) KGE_V8R(SC, 256, 1); else KGE_V8R(SC, 256, 0); } \
  else { if (split) KGE_V8R(SC, 128, 1); else KGE_V8R(SC, 128, 0); }
  if (scorer == KGE_COMPLEX) { KGE_V8R2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_V8R2(KGE_DISTMULT) }
  else return KGE_ERR_UNSUPPORTED;
  __atomic_fetch_add(&g_v8_launches[1], 1, __ATOMIC_RELAXED);
#undef KGE_V8R2
#undef KGE_V8RB
#undef KGE_V8R
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

#undef KGE_V8_DMA

// ---- The true scores of a batch from its prepared query fragments: score of (s_i, p_i) against o_i and of (p_i, o_i)
// against s_i -- the elements (i, o_i) / (i, s_i) of the score matrices, bit for bit (each score is one accumulation
// chain in K order: the chain of the kernels above; MFMA(table rows, query rows), the counting kernel's orientation).
// kge_eval_batch spent a whole scoring launch on them (17 us for an [n, 2 n] block per side on listed targets, of
// which the diagonals were used); here one wave takes the 32 operand rows of a fragment block, gathers their 32 target
// rows into LDS in the units' swizzled layout, runs ONE chain and keeps the diagonal.
struct V8TrueArgs {
  Operand TG;           // the entity table (identity rows)
  Index tgt[2];         // side 0: the true objects, side 1: the true subjects
  long long n;
  int rgn1;
  const u32x4* qf;
  float* out[2];        // [n] each
};

template <int SCORER, int HH, int SPLIT>
__global__ __launch_bounds__(256) void pairs_bf16_true_kernel(V8TrueArgs a) {
  constexpr int NKB = 2 * HH / 16, ROWB = 4 * HH, SPR = ROWB / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4][32 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int blk = (int)blockIdx.x * 4 + wave;     // fragment block: (side, group of 128 operand rows, 32 of them)
  const int per_side = a.rgn1 * 4;
  if (blk >= 2 * per_side) return;
  const int side = blk / per_side, g = (blk % per_side) >> 2, sub = blk & 3;
  // ---- the lane's real row and its fragments (the addressing of pairs_bf16_v8_rank_kernel)
  const int rr = SPLIT ? 8 * (fi >> 4) + (fi & 7) : fi;  // real row within the wave = its target's slot
  const long long lrow = SPLIT ? (long long)g * 64 + 16 * sub + rr : (long long)g * 128 + 32 * sub + fi;
  const unsigned char* const gbase = (const unsigned char*)(a.qf + (long long)(side * a.rgn1 + g) * 4 * NKB * 64);
  unsigned int flo;
  const unsigned char* fb;
  if constexpr (SPLIT) {
    const int part = (fi >> 3) & 1, r64 = 16 * sub + rr;
    flo = (unsigned int)((((2 * part + (r64 >> 5)) * NKB) * 64 + (r64 & 31) + 32 * fh) * 16);
    fb = gbase;
  } else {
    flo = (unsigned int)(lane * 16);
    fb = gbase + sub * (NKB * 1024);
  }
  bf16x8 afr[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) afr[kb] = *reinterpret_cast<const bf16x8*>(fb + flo + kb * 1024);
  // ---- the 32 target rows of this wave into its LDS block (slot j = the target of real row j; split: 16 real rows,
  // slots 16 .. 31 repeat them), 16-byte slot `c ^ (row & 15)` of row `row` as in the units of the kernels above
  long long myrow = SPLIT ? (long long)g * 64 + 16 * sub + (lane & 15) : (long long)g * 128 + 32 * sub + (lane & 31);
  if (myrow >= a.n) myrow = a.n - 1;
  const int my_t = (int)index_at(a.tgt[side], myrow);  // lane j (< 32): the table row of slot j
  unsigned char* const lds = smem[wave];
  const unsigned char* const tgb = (const unsigned char*)a.TG.base;
  const long long tld2 = a.TG.ld * 2;
#pragma unroll 4
  for (int it = 0; it < 32 * SPR / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / SPR, c = idx % SPR;
    const int trow = __shfl(my_t, row, 64);
    const u32x4 v = *reinterpret_cast<const u32x4*>(tgb + (long long)trow * tld2 + c * 16);
    *reinterpret_cast<u32x4*>(lds + row * ROWB + ((c ^ (row & 15)) << 4)) = v;
  }
  __syncthreads();  // (every wave of the block reaches it: blocks beyond the work return whole or not at all -- see below)
  // ---- one chain
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const bf16x8 bq = *reinterpret_cast<const bf16x8*>(lds + fi * ROWB + (((2 * kb + fh) ^ (fi & 15)) << 4));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, afr[kb], acc, 0, 0, 0);
  }
  // ---- the diagonal: this lane's query row against target slot rr = element 8 (r >> 2) + 4 fh + (r & 3) of its 32
  const int want_r = 4 * (rr >> 3) + (rr & 3);
  float v = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v = r == want_r ? acc[r] : v;
  if constexpr (SPLIT) {  // (sum q_hi t) + (sum q_lo t): the partner lane (fi ^ 8) holds the other part
    const float o = __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    v = v + o;
  }
  const bool owner = fh == ((rr >> 2) & 1) && (SPLIT ? ((fi >> 3) & 1) == 0 : true);
  if (owner && lrow < a.n) a.out[side][lrow] = v;
}

// KGE_ERR_UNSUPPORTED: not this kernel's case (the caller scores the listed targets with a scoring launch)
int run_pairs_bf16_true(int scorer, bool split, const Operand& TG, int d, long long n, const void* qf, const Index& t_sp,
                        const Index& t_po, float* true_sp, float* true_po, hipStream_t st) {
  if ((d != 512 && d != 256) || TG.idx.ptr != nullptr || qf == nullptr || ((uintptr_t)qf & 15)) return KGE_ERR_UNSUPPORTED;
  if (n <= 0) return KGE_OK;
  const long long rgr = split ? 64 : 128;
  const long long rgn1 = (n + rgr - 1) / rgr;
  if (rgn1 > (1 << 20) || TG.ld * 2 >= (1LL << 28)) return KGE_ERR_UNSUPPORTED;
  V8TrueArgs a{};
  a.TG = TG;
  a.tgt[0] = t_sp;
  a.tgt[1] = t_po;
  a.n = n;
  a.rgn1 = (int)rgn1;
  a.qf = (const u32x4*)qf;
  a.out[0] = true_sp;
  a.out[1] = true_po;
  const dim3 grid((unsigned)(2 * rgn1)), block(256);  // 4 fragment blocks (waves) per workgroup: exactly one group
#define KGE_V8T(SC, HHV, SP) hipLaunchKernelGGL((pairs_bf16_true_kernel<SC, HHV, SP>), grid, block, 0, st, a)
#define KGE_V8T2(SC)                                                \
  if (d == 512) { if (split) KGE_V8T(SC, 256, 1); else KGE_V8T(SC, 256, 0); } \
  else { if (split) KGE_V8T(SC, 128, 1); else KGE_V8T(SC, 128, 0); }
  if (scorer == KGE_COMPLEX) { KGE_V8T2(KGE_COMPLEX) } else if (scorer == KGE_DISTMULT) { KGE_V8T2(KGE_DISTMULT) }
  else return KGE_ERR_UNSUPPORTED;
#undef KGE_V8T2
#undef KGE_V8T
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- Band-and-rescore, second launch: the pairs the band launch left undecided, counted with BOTH chains.
// One wave per list (= one wave of the first launch and one (side, 256-row chunk): 32 query rows).  Per batch of up to
// 32 listed pairs the wave gathers the pairs' table rows into LDS in the units' swizzled layout (slot k = the column of
// pair k: pairs_bf16_true_kernel's gather), runs the q_hi and the q_lo chain of its 32 rows' split fragments -- each
// chain the K order and the instruction of pairs_bf16_v8_rank_kernel<SPLIT>, so element (row, slot) is
// fl(sum q_hi t) + fl(sum q_lo t), the bits that kernel counts -- and applies count_one to element (row_k, k) of every
// listed pair, with the filter bits that travelled with it.  ~15 pairs per list on a trained model: one batch, one
// gather of 32 x 512 bytes.  The header's count is zeroed on the way out (lists are all-empty between calls).
struct V8RescoreArgs {
  Operand TG;
  long long n, m;
  int rgn1;              // 64-row split fragment groups per side
  long long nlists;
  const u32x4* qf;
  CeArgs ce;
};

template <int HH>
__global__ __launch_bounds__(256) void pairs_bf16_rescore_kernel(V8RescoreArgs a) {
  constexpr int NKB = 2 * HH / 16, ROWB = 4 * HH, SPR = ROWB / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4][32 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const CeArgs& ce = a.ce;
  const long long id = (long long)blockIdx.x * 4 + wave;
  if (id >= a.nlists) return;
  u32x4* const list = ce.rk_list + id * (V8_BAND_CAP + 1);
  const u32x4 h = list[0];
  const int cnt = (int)h[0];
  if (cnt <= 0) return;  // (wave-uniform; no workgroup barrier below)
  const int side = (int)h[1];
  const long long row0 = (long long)h[2];
  const long long lrow = row0 + fi;
  const long long orow = lrow < a.n ? lrow : a.n - 1;
  float t = ce.rk_true[side][orow * ce.rk_true_stride];
  if (t != t) t = -__builtin_inff();
  unsigned char* const lds = smem[wave];
  const unsigned char* const tgb = (const unsigned char*)a.TG.base;
  const long long tld2 = a.TG.ld * 2;
  long long grp = row0 >> 6;
  if (grp >= a.rgn1) grp = a.rgn1 - 1;
  const unsigned char* const gbase = (const unsigned char*)(a.qf + ((long long)side * a.rgn1 + grp) * 4 * NKB * 64);
  const int blk = (int)((row0 >> 5) & 1);
  int G = 0, C = 0, FG[2] = {0, 0}, FC[2] = {0, 0};
  for (int b0 = 0; b0 < cnt; b0 += 32) {
    // lane k (and k + 32) holds pair b0 + k: (row within the wave's 32, column, filter bits); beyond the list: none
    const int k = b0 + (lane & 31);
    const u32x4 rec = k < cnt ? list[1 + k] : u32x4{0xffffffffu, 0u, 0u, 0u};
    long long col = (long long)rec[1];
    if (col >= a.m) col = a.m - 1;
    const int my_t = (int)col;
    // (all of a batch's row pieces requested before the first is stored: one memory latency per batch)
    u32x4 gv[32 * SPR / 64];
#pragma unroll
    for (int it = 0; it < 32 * SPR / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / SPR, c = idx % SPR;
      const int trow = __shfl(my_t, row, 64);
      gv[it] = *reinterpret_cast<const u32x4*>(tgb + (long long)trow * tld2 + c * 16);
    }
#pragma unroll
    for (int it = 0; it < 32 * SPR / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / SPR, c = idx % SPR;
      *reinterpret_cast<u32x4*>(lds + row * ROWB + ((c ^ (row & 15)) << 4)) = gv[it];
    }
    // (the block is this wave's own and a wave's LDS operations execute in order: the compiler must keep the other
    // lanes' stores in front of this lane's reads, and the reads in front of the next batch's stores)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // both chains; a chain's NKB fragments are requested back to back (one L2 latency per chain, not one per K-block)
    f32x16 acc[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const unsigned char* fb = gbase + (2 * part + blk) * (NKB * 1024) + lane * 16;
      bf16x8 af[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) af[kb] = *reinterpret_cast<const bf16x8*>(fb + kb * 1024);
      f32x16 ac = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(lds + fi * ROWB + (((2 * kb + fh) ^ (fi & 15)) << 4));
        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, af[kb], ac, 0, 0, 0);
      }
      acc[part] = ac;
    }
    // element r of this lane = (query row fi, slot 8 (r >> 2) + 4 fh + (r & 3)): counted if that slot's pair names row fi
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot = 8 * (r >> 2) + 4 * fh + (r & 3);
      const unsigned int prow = (unsigned int)__shfl((int)rec[0], slot, 64);
      const unsigned int pfw = (unsigned int)__shfl((int)rec[2], slot, 64);
      if (prow == (unsigned int)fi) {
        int g1 = 0, c1 = 0;
        count_one(acc[0][r] + acc[1][r], t, ce.rk_atol, ce.rk_rtol, g1, c1);
        G += g1;
        C += c1;
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (q < ce.rk_nfilt && ((pfw >> q) & 1u) != 0u) {
            FG[q] += g1;
            FC[q] += c1;
          }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  G += __shfl_xor(G, 32, 64);
  C += __shfl_xor(C, 32, 64);
  const bool wr = fh == 0 && lrow < a.n;
  unsigned long long* rank = ce.rk_rank[side] + lrow;
  unsigned long long* ties = ce.rk_ties[side] + lrow;
  if (wr && G != 0) atomicAdd(rank, (unsigned long long)G);
  if (wr && C != 0) atomicAdd(ties, (unsigned long long)C);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q < ce.rk_nfilt) {
      const int fg = FG[q] + __shfl_xor(FG[q], 32, 64), fc = FC[q] + __shfl_xor(FC[q], 32, 64);
      // (a filtered column inside the band leaves the filtered ranking; "-inf is close to a true score of -inf" was
      // added for ALL filtered columns by the first launch)
      if (wr && G - fg != 0) atomicAdd(rank + (q + 1) * ce.rk_ld, (unsigned long long)(long long)(G - fg));
      if (wr && C - fc != 0) atomicAdd(ties + (q + 1) * ce.rk_ld, (unsigned long long)(long long)(C - fc));
    }
  }
  if (lane == 0) list[0] = u32x4{0u, 0u, 0u, 0u};
}

// Bytes of list space that hold the lists of a batch of n rows per side whatever the launch geometry: a launch of
// 8 wpx workgroups (wpx <= 32) writes ceil(P / wpx) * 8 wpx * 8 <= 64 (P + 32) lists for its P = 2 ceil(n / 256) pairs.
long long pairs_bf16_band_list_bytes(long long n) {
  return n <= 0 ? 0 : 64 * (2 * ((n + 255) / 256) + 32) * (long long)(V8_BAND_CAP + 1) * 16;
}

int run_pairs_bf16_rescore(const Operand& TG, int d, long long n, long long m, const void* qf, const CeArgs& ce,
                           long long nlists, hipStream_t st) {
  if ((d != 512 && d != 256) || TG.idx.ptr != nullptr || qf == nullptr || ((uintptr_t)qf & 15)) return KGE_ERR_UNSUPPORTED;
  V8RescoreArgs a{};
  a.TG = TG;
  a.n = n;
  a.m = m;
  a.rgn1 = (int)((n + 63) / 64);
  a.nlists = nlists;
  a.qf = (const u32x4*)qf;
  a.ce = ce;
  const dim3 grid((unsigned)((a.nlists + 3) / 4)), block(256);
  if (d == 512) hipLaunchKernelGGL((pairs_bf16_rescore_kernel<256>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((pairs_bf16_rescore_kernel<128>), grid, block, 0, st, a);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- kge_table_max_row_norm: the largest Euclidean row norm of a bf16 table, x 1.001 (the rounding of the sums), as
// one float -- the table-wide factor of the band's Cauchy-Schwarz bound.  One wave per row, float atomicMax on the
// bits (norms are >= 0: the integer order is the float order); `out` zeroed by the launcher's fill.
__global__ __launch_bounds__(256) void table_max_norm_kernel(Operand TG, long long m, int d, unsigned int* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float best = 0.0f;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += (long long)gridDim.x * 4) {
    const unsigned short* p = (const unsigned short*)TG.base + row * TG.ld;
    float s2 = 0.0f;
    for (int c = lane * 8; c < d; c += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(p + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f0 = __uint_as_float(v[e] << 16), f1 = __uint_as_float(v[e] & 0xffff0000u);
        s2 += f0 * f0;
        s2 += f1 * f1;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s2 += __shfl_xor(s2, off, 64);
    float nrm = 1.001f * __builtin_sqrtf(s2);
    if (nrm != nrm) nrm = __builtin_inff();  // a NaN in the table: no finite band
    best = fmaxf(best, nrm);
  }
  if (lane == 0) atomicMax(out, __float_as_uint(best));
}

int run_table_max_norm(const Operand& TG, long long m, int d, float* out, hipStream_t st) {
  if (TG.idx.ptr != nullptr || (d % 8) != 0 || (TG.ld % 8) != 0 || ((uintptr_t)TG.base & 15)) return KGE_ERR_UNSUPPORTED;
  if (!fill_words_async(out, 0, sizeof(float), st)) return KGE_ERR_LAUNCH;
  if (m <= 0) return KGE_OK;
  long long blocks = (m + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(table_max_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, TG, m, d, (unsigned int*)out);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---- kge_debug_mfma_rate: the matrix pipe alone.  The grid and wave layout of the kernels above (one workgroup of
// eight waves per CU = two waves per SIMD), each wave issuing `iters` x 16 v_mfma_f32_32x32x16_bf16 on TWO independent
// accumulators from operands loaded once (random bf16 values: the pipe's power draw depends on the data) -- no LDS, no
// vector memory, no comparisons.  What this reaches is what the chip sustains under matrix load (the clock drops to
// ~1.3 GHz on the boxes measured: 0.55 of the nominal dense peak), i.e. the floor the counting kernel and the split
// store kernel are compared with.
__global__ __launch_bounds__(512, 1) void mfma_rate_kernel(const u32x4* __restrict__ rnd, int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16x8 a0 = __builtin_bit_cast(bf16x8, rnd[(wave * 4 + 0) * 64 + lane]);
  const bf16x8 a1 = __builtin_bit_cast(bf16x8, rnd[(wave * 4 + 1) * 64 + lane]);
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, rnd[(wave * 4 + 2) * 64 + lane]);
  const bf16x8 b1 = __builtin_bit_cast(bf16x8, rnd[(wave * 4 + 3) * 64 + lane]);
  f32x16 c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c1, 0, 0, 0);
    }
  }
  float t = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) t += c0[r] + c1[r];
  if (t == 12345.678f) sink[0] = t;  // (keeps the chains alive; never true for the operands handed in)
}

// flops of the launch, or a negative kge_status
double run_mfma_rate(const void* rnd, int iters, float* sink, hipStream_t st) {
  if (rnd == nullptr || sink == nullptr || iters <= 0 || ((uintptr_t)rnd & 15)) return (double)KGE_ERR_INVALID_ARG;
  int cus = v8_cu_count();
  if (cus > 256) cus = 256;
  hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(512), 0, st, (const u32x4*)rnd, iters, sink);
  if (hipGetLastError() != hipSuccess) return (double)KGE_ERR_LAUNCH;
  return (double)cus * 8.0 * (double)iters * 16.0 * (2.0 * 32.0 * 32.0 * 16.0);
}
}  // namespace kge
