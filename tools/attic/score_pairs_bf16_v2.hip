// score_pairs_bf16_v2.hip -- "row-persistent" ComplEx / DistMult sp_/_po kernel for bf16
// tables, d in {128, 256, 512}: the BASELINE.json headline path on gfx950.
//
// v1 (score_pairs_bf16.hip) walks K in 8 dependent load->barrier->MFMA stages per tile and
// rebuilds the query tile for every target tile: latency bound (29.5 us at C2).  Here:
//
//   * a workgroup = 4 waves owns 128 query rows (32 per wave) and a contiguous RANGE of
//     target tiles; it builds the bf16 query fragments q = s (x) r ONCE, for the whole
//     reduction dimension, directly in MFMA A-operand registers (d=512: 32 K-blocks x 4
//     VGPRs = 128 VGPRs per lane), and keeps them for every target tile;
//   * the gather of the s / r rows is staged through LDS in 128-byte segments (full cache
//     lines: two rows x four row-halves per wave-instruction) and re-read in fragment
//     shape with an XOR swizzle (conflict-free ds_read_b128);
//   * target tiles (32 rows x d bf16) stream HBM -> LDS by LDS-DMA (global_load_lds, 16 B per
//     lane, no VGPRs) through a 3-deep LDS ring, two tiles ahead, ONE raw s_barrier per tile
//     and a COUNTED s_waitcnt vmcnt(16): the in-order VMEM counter also counts the score
//     stores of each tile, so every wave issues exactly 8 DMA ops + 4 stores per full tile
//     (row tails are clamped, never predicated: a clamped lane recomputes and rewrites the
//     bits of the last valid row) and stores are never waited for;
//   * per tile 32 v_mfma_f32_32x32x16_bf16 with the TARGET fragment (from LDS, XOR swizzle
//     applied on the DMA source address, lane-linear LDS image) as the "A" operand and the
//     query fragment as "B": each lane then owns 4 x 4 consecutive targets of one query row
//     and writes them with 16-byte stores (dword stores are issue-bound: 1.3 us per tile);
//   * 1-D XCD-aware grid: all row groups of a target range run on the same XCD (block id
//     mod 8), so each XCD pulls its 1/8 of the table through its own L2 once.
//
// K-block <-> data: block kb < HH/16 holds first-half coordinates [16kb, 16kb+16) (lane half
// h = lane>>5 holds 8 of them), block HH/16+kb the same coordinates of the second half;
// A and B fragments use the same map, so the contraction is over all d coordinates.
#include "common.hpp"

namespace kge {

constexpr int V2_ROWS = 128, V2_TN = 32;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ long long shfl64(long long v, int src) {
  int lo = __shfl((int)(v & 0xffffffffLL), src, 64);
  int hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// ABL (ablation, debug entry only): bit 0 = drop the global score stores, bit 1 = drop the
// in-loop tile DMA (results are wrong; used to attribute time, tools/v2_phases.py)
// PREQ: the query fragments were built by build_queries_kernel into `qf` (workspace);
// otherwise the prologue gathers and builds them itself (fused, no workspace).
template <int SCORER, int HH, int TGMODE, int ABL = 0, bool PREQ = false>
__global__ __launch_bounds__(256, 1) void pairs_bf16_v2_kernel(
    Operand A, Operand R, Operand TG, int dir, long long n, long long m, int rgn, int ncg,
    int tiles_per_cg, int ntiles, float* __restrict__ out, long long ldo,
    unsigned long long* __restrict__ dbg, const u32x4* __restrict__ qf) {
  constexpr int NKB = 2 * HH / 16;        // K-blocks of 16
  constexpr int NKH = HH / 16;            // K-blocks per half
  constexpr int ROWB = 4 * HH;            // bytes per table row (2*HH bf16)
  constexpr int SPR = HH / 4;             // 16-byte slots per row
  constexpr int TILEB = V2_TN * ROWB;     // bytes per target tile
  constexpr int NL = TILEB / 16 / 256;    // 16-byte DMA ops per wave-lane per tile
  constexpr int PASSES = HH / 64;         // prologue passes of 64 coordinates
  constexpr int STAGE = 4 * 16384;        // one prologue staging slot: 16 KiB per wave
  constexpr int STG0 = TILEB;             // two slots behind ring buffer 0
  constexpr int CST = 4 * 32 * 144;       // per-wave C-tile transpose buffers (after the ring)
  constexpr int SMEM = (3 * TILEB + CST > STG0 + 2 * STAGE) ? 3 * TILEB + CST : STG0 + 2 * STAGE;
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
  const long long row0 = (long long)rg * V2_ROWS + 32 * wave;
  const unsigned short* tgb = (const unsigned short*)TG.base;

  int dbg_i = 0;
  auto stamp = [&]() {  // optional per-phase timestamps (tools/v2_phases.py); dbg == NULL in production
    if (dbg != nullptr && tid == 0 && dbg_i < 64)
      dbg[(long long)blockIdx.x * 64 + dbg_i] = __builtin_readcyclecounter();
    ++dbg_i;
  };
  stamp();  // 0: kernel start

  // ---- target tile DMA (HBM -> LDS, no registers), one 1-KiB piece per call
  // element offset of this lane's 16-B chunk inside a full tile of consecutive table rows
  unsigned int toff[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int L = (wave * NL + k) * 64 + lane;  // linear 16-B slot of the tile image
    const int row = L / SPR, slot = L % SPR;
    toff[k] = (unsigned int)(row * (int)TG.ld + ((slot ^ (row & 15)) << 3));
  }
  auto dma_piece = [&](int tt, int buf, int k) {
    if (tt >= ntl) tt = ntl - 1;  // keep the VMEM op count per step constant (see header)
    const long long trow0 = (long long)(tile_lo + tt) * V2_TN;
    unsigned char* dst = smem + buf * TILEB + (wave * NL + k) * 1024;  // wave-uniform
    const unsigned short* src;
    if (TGMODE == 0 && trow0 + V2_TN <= m) {  // all entities, full tile: uniform base + lane offset
      src = tgb + trow0 * TG.ld + toff[k];
    } else {
      const int L = (wave * NL + k) * 64 + lane;
      const int row = L / SPR, slot = L % SPR;
      long long tr = trow0 + row;
      if (tr >= m) tr = m - 1;
      src = tgb + index_mode<TGMODE>(TG.idx, tr) * TG.ld + ((slot ^ (row & 15)) << 3);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto tile_dma = [&](int tt, int buf) {
#pragma unroll
    for (int k = 0; k < NL; ++k) dma_piece(tt, buf, k);
  };

  // ---- prologue: build the query fragments of this wave's 32 rows in registers.
  // Tile 0 streams first (it needs no index), then the s / r rows are gathered in passes of
  // 64 coordinates: 128-byte segments (full cache lines) -> registers -> wave-private LDS
  // staging -> re-read in fragment shape.  The loads of pass p+1 fly while pass p is built.
  bf16x8 afr[NKB];
  if constexpr (PREQ) {
    // fragment-major workspace: K-block kb of 32-row block rb is 64 lanes x 16 B, contiguous
    const u32x4* src = qf + ((long long)(rg * (V2_ROWS / 32) + wave) * NKB) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) afr[kb] = __builtin_bit_cast(bf16x8, src[kb * 64]);
    tile_dma(0, 0);
    tile_dma(1, 1);
    stamp();  // 1: fragment loads and tiles 0, 1 issued
  } else
  {
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
      const long long ao = shfl64(aoff, rr), ro = shfl64(roff, rr);
      gsrc[k] = ((p5 < 16) ? ab + ao : rb + ro) + ((p5 >> 3) & 1) * HH + (p5 & 7) * 8;
    }
    stamp();  // 1: indices loaded, source pointers built
    // gather of pass p: 16 plain 16-B loads per lane (an LDS-DMA piece costs ~100 issue
    // cycles, a global_load ~8), written to the wave-private staging slot p & 1 lane-linearly
    u32x4 g[16];
    auto gather = [&](int p) {
#pragma unroll
      for (int k = 0; k < 16; ++k) g[k] = *reinterpret_cast<const u32x4*>(gsrc[k] + 64 * p);
    };
    auto stage_write = [&](int p) {
      unsigned char* dst = smem + STG0 + (p & 1) * STAGE + wave * 16384 + lane * 16;
#pragma unroll
      for (int k = 0; k < 16; ++k) *reinterpret_cast<u32x4*>(dst + k * 1024) = g[k];
    };
    auto build = [&](int p) {
      const unsigned char* stage = smem + STG0 + (p & 1) * STAGE + wave * 16384 + fi * 512;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        u32x4 v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int p5 = a * 8 + 2 * jj + fh;
          v[a] = *reinterpret_cast<const u32x4*>(stage + ((p5 ^ (fi & 15)) << 4));
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
      stage_write(p);  // waits for the loads of pass p (compiler-tracked)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      stamp();  // 3+2p: pass p landed and staged
      if (p + 1 < PASSES) gather(p + 1);  // flies while pass p is built
      build(p);
      stamp();  // 4+2p: pass p built
    }
    // staging overlaps ring buffers 1 and 2: everyone must be done before tile 1 streams in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    tile_dma(1, 1);
    stamp();  // 3+2*PASSES: prologue done, all waves synchronised
  }

  // ---- main loop over this workgroup's target tiles
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(((2 * t + fh) ^ (fi & 15)) << 4);
  // this lane's output row (query row0 + fi; clamped rows rewrite the bits of row n-1)
  long long orow = row0 + fi;
  if (orow >= n) orow = n - 1;
  float* const orow_ptr = out + orow * ldo;

  // acc[4g + e] = score(query fi, target col0 + 8g + 4fh + e).  The finished tile goes through
  // a wave-private LDS transpose so that every store instruction writes 8 rows x 128
  // contiguous bytes, and its ~60 instructions are spread between the MFMAs of the NEXT tile
  // (two accumulator sets, ping-pong), where they are free.
  unsigned char* const cst = smem + 3 * TILEB + wave * (32 * 144);  // free after the prologue
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
  auto ep_store = [&](int tt, int i, const f32x4& v) {
    if (ABL & 1) {
      asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
      return;
    }
    long long orow_i = row0 + 8 * i + (lane >> 3);
    if (orow_i >= n) orow_i = n - 1;  // clamped rows rewrite the bits of row n-1
    *reinterpret_cast<f32x4u*>(out + orow_i * ldo + (long long)(tile_lo + tt) * V2_TN + 4 * (lane & 7)) = v;
  };
  auto store_ragged = [&](int tt, const f32x16& acc) {  // last tile of the table, m % 32 != 0
    const long long col0 = (long long)(tile_lo + tt) * V2_TN;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      long long oc = col0 + 8 * (r >> 2) + 4 * fh + (r & 3);
      if (oc >= m) oc = m - 1;
      orow_ptr[oc] = acc[r];
    }
  };
  auto tile_full = [&](int tt) { return (long long)(tile_lo + tt + 1) * V2_TN <= m; };

  // One tile: wait + barrier, then the MFMA chain; between its MFMAs: the B-fragment reads
  // (double-buffered batches), the DMA pieces of tile tt+2, and the epilogue of tile tt-1.
  // VMEM order per wave: T1 | T2 | S0 T3 | S1 T4 | ...  -> newer than T(tt) at the wait of
  // tile tt: NL for tt < 2 (T(tt+1) only), NL + 4 from tt == 2 on;
  // only the last tile of the table can be ragged, and it is stored after the loop.
  auto tile_body = [&](int tt, f32x16& acc, const f32x16& accp, bool store_prev) {
    if (tt < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL + 4) : "memory");
    __builtin_amdgcn_s_barrier();  // tile tt visible to all; everyone finished reading tile tt-1
    __builtin_amdgcn_sched_barrier(0);
    stamp();  // tile tt released
    // B fragment of K-block kb sits at 16-B slot s = s0(kb) + fh of target row fi, stored at
    // slot s ^ (fi & 15): with s = 16*a + b the swizzle only touches b, so the address is
    // (per-lane base for b) + immediate a*256 -- 8 address registers instead of 32.
    const unsigned int bt = (unsigned int)((tt % 3) * TILEB + fi * ROWB);
    unsigned int bp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bp[t] = bt + boff[t];
    // The TARGET fragment is the MFMA "A" operand and the query fragment the "B" operand,
    // so the accumulator holds, for query row fi, 4 x 4 CONSECUTIVE targets.
    // B fragments: PF ds_read_b128 in flight, issued by inline asm and retired with COUNTED
    // lgkmcnt waits (LDS ops return in order).  hipcc only ever emits lgkmcnt(0) for the LDS
    // reads of this kernel, which stalls every few MFMAs on the newest read (measured: 2000
    // cycles per tile for a 1056-cycle MFMA chain).  Extra compiler-issued LDS ops between
    // the reads (C-tile transpose) only make a counted wait stricter, never weaker.
    constexpr int PF = (NKB < 8) ? NKB : 8;
    constexpr int DSTEP = NKB / NL;  // one DMA piece every DSTEP MFMAs
    bf16x8 bq[PF];
    f32x4 cv[4];
    auto bread = [&](bf16x8& dst, int kb) {
      const int s0 = (kb < NKH) ? (2 * kb) : (HH / 8 + 2 * (kb - NKH));
      asm volatile("ds_read_b128 %0, %1 offset:%2"
                   : "=v"(dst)
                   : "v"(bp[(s0 & 15) >> 1]), "i"((s0 >> 4) * 256)
                   : "memory");
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) bread(bq[j], j);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      // reads newer than read kb: min(PF - 1, NKB - 1 - kb)
      if (NKB - 1 - kb >= PF - 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(PF - 1) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NKB - 1 - kb) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % PF], afr[kb], acc, 0, 0, 0);
      if (kb + PF < NKB) bread(bq[kb % PF], kb + PF);
      if (!(ABL & 2) && (kb % DSTEP) == DSTEP - 1) dma_piece(tt + 2, (tt + 2) % 3, kb / DSTEP);
      if (store_prev) {  // epilogue of tile tt-1, spread over the chain (positions scale with NKB)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (kb == g * NKB / 32) ep_write(accp, g);
        if (kb == 4 * NKB / 32) ep_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb == (6 + i) * NKB / 32) cv[i] = ep_read(i);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb == (14 + 2 * i) * NKB / 32) ep_store(tt - 1, i, cv[i]);
      }
    }
    stamp();  // tile tt: MFMA chain issued
  };

  f32x16 acc_a = {}, acc_b = {};
  tile_body(0, acc_a, acc_b, false);
  int tt = 1;
  for (; tt + 1 < ntl; tt += 2) {
    tile_body(tt, acc_b, acc_a, true);
    tile_body(tt + 1, acc_a, acc_b, true);
  }
  if (tt < ntl) {  // odd tail: last tile accumulates in acc_b
    tile_body(tt, acc_b, acc_a, true);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_a[r] = acc_b[r];
    ++tt;
  }
  // the workgroup's last tile (tt - 1 == ntl - 1) is in acc_a
  if (tile_full(ntl - 1)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) ep_write(acc_a, g);
    ep_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) ep_store(ntl - 1, i, ep_read(i));
  } else {
    store_ragged(ntl - 1, acc_a);
  }
}


// ---- query-fragment builder (workspace path) ----------------------------------------------
// One thread per (query row, group of 8 coordinates): gather 4 x 16 B (both halves of the s and
// r rows), build q = s (x) r in f32, round to bf16 and write the two 16-B fragments where the
// scoring kernel loads them: qf[((row/32 * NKB + kb) * 64 + lane)], lane = row%32 + 32*(cg&1),
// kb = cg/2 (first half) and NKH + cg/2 (second half).  n * d/16 threads: fully parallel, two
// dependent memory latencies.  Removes the 64-fold redundant query build (8,400 VALU cycles
// per wave) from the scoring kernel's prologue.
template <int SCORER, int HH>
__global__ __launch_bounds__(256) void build_queries_kernel(Operand A, Operand R, int dir,
                                                            long long n, long long nrows,
                                                            u32x4* __restrict__ qf) {
  constexpr int NKB = 2 * HH / 16, NKH = HH / 16, CG = HH / 8;  // coordinate groups per row
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = t / CG;
  const int cg = (int)(t % CG);
  if (row >= nrows) return;
  long long qrow = row < n ? row : n - 1;  // padded rows repeat row n-1
  const unsigned short* a = (const unsigned short*)A.base + index_at(A.idx, qrow) * A.ld + cg * 8;
  const unsigned short* r = (const unsigned short*)R.base + index_at(R.idx, qrow) * R.ld + cg * 8;
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
  u32x4* dst = qf + ((row >> 5) * NKB) * 64 + (row & 31) + 32 * (cg & 1);
  dst[(cg >> 1) * 64] = q0;
  dst[(NKH + (cg >> 1)) * 64] = q1;
}

static inline bool v2_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool pairs_bf16_v2_supported(int scorer, int dtype, int d, const Operand& A, const Operand& R,
                             const Operand& TG) {
  if (dtype != KGE_BF16) return false;
  if (scorer != KGE_COMPLEX && scorer != KGE_DISTMULT) return false;
  if (d != 128 && d != 256 && d != 512) return false;
  if (!v2_al16(A.base) || !v2_al16(R.base) || !v2_al16(TG.base)) return false;
  if ((A.ld % 8) || (R.ld % 8) || (TG.ld % 8)) return false;
  return true;
}

template <int SCORER, int HH>
static int launch_v2(const Operand& A, const Operand& R, const Operand& TG, int dir, long long n,
                     long long m, float* out, long long ldo, hipStream_t st,
                     unsigned long long* dbg, void* ws, long long ws_bytes) {
  const int rgn = (int)((n + V2_ROWS - 1) / V2_ROWS);
  const int ntiles = (int)((m + V2_TN - 1) / V2_TN);
  // one workgroup per CU (256 CUs): split the target tiles into column groups
  int ncg = 256 / rgn;
  if (ncg < 1) ncg = 1;
  int tpc = (ntiles + ncg - 1) / ncg;
  if (tpc < 1) tpc = 1;
  ncg = (ntiles + tpc - 1) / tpc;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  const int tgmode = TG.idx.ptr == nullptr ? 0 : (TG.idx.itype ? 2 : 1);
  const long long rb32 = (long long)rgn * (V2_ROWS / 32);  // every wave of every row group loads a block
  const bool preq = ws != nullptr && v2_al16(ws) && ws_bytes >= rb32 * 32 * (long long)HH * 4;
  u32x4* qf = (u32x4*)ws;
  if (preq) {
    const long long nrows = rb32 * 32, nthreads = nrows * (HH / 8);
    hipLaunchKernelGGL((build_queries_kernel<SCORER, HH>), dim3((unsigned)((nthreads + 255) / 256)),
                       dim3(256), 0, st, A, R, dir, n, nrows, qf);
  }
#define KGE_V2L(MODE)                                                                             \
  if (preq)                                                                                       \
    hipLaunchKernelGGL((pairs_bf16_v2_kernel<SCORER, HH, MODE, 0, true>), dim3(grid), dim3(256), 0, \
                       st, A, R, TG, dir, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg, qf);        \
  else                                                                                            \
    hipLaunchKernelGGL((pairs_bf16_v2_kernel<SCORER, HH, MODE, 0, false>), dim3(grid), dim3(256),   \
                       0, st, A, R, TG, dir, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg, qf)
  if (tgmode == 0) { KGE_V2L(0); }
  else if (tgmode == 1) { KGE_V2L(1); }
  else { KGE_V2L(2); }
#undef KGE_V2L
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// debug: ComplEx d=512, all entities, with ablation mode
int run_pairs_bf16_v2_ablate(int abl, const Operand& A, const Operand& R, const Operand& TG, long long n,
                             long long m, float* out, long long ldo, hipStream_t st,
                             unsigned long long* dbg) {
  const int rgn = (int)((n + V2_ROWS - 1) / V2_ROWS);
  const int ntiles = (int)((m + V2_TN - 1) / V2_TN);
  int ncg = 256 / rgn;
  if (ncg < 1) ncg = 1;
  int tpc = (ntiles + ncg - 1) / ncg;
  ncg = (ntiles + tpc - 1) / tpc;
  const int grid = 8 * rgn * ((ncg + 7) / 8);
#define KGE_ABL(X)                                                                              \
  hipLaunchKernelGGL((pairs_bf16_v2_kernel<KGE_COMPLEX, 256, 0, X>), dim3(grid), dim3(256), 0, st, \
                     A, R, TG, (int)KGE_SP_, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg, nullptr)
  if (abl == 1) KGE_ABL(1);
  else if (abl == 2) KGE_ABL(2);
  else if (abl == 3) KGE_ABL(3);
  else KGE_ABL(0);
#undef KGE_ABL
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_pairs_bf16_v2(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir,
                      int d, long long n, long long m, float* out, long long ldo,
                      hipStream_t st, unsigned long long* dbg, void* ws, long long ws_bytes) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V2(SC)                                                                   \
  switch (d) {                                                                       \
    case 128: return launch_v2<SC, 64>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes); \
    case 256: return launch_v2<SC, 128>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes); \
    case 512: return launch_v2<SC, 256>(A, R, TG, dir, n, m, out, ldo, st, dbg, ws, ws_bytes); \
  }
  if (scorer == KGE_COMPLEX) { KGE_V2(KGE_COMPLEX) } else { KGE_V2(KGE_DISTMULT) }
#undef KGE_V2
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
