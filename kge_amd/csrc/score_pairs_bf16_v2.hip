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

__device__ __forceinline__ unsigned int v2_pack(float lo, float hi) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// two coordinates per dword: (a0,a1) entity halves, (r0,r1) relation halves -> (q0,q1)
template <int SCORER>
__device__ __forceinline__ void v2_qpair(int dir, unsigned int a0, unsigned int a1,
                                         unsigned int r0, unsigned int r1, unsigned int& q0,
                                         unsigned int& q1) {
  const float a0l = __uint_as_float(a0 << 16), a0h = __uint_as_float(a0 & 0xffff0000u);
  const float a1l = __uint_as_float(a1 << 16), a1h = __uint_as_float(a1 & 0xffff0000u);
  const float r0l = __uint_as_float(r0 << 16), r0h = __uint_as_float(r0 & 0xffff0000u);
  const float r1l = __uint_as_float(r1 << 16), r1h = __uint_as_float(r1 & 0xffff0000u);
  float q0l, q0h, q1l, q1h;
  if (SCORER == KGE_DISTMULT) {
    q0l = a0l * r0l; q0h = a0h * r0h; q1l = a1l * r1l; q1h = a1h * r1h;
  } else if (dir == KGE_SP_) {
    q0l = a0l * r0l - a1l * r1l; q0h = a0h * r0h - a1h * r1h;
    q1l = a1l * r0l + a0l * r1l; q1h = a1h * r0h + a0h * r1h;
  } else {
    q0l = r0l * a0l + r1l * a1l; q0h = r0h * a0h + r1h * a1h;
    q1l = r0l * a1l - r1l * a0l; q1h = r0h * a1h - r1h * a0h;
  }
  q0 = v2_pack(q0l, q0h);
  q1 = v2_pack(q1l, q1h);
}

__device__ __forceinline__ long long shfl64(long long v, int src) {
  int lo = __shfl((int)(v & 0xffffffffLL), src, 64);
  int hi = __shfl((int)(v >> 32), src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// row index through an index vector; MODE 0 = identity, 1 = int32, 2 = int64 (no branches)
template <int MODE>
__device__ __forceinline__ long long v2_index(const Index& ix, long long i) {
  if (MODE == 0) return i;
  if (MODE == 1) return (long long)((const int*)ix.ptr)[i * ix.stride];
  return ((const long long*)ix.ptr)[i * ix.stride];
}

template <int SCORER, int HH, int TGMODE>
__global__ __launch_bounds__(256, 1) void pairs_bf16_v2_kernel(
    Operand A, Operand R, Operand TG, int dir, long long n, long long m, int rgn, int ncg,
    int tiles_per_cg, int ntiles, float* __restrict__ out, long long ldo,
    unsigned long long* __restrict__ dbg) {
  constexpr int NKB = 2 * HH / 16;        // K-blocks of 16
  constexpr int NKH = HH / 16;            // K-blocks per half
  constexpr int ROWB = 4 * HH;            // bytes per table row (2*HH bf16)
  constexpr int SPR = HH / 4;             // 16-byte slots per row
  constexpr int TILEB = V2_TN * ROWB;     // bytes per target tile
  constexpr int NL = TILEB / 16 / 256;    // 16-byte DMA ops per wave-lane per tile
  constexpr int PASSES = HH / 64;         // prologue passes of 64 coordinates
  constexpr int STAGE = 4 * 16384;        // one prologue staging slot: 16 KiB per wave
  constexpr int STG0 = TILEB;             // two slots behind ring buffer 0
  constexpr int SMEM = (3 * TILEB > STG0 + 2 * STAGE) ? 3 * TILEB : STG0 + 2 * STAGE;
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
      src = tgb + v2_index<TGMODE>(TG.idx, tr) * TG.ld + ((slot ^ (row & 15)) << 3);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto tile_dma = [&](int tt, int buf) {
#pragma unroll
    for (int k = 0; k < NL; ++k) dma_piece(tt, buf, k);
  };

  // ---- prologue: build the query fragments of this wave's 32 rows in registers.
  // Tile 0 streams first (it needs no index), then the s / r rows are gathered by LDS-DMA
  // into a wave-private staging area, two passes (of 64 coordinates) in flight.  The VMEM
  // counter retires in order, so the gathers are waited with counted vmcnt: issue order
  // T0, G0, G1, then G(p+2) once pass p has been consumed.
  tile_dma(0, 0);
  bf16x8 afr[NKB];
  {
    long long qrow = row0 + fi;
    if (qrow >= n) qrow = n - 1;
    const long long aoff = index_at(A.idx, qrow) * A.ld;  // element offsets of row `fi`
    const long long roff = index_at(R.idx, qrow) * R.ld;
    const unsigned short* ab = (const unsigned short*)A.base;
    const unsigned short* rb = (const unsigned short*)R.base;
    // source pointer of DMA instruction k: row rr = 2k + fh, LDS slot (lane & 31) holds the
    // logical slot p5 = (lane & 31) ^ (rr & 15) = array (p5 >> 3), 16-B chunk (p5 & 7)
    const unsigned short* gsrc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int rr = 2 * k + fh;
      const int p5 = (lane & 31) ^ (rr & 15);
      const long long ao = shfl64(aoff, rr), ro = shfl64(roff, rr);
      gsrc[k] = ((p5 < 16) ? ab + ao : rb + ro) + ((p5 >> 3) & 1) * HH + (p5 & 7) * 8;
    }
    stamp();  // 1: tile 0 issued, indices loaded, source pointers built
    auto gather = [&](int p) {
      unsigned char* dst = smem + STG0 + (p & 1) * STAGE + wave * 16384;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(gsrc[k] + 64 * p),
            (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, 0, 0);
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
          v2_qpair<SCORER>(dir, v[0][e], v[1][e], v[2][e], v[3][e], x0, x1);
          q0[e] = x0;
          q1[e] = x1;
        }
        afr[4 * p + jj] = __builtin_bit_cast(bf16x8, q0);
        afr[NKH + 4 * p + jj] = __builtin_bit_cast(bf16x8, q1);
      }
    };
    gather(0);
    if (PASSES > 1) gather(1);
    stamp();  // 2: gathers issued
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      // ops newer than G_p: G_{p+1} (16) if it exists
      if (p + 1 < PASSES) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stamp();  // 3+2p: pass p landed
      build(p);
      // the staging slot of pass p is free again once this wave's reads have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stamp();  // 4+2p: pass p built
      if (p + 2 < PASSES) gather(p + 2);
    }
  }
  // staging overlaps ring buffers 1 and 2: everyone must be done before tile 1 streams in
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  tile_dma(1, 1);
  stamp();  // 3+2*PASSES: prologue done, all waves synchronised

  // ---- main loop over this workgroup's target tiles
  unsigned int boff[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) boff[t] = (unsigned int)(((2 * t + fh) ^ (fi & 15)) << 4);
  // this lane's output row (query row0 + fi; clamped rows rewrite the bits of row n-1)
  long long orow = row0 + fi;
  if (orow >= n) orow = n - 1;
  float* const orow_ptr = out + orow * ldo;

  // acc[4g + e] = score(query fi, target col0 + 8g + 4fh + e)
  auto store_tile = [&](int tt, const f32x16& acc) {
    const long long col0 = (long long)(tile_lo + tt) * V2_TN;
    if (col0 + V2_TN <= m) {  // wave-uniform: full tile -> 4 x 16-byte stores per lane
      float* p = orow_ptr + col0 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4u v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4u*>(p + 8 * g) = v;
      }
    } else {  // ragged last tile of the table: scalar stores, clamped to column m-1
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        long long oc = col0 + 8 * (r >> 2) + 4 * fh + (r & 3);
        if (oc >= m) oc = m - 1;
        orow_ptr[oc] = acc[r];
      }
    }
  };

  // One tile: wait + barrier, first batch of B reads, the PREVIOUS tile's stores (their
  // VALU / VMEM issue hides under the LDS latency), then the MFMA chain with the DMA of
  // tile tt+2 interleaved piece by piece (an LDS-DMA piece costs ~100 issue cycles: behind
  // 4 MFMAs it is free, in front of the chain it was 800 cycles per tile).
  // VMEM order per wave: T1 | T2 | S0 T3 | S1 T4 | ...  -> newer than T(tt) at the wait of
  // tile tt: NL at tt == 1, NL + 4 from tt == 2 on (tile 0 landed before the barrier above).
  f32x16 accp;
#pragma unroll
  for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
  auto tile_body = [&](int tt, bool store_prev) {
    if (tt == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NL) : "memory");
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
    auto bfrag = [&](int kb) {
      const int s0 = (kb < NKH) ? (2 * kb) : (HH / 8 + 2 * (kb - NKH));
      return *reinterpret_cast<const bf16x8*>(smem + bp[(s0 & 15) >> 1] + (s0 >> 4) * 256);
    };
    // The TARGET fragment is the MFMA "A" operand and the query fragment the "B" operand,
    // so the accumulator holds, for query row fi, 4 x 4 CONSECUTIVE targets: 16-byte stores.
    // B fragments are double-buffered in batches of BB reads (hipcc retires LDS reads with
    // lgkmcnt(0) here, so a wait also covers the newest read: batch b+1 is issued ahead of
    // the MFMAs of batch b).
    constexpr int BB = (NKB >= 16) ? 8 : 4;
    constexpr int NB = NKB / BB;
    constexpr int DSTEP = NKB / NL;  // one DMA piece every DSTEP MFMAs
    bf16x8 bq[2][BB];
#pragma unroll
    for (int j = 0; j < BB; ++j) bq[0][j] = bfrag(j);
    if (store_prev) store_tile(tt - 1, accp);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) {
#pragma unroll
        for (int j = 0; j < BB; ++j) bq[(b + 1) & 1][j] = bfrag((b + 1) * BB + j);
      }
#pragma unroll
      for (int j = 0; j < BB; ++j) {
        const int kb = b * BB + j;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[b & 1][j], afr[kb], acc, 0, 0, 0);
        if ((kb % DSTEP) == DSTEP - 1) dma_piece(tt + 2, (tt + 2) % 3, kb / DSTEP);
      }
    }
    stamp();  // tile tt: MFMA chain issued
    accp = acc;
  };

  tile_body(0, false);
  for (int tt = 1; tt < ntl; ++tt) tile_body(tt, true);
  store_tile(ntl - 1, accp);
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
                     unsigned long long* dbg) {
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
#define KGE_V2L(MODE)                                                                          \
  hipLaunchKernelGGL((pairs_bf16_v2_kernel<SCORER, HH, MODE>), dim3(grid), dim3(256), 0, st, A, \
                     R, TG, dir, n, m, rgn, ncg, tpc, ntiles, out, ldo, dbg)
  if (tgmode == 0) KGE_V2L(0);
  else if (tgmode == 1) KGE_V2L(1);
  else KGE_V2L(2);
#undef KGE_V2L
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

int run_pairs_bf16_v2(int scorer, const Operand& A, const Operand& R, const Operand& TG, int dir,
                      int d, long long n, long long m, float* out, long long ldo,
                      hipStream_t st, unsigned long long* dbg) {
  if (n == 0 || m == 0) return KGE_OK;
#define KGE_V2(SC)                                                                   \
  switch (d) {                                                                       \
    case 128: return launch_v2<SC, 64>(A, R, TG, dir, n, m, out, ldo, st, dbg);      \
    case 256: return launch_v2<SC, 128>(A, R, TG, dir, n, m, out, ldo, st, dbg);     \
    case 512: return launch_v2<SC, 256>(A, R, TG, dir, n, m, out, ldo, st, dbg);     \
  }
  if (scorer == KGE_COMPLEX) { KGE_V2(KGE_COMPLEX) } else { KGE_V2(KGE_DISTMULT) }
#undef KGE_V2
  return KGE_ERR_UNSUPPORTED;
}

}  // namespace kge
