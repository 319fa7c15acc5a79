// ce_pairs_v8.hip -- the fused 1vsAll / KvsAll loss passes (kge_ce_* / kge_kl_*: ce_loss.hip) on the persistent
// two-consumer-waves-per-SIMD structure of pairs_bf16_v8_rank_kernel (score_pairs_bf16_v8.hip), round 6.
//
// Reference: TrainingJob1vsAll scores a batch against all entities and hands the [n, E] matrix to
// KLDivWithSoftmaxKgeLoss = CrossEntropyLoss(reduction="sum") (kge/job/train_1vsAll.py:48-82, kge/util/loss.py:192-207).
// Until round 5 both passes of the fused form -- the row statistics of the forward (V3_LSE) and d loss / d score of the
// backward (V3_DS) -- ran on pairs_bf16_v4_kernel: ONE consumer wave per SIMD that alternated a chain of MFMAs with
// its epilogue arithmetic while the matrix pipe idled, behind an in-launch cooperative query build (~12 k cycles in
// which nothing is scored): 23.6 + 28.5 us of the 149 us of kernels of a training step at the FB15k-237 shape
// (profiles/r5_train_step_kernels.txt) where the counting kernel does the same contraction in 15-17.
//
// Here: prepared query fragments (query_build_kernel, bf16_queries.hpp), 8 x (CUs / 8) persistent workgroups of eight
// waves = two consumer waves per SIMD, the halves of a workgroup half a chain apart so that one wave's epilogue burst
// (exp2 + adds: ~60 VALU operations per 32 columns) runs under the other wave's MFMAs; XCD x owns the column slice x
// of the table; a lane owns ONE query row (MFMA(targets, queries): the row's running (max, sum exp), label, logsumexp
// and upstream gradient are per-lane state).  Same products, same K order: the scores inside are the bits the store
// kernels write.
//
//   V3_LSE  per row and column group (max, sum exp) by online softmax in the log2 domain + the label's score;
//           ce_combine_kernel / kl_*_combine merge the column groups (a column group = one workgroup's share of the
//           row's columns: ncg = 8 XCD slices x the sub-ranges a slice is cut into);
//   V3_DS   g_i (exp(score - lse_i) - row_bias_i - [j == label_i]) as bf16 into G16: a lane holds four runs of four
//           consecutive columns; one v_permlane32_swap per packed dword pairs the two lanes of a row so that every
//           lane stores 16 contiguous bytes (a row's 64 bytes of a unit = two whole 32-byte sectors per instruction).
//
// Synchronisation as in pairs_bf16_v8_rank_kernel: ring of four 32 KiB units in LDS, one workgroup barrier per unit,
// counted s_waitcnt vmcnt(N) with N = the vector-memory operations a wave issues between the pieces it waits for and
// the wait (pieces, and -- V3_DS -- the stores of the two bursts between); every pair of (side, 256-row chunk)
// starts behind a full vmcnt(0).
#include "common.hpp"
#include "bf16_queries.hpp"
#include <atomic>

namespace kge {

constexpr int V8C_UT = 32;  // table rows per sub-unit

struct V8CeArgs {
  Operand TG;
  long long n, m;          // rows per side, table rows (columns)
  int rgn1, sides, chunks; // 128-row fragment groups per side; sides; 256-row chunks per side
  int nunits, su, wpx;     // units in all (V3_DS: covering the padded pitch ld16), per XCD slice; workgroups per XCD
  int nsub, ncg;           // sub-ranges of a slice (pairs < workgroups per XCD), column groups per row
  const u32x4* qf;
  CeArgs ce;
  // V3_STORE (the score store of d = 256 tables, round 6): a GROUP of `nbatch` equally shaped batches (0 = 1), batch
  // l's fragments `q_stride` 16-byte words and its score block `out_stride` floats behind batch l - 1's; rows on the
  // pitch `ldo`, the second side's block `out2_off` floats into a row
  int nbatch;
  long long q_stride, out_stride, out2_off, ldo;
  float* out;
};

#define KGE_V8C_DMA(D, VO, P) \
  KGE_STALL(__LINE__ + 8000); \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(D), "v"(VO), "s"(P) : "memory", "m0")

// EPI == V3_STORE (round 6; VERDICT r5 missing 3): the same structure as the score STORE path of d = 256 tables -- the
// persistent store kernel pairs_bf16_v8_kernel is scheduled by hand for d = 512's 32-slot chains and had no d = 256 form,
// so `configs[4]` (Wikidata5M, d = 256) stored its scores through the round-2/3 kernels.  The operands are swapped for
// this epilogue -- MFMA(queries, targets), the store kernel's orientation: a lane holds ONE column and sixteen rows, a
// dword store writes two rows x 128 contiguous bytes (whole lines; the first form kept the loss kernels' orientation
// and wrote 32 rows x 32 bytes per instruction: 9.9 us per one-sided batch at the FB15k-237 shape, of which the L2's
// merging of partial lines was the larger part) -- and a unit is ONE 32-column sub-unit: 16 stores per burst keep the
// count of operations in flight across two bursts inside the 6-bit vmcnt.  AUX: the stores' cache policy.
// SPLIT (V3_STORE only; KGE_FLAG_SPLIT_QUERY: q = q_hi + q_lo, the parity mode of the score store): the q_hi and q_lo
// rows of 16 real rows are a wave's 32 operand rows, operand row 16 a + 8 part + j = real row 8 a + j (the arrangement of
// pairs_bf16_v8_kernel<SPLIT>): accumulator elements r and r + 4 (r & 4 == 0) of a lane are the two partial scores of
// one (row, column) -- one add, eight stores per sub-unit; a chunk is 128 real rows, fragment groups hold 64.
template <int HH, int EPI, int AUX = 0, int SPLIT = 0>
__global__ __launch_bounds__(512, 1) void pairs_bf16_v8_ce_kernel(V8CeArgs a) {
  static_assert(EPI == V3_LSE || EPI == V3_DS || EPI == V3_STORE, "forward row statistics, the gradient of the scores, or the scores");
  constexpr bool ST = EPI == V3_STORE;
  static_assert(!SPLIT || ST, "split queries: the score store only");
  constexpr int RW = SPLIT ? 16 : 32;     // real query rows per wave
  constexpr int NT = (HH == 128 && !ST) ? 2 : 1;   // 32-row sub-units of a unit, one accumulator each
  constexpr int UT = V8C_UT * NT;         // table rows per unit: 32 / 64
  constexpr int NKB = 2 * HH / 16;        // 32 / 16 K-blocks
  constexpr int ROWB = 4 * HH;            // bytes per table row
  constexpr int SPR = ROWB / 16;          // 16-byte slots per row
  constexpr int RPP = 64 / SPR;           // table rows per 1-KiB piece: 1 / 2
  constexpr int UNITB = UT * ROWB;        // 32 KiB
  constexpr int SUBB = V8C_UT * ROWB;
  constexpr int NBUF = 4;
  constexpr int SMEM = NBUF * UNITB;
  constexpr int NP = UNITB / 1024 / 8;    // pieces per unit and wave: 4
  constexpr int PF = 4;                   // K-blocks read ahead
  constexpr int PB = NKB == 32 ? 14 : 6;  // K-block of the barrier (first half of the workgroup)
  constexpr int NSTORE = EPI == V3_DS ? 2 * NT : (ST ? (SPLIT ? 8 : 16) * NT : 0);  // vector stores of a burst (at least)
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  if (a.n < 0) smem[threadIdx.x] = 0;  // (never: keeps the allocation -- only asm names the array)

  const int b = blockIdx.x;
  const int x = b & 7, j = b >> 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const CeArgs& ce = a.ce;
  int u_lo = x * a.su;
  int sux = a.nunits - u_lo;
  if (sux > a.su) sux = a.su;
  if (sux < 0) sux = 0;
  const int P = (a.nbatch > 1 ? a.nbatch : 1) * a.sides * a.chunks;
  // the workgroup's list of (pair, unit) positions: see pairs_bf16_v8_kernel.  cg = its column group.
  int g1 = 0, pair0 = 0, npairs = 0, cg = x;
  const int pstep = a.wpx;
  if (P >= a.wpx) {
    pair0 = j;
    npairs = j < P ? (P - j + a.wpx - 1) / a.wpx : 0;
    g1 = npairs * sux;
  } else if (j < P * a.nsub) {
    pair0 = j % P;
    npairs = 1;
    const int sub = j / P;
    const int lo = (int)((long long)sux * sub / a.nsub), hi = (int)((long long)sux * (sub + 1) / a.nsub);
    u_lo += lo;
    sux = hi - lo;
    g1 = sux;
    cg = x * a.nsub + sub;
  }
  if (npairs == 0) return;
  if (g1 <= 0) {
    // no unit in this workgroup's range (a table of a few units): its column group still exists for ce_combine
    if constexpr (EPI == V3_LSE) {
      for (int k = 0; k < npairs; ++k) {
        const int pair = pair0 + k * pstep;
        const int side = pair / a.chunks, ch = pair - side * a.chunks;
        const long long row = (long long)ch * 256 + (tid >> 1);
        if ((tid & 1) == 0 && row < a.n) {
          float* pp = ce.part + ((row + (side ? ce.side2_off : 0)) * a.ncg + cg) * 2;
          pp[0] = -__builtin_inff();
          pp[1] = 0.0f;
        }
      }
    }
    return;
  }

  // ---------------- the table stream ----------------
  const unsigned char* const tgb = (const unsigned char*)a.TG.base;
  const long long tld2 = a.TG.ld * 2;
  const long long m = a.m;
  const int lr = lane / SPR, slot = lane % SPR;
  const int rp0 = wave * NP * RPP;  // this wave's first row of every unit
  int dq = 0, du = 0;
  const int ulast = (g1 - 1) % sux;
  auto dma_piece = [&](int un, int ks, auto kc) __attribute__((always_inline)) {
    constexpr int kk = decltype(kc)::value;
    const int ru = rp0 + kk * RPP;  // the piece's first row within the unit
    const long long r0 = (long long)(u_lo + un) * UT + ru;
    const unsigned int dk = (unsigned int)((ks & (NBUF - 1)) * UNITB + ru * ROWB);
    // lane (lr, slot) fetches the 16-byte slot `slot ^ (row & 15)` of row ru + lr into slot `slot` of its LDS row;
    // rows beyond the table repeat its last row (their columns are masked in the epilogue)
    const unsigned int sw = (unsigned int)((slot ^ ((ru + lr) & 15)) << 4);
    const long long rb = r0 < m ? r0 : m - 1;
    const unsigned char* pk = tgb + rb * tld2;
    unsigned int vo = sw;
    if constexpr (RPP > 1) {
      const int left = (int)(m - 1 - rb < RPP - 1 ? m - 1 - rb : RPP - 1);
      vo += (unsigned int)((lr < left ? lr : left) * (int)tld2);
    }
    KGE_V8C_DMA(dk, vo, pk);
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
  bf16x8 afr[NKB];
  unsigned int bp[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bp[t] = (unsigned int)(fi * ROWB + (((2 * t + fh) ^ (fi & 15)) << 4));
  bf16x8 bq[NT][PF];
  auto bread = [&](bf16x8& dst, auto kc, auto ac) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value, sub = decltype(ac)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256 + sub * SUBB) : "memory");
  };

  // ---- per-row state.  Accumulator element r of a sub-unit = column c0 + 8 (r >> 2) + 4 fh + (r & 3).
  long long lab = -1;  // the row's label column (none: -1)
  // V3_LSE, log2 domain: m2 = running max x log2(e) (a float the exponents are taken against), rsum = sum 2^(x log2e - m2)
  float rmax = -__builtin_inff(), m2 = -__builtin_inff(), rsum = 0.0f, tsc = 0.0f;
  bool tfound = false;
  // V3_DS: l2 = lse_i log2(e), g_i = upstream gradient, gb_i = g_i row_bias_i
  float l2 = 0.0f, g_i = 0.0f, gb_i = 0.0f;
  __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)a.qf, 0, 0, 0x00020000);
  unsigned int gvo = 0;  // the lane's byte offset in the pair's G16 rows: row fi, 8 fh columns in
  const unsigned int gld4 = ST ? (unsigned int)(a.ldo * 4) : 0u;  // V3_STORE: bytes per score row

  auto lse_sub = [&](f32x16& v, long long c0u) __attribute__((always_inline)) {
    // c0u: first column of the sub-unit
    const long long c0 = c0u + 4 * fh;
    if (c0u + V8C_UT > m) {  // the ragged last sub-unit of the table: columns beyond m do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = c0 + 8 * (r >> 2) + (r & 3) < m ? v[r] : -__builtin_inff();
    }
    float mx = rmax;
#pragma unroll
    for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %1, %2, %0" : "+v"(mx) : "v"(v[r]), "v"(v[r + 1]));
    // exponents against n2 = mx log2(e); all of this lane's columns masked and nothing before them (mx = -inf): a
    // finite reference keeps inf - inf out of the exponents.  m2 (= -inf before the first column) is the reference
    // of the terms summed so far: 2^(m2 - n2) rescales them, 0 for the empty sum.
    const float n2 = mx == -__builtin_inff() ? 0.0f : mx * V3_LOG2E;
    float sm = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sm += __builtin_amdgcn_exp2f(__builtin_fmaf(v[r], V3_LOG2E, -n2));
    rsum = __builtin_fmaf(rsum, __builtin_amdgcn_exp2f(m2 - n2), sm);
    m2 = mx * V3_LOG2E;
    rmax = mx;
    const long long rel = lab - c0;
    const bool hit = rel >= 0 && rel < V8C_UT && (rel & 7) < 4 && lab < m;
    if (__any(hit)) {  // rare: a row's label lies in exactly one sub-unit of the table
#pragma unroll
      for (int r = 0; r < 16; ++r) tsc = (hit && rel == 8 * (r >> 2) + (r & 3)) ? v[r] : tsc;
      tfound = tfound || hit;
    }
  };
  auto ds_sub = [&](const f32x16& v, long long c0u, unsigned int colb) __attribute__((always_inline)) {
    const long long c0 = c0u + 4 * fh;
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      p[r] = __builtin_fmaf(__builtin_amdgcn_exp2f(__builtin_fmaf(v[r], V3_LOG2E, -l2)), g_i, -gb_i);
    const long long rel = lab - c0;
    const bool hit = rel >= 0 && rel < V8C_UT && (rel & 7) < 4;
    if (__any(hit)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = (hit && rel == 8 * (r >> 2) + (r & 3)) ? p[r] - g_i : p[r];
    }
    if (c0u + V8C_UT > m) {  // pad columns of G16 are zeros: the gradient products read whole 16-byte chunks
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = c0 + 8 * (r >> 2) + (r & 3) < m ? p[r] : 0.0f;
    }
    // pack: run q (columns 8 q + 4 fh + 0..3) -> two dwords; then the lane pair (fi, 0) / (fi, 1) trades runs so that
    // fh = 0 holds columns [0, 8) and [16, 24), fh = 1 holds [8, 16) and [24, 32): v_permlane32_swap(d, s) swaps the
    // upper 32 lanes of d with the lower 32 lanes of s
    unsigned int w[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w[q][0] = bf16_pack_hw(f32x2q{p[4 * q], p[4 * q + 1]});
      w[q][1] = bf16_pack_hw(f32x2q{p[4 * q + 2], p[4 * q + 3]});
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // runs (0, 1) and (2, 3)
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(w[2 * h][e], w[2 * h + 1][e], false, false);
        o[e] = sw[0];      // fh = 0: own run 2h (columns 16 h + 0..3)      fh = 1: partner's run 2h + 1 (16 h + 8..11)
        o[2 + e] = sw[1];  // fh = 0: partner's run 2h (columns 16 h + 4..7)  fh = 1: own run 2h + 1 (16 h + 12..15)
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, grs, gvo + (unsigned int)(32 * h), colb, 0);
    }
  };
  // V3_STORE: acc[r] = score(row 8 (r >> 2) + 4 fh + (r & 3) of the wave's 32, column c0u + fi).  Every store instruction
  // is issued UNCONDITIONALLY (the counted waits count them): rows beyond n fall outside the descriptor, and a lane whose
  // column lies beyond m (the table's ragged last unit) stores through an offset beyond every descriptor -- the hardware
  // drops both.  gvo = the lane's byte offset ((4 fh) rows down, fi columns in); the row of element r rides on top.
  auto sc_sub = [&](const f32x16& v, long long c0u, unsigned int colb) __attribute__((always_inline)) {
    const unsigned int vo = (c0u + V8C_UT <= m || c0u + fi < m) ? gvo : 0x80000000u;
    if constexpr (SPLIT) {
      // element r (r & 4 == 0): real row 8 (r >> 3) + 4 fh + (r & 3) of the wave's 16; score = (sum q_hi t) + (sum q_lo t)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = (q & 3) + 8 * (q >> 2);
        const float hi = v[r], lo = v[r + 4];
        const float x = hi + lo;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), grs,
                                              vo + (unsigned int)(8 * (r >> 3) + (r & 3)) * gld4, colb, AUX);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = v[r];  // (a copy first: __builtin_bit_cast straight on a vector element takes element 0 every time)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), grs,
                                              vo + (unsigned int)(8 * (r >> 2) + (r & 3)) * gld4, colb, AUX);
      }
    }
  };
  // the row's results out (V3_LSE): the two lanes of a row -> one (max, sum exp) per row and column group
  int side_cur = 0;
  long long lrow_cur = 0;
  auto lse_flush = [&]() __attribute__((always_inline)) {
    const float o2 = __shfl_xor(m2, 32, 64), osum = __shfl_xor(rsum, 32, 64);
    const float M2 = __builtin_fmaxf(m2, o2);
    const float ref = M2 == -__builtin_inff() ? 0.0f : M2;  // (neither lane has a column: the empty sum)
    const float L = rsum * __builtin_amdgcn_exp2f(m2 - ref) + osum * __builtin_amdgcn_exp2f(o2 - ref);
    if (lrow_cur < a.n) {
      const long long roff = side_cur ? ce.side2_off : 0;
      if (fh == 0) {
        float* pp = ce.part + ((lrow_cur + roff) * a.ncg + cg) * 2;
        // (max, sum exp(x - max)) with max = M2 / log2(e): M2 = fl(M log2e), so 2^(x log2e - M2) = exp(x - max) up to the
        // rounding of M2 -- a relative 2^-24 |M2| on the sum, the size of the rounding of the score itself
        pp[0] = M2 * V3_LN2;
        pp[1] = L;
      }
      if (tfound) ce.true_score[lrow_cur + roff] = tsc;
    }
    rmax = -__builtin_inff();
    m2 = -__builtin_inff();
    rsum = 0.0f;
    tfound = false;
  };

  auto run = [&](auto half) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half)::value;
    constexpr int PBH = HALF ? 0 : PB;  // slot of the barrier
    // vector-memory operations of a wave in order, per chain: NP pieces (behind the barrier), then -- behind the last
    // MFMA -- the NSTORE stores of the burst.  Behind the pieces of unit k + 1 (chain k - 2) until P(k): burst k - 2,
    // the pieces and the burst of chain k - 1.
    constexpr int VMB = NP + 2 * NSTORE;
    static_assert(VMB < 64, "vmcnt is a 6-bit counter");
    f32x16 acc[NT];
    auto chain = [&](int ks, int cu) __attribute__((always_inline)) {
      const unsigned int bdelta = ((ks + 1) & (NBUF - 1)) ? (unsigned int)UNITB : (unsigned int)(-(NBUF - 1) * UNITB);
      const int un = dq < g1 ? du : ulast;
      v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kc)::value;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"((PF - 1) * NT) : "memory");
        if constexpr (kb == PBH) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(VMB) : "memory");  // this wave's pieces of unit ks + 1 have landed
          KGE_BARRIER();                                // P(ks)
        }
        __builtin_amdgcn_sched_barrier(0);
        v4_static_for<0, NT>([&](auto ac) __attribute__((always_inline)) {
          constexpr int sub = decltype(ac)::value;
          if constexpr (kb == 0) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (ST) acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], bq[sub][0], zero, 0, 0, 0);
            else acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[sub][0], afr[0], zero, 0, 0, 0);
          } else {
            if constexpr (ST) acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[kb], bq[sub][kb % PF], acc[sub], 0, 0, 0);
            else acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[sub][kb % PF], afr[kb], acc[sub], 0, 0, 0);
          }
        });
        if constexpr (kb + PF == NKB) {
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("v_add_u32 %0, %1, %0" : "+v"(bp[t]) : "s"(bdelta));
        }
        v4_static_for<0, NT>([&](auto ac) __attribute__((always_inline)) {
          bread(bq[decltype(ac)::value][kb % PF], std::integral_constant<int, (kb + PF) % NKB>{}, ac);
        });
        if constexpr (kb > PBH && kb <= PBH + NP) dma_piece(un, ks + NBUF - 1, std::integral_constant<int, kb - PBH - 1>{});
      });
      dma_advance();
      // the burst
#pragma unroll
      for (int sub = 0; sub < NT; ++sub) {
        const long long c0u = ((long long)(u_lo + cu) * NT + sub) * V8C_UT;
#if defined(KGE_V8C_PROBE) && KGE_V8C_PROBE == 1  // timing probe: the chains alone (wrong results)
        asm volatile("" : : "v"(acc[sub][0]), "v"(acc[sub][15]));
        (void)c0u;
#else
        if constexpr (EPI == V3_LSE) lse_sub(acc[sub], c0u);
        else if constexpr (EPI == V3_STORE) sc_sub(acc[sub], c0u, (unsigned int)(c0u * 4));
        else ds_sub(acc[sub], c0u, (unsigned int)(c0u * 2));
#endif
      }
    };

    int g = 0, ks = 0;
    int pair = pair0;
    bool first = true;
    while (g < g1) {
      const int cnt = sux;
      const int per_b = a.sides * a.chunks;
      const int lb = pair / per_b, prem = pair - lb * per_b;  // (a group of batches: V3_STORE only)
      const int side = prem / a.chunks, ch = prem - side * a.chunks;
      // ---- this lane's row of the pair
      const long long rb = (long long)ch * (8 * RW) + RW * wave;  // the wave's first row (of the side)
      const long long lrow = rb + fi;
      const long long orow = lrow < a.n ? lrow : a.n - 1;  // padded rows repeat row n - 1 (and never write)
      const long long roff = side ? ce.side2_off : 0;
      side_cur = side;
      lrow_cur = lrow;
      if constexpr (EPI != V3_STORE) {
        const Index& lix = side ? ce.label2 : ce.label;
        lab = lix.ptr != nullptr ? index_at(lix, orow) : -1;
      }
      if constexpr (EPI == V3_STORE) {
        // the wave's rows of the score block: rows beyond n fall outside the descriptor and are dropped by the hardware
        const long long rows_here = rb < a.n ? (a.n - rb < RW ? a.n - rb : RW) : 0;
        float* const ob = a.out + (long long)lb * a.out_stride + (side ? a.out2_off : 0) + (rows_here > 0 ? rb : 0) * a.ldo;
        grs = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, (int)(rows_here * a.ldo * 4), 0x00020000);
        gvo = (unsigned int)(((long long)(4 * fh) * a.ldo + fi) * 4);
      }
      if constexpr (EPI == V3_DS) {
        l2 = ce.lse[orow + roff] * V3_LOG2E;
        g_i = ce_row_gradient(ce, orow + roff);
        if (ce.rowptr != nullptr && ce.rowptr[orow + 1] == ce.rowptr[orow]) g_i = 0.0f;  // (one-sided multi-label loss)
        gb_i = ce.row_bias != nullptr ? g_i * ce.row_bias[orow + roff] : 0.0f;
        // the wave's rows of G16: rows beyond n fall outside the descriptor and are dropped by the hardware
        const long long rows_here = rb < a.n ? (a.n - rb < 32 ? a.n - rb : 32) : 0;
        unsigned short* const gb = ce.g16 + (roff + (rows_here > 0 ? rb : 0)) * ce.ld16;
        grs = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, (int)(rows_here * ce.ld16 * 2), 0x00020000);
        gvo = (unsigned int)(((long long)fi * ce.ld16 + 8 * fh) * 2);
      }
      // ---- fragments: groups of 128 operand rows = four blocks of 32 (bf16_queries.hpp); a chunk = two groups, wave w
      // takes block w & 3 of group w >> 2
      int grp = 2 * ch + (wave >> 2);
      if (grp >= a.rgn1) grp = a.rgn1 - 1;
      grp += side * a.rgn1;
      const unsigned char* const gbase = (const unsigned char*)(a.qf + (long long)lb * a.q_stride + (long long)grp * 4 * NKB * 64);
      unsigned int flo;
      const unsigned char* fb;
      int frange;
      if constexpr (SPLIT) {
        // operand row fi = 16 a + 8 part + jj of wave w: real row 16 (w & 3) + 8 a + jj of the group's 64, whose q_hi
        // sits in block (row >> 5), its q_lo in block 2 + (row >> 5) (bf16_queries.hpp)
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
      v4_static_for<0, NKB>([&](auto kc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kc)::value;
        afr[kb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, flo + kb * 1024, 0, 16 /* sc1 */));
      });
      // fragments, row state, pieces of the ring fill, the stores of the pair before: everything of this wave has landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // The compiler does not read the wait above: without a use of the fragments HERE it puts its own waits in front
      // of their first uses -- inside the chain loop, draining the table pieces requested for the units ahead.
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(afr[kb]));
      if constexpr (EPI == V3_DS) asm volatile("" : "+v"(l2), "+v"(g_i), "+v"(gb_i));
      {
        int lab_lo = (int)lab, lab_hi = (int)(lab >> 32);
        asm volatile("" : "+v"(lab_lo), "+v"(lab_hi));
        lab = ((long long)lab_hi << 32) | (unsigned int)lab_lo;
      }
      if (first) {
        KGE_BARRIER();  // R0: units 0 .. 2 of the list have landed
        v4_static_for<0, PF>([&](auto jc) __attribute__((always_inline)) {
          v4_static_for<0, NT>([&](auto ac) __attribute__((always_inline)) {
            bread(bq[decltype(ac)::value][decltype(jc)::value], jc, ac);
          });
        });
        first = false;
      }
      for (int i = 0; i < cnt; ++i) {
        chain(ks, i);
        ++ks;
      }
      if constexpr (EPI == V3_LSE) lse_flush();
      g += cnt;
      pair += pstep;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int sub = 0; sub < NT; ++sub)
#pragma unroll
      for (int jj = 0; jj < PF; ++jj) asm volatile("" : : "v"(bq[sub][jj]));
  };
  if (wave < 4) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

#undef KGE_V8C_DMA

static int v8c_cu_count() {
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

// Geometry of a launch over `n` rows per side (`two_sided`: two sides) against m columns: how many column groups a
// row's statistics come in (V3_LSE: the layout of CeArgs::part), 0 = not this kernel's case.  A function of the
// shape alone (the device's CU count is clamped to 256 as in the launch below).
static bool v8c_geometry(int d, long long n, long long m, bool two_sided, int epi, long long ld16, V8CeArgs& a,
                         int nbatch = 1, int reserve_cus = 0, bool split = false) {
  if ((d != 512 && d != 256) || n < 1 || m < 1) return false;
  const long long rgr = split ? 64 : 128;  // real rows per fragment group (a chunk = two groups)
  const long long rgn1 = (n + rgr - 1) / rgr;
  const long long ut = (d == 256 && epi != V3_STORE) ? 2 * V8C_UT : V8C_UT;
  const long long cols = epi == V3_DS ? ld16 : m;  // the gradient pass also writes the pad columns of the pitch
  const long long nunits = (cols + ut - 1) / ut;
  if (rgn1 > (1 << 20) || (rgn1 + 1) * nunits >= (1LL << 30)) return false;
  int cus = v8c_cu_count() - reserve_cus;
  if (cus > 256) cus = 256;
  if (cus < 8) return false;
  a.n = n;
  a.m = m;
  a.rgn1 = (int)rgn1;
  a.sides = two_sided ? 2 : 1;
  a.chunks = (int)((rgn1 + 1) / 2);
  a.nunits = (int)nunits;
  a.su = (int)((nunits + 7) / 8);
  a.wpx = cus / 8;
  a.nbatch = nbatch;
  if ((long long)nbatch * (rgn1 + 1) * nunits >= (1LL << 30)) return false;
  const int P = nbatch * a.sides * a.chunks;
  a.nsub = P >= a.wpx ? 1 : a.wpx / P;
  if (a.nsub > a.su) a.nsub = a.su > 0 ? a.su : 1;  // no more sub-ranges than a slice has units
  a.ncg = 8 * a.nsub;
  return true;
}

// Does the persistent kernel take this pass?  Rows per side above half a 256-row chunk (below, half of every
// workgroup's operand rows would be padding: pairs_bf16_v4_kernel's 128-row groups fit better), d in {256, 512}.
bool pairs_bf16_v8_ce_takes(int d, long long n, long long m) {
  if (d != 512 && d != 256) return false;
  const long long tail = n % 256;
  return n > 128 && (tail == 0 || tail > 128 || n >= 1024) && m >= 1;
}

// column groups of a row's (max, sum exp) partials when pairs_bf16_v8_ce_kernel<V3_LSE> runs the pass
int pairs_bf16_v8_ce_column_groups(int d, long long n, long long m, bool two_sided) {
  V8CeArgs a{};
  if (!v8c_geometry(d, n, m, two_sided, V3_LSE, 0, a)) return 0;
  return a.ncg;
}

// One loss pass from PREPARED query fragments `qf` (v4_build_queries' layout: the row groups of side 1, then side 2):
// epi = V3_LSE (ce.part with pairs_bf16_v8_ce_column_groups column groups, ce.true_score) or V3_DS (ce.g16, ce.ld16:
// a multiple of 64).  Two sides: ce.side2_off, ce.label2 describe the second.  KGE_ERR_UNSUPPORTED: not this kernel's case.
int run_pairs_bf16_v8_ce(int epi, const Operand& TG, int d, long long n, long long m, bool two_sided, const void* qf,
                         const CeArgs& ce, hipStream_t st) {
  if (TG.idx.ptr != nullptr || qf == nullptr || ((uintptr_t)qf & 15)) return KGE_ERR_UNSUPPORTED;
  if (epi != V3_LSE && epi != V3_DS) return KGE_ERR_UNSUPPORTED;
  if (TG.ld * 2 >= (1LL << 28)) return KGE_ERR_UNSUPPORTED;
  if (epi == V3_DS && (ce.g16 == nullptr || (ce.ld16 & 63) || ce.ld16 < m || ce.ld16 * 2 * 32 >= (1LL << 31) ||
                       ((uintptr_t)ce.g16 & 15)))
    return KGE_ERR_UNSUPPORTED;
  V8CeArgs a{};
  if (!v8c_geometry(d, n, m, two_sided, epi, ce.ld16, a)) return KGE_ERR_UNSUPPORTED;
  a.TG = TG;
  a.qf = (const u32x4*)qf;
  a.ce = ce;
  const dim3 grid(8 * a.wpx), block(512);
  if (d == 512) {
    if (epi == V3_LSE) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<256, V3_LSE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<256, V3_DS>), grid, block, 0, st, a);
  } else {
    if (epi == V3_LSE) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_LSE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_DS>), grid, block, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// Scores of `nbatch` prepared batches (single-pass or split queries) against the identity-indexed bf16 table TG at d = 256 -- what
// run_pairs_bf16_v8 is at d = 512 (same arguments; no in-launch build of the next group: the caller launches it).
// sc1: the stores' cache policy as there (0 plain, 1 write-through, 2 non-temporal).
int run_pairs_bf16_v8_store256(const Operand& TG, bool split, bool two_sided, long long n, long long m, int nbatch,
                               const void* qf, long long q_stride_bytes, float* out, long long out_stride, long long ldo,
                               long long out2_off, int sc1, int reserve_cus, hipStream_t st) {
  if (TG.idx.ptr != nullptr || qf == nullptr || nbatch < 1 || ((uintptr_t)qf & 15) || (q_stride_bytes & 15))
    return KGE_ERR_UNSUPPORTED;
  if (TG.ld * 2 >= (1LL << 28) || ldo >= (1LL << 24)) return KGE_ERR_UNSUPPORTED;
  V8CeArgs a{};
  if (!v8c_geometry(256, n, m, two_sided, V3_STORE, 0, a, nbatch, reserve_cus, split)) return KGE_ERR_UNSUPPORTED;
  a.TG = TG;
  a.qf = (const u32x4*)qf;
  a.q_stride = q_stride_bytes / 16;
  a.out = out;
  a.out_stride = out_stride;
  a.ldo = ldo;
  a.out2_off = out2_off;
  const dim3 grid(8 * a.wpx), block(512);
  if (split) {
    if (sc1 == 1) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 16, 1>), grid, block, 0, st, a);
    else if (sc1 == 2) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 2, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 0, 1>), grid, block, 0, st, a);
  } else if (sc1 == 1) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 16>), grid, block, 0, st, a);
  else if (sc1 == 2) hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((pairs_bf16_v8_ce_kernel<128, V3_STORE, 0>), grid, block, 0, st, a);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
