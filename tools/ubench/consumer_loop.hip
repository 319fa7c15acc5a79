// Micro-benchmark of the consumer-wave loop of pairs_bf16_v4_kernel (d = 512): per tile 64 x
// {s_waitcnt lgkmcnt, v_mfma_f32_32x32x16_bf16, ds_read_b128} against a 64 KiB target tile resident in LDS,
// query fragments in registers.  The production loop issues one MFMA every 44-48 cycles instead of 32
// (profiles/r12_phase_timestamps.txt); this isolates why, one variable at a time:
//
//   WAVES   4: the four consumer waves alone (one per SIMD); 8: + four partner waves that only keep the barriers;
//           9: + partner waves that stream a 64 KiB tile per tile period into a second LDS buffer with LDS-DMA
//   NSET    fragment register sets (8 = production: the read issued behind MFMA q overwrites the operand MFMA q
//           has just been issued with; 10/12: the read targets a set last used 2/4 MFMAs earlier)
//   ORDER   0: wait, MFMA, read (production); 1: read, wait, MFMA
//   BAR     1: the two workgroup barriers per tile of the production loop; 0: none
//   STG     1: the eight 16-byte staging writes at the start of a tile; 0: none
//   NOLDS   1: no fragment reads at all (MFMA-only floor)
// Prints cycles per MFMA (wave 0 of block 0, s_memtime) for 1 and 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o consumer_loop consumer_loop.hip && ./consumer_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

constexpr int HH = 256, NKB = 32, NKH = 16, ROWB = 1024, TILEB = 65536, NQ = 64, QB2 = 48;
constexpr int NT = 16;

template <int WAVES, int NSET, int PF, int ORDER, int BAR, int STG, int NOLDS>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ g, float* __restrict__ out,
                                            unsigned long long* __restrict__ t, const unsigned char* __restrict__ gt) {
  static_assert(ORDER == 0 ? PF <= NSET : PF < NSET, "a read must not target the operand of an MFMA still to be issued");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILEB + 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < TILEB / 16; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 511];
  __syncthreads();
  if (wave >= 4) {
    // partner waves: barriers only (WAVES == 8) or barriers + a tile of LDS-DMA per period (WAVES == 9)
    const unsigned int voff = (unsigned int)(lane * 16);
    for (int tt = 0; tt < NT; ++tt) {
      if (BAR) __builtin_amdgcn_s_barrier();
      if (WAVES == 9) {
        unsigned int d = (unsigned int)(TILEB + (wave - 4) * 16384);
        const unsigned char* p = gt + (size_t)(wave - 4) * 16384;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(voff), "s"(p) : "memory", "m0");
          d += 1024;
          p += 1024;
        }
      }
      if (BAR) __builtin_amdgcn_s_barrier();
      if (WAVES == 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return;
  }
  const int fi = lane & 31, fh = lane >> 5, w4 = wave;
  bf16x8 afr[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) afr[kb] = g[(lane + kb * 7) & 511];
  unsigned int boff[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) boff[q] = (unsigned int)(fi * ROWB + (((2 * q + fh) ^ (fi & 15)) << 4));
  const unsigned int cwr = (unsigned int)(2 * TILEB + w4 * 8192 + fi * 256);
  const int y = fh ^ (fi & 15);
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
  unsigned long long t0 = 0, t1 = 0;
  for (int tt = 0; tt < NT; ++tt) {
    if (tt == 2) t0 = __builtin_readcyclecounter();
    if (BAR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned int bp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bp[q] = boff[q];
    bf16x8 bq[NSET];
#pragma unroll
    for (int i = 0; i < NSET; ++i) bq[i] = afr[i];
    auto bread = [&](bf16x8& dst, auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int kb = q >> 1, hf = q & 1;
      constexpr int s0 = (kb < NKH) ? (2 * kb) : (HH / 8 + 2 * (kb - NKH));
      const unsigned int addr = bp[(s0 & 15) >> 1];
      if (!NOLDS)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((s0 >> 4) * 256 + hf * 32 * ROWB) : "memory");
    };
    sfor<0, PF>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value % NSET], jc); });
    if (STG) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 v = {acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]};
        *reinterpret_cast<f32x4*>(smem + cwr + (((2 * gq) ^ y) << 4)) = v;
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 v = {acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]};
        *reinterpret_cast<f32x4*>(smem + cwr + (((8 + 2 * gq) ^ y) << 4)) = v;
      }
    }
    sfor<0, NQ>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      if constexpr (ORDER == 1) {
        if constexpr (q + PF < NQ) bread(bq[(q + PF) % NSET], std::integral_constant<int, q + PF>{});
      }
      // reads younger than read q at this point
      constexpr int rest = NQ - 1 - q;
      constexpr int inflight = ORDER == 1 ? (rest >= PF ? PF : rest) : (rest >= PF - 1 ? PF - 1 : rest);
      constexpr int younger = q < PF ? ((STG ? 8 : 0) + (ORDER == 1 ? (rest >= PF ? PF : rest) : PF - 1)) : inflight;
      if (!NOLDS) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(younger > 15 ? 15 : younger) : "memory");
      if constexpr (q == QB2 && BAR) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (q == 0) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], afr[0], zero, 0, 0, 0);
      } else if constexpr (q == 1) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[1 % NSET], afr[0], zero, 0, 0, 0);
      } else if constexpr (q & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % NSET], afr[q >> 1], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[q % NSET], afr[q >> 1], acc0, 0, 0, 0);
      }
      if constexpr (ORDER == 0) {
        if constexpr (q + PF < NQ) bread(bq[(q + PF) % NSET], std::integral_constant<int, q + PF>{});
      }
    });
  }
  t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) t[blockIdx.x] = t1 - t0;
}

static bf16x8* g;
static float* outp;
static unsigned long long* tp;
static unsigned char* gt;

template <int WAVES, int NSET, int PF, int ORDER, int BAR, int STG, int NOLDS>
static void run(const char* name) {
  for (int blocks : {1, 256}) {
    const int threads = WAVES == 4 ? 256 : 512;
    for (int w = 0; w < 3; ++w)
      hipLaunchKernelGGL((k<WAVES, NSET, PF, ORDER, BAR, STG, NOLDS>), dim3(blocks), dim3(threads), 0, 0, g, outp, tp, gt);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, tp, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; ++b) avg += (double)h[b];
    avg /= blocks;
    printf("%-64s blocks=%3d  %6.1f cycles/MFMA (block 0), %6.1f (mean over blocks)\n", name, blocks,
           h[0] / (double)((NT - 2) * NQ), avg / (double)((NT - 2) * NQ));
  }
}

int main() {
  hipMalloc(&g, 512 * 16);
  hipMemset(g, 0x3c, 512 * 16);
  hipMalloc(&outp, 256 * 512 * 4);
  hipMalloc(&tp, 4096 * 8);
  hipMalloc(&gt, 1 << 20);
  hipMemset(gt, 0x3c, 1 << 20);
  //   WAVES NSET PF ORDER BAR STG NOLDS
  run<4, 8, 8, 0, 0, 0, 1>("MFMA only (floor), 4 waves");
  run<4, 8, 8, 0, 0, 0, 0>("production slot, 4 waves, no barriers, no staging");
  run<4, 8, 8, 0, 1, 1, 0>("production slot, 4 waves, barriers + staging");
  run<8, 8, 8, 0, 1, 1, 0>("production slot, 8 waves (idle partners)");
  run<9, 8, 8, 0, 1, 1, 0>("production slot, 8 waves, partners stream a tile by LDS-DMA");
  run<4, 10, 8, 0, 0, 0, 0>("read target last used 2 MFMAs earlier (10 sets, 8 in flight), 4 waves");
  run<4, 12, 8, 0, 0, 0, 0>("read target last used 4 MFMAs earlier (12 sets, 8 in flight), 4 waves");
  run<4, 8, 6, 0, 0, 0, 0>("8 sets, 6 in flight (target last used 2 MFMAs earlier), 4 waves");
  run<4, 8, 4, 0, 0, 0, 0>("8 sets, 4 in flight (target last used 4 MFMAs earlier), 4 waves");
  run<4, 8, 7, 1, 0, 0, 0>("read BEFORE the MFMA, 8 sets, 7 ahead, 4 waves");
  run<4, 10, 8, 1, 0, 0, 0>("read BEFORE the MFMA, 10 sets, 8 ahead, 4 waves");
  run<9, 10, 8, 0, 1, 1, 0>("10 sets, 8 waves, partners stream");
  run<9, 8, 6, 0, 1, 1, 0>("8 sets, 6 in flight, 8 waves, partners stream");
  run<9, 8, 4, 0, 1, 1, 0>("8 sets, 4 in flight, 8 waves, partners stream");
  run<9, 10, 8, 1, 1, 1, 0>("read before MFMA, 10 sets, 8 waves, partners stream");
  return 0;
}
