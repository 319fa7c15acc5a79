// Micro-benchmark of the consumer loop of pairs_bf16_v6_kernel: per 32-target unit 32 x {s_waitcnt lgkmcnt,
// v_mfma_f32_32x32x16_bf16, ds_read_b128} against units resident in LDS, the read stream continuous across units,
// four staging writes per unit.  What does the per-unit synchronisation cost the matrix pipe?
//   SYNC 0: none;  1: one s_barrier per unit (slot 14), partner waves only keep the barrier;
//        2: LDS counters instead -- the consumer publishes "staged" / "done" with ds_write_b32 and looks at a
//           "landed" / "drained" counter with a ds_read issued ten slots ahead of its use (never blocks here: the
//           partner waves publish far ahead); partner waves poll the consumers' counters with s_sleep between polls
//   DMA  1: the partner waves 4, 5 also stream 32 KiB per unit into a third LDS buffer with LDS-DMA
// Prints cycles per MFMA (wave 0 of block 0) for 1 and 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o unit_loop unit_loop.hip && ./unit_loop
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

constexpr int NKB = 32, ROWB = 1024, UNITB = 32768, NU = 24;
constexpr int STG0 = 3 * UNITB, FLG = STG0 + 32768;

template <int SYNC, int STG, int DMA>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ g, float* __restrict__ out,
                                            unsigned long long* __restrict__ t, const unsigned char* __restrict__ gt,
                                            float* __restrict__ scores) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FLG + 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * UNITB / 16; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 511];
  if (tid < 64) reinterpret_cast<unsigned int*>(smem + FLG)[tid] = 0u;
  __syncthreads();
  volatile unsigned int* flg = reinterpret_cast<volatile unsigned int*>(smem + FLG);
  if (wave >= 4) {
    // partners: [0..3] staged counters of the consumers, [4..7] done counters, [8] landed, [9] drained
    const unsigned int voff = (unsigned int)(lane * 16);
    for (int u = 0; u < NU; ++u) {
      if (SYNC == 1) __builtin_amdgcn_s_barrier();
      if (SYNC == 2 && wave == 4 && lane == 0) flg[8] = u + 2;  // "unit u + 1 landed" (far ahead of need)
      if (SYNC == 2 && wave == 6 && lane == 0) flg[9] = u + 1;
      if (DMA == 1 && wave < 6) {
        unsigned int d = (unsigned int)(2 * UNITB + (wave - 4) * 16384);
        const unsigned char* p = gt + (size_t)(wave - 4) * 16384;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(voff), "s"(p) : "memory", "m0");
          d += 1024;
          p += 1024;
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      if (DMA == 2) {  // all four partner waves load: 8 pieces per unit each (a fresh 32 KiB of the table per unit)
        unsigned int d = (unsigned int)(2 * UNITB + (wave - 4) * 8192);
        const unsigned char* p = gt + ((size_t)((blockIdx.x >> 2) * NU + u) & 255) * 32768 + (size_t)(wave - 4) * 8192;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(voff), "s"(p) : "memory", "m0");
          d += 1024;
          p += 1024;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      if (SYNC == 2 && wave >= 6) {
        // a store wave waits until its two consumers have staged unit u
        const int w0 = 2 * (wave & 1);
        while (true) {
          const unsigned int a = flg[w0], b = flg[w0 + 1];
          if (a > (unsigned)u && b > (unsigned)u) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int fi = lane & 31, fh = lane >> 5, w4 = wave;
  bf16x8 afr[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) afr[kb] = g[(lane + kb * 7) & 511];
  unsigned int boff[8], bp[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bp[q] = boff[q] = (unsigned int)(fi * ROWB + (((2 * q + fh) ^ (fi & 15)) << 4));
  const unsigned int cwr = (unsigned int)(STG0 + w4 * 4096 + fi * 128);
  const int y = fh ^ (fi & 7);
  bf16x8 bq[8];
  auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256) : "memory");
  };
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)scores, 0, 0x7fffffff, 0x00020000);
  const unsigned int svo = (unsigned int)((((long long)(blockIdx.x & 3) * 128 + 32 * w4 + 4 * fh) * 16640 + fi) * 4);
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
  sfor<0, 8>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
  unsigned long long t0 = 0, t1 = 0;
  unsigned int seen = 0;
  auto chain = [&](int u, f32x16& acc, const f32x16& prev) __attribute__((always_inline)) {
    const unsigned int bn = (unsigned int)(((u + 1) & 1) * UNITB);
    unsigned int landed = 0, drained = 0;
    sfor<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      constexpr int w_lo = 3;
      constexpr auto younger = [](int k) constexpr {
        int yg = 7;
        if (STG == 1)
          for (int j = w_lo; j < w_lo + 4; ++j)
            if (j >= k - 8 && j <= k - 1) ++yg;
        if (SYNC == 2) {
          const int extras[4] = {-1, 7, 12, -6};
          for (int e = 0; e < 4; ++e)
            if (extras[e] >= k - 8 && extras[e] <= k - 1) ++yg;
        }
        return yg > 15 ? 15 : yg;
      };
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(younger(kb)) : "memory");
      if constexpr (SYNC == 1 && kb == 14) __builtin_amdgcn_s_barrier();
      if constexpr (SYNC == 2 && kb == 3) seen += drained;   // (the real kernel compares and, rarely, spins)
      if constexpr (SYNC == 2 && kb == 22) seen += landed;
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % 8], afr[kb], acc, 0, 0, 0);
      if constexpr (kb == 24) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_add_u32 %0, %1, %2" : "=v"(bp[q]) : "s"(bn), "v"(boff[q]));
      }
      bread(bq[kb % 8], std::integral_constant<int, (kb + 8) % NKB>{});
      if constexpr (STG == 2 && (kb & 1) == 0) {  // direct dword stores of the previous unit: element kb / 2
        constexpr int r = kb >> 1;
        const unsigned int so = (unsigned int)((((blockIdx.x >> 2) * NU + u) & 127) * 128) +
                                (unsigned int)((8 * (r >> 2) + (r & 3)) * 16640 * 4);
        const float pv = prev[r];  // (bit_cast straight on the vector element stores element 0: clang quirk)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, pv), srs, svo, so, 0);
      }
      if constexpr (STG == 1 && kb >= w_lo && kb < w_lo + 4) {
        constexpr int gq = kb - w_lo;
        f32x4 v = {prev[4 * gq], prev[4 * gq + 1], prev[4 * gq + 2], prev[4 * gq + 3]};
        *reinterpret_cast<f32x4*>(smem + cwr + (u & 1) * 16384 + (((2 * gq) ^ y) << 4)) = v;
      }
      if constexpr (SYNC == 2 && kb == 7) asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(FLG + 4 * w4)), "v"((unsigned)(u + 1)) : "memory");
      if constexpr (SYNC == 2 && kb == 12) asm volatile("ds_read_b32 %0, %1" : "=v"(landed) : "v"((unsigned)(FLG + 32)) : "memory");
      if constexpr (SYNC == 2 && kb == 26) asm volatile("ds_read_b32 %0, %1" : "=v"(drained) : "v"((unsigned)(FLG + 36)) : "memory");
      if constexpr (SYNC == 2 && kb == 31) asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(FLG + 16 + 4 * w4)), "v"((unsigned)(u + 1)) : "memory");
    });
  };
  for (int u = 0; u < NU; u += 2) {
    if (u == 2) t0 = __builtin_readcyclecounter();
    chain(u, acc0, acc1);
    chain(u + 1, acc1, acc0);
  }
  t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = (float)seen;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) t[blockIdx.x] = t1 - t0;
}

static bf16x8* g;
static float* outp;
static unsigned long long* tp;
static unsigned char* gt;
static float* scores;

template <int SYNC, int STG, int DMA>
static void run(const char* name) {
  for (int blocks : {1, 256}) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SYNC, STG, DMA>), dim3(blocks), dim3(512), 0, 0, g, outp, tp, gt, scores);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, tp, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; ++b) avg += (double)h[b];
    avg /= blocks;
    printf("%-64s blocks=%3d  %6.1f cycles/MFMA (block 0), %6.1f (mean)\n", name, blocks, h[0] / (double)((NU - 2) * NKB),
           avg / (double)((NU - 2) * NKB));
  }
}

int main() {
  hipMalloc(&g, 512 * 16);
  hipMemset(g, 0x3c, 512 * 16);
  hipMalloc(&outp, 256 * 512 * 4);
  hipMalloc(&tp, 4096 * 8);
  hipMalloc(&gt, 8 << 20);
  hipMemset(gt, 0x3c, 8 << 20);
  hipMalloc(&scores, (size_t)520 * 16640 * 4);
  run<0, 0, 0>("no sync, no staging");
  run<0, 1, 0>("no sync, staging writes");
  run<1, 1, 0>("s_barrier per unit, staging");
  run<2, 1, 0>("LDS counters, staging");
  run<0, 1, 1>("no sync, staging, partners stream 32 KiB per unit");
  run<1, 1, 1>("s_barrier per unit, staging, partners stream");
  run<2, 1, 1>("LDS counters, staging, partners stream");
  run<1, 2, 0>("s_barrier per unit, DIRECT dword stores (16 / unit / wave)");
  run<1, 2, 1>("s_barrier, direct stores, two partners stream");
  run<1, 2, 2>("s_barrier, direct stores, FOUR partners stream fresh units");
  run<1, 1, 2>("s_barrier, staging (no store waves), four partners stream");
  return 0;
}
