// Micro-benchmark for a "wide" consumer: what would a scoring workgroup of FOUR waves cost per unit if every wave held
// the query fragments of 64 rows (two 32-row blocks: 256 registers of the 512 a wave has at one wave per SIMD) and
// used every target fragment it reads from LDS for TWO MFMAs?  pairs_bf16_v6_kernel reads 4 x 32 KiB of LDS per
// 32-target unit for 128 rows (128 B/clk: the LDS is its pacemaker, 1.41-1.5 k cycles per unit = 2.8-3.0 k per
// 256 rows); here 4 x 32 KiB serve 256 rows.  The waves do their own loading (8 LDS-DMA pieces per unit each) and
// store straight from the accumulators (32 dword stores per unit each: 2 x 128 contiguous bytes per instruction).
//   ST  0: no stores  1: plain  2: sc1 (write-through)       DMA 0 / 1 / 2 (2: no wait for the pieces -- timing only)
//   BAR 0 / 1 (one s_barrier per unit)
// Prints cycles per unit (64 MFMAs per wave: floor 2048) for 1 and 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o wide_loop wide_loop.hip && ./wide_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

constexpr int NKB = 32, ROWB = 1024, UNITB = 32768, NBUF = 4, NU = 16;
constexpr long long PITCH = 16640;  // floats per score row

template <int ST, int DMA, int BAR>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ g, float* __restrict__ out, float* __restrict__ sink,
                                         unsigned long long* __restrict__ t, const unsigned char* __restrict__ gt) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * UNITB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, rg = b & 3, cg = b >> 2;
  for (int i = tid; i < NBUF * UNITB / 16; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 511];
  __syncthreads();
  const int fi = lane & 31, fh = lane >> 5;
  bf16x8 a0[NKB], a1[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    a0[kb] = g[(lane + kb * 7) & 511];
    a1[kb] = g[(lane + kb * 11 + 3) & 511];
  }
  unsigned int boff[8], bp[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bp[q] = boff[q] = (unsigned int)(fi * ROWB + (((2 * q + fh) ^ (fi & 15)) << 4));
  bf16x8 bq[8];
  auto bread = [&](bf16x8& dst, auto kc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kc)::value;
    const unsigned int addr = bp[kb & 7];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"((kb >> 3) * 256) : "memory");
  };
  // stores: element r of block bl of this lane = row 64 wave + 32 bl + 8 (r >> 2) + 4 fh + (r & 3), column lane & 31
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
  const unsigned int svo = (unsigned int)((((long long)rg * 256 + 64 * wave + 4 * fh) * PITCH + fi) * 4);
  // loads: 8 rows of 1 KiB per unit and wave, lane l fetching the 16-byte slot l ^ (row & 15)
  const unsigned int lane16 = (unsigned int)lane << 4;
  f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.0f;
  sfor<0, 8>([&](auto jc) __attribute__((always_inline)) { bread(bq[decltype(jc)::value], jc); });
  unsigned long long t0 = 0, t1 = 0;
  auto chain = [&](int u, f32x16& x0, f32x16& x1, const f32x16& p0, const f32x16& p1) __attribute__((always_inline)) {
    const unsigned int bn = (unsigned int)(((u + 1) & (NBUF - 1)) * UNITB);
    const unsigned int col_prev = (unsigned int)((cg * NU + (u - 1)) * 32 * 4);
    const unsigned char* src = gt + ((long long)(cg * NU + u + 3) * 32 + 8 * wave) * ROWB;
    const unsigned int dst = (unsigned int)(((u + 3) & (NBUF - 1)) * UNITB + 8 * wave * ROWB);
    sfor<0, NKB>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kc)::value;
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      if constexpr (BAR && kb == 14) {
        if (DMA == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ST ? 57 : 11) : "memory");  // unit u + 1 has landed
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kb == 0) {
        const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], a0[0], z, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[0], a1[0], z, 0, 0, 0);
      } else {
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % 8], a0[kb], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kb % 8], a1[kb], x1, 0, 0, 0);
      }
      if constexpr (kb == 24) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_add_u32 %0, %1, %2" : "=v"(bp[q]) : "s"(bn), "v"(boff[q]));
      }
      bread(bq[kb % 8], std::integral_constant<int, (kb + 8) % NKB>{});
      if constexpr (ST != 0) {  // one element of the previous unit per slot: block kb >> 4, element kb & 15
        constexpr int bl = kb >> 4, r = kb & 15;
        const float v = bl ? p1[r] : p0[r];
        const unsigned int so = col_prev + (unsigned int)((32 * bl + 8 * (r >> 2) + (r & 3)) * PITCH * 4);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rs, svo, so, ST == 2 ? 16 : 0);
      }
      if constexpr (DMA != 0 && (kb & 3) == 2) {
        constexpr int j = kb >> 2;  // piece j of 8
        const unsigned int dk = dst + j * ROWB;
        const unsigned char* pk = src + j * ROWB;
        const unsigned int vo = lane16 ^ (unsigned int)(((8 * wave + j) & 15) << 4);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dk), "v"(vo), "s"(pk) : "memory", "m0");
      }
    });
  };
  for (int u = 0; u < NU; u += 2) {
    if (u == 2) t0 = __builtin_readcyclecounter();
    chain(u, c00, c01, c10, c11);
    chain(u + 1, c10, c11, c00, c01);
  }
  t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
  sink[blockIdx.x * 256 + tid] = s;
  if (tid == 0) t[blockIdx.x] = t1 - t0;
}

static bf16x8* g;
static float *outp, *sinkp;
static unsigned long long* tp;
static unsigned char* gt;

template <int ST, int DMA, int BAR>
static void run(const char* name) {
  for (int blocks : {1, 256}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<ST, DMA, BAR>), dim3(blocks), dim3(256), 0, 0, g, outp, sinkp, tp, gt);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ST, DMA, BAR>), dim3(blocks), dim3(256), 0, 0, g, outp, sinkp, tp, gt);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, tp, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; ++b) avg += (double)h[b];
    avg /= blocks;
    printf("%-52s blocks=%3d  %7.1f cycles/unit (block 0), %7.1f (mean); launch %.1f us\n", name, blocks,
           h[0] / (double)(NU - 2), avg / (double)(NU - 2), ms * 1e3);
  }
}

int main() {
  hipMalloc(&g, 512 * 16);
  hipMemset(g, 0x3c, 512 * 16);
  hipMalloc(&outp, (size_t)1024 * PITCH * 4);
  hipMalloc(&sinkp, 256 * 256 * 4);
  hipMalloc(&tp, 4096 * 8);
  hipMalloc(&gt, (size_t)(64 * NU + 8) * 32 * 1024);
  hipMemset(gt, 0x3c, (size_t)(64 * NU + 8) * 32 * 1024);
  run<0, 0, 0>("two MFMAs per fragment read");
  run<0, 0, 1>("  + barrier per unit");
  run<0, 1, 1>("  + barrier + own LDS-DMA (8 pieces / unit / wave)");
  run<1, 0, 1>("  + barrier + dword stores (plain)");
  run<2, 0, 1>("  + barrier + dword stores (sc1)");
  run<1, 1, 1>("  + barrier + own LDS-DMA + stores (plain)");
  run<2, 1, 1>("  + barrier + own LDS-DMA + stores (sc1)");
  run<0, 2, 1>("  + barrier + own LDS-DMA, NO vmcnt wait");
  run<1, 2, 1>("  + barrier + own LDS-DMA + stores, NO vmcnt wait");
  run<2, 2, 1>("  + barrier + own LDS-DMA + sc1 stores, NO vmcnt wait");
  return 0;
}
