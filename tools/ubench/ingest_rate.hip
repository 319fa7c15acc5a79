// Micro-benchmark: how fast can ONE compute unit pull an L2-resident table into LDS / registers when every compute
// unit does the same (the ingest side of pairs_bf16_v6_kernel without consumers and stores)?  256 workgroups, each
// streaming its own 256 KiB slice of a 14.9 MB bf16 table (1 KiB rows), round after round, with W waves issuing:
//   MODE 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces into a 128 KiB ring), waits counted so that 32 pieces per
//           wave stay in flight
//   MODE 1: plain global_load_dwordx4 into registers (16 in flight per wave), results folded with v_or
//   MODE 2: as 1 with sc1 (served by L2, this CU's L1 bypassed)
// Prints cycles per 32 KiB unit per compute unit and B/clk per compute unit.
//   hipcc --offload-arch=gfx950 -O3 -o ingest_rate ingest_rate.hip && ./ingest_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int W>
__global__ __launch_bounds__(64 * W) void k(const unsigned char* __restrict__ tab, long long rows, unsigned long long* t,
                                            unsigned int* sink, int rounds) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[MODE == 0 ? 131072 : 16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
    // XCD-local slices, as in the scoring kernel: workgroup b runs on XCD b % 8 and reads only that XCD's eighth of
  // the table (1.9 MB: L2-resident after the first round)
  const long long per = rows / 8;
  const unsigned char* base = tab + ((long long)(b & 7) * per + ((long long)(b >> 3) * 57) % (per - 256)) * 1024;
  const unsigned int vo = (unsigned int)lane << 4;
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  constexpr int PER = 256 / W;  // rows per wave and round
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 0) {
#pragma unroll 8
      for (int i = 0; i < PER; ++i) {
        const int row = wave * PER + i;
        const unsigned char* p = base + (long long)row * 1024;
        const unsigned int d = (unsigned int)((row & 127) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(vo), "s"(p) : "memory", "m0");
        if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      }
    } else {
#pragma unroll 8
      for (int i = 0; i < PER; ++i) {
        const int row = wave * PER + i;
        const unsigned char* p = base + (long long)row * 1024;
        u32x4 v;
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(vo), "s"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(vo), "s"(p) : "memory");
        if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        acc |= v;  // (the value may be stale by up to 8 loads: only the traffic matters)
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[b] = t1 - t0;
  if (acc[0] == 0x12345678u) sink[0] = acc[1] + (MODE == 0 ? smem[lane] : 0);
}

template <int MODE, int W>
static void run(const unsigned char* tab, long long rows, unsigned long long* tp, unsigned int* sink, const char* name) {
  const int rounds = 8;
  for (int blocks : {1, 256}) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(64 * W), 0, 0, tab, rows, tp, sink, rounds);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, tp, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; ++b) avg += (double)h[b];
    avg /= blocks;
    const double bytes = 256.0 * 1024 * rounds;
    printf("%-40s waves=%d blocks=%3d  %7.0f cycles per 32 KiB  %5.1f B/clk per CU\n", name, W, blocks, avg / (bytes / 32768), bytes / avg);
  }
}

int main() {
  const long long rows = 14541;
  unsigned char* tab;
  unsigned long long* tp;
  unsigned int* sink;
  hipMalloc(&tab, rows * 1024);
  hipMemset(tab, 1, rows * 1024);
  hipMalloc(&tp, 4096 * 8);
  hipMalloc(&sink, 64);
  run<0, 1>(tab, rows, tp, sink, "LDS-DMA");
  run<0, 2>(tab, rows, tp, sink, "LDS-DMA");
  run<0, 4>(tab, rows, tp, sink, "LDS-DMA");
  run<0, 8>(tab, rows, tp, sink, "LDS-DMA");
  run<1, 1>(tab, rows, tp, sink, "global_load_dwordx4 -> VGPR");
  run<1, 2>(tab, rows, tp, sink, "global_load_dwordx4 -> VGPR");
  run<1, 4>(tab, rows, tp, sink, "global_load_dwordx4 -> VGPR");
  run<1, 8>(tab, rows, tp, sink, "global_load_dwordx4 -> VGPR");
  run<2, 2>(tab, rows, tp, sink, "global_load_dwordx4 sc1 -> VGPR");
  run<2, 4>(tab, rows, tp, sink, "global_load_dwordx4 sc1 -> VGPR");
  return 0;
}
