// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 / 16x16x32 with 1, 2, 4 independent
// accumulator chains, one wave per SIMD, timed with s_memtime.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, int BIG>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* g, float* out, unsigned long long* t) {
  bf16x8 a = g[threadIdx.x], b = g[threadIdx.x + 256];
  f32x16 acc[4];
  f32x4 acs[4];
  for (int c = 0; c < 4; ++c) { for (int r = 0; r < 16; ++r) acc[c][r] = 0; for (int r = 0; r < 4; ++r) acs[c][r] = 0; }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (BIG) acc[u % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % CH], 0, 0, 0);
      else acs[u % CH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acs[u % CH], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < 4; ++c) { for (int r = 0; r < 16; ++r) s += acc[c][r]; for (int r = 0; r < 4; ++r) s += acs[c][r]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int CH, int BIG>
static void run(const char* name, const bf16x8* g, float* out, unsigned long long* t, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<CH, BIG>), dim3(blocks), dim3(256), 0, 0, g, out, t);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CH, BIG>), dim3(blocks), dim3(256), 0, 0, g, out, t);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s chains=%d blocks=%d: %.1f ticks per MFMA (block0), kernel %.1f us -> %.1f ticks/us\n", name, CH, blocks,
         h[0] / 512.0, ms * 1e3, h[0] / (ms * 1e3));
}

int main() {
  bf16x8* g; float* out; unsigned long long* t;
  hipMalloc(&g, 512 * 16); hipMemset(g, 0x3c, 512 * 16); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&t, 4096 * 8);
  for (int blocks : {1, 256, 1024}) {
    run<1, 1>("32x32x16", g, out, t, blocks); run<2, 1>("32x32x16", g, out, t, blocks); run<4, 1>("32x32x16", g, out, t, blocks);
    run<1, 0>("16x16x32", g, out, t, blocks); run<2, 0>("16x16x32", g, out, t, blocks); run<4, 0>("16x16x32", g, out, t, blocks);
  }
  return 0;
}
