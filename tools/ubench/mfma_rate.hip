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

// the same with the B operand of every MFMA freshly read from LDS (1 KiB per wave and MFMA, conflict-free
// ds_read_b128, 4 waves per CU): what the consumer waves of the scoring kernels do
template <int CH>
__global__ __launch_bounds__(256, 1) void kl(const bf16x8* g, float* out, unsigned long long* t) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 511];
  bf16x8 a = g[threadIdx.x];
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    bf16x8 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const bf16x8*>(smem + (((it * 8 + u) & 63) << 10) + lane * 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u], a, acc[u % CH], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// the B operand as TWO ds_read_b64 per lane instead of one ds_read_b128, laid out like a target tile of the
// scoring kernels (row fi at fi * 1 KiB, 16-byte slot s stored at s ^ (fi & 15)): lanes of target rows 16-31
// fetch the halves of their slot in the opposite order, so that the 32 lanes of a pass hit 32 different
// 8-byte bank pairs; a v_cndmask per dword puts the halves back in order.  (ds_read_b128 collapses when
// four waves of a CU read at once -- tools/ubench/lds_rate.hip -- ds_read_b64 scales.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 1) void kl2(const bf16x8* g, float* out, unsigned long long* t) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 511];
  bf16x8 a = g[threadIdx.x];
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, fi = lane & 31, fh = lane >> 5;
  const bool hi = (fi >> 4) & 1;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    u32x2 x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned slot = (unsigned)(((2 * u + fh) ^ (fi & 15)) << 4);
      const unsigned addr = (unsigned)(fi * 1024) + slot + (((it * 8) & 63) << 4 & 0);
      const unsigned a1 = addr + (hi ? 8u : 0u), a2 = addr + (hi ? 0u : 8u);
      asm volatile("ds_read_b64 %0, %1" : "=v"(x[u]) : "v"(a1) : "memory");
      asm volatile("ds_read_b64 %0, %1" : "=v"(y[u]) : "v"(a2) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      u32x4 f = {hi ? y[u][0] : x[u][0], hi ? y[u][1] : x[u][1], hi ? x[u][0] : y[u][0], hi ? x[u][1] : y[u][1]};
      acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), a, acc[u & 1], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int CH>
static void runl(const bf16x8* g, float* out, unsigned long long* t, int blocks) {
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((kl<CH>), dim3(blocks), dim3(256), 0, 0, g, out, t);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("32x32x16 + ds_read_b128 per MFMA, chains=%d blocks=%d: %.1f ticks per MFMA (block0)\n", CH, blocks, h[0] / 512.0);
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
  for (int blocks : {1, 256}) { runl<1>(g, out, t, blocks); runl<2>(g, out, t, blocks); runl<4>(g, out, t, blocks); }
  for (int blocks : {1, 256}) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kl2, dim3(blocks), dim3(256), 0, 0, g, out, t);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("32x32x16 + 2 x ds_read_b64 (tile layout, half order by row) + 4 v_cndmask per MFMA, blocks=%d: %.1f ticks per MFMA\n", blocks, h[0] / 512.0);
  }
  return 0;
}
