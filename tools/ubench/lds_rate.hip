// LDS read throughput per CU by instruction width and wave count (conflict-free, contiguous per wave):
//   hipcc --offload-arch=gfx950 -O2 -o lds_rate lds_rate.hip && ./lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(512, 1) void k(unsigned int* out, unsigned long long* t, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned int*>(smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned base = ((it * 8 + wave) & 31) << 11;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (W == 16) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base + lane * 16), "i"(0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc += v[0];
      } else if (W == 8) {
        u32x2 v0, v1;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(base + lane * 8) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(v1) : "v"(base + lane * 8) : "memory");
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc += v0[0] + v1[0];
      } else {
        unsigned v0, v1, v2, v3;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(base + lane * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(v1) : "v"(base + lane * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(v2) : "v"(base + lane * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(v3) : "v"(base + lane * 4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc += v0 + v1 + v2 + v3;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int W>
static void run(int waves, unsigned int* out, unsigned long long* t) {
  const int iters = 256;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<W>), dim3(8), dim3(64 * waves), 0, 0, out, t, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes = (double)iters * 8 * 1024 * waves;  // 1 KiB per wave and step
  printf("ds_read_b%-3d %d waves per CU: %.1f B/clk per CU (%.1f cycles per KiB and wave)\n", W * 8, waves,
         bytes / h[0], (double)h[0] / (iters * 8));
}

int main() {
  unsigned int* out;
  unsigned long long* t;
  hipMalloc(&out, 8 * 512 * 4);
  hipMalloc(&t, 64);
  for (int waves : {1, 2, 4, 8}) {
    run<16>(waves, out, t);
    run<8>(waves, out, t);
    run<4>(waves, out, t);
  }
  return 0;
}
