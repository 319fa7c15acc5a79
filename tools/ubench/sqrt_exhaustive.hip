// sqrt_exhaustive.hip -- is sqrt_rn_fast (kge_amd/csrc/common.hpp) the correctly rounded square root?  Every float bit
// pattern from +0 to +inf (2^31 - 2^23 + 1 values) against the IEEE sequence the compiler emits for __builtin_sqrtf,
// bit for bit, on the device; prints the number of mismatches and the first few.  Timing of both forms at the end.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -I kge_amd/csrc tools/ubench/sqrt_exhaustive.hip -o tools/ubench/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "common.hpp"

__global__ void check(unsigned int lo, unsigned long long count, unsigned long long* nbad, unsigned int* first) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const unsigned int b = lo + (unsigned int)i;
    const float x = __builtin_bit_cast(float, b);
    const float a = kge::sqrt_rn_fast(x), r = __builtin_sqrtf(x);
    if (__builtin_bit_cast(unsigned int, a) != __builtin_bit_cast(unsigned int, r)) {
      const unsigned long long k = atomicAdd(nbad, 1ull);
      if (k < 16) first[k] = b;
    }
  }
}

template <int FAST>
__global__ void rate(const float* in, float* out, int iters) {
  float x = in[threadIdx.x & 63] + (float)blockIdx.x, acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc += FAST ? kge::sqrt_rn_fast(x) : __builtin_sqrtf(x);
      x += 1.25f;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  unsigned long long* nbad;
  unsigned int* first;
  hipMalloc(&nbad, 8);
  hipMalloc(&first, 64);
  hipMemset(nbad, 0, 8);
  hipMemset(first, 0, 64);
  const unsigned long long count = 0x7f800000ull + 1ull;  // +0 ... +inf
  hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, 0u, count, nbad, first);
  // the negative half and the NaNs: the same answers as the IEEE form (NaN results compared as bit patterns)
  hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, 0x7f800001u, 0x80000000ull - 1ull, nbad, first);
  unsigned long long h = 0;
  unsigned int f[16];
  hipMemcpy(&h, nbad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(f, first, 64, hipMemcpyDeviceToHost);
  printf("sqrt_rn_fast vs __builtin_sqrtf over all 2^32 bit patterns: %llu mismatches\n", h);
  for (int i = 0; i < 16 && i < (int)h; ++i) printf("  0x%08x\n", f[i]);
  float *in, *out;
  hipMalloc(&in, 256);
  hipMalloc(&out, 1024 * 256 * 4);
  float hin[64];
  for (int i = 0; i < 64; ++i) hin[i] = 1.0f + i * 0.37f;
  hipMemcpy(in, hin, 256, hipMemcpyHostToDevice);
  for (int fast = 0; fast < 2; ++fast) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      if (fast) hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, in, out, 2000);
      else hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, in, out, 2000);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms for %.2f G square roots -> %.1f G/s\n", fast ? "sqrt_rn_fast   " : "__builtin_sqrtf",
           ms, 1024.0 * 256 * 2000 * 8 / 1e9, 1024.0 * 256 * 2000 * 8 / 1e6 / ms);
  }
  return h == 0 ? 0 : 1;
}
