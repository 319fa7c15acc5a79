// Micro-benchmark: the write stream of the scoring kernel WITHOUT the scoring.  256 workgroups write an [n, m] f32
// block (row pitch ld) the way pairs_bf16_v6_kernel's store waves do -- workgroup (rg, cg) owns 128 rows x a
// contiguous column range and walks it in chunks -- with different shapes of one store instruction:
//   SEG = 128 B: 8 rows x 128 B per instruction (v6, 32-column units)
//   SEG = 256 B: 4 rows x 256 B (v4, 64-column tiles)
//   SEG = 512 B / 1024 B: 2 rows x 512 B, 1 row x 1 KiB
// WAVES store waves per workgroup, each issuing its instructions back to back.  Prints us and TB/s per variant:
// what the memory system takes from this geometry when nothing else limits it.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int WAVES, int SC1>
__global__ __launch_bounds__(64 * WAVES) void k(float* __restrict__ out, long long ld, int n, int m, int rgn, int ncg) {
  const int b = blockIdx.x, q8 = b >> 3;
  const int rg = q8 % rgn, cg = (q8 / rgn) * 8 + (b & 7);
  if (cg >= ncg) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPI = 64 / LPR;        // rows per instruction
  constexpr int COLS = SEG / 4;        // columns per chunk
  const int cpc = (m / COLS + ncg - 1) / ncg;  // chunks per column group
  const int c_lo = cg * cpc;
  const int lr = lane / LPR, lc = lane % LPR;
  const f32x4 v = {1.0f, 2.0f, 3.0f, (float)b};
  for (int c = c_lo; c < c_lo + cpc && (c + 1) * COLS <= m; ++c) {
    // the workgroup's 128 rows of this chunk: 128 / RPI instructions, dealt to the waves
    for (int i = wave; i < 128 / RPI; i += WAVES) {
      const long long row = (long long)rg * 128 + i * RPI + lr;
      if (row < n) {
        float* p = out + row * ld + (long long)c * COLS + lc * 4;
        if (SC1) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), rs, 0, 0, 16);
        } else {
          *reinterpret_cast<f32x4*>(p) = v;
        }
      }
    }
  }
}

template <int SEG, int WAVES, int SC1>
static void run(float* out, long long ld, int n, int m, const char* name) {
  const int rgn = (n + 127) / 128;
  int ncg = 8 * (256 / 8 / rgn > 0 ? 256 / 8 / rgn : 1);
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<SEG, WAVES, SC1>), dim3(grid), dim3(64 * WAVES), 0, 0, out, ld, n, m, rgn, ncg);
  hipEventRecord(e0, 0);
  const int reps = 50;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SEG, WAVES, SC1>), dim3(grid), dim3(64 * WAVES), 0, 0, out, ld, n, m, rgn, ncg);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes = (double)n * (m / (SEG / 4) * (SEG / 4)) * 4;
  printf("n=%5d ld=%6lld %-44s %8.2f us  %5.2f TB/s\n", n, ld, name, us, bytes / us * 1e-6);
}

int main() {
  float* out;
  const int m = 14541;
  hipMalloc(&out, 4096ull * 16384 * 4);
  for (int n : {512, 1024, 2048, 4096}) {
    for (long long ld : {14592ll, 14656ll}) {
      run<128, 2, 1>(out, ld, n, m, "8 rows x 128 B, 2 waves, write-through");
      run<128, 2, 0>(out, ld, n, m, "8 rows x 128 B, 2 waves, plain");
      run<128, 4, 1>(out, ld, n, m, "8 rows x 128 B, 4 waves, write-through");
      run<128, 8, 1>(out, ld, n, m, "8 rows x 128 B, 8 waves, write-through");
      run<256, 2, 1>(out, ld, n, m, "4 rows x 256 B, 2 waves, write-through");
      run<256, 4, 1>(out, ld, n, m, "4 rows x 256 B, 4 waves, write-through");
      run<512, 2, 1>(out, ld, n, m, "2 rows x 512 B, 2 waves, write-through");
      run<1024, 2, 1>(out, ld, n, m, "1 row x 1 KiB, 2 waves, write-through");
      run<1024, 4, 1>(out, ld, n, m, "1 row x 1 KiB, 4 waves, write-through");
      run<1024, 4, 0>(out, ld, n, m, "1 row x 1 KiB, 4 waves, plain");
    }
  }
  return 0;
}
