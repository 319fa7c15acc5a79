// Micro-benchmark (round 4): what the chip takes from the WRITE stream of the direct-store scoring kernel
// (pairs_bf16_v7_kernel) by width of one store instruction, cache policy and -- the question of this round -- by
// whether the score block lives in the Infinity Cache (the same 30-60 MB buffer rewritten launch after launch, what
// bench.py's single-batch launches do) or streams to HBM (R rotating buffers, 1 GiB in all).
//
// Geometry of the kernel: 256 workgroups of 4 storing waves; a workgroup owns 128 rows x a contiguous column range
// and walks it in 32-column units; wave w owns rows 32 w .. 32 w + 31 of the unit:
//   W = 1  dword stores: one instruction = 2 rows x 128 B  (v7: element r of all lanes)         16 per unit and wave
//   W = 2  dwordx2:      one instruction = 2 rows x 256 B  (64-column units)                      16 per 64 columns
//   W = 4  dwordx4:      one instruction = 2 rows x 512 B  (128-column units)
//   W = 5  dwordx4:      one instruction = 8 rows x 128 B  (v6's staged pattern)
// All through ONE buffer descriptor per wave (uniform base, per-lane offset), aux = 0 (plain) or 16 (sc1).
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int W, int SC1>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, long long ld, int n, int m, int rgn, int ncg) {
  const int b = blockIdx.x, q8 = b >> 3;
  const int rg = q8 % rgn, cg = (q8 / rgn) * 8 + (b & 7);
  if (cg >= ncg) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int COLS = W == 5 ? 32 : 32 * W;  // columns per unit
  const int nunits = m / COLS;
  const int upc = (nunits + ncg - 1) / ncg;
  const int u_lo = cg * upc;
  const long long rb = (long long)rg * 128 + 32 * wave;
  if (rb >= n) return;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(out + rb * ld), 0, (int)(32 * ld * 4), 0x00020000);
  const unsigned int ld4 = (unsigned int)(ld * 4);
  const int fi = lane & 31, fh = lane >> 5;
  for (int u = u_lo; u < u_lo + upc && u < nunits; ++u) {
    const unsigned int colb = (unsigned int)(u * COLS * 4);
    if constexpr (W == 5) {
      const int rq = lane >> 3, cl = lane & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 v = {1u, 2u, (unsigned)u, (unsigned)i};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned int)(8 * i + rq) * ld4 + cl * 16, colb, SC1 ? 16 : 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned int vo = (unsigned int)(8 * (r >> 2) + 4 * fh + (r & 3)) * ld4 + fi * 4 * W;
        if constexpr (W == 1) __builtin_amdgcn_raw_buffer_store_b32((unsigned)r, rs, vo, colb, SC1 ? 16 : 0);
        if constexpr (W == 2) {
          const u32x2 v = {(unsigned)r, (unsigned)u};
          __builtin_amdgcn_raw_buffer_store_b64(v, rs, vo, colb, SC1 ? 16 : 0);
        }
        if constexpr (W == 4) {
          const u32x4 v = {(unsigned)r, (unsigned)u, 3u, 4u};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, vo, colb, SC1 ? 16 : 0);
        }
      }
    }
  }
}

template <int W, int SC1>
static void run(float* out, long long ld, int n, int m, int nbuf, const char* name) {
  const int rgn = (n + 127) / 128;
  int ncg = 8 * (256 / 8 / rgn > 0 ? 256 / 8 / rgn : 1);
  const int grid = 8 * rgn * ((ncg + 7) / 8);
  const long long bufs = (long long)n * ld;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 64;
  for (int w = 0; w < 8; ++w)
    hipLaunchKernelGGL((k<W, SC1>), dim3(grid), dim3(256), 0, 0, out + (w % nbuf) * bufs, ld, n, m, rgn, ncg);
  hipEventRecord(e0, 0);
  for (int w = 0; w < reps; ++w)
    hipLaunchKernelGGL((k<W, SC1>), dim3(grid), dim3(256), 0, 0, out + (w % nbuf) * bufs, ld, n, m, rgn, ncg);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int cols = W == 5 ? 32 : 32 * W;
  const double us = ms * 1e3 / reps, bytes = (double)n * (m / cols * cols) * 4;
  printf("n=%5d bufs=%2d %-34s %-5s %8.2f us  %5.2f TB/s\n", n, nbuf, name, SC1 ? "sc1" : "plain", us, bytes / us * 1e-6);
}

int main() {
  float* out;
  const int m = 14541;
  const long long ld = 14656;
  const size_t total = 2ull << 30;
  if (hipMalloc(&out, total) != hipSuccess) return 1;
  hipMemset(out, 0, total);
  for (int n : {512, 1024, 4096}) {
    const long long per = (long long)n * ld * 4;
    const int rot = (int)((total / per) > 32 ? 32 : (total / per));
    for (int nbuf : {1, rot}) {
      run<1, 0>(out, ld, n, m, nbuf, "dword    2 rows x 128 B");
      run<1, 1>(out, ld, n, m, nbuf, "dword    2 rows x 128 B");
      run<2, 0>(out, ld, n, m, nbuf, "dwordx2  2 rows x 256 B");
      run<2, 1>(out, ld, n, m, nbuf, "dwordx2  2 rows x 256 B");
      run<4, 0>(out, ld, n, m, nbuf, "dwordx4  2 rows x 512 B");
      run<4, 1>(out, ld, n, m, nbuf, "dwordx4  2 rows x 512 B");
      run<5, 0>(out, ld, n, m, nbuf, "dwordx4  8 rows x 128 B");
      run<5, 1>(out, ld, n, m, nbuf, "dwordx4  8 rows x 128 B");
    }
  }
  return 0;
}
