// Is v_dot2_f32_bf16 (D = a.lo*b.lo + a.hi*b.hi + c) bit-identical to the canonical query build
// fl(fl(x0*y0) + fl(x1*y1)) for bf16 inputs?  (bf16 x bf16 products are exact in f32, so the
// question is only whether the instruction rounds the two-term sum once, like the canonical
// form.)  Random bf16 bit patterns incl. denormals, huge exponents, signs; counts mismatches.
// Build: hipcc --offload-arch=gfx950 dot2_exact.hip -o dot2_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k(const unsigned int* a, const unsigned int* b, unsigned int* bad, unsigned int* ex, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int x = a[i], y = b[i];
  float x0 = __uint_as_float(x << 16), x1 = __uint_as_float(x & 0xffff0000u);
  float y0 = __uint_as_float(y << 16), y1 = __uint_as_float(y & 0xffff0000u);
  float p0 = x0 * y0, p1 = x1 * y1;  // exact
  float canon = p0 + p1;
  float d;
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, 0\n\ts_nop 7" : "=v"(d) : "v"(x), "v"(y));
  unsigned int cb = __float_as_uint(canon), db = __float_as_uint(d);
  const bool nc = (cb & 0x7fffffffu) > 0x7f800000u, nd = (db & 0x7fffffffu) > 0x7f800000u;
  const bool same = (cb == db) || (nc && nd);
  if (!same) {
    unsigned int slot = atomicAdd(bad, 1u);
    if (slot < 8) { ex[4 * slot] = x; ex[4 * slot + 1] = y; ex[4 * slot + 2] = cb; ex[4 * slot + 3] = db; }
  }
}

int main() {
  const int n = 1 << 24;
  unsigned int *ha = (unsigned int*)malloc(4 * n), *hb = (unsigned int*)malloc(4 * n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    ha[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
    hb[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
    if (i % 3 == 0) {  // moderate magnitudes (exponent near 127): the interesting cancellation cases
      ha[i] = (ha[i] & 0x807f807fu) | 0x3f003f00u | ((rand() & 7) << 23) | ((rand() & 7) << 7);
      hb[i] = (hb[i] & 0x807f807fu) | 0x3f003f00u | ((rand() & 7) << 23) | ((rand() & 7) << 7);
    }
  }
  unsigned int *da, *db, *dbad, *dex;
  hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dbad, 4); hipMalloc(&dex, 4 * 32);
  hipMemcpy(da, ha, 4 * n, hipMemcpyHostToDevice); hipMemcpy(db, hb, 4 * n, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dbad, dex, n);
  unsigned int bad, ex[32];
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost); hipMemcpy(ex, dex, 128, hipMemcpyDeviceToHost);
  printf("v_dot2_f32_bf16 vs fl(x0*y0 + x1*y1): %u mismatches of %d\n", bad, n);
  for (unsigned i = 0; i < bad && i < 8; ++i)
    printf("  a=%08x b=%08x canonical=%08x dot2=%08x\n", ex[4 * i], ex[4 * i + 1], ex[4 * i + 2], ex[4 * i + 3]);
  return 0;
}
