// xcd_handoff.hip -- how fast can one workgroup hand a 16 KiB block to another one, and does the
// hand-off need memory-side (agent-scope write-through) traffic when both sit on the same XCD?
//
//   hipcc --offload-arch=gfx950 -O2 -o xcd_handoff xcd_handoff.hip && ./xcd_handoff
//
// 256 workgroups x 64 threads, one per CU.  Workgroup b reads its XCC id (hardware register) and
// is paired with workgroup b + DIST: DIST = 8 -> the same XCD under round-robin dispatch, DIST = 1 ->
// the neighbouring XCD.  Ping-pong, ROUNDS times: the producer writes 16 KiB whose content depends
// on the round, publishes a flag; the consumer polls the flag, loads the 16 KiB, checks every
// dword, publishes its own flag back.  Reported per variant: round-trip cycles / 2 (one direction)
// and the number of stale dwords seen.
//
// variants (store policy of the payload / load policy of the payload):
//   0  sc1 stores (agent-scope write-through) + wait for the acknowledgement; sc1 loads   [what
//      pairs_bf16_v4_kernel does today]
//   1  plain stores (stay dirty in the producer's L2) + wait; sc1 loads (bypass the consumer CU's L1)
//   2  plain stores + wait; sc0 sc1 loads
//   3  plain stores + wait; plain loads after a buffer_inv sc0 (L1 invalidate)
//   4  plain stores + wait; plain loads (no invalidate: expected stale through L1)
// flags: always agent-scope atomics (relaxed), one 64-byte line per direction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                             \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int PAYLOAD_VEC = 1024;  // 16-byte vectors per block (16 KiB), 16 per lane
constexpr int ROUNDS = 64;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

template <int VARIANT>
__device__ __forceinline__ void put(u32x4* dst, u32x4 v) {
  if (VARIANT == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
}

template <int VARIANT>
__device__ __forceinline__ u32x4 get(const u32x4* src) {
  u32x4 v;
  if (VARIANT == 0 || VARIANT == 1)
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  else if (VARIANT == 2)
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  return v;
}

__device__ __forceinline__ void wait_flag(unsigned long long* f, unsigned long long want, int* timeouts) {
  for (int spin = 0;; ++spin) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return;
    if (spin > (1 << 20)) {
      if (threadIdx.x == 0) atomicAdd(timeouts, 1);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// buf: per pair 2 blocks of PAYLOAD_VEC vectors (a -> b, b -> a); flags: per pair 2 x 8 u64 (64-B lines)
template <int VARIANT>
__global__ __launch_bounds__(64) void pingpong(u32x4* buf, unsigned long long* flags, int dist,
                                               unsigned long long epoch0, long long* cycles, int* stale,
                                               int* timeouts, int* xcc_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned xcc = xcc_id();
  if (lane == 0) xcc_out[b] = (int)xcc;
  // pairs: within groups of 2*dist consecutive blocks, block i < dist is the producer of pair, i + dist its partner
  const int grp = b / (2 * dist), pos = b % (2 * dist);
  const bool first = pos < dist;
  const int pair = grp * dist + (first ? pos : pos - dist);
  u32x4* mine = buf + ((long long)pair * 2 + (first ? 0 : 1)) * PAYLOAD_VEC;    // I write here
  u32x4* theirs = buf + ((long long)pair * 2 + (first ? 1 : 0)) * PAYLOAD_VEC;  // I read here
  unsigned long long* fmine = flags + ((long long)pair * 2 + (first ? 0 : 1)) * 8;
  unsigned long long* ftheirs = flags + ((long long)pair * 2 + (first ? 1 : 0)) * 8;
  int bad = 0;
  long long t0 = 0;
  for (int r = 0; r < ROUNDS; ++r) {
    const unsigned long long ep = epoch0 + r + 1;
    if (r == 8 && first) t0 = (long long)__builtin_readcyclecounter();  // 8 warm-up rounds
    if (!first) {  // partner: wait for the block, check it
      wait_flag(ftheirs, ep, timeouts);
      if (VARIANT == 3) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll 4
      for (int k = 0; k < PAYLOAD_VEC / 64; ++k) {
        const u32x4 v = get<VARIANT>(theirs + k * 64 + lane);
        const unsigned want = (unsigned)ep * 2654435761u + (unsigned)(k * 64 + lane);
        bad += (v[0] != want) + (v[1] != (want ^ 1u)) + (v[2] != (want ^ 2u)) + (v[3] != (want ^ 3u));
      }
    }
    // write my block for this round, wait until the stores are acknowledged, publish
#pragma unroll 4
    for (int k = 0; k < PAYLOAD_VEC / 64; ++k) {
      const unsigned w = (unsigned)ep * 2654435761u + (unsigned)(k * 64 + lane);
      const u32x4 v = {w, w ^ 1u, w ^ 2u, w ^ 3u};
      put<VARIANT>(mine + k * 64 + lane, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(fmine, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first) {  // producer: wait for the partner's block of this round, check it
      wait_flag(ftheirs, ep, timeouts);
      if (VARIANT == 3) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll 4
      for (int k = 0; k < PAYLOAD_VEC / 64; ++k) {
        const u32x4 v = get<VARIANT>(theirs + k * 64 + lane);
        const unsigned want = (unsigned)ep * 2654435761u + (unsigned)(k * 64 + lane);
        bad += (v[0] != want) + (v[1] != (want ^ 1u)) + (v[2] != (want ^ 2u)) + (v[3] != (want ^ 3u));
      }
    }
  }
  if (first && lane == 0) cycles[pair] = (long long)__builtin_readcyclecounter() - t0;
  if (bad) atomicAdd(stale, bad);
}

template <int V>
static void run(const char* what, int dist, u32x4* buf, unsigned long long* flags, long long* cycles, int* stale,
                int* timeouts, int* xcc, unsigned long long& epoch) {
  const int nb = 256;
  CHECK(hipMemset(stale, 0, sizeof(int)));
  CHECK(hipMemset(timeouts, 0, sizeof(int)));
  CHECK(hipMemset(cycles, 0, 128 * sizeof(long long)));
  hipLaunchKernelGGL(pingpong<V>, dim3(nb), dim3(64), 0, 0, buf, flags, dist, epoch, cycles, stale, timeouts, xcc);
  CHECK(hipDeviceSynchronize());
  epoch += ROUNDS + 16;
  std::vector<long long> c(128);
  std::vector<int> x(nb);
  int st, to;
  CHECK(hipMemcpy(c.data(), cycles, 128 * sizeof(long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x.data(), xcc, nb * sizeof(int), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&st, stale, sizeof(int), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&to, timeouts, sizeof(int), hipMemcpyDeviceToHost));
  double sum = 0;
  long long mx = 0;
  for (int i = 0; i < 128; ++i) {
    sum += (double)c[i];
    if (c[i] > mx) mx = c[i];
  }
  int same = 0, rr = 0;
  for (int b = 0; b < nb; ++b) rr += x[b] == (b & 7);
  for (int p = 0; p < 128; ++p) {
    const int grp = p / dist, pos = p % dist, a = grp * 2 * dist + pos;
    same += x[a] == x[a + dist];
  }
  printf("variant %d dist %d  %-58s one-way %7.0f cycles (max %7.0f)  stale dwords %d  timeouts %d  pairs on one XCD %d/128  "
         "blocks with xcc == b%%8: %d/256\n",
         V, dist, what, sum / 128 / (ROUNDS - 8) / 2, (double)mx / (ROUNDS - 8) / 2, st, to, same, rr);
}

int main() {
  u32x4* buf;
  unsigned long long* flags;
  long long* cycles;
  int *stale, *timeouts, *xcc;
  CHECK(hipMalloc(&buf, 128LL * 2 * PAYLOAD_VEC * 16));
  CHECK(hipMalloc(&flags, 128 * 2 * 64));
  CHECK(hipMalloc(&cycles, 128 * sizeof(long long)));
  CHECK(hipMalloc(&stale, sizeof(int)));
  CHECK(hipMalloc(&timeouts, sizeof(int)));
  CHECK(hipMalloc(&xcc, 256 * sizeof(int)));
  CHECK(hipMemset(buf, 0, 128LL * 2 * PAYLOAD_VEC * 16));
  CHECK(hipMemset(flags, 0, 128 * 2 * 64));
  unsigned long long epoch = 1000;
  for (int rep = 0; rep < 2; ++rep)
    for (int dist : {8, 1}) {
      run<0>("sc1 stores + ack, sc1 loads (today)", dist, buf, flags, cycles, stale, timeouts, xcc, epoch);
      run<1>("plain stores + ack, sc1 loads", dist, buf, flags, cycles, stale, timeouts, xcc, epoch);
      run<2>("plain stores + ack, sc0 sc1 loads", dist, buf, flags, cycles, stale, timeouts, xcc, epoch);
      run<3>("plain stores + ack, buffer_inv sc0 + plain loads", dist, buf, flags, cycles, stale, timeouts, xcc, epoch);
      run<4>("plain stores + ack, plain loads (no invalidate)", dist, buf, flags, cycles, stale, timeouts, xcc, epoch);
    }
  return 0;
}
