// embed.hip -- LookupEmbedder.embed (kge/model/embedder/lookup_embedder.py:96-105) as a stand-alone
// row gather: out[i, :] = table[idx[i], :].  The scoring kernels gather on the fly and never need
// it; the sharded path does (the query rows a rank owns are gathered, then exchanged with ONE
// all-gather): two (table, index, out) jobs per launch, so the entity rows and the relation rows
// of a batch are one kernel.  HBM gather bound: 16-byte loads / stores, one wave per row slice.
#include "common.hpp"

namespace kge {

// rowbytes: bytes per row (multiple of 16); one thread per 16-byte chunk
__global__ __launch_bounds__(256) void embed_kernel(EmbedJob j0, EmbedJob j1, int rowbytes, int esize) {
  const int cpr = rowbytes >> 4;  // chunks per row
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n0c = j0.n * cpr;
  const bool second = t >= n0c;
  const EmbedJob& j = second ? j1 : j0;
  const long long tt = second ? t - n0c : t;
  const long long i = tt / cpr;
  const int c = (int)(tt % cpr);
  if (i >= j.n) return;
  const char* src = (const char*)j.table + index_at(j.idx, i) * j.ld * esize + c * 16;
  char* dst = (char*)j.out + i * j.ldo * esize + c * 16;
  *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
}

int run_embed2(const EmbedJob& a, const EmbedJob& b, int rowbytes, int esize, hipStream_t st) {
  const long long chunks = (a.n + b.n) * (rowbytes >> 4);
  if (chunks == 0) return KGE_OK;
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, a, b, rowbytes,
                     esize);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
