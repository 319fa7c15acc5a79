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

// ---- the two row moves of the entity-sharded exchange (kge_amd/sharded.py) -------------------
// Up to three (table, ids, out) jobs per launch, rows addressed through an id map evaluated in the kernel (the
// torch-op form -- cat, sub, clamp, div, arange, add: six launches and ~40 us of host time per batch -- was the
// host-side bound of the sharded step):
//   own:   row = clamp(id - lo, 0, rows - 1)          a rank's rows of the batch -> its block of the all-gather
//          (ids it does not own read some local row; that block entry is never picked)
//   pick:  row = (id / shard) * stride + offset + i   the owner's copy of row i out of the gathered blocks
__global__ __launch_bounds__(256) void shard_rows_kernel(ShardJob j0, ShardJob j1, ShardJob j2, int rowbytes0,
                                                         int rowbytes2, int esize) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int cpr0 = rowbytes0 >> 4, cpr2 = rowbytes2 >> 4;
  const long long c0 = j0.n * cpr0, c1 = j1.n * cpr0;
  const ShardJob* j;
  long long tt;
  int cpr;
  if (t < c0) {
    j = &j0, tt = t, cpr = cpr0;
  } else if (t < c0 + c1) {
    j = &j1, tt = t - c0, cpr = cpr0;
  } else {
    j = &j2, tt = t - c0 - c1, cpr = cpr2;
    if (cpr == 0 || tt >= j2.n * cpr2) return;
  }
  const long long i = tt / cpr;
  const int c = (int)(tt % cpr);
  const long long g = index_at(j->idx, i);
  long long row;
  if (j->div > 0) {
    row = (g / j->div) * j->mul + j->add + i;
  } else {
    row = g - j->sub;
    row = row < 0 ? 0 : (row > j->hi ? j->hi : row);
  }
  const char* src = (const char*)j->table + row * j->ld * esize + c * 16;
  char* dst = (char*)j->out + i * j->ldo * esize + c * 16;
  *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
}

int run_shard_rows(const ShardJob& a, const ShardJob& b, const ShardJob& c, int rowbytes01, int rowbytes2, int esize,
                   hipStream_t st) {
  const long long chunks = (a.n + b.n) * (rowbytes01 >> 4) + c.n * (rowbytes2 >> 4);
  if (chunks == 0) return KGE_OK;
  hipLaunchKernelGGL(shard_rows_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, a, b, c,
                     rowbytes01, rowbytes2, esize);
  return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

}  // namespace kge
