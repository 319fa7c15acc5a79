// torch_ext.cpp -- the PyTorch-ROCm C++ extension over the C ABI (include/kge_amd.h): `kge_amd._C`.
//
// north_star asks for the engine "exposed through a PyTorch-ROCm C++/HIP extension that keeps the RelationalScorer /
// KgeModel plugin API".  The kernels and the drop-in boundary stay in libkge_amd.so (plain C, no torch types); this
// file is the thin torch side of it: tensors in, tensors out, outputs from torch's caching allocator, launches on
// torch's CURRENT HIP stream, C status codes turned into TORCH_CHECK failures (RuntimeError in Python).  An allocation
// failure of an output is re-raised as RuntimeError("CUDA out of memory ...") -- on ROCm torch's own text is "HIP out
// of memory", and the reference's sub-batch auto-tuner string-matches the CUDA spelling (kge/job/train.py:384-391):
// `empty_f32` below; tests/test_gpu_queries.py::test_out_of_memory_keeps_the_text_the_reference_greps_for.
// One C++ call per scoring call instead of a ctypes call with a dozen boxed arguments: ~2 us of host time instead of ~9
// (tools/host_overhead.py).
//
// Mirrors (paths in the reference tree):
//   score_spo / score_sp / score_po / score_sp_po   KgeModel.score_*      kge/model/kge_model.py:663-789
//   build_queries_group / score_queries_group       the same scores for a GROUP of batches in one persistent launch
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <vector>

#include "../../include/kge_amd.h"

namespace {

void check(int rc, const char* what) {
  TORCH_CHECK(rc == KGE_OK, "kge_amd: ", what, " failed: ", kge_status_string(rc), " (kge_status ", rc, ")");
}

// Output allocation through torch's caching allocator with the reference's OOM text (SURVEY.md 8b "Error conventions").
at::Tensor empty_f32(at::IntArrayRef shape, const at::Tensor& like) {
  try {
    return at::empty(shape, like.options().dtype(at::kFloat));
  } catch (const c10::OutOfMemoryError& e) {
    TORCH_CHECK(false, "CUDA out of memory (kge_amd, ROCm: ", e.what_without_backtrace(), ")");
  }
  return at::Tensor();  // not reached
}

void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

kge_tables tables_of(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, double l_norm, int64_t flags) {
  TORCH_CHECK(ent.is_cuda() && rel.is_cuda() && ent.get_device() == rel.get_device(),
              "kge_amd: embedding tables must live on one GPU (no CPU path)");
  TORCH_CHECK(ent.dim() == 2 && rel.dim() == 2 && ent.stride(1) == 1 && rel.stride(1) == 1, "kge_amd: tables are [rows, dim]");
  TORCH_CHECK(ent.scalar_type() == rel.scalar_type() &&
                  (ent.scalar_type() == at::kFloat || ent.scalar_type() == at::kBFloat16),
              "kge_amd: tables must be float32 or bfloat16");
  kge_tables t{};
  t.ent = ent.data_ptr();
  t.rel = rel.data_ptr();
  t.dtype = ent.scalar_type() == at::kBFloat16 ? KGE_BF16 : KGE_F32;
  t.scorer = (int32_t)scorer;
  t.num_ent = ent.size(0);
  t.num_rel = rel.size(0);
  t.dim = ent.size(1);
  t.rel_dim = rel.size(1);
  t.ent_ld = ent.stride(0);
  t.rel_ld = rel.stride(0);
  t.l_norm = (float)l_norm;
  t.flags = (int32_t)flags;
  return t;
}

// 1-D int32 / int64 index tensor of any stride (the trainers pass triples[:, k]); anything else is converted (kept alive)
kge_index index_of(const c10::optional<at::Tensor>& ix, const at::Tensor& like, std::vector<at::Tensor>& keep, int64_t* n) {
  kge_index k{nullptr, KGE_I64, 0, 1};
  if (!ix.has_value() || !ix->defined()) return k;
  at::Tensor t = *ix;
  TORCH_CHECK(t.is_cuda() && t.get_device() == like.get_device(), "kge_amd: index tensor on ", t.device(), ", tables on ",
              like.device());
  if (t.dim() != 1) t = t.reshape({-1});
  if (t.scalar_type() != at::kInt && t.scalar_type() != at::kLong) t = t.to(at::kLong);
  int64_t stride = t.numel() > 1 ? t.stride(0) : 1;
  if (stride < 1) {
    t = t.contiguous();
    stride = 1;
  }
  keep.push_back(t);
  k.ptr = t.data_ptr();
  k.itype = t.scalar_type() == at::kInt ? KGE_I32 : KGE_I64;
  k.stride = stride;
  if (n != nullptr) {
    TORCH_CHECK_VALUE(*n < 0 || *n == t.numel(), "kge_amd: index vectors of different lengths (", *n, " and ", t.numel(), ")");
    *n = t.numel();
  }
  return k;
}

at::Tensor score_spo(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, double l_norm, int64_t flags,
                     const at::Tensor& s, const at::Tensor& p, const at::Tensor& o) {
  const kge_tables t = tables_of(ent, rel, scorer, l_norm, flags);
  std::vector<at::Tensor> keep;
  int64_t n = -1;
  const kge_index si = index_of(s, ent, keep, &n), pi = index_of(p, ent, keep, &n), oi = index_of(o, ent, keep, &n);
  at::Tensor out = empty_f32({n}, ent);
  check(kge_score_spo(&t, si, pi, oi, n, out.data_ptr<float>(), stream_of(ent)), "kge_score_spo");
  return out;
}

// combine: 1 = sp_ (a = s), 2 = _po (a = o), 3 = both blocks [n, 2 m] (a = s, b = o)
at::Tensor score_pairs(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, double l_norm, int64_t flags,
                       int64_t combine, const at::Tensor& a, const at::Tensor& p, const c10::optional<at::Tensor>& b,
                       const c10::optional<at::Tensor>& targets, const c10::optional<at::Tensor>& workspace) {
  const kge_tables t = tables_of(ent, rel, scorer, l_norm, flags);
  std::vector<at::Tensor> keep;
  int64_t n = -1, m = -1;
  const kge_index ai = index_of(a, ent, keep, &n), pi = index_of(p, ent, keep, &n);
  const kge_index bi = combine == KGE_SP_PO ? index_of(b, ent, keep, &n) : kge_index{nullptr, KGE_I64, 0, 1};
  const kge_index ti = index_of(targets, ent, keep, &m);
  if (m < 0) m = t.num_ent;
  void* ws = nullptr;
  int64_t wsb = 0;
  if (workspace.has_value() && workspace->defined()) {
    TORCH_CHECK(workspace->is_cuda() && workspace->is_contiguous(), "kge_amd: workspace must be a contiguous GPU tensor");
    ws = workspace->data_ptr();
    wsb = workspace->numel() * workspace->element_size();
  }
  const int64_t width = combine == KGE_SP_PO ? 2 * m : m;
  at::Tensor out = empty_f32({n, width}, ent);
  const int64_t ldo = width > 0 ? width : 1;
  void* st = stream_of(ent);
  if (combine == KGE_SP_)
    check(kge_score_sp(&t, ai, pi, n, ti, m, out.data_ptr<float>(), ldo, ws, wsb, st), "kge_score_sp");
  else if (combine == KGE_PO_)
    check(kge_score_po(&t, pi, ai, n, ti, m, out.data_ptr<float>(), ldo, ws, wsb, st), "kge_score_po");
  else if (combine == KGE_SP_PO)
    check(kge_score_sp_po(&t, ai, pi, bi, n, ti, m, out.data_ptr<float>(), ldo, ws, wsb, st), "kge_score_sp_po");
  else
    TORCH_CHECK(false, "kge_amd: combine must be 1 (sp_), 2 (_po) or 3 (sp_po)");
  return out;
}

int64_t queries_bytes(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, int64_t flags, int64_t combine, int64_t n) {
  const kge_tables t = tables_of(ent, rel, scorer, 1.0, flags);
  return kge_queries_bytes(&t, (int)combine, n);
}

void build_queries_group(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, int64_t flags, int64_t combine,
                         const c10::optional<at::Tensor>& s, const at::Tensor& p, const c10::optional<at::Tensor>& o,
                         int64_t n, int64_t num_batches, const at::Tensor& queries, int64_t stride) {
  const kge_tables t = tables_of(ent, rel, scorer, 1.0, flags);
  std::vector<at::Tensor> keep;
  int64_t len = -1;
  const kge_index si = index_of(s, ent, keep, &len), pi = index_of(p, ent, keep, &len), oi = index_of(o, ent, keep, &len);
  TORCH_CHECK_VALUE(len == n * num_batches, "kge_amd: a group of ", num_batches, " batches of ", n, " rows needs ", n * num_batches,
              " index entries, got ", len);
  TORCH_CHECK(queries.is_cuda() && queries.is_contiguous() && queries.scalar_type() == at::kByte, "kge_amd: queries buffer");
  check(kge_build_queries_multi(&t, (int)combine, si, pi, oi, n, num_batches, queries.data_ptr(), stride, queries.numel(),
                                stream_of(ent)),
        "kge_build_queries_multi");
}

// out: [L, n, m] / [L, n, 2 m] (any row pitch, unit inner stride) or [L, n, 2, m]
// next_*: the NEXT group's index vectors (num_batches * next_n entries each) and fragment buffer -- built by the same
// launch behind its last unit (kge_next_queries); next_queries undefined: none
void score_queries_group(const at::Tensor& ent, const at::Tensor& rel, int64_t scorer, int64_t flags, int64_t combine,
                         const at::Tensor& queries, int64_t stride, int64_t n, int64_t num_batches, at::Tensor out,
                         const c10::optional<at::Tensor>& next_s, const c10::optional<at::Tensor>& next_p,
                         const c10::optional<at::Tensor>& next_o, int64_t next_n,
                         const c10::optional<at::Tensor>& next_queries, int64_t next_stride) {
  const kge_tables t = tables_of(ent, rel, scorer, 1.0, flags);
  const int64_t m = t.num_ent;
  TORCH_CHECK_VALUE(out.is_cuda() && out.get_device() == ent.get_device() && out.scalar_type() == at::kFloat,
                    "kge_amd: `out` must be a float32 tensor on the tables' GPU");
  TORCH_CHECK_VALUE(out.dim() >= 3 && out.size(0) == num_batches && out.size(1) == n, "kge_amd: `out` is [L, n, ...]");
  int64_t ldo = out.stride(1), b2 = 0;
  const int64_t width = combine == KGE_SP_PO ? 2 * m : m;
  if (out.dim() == 4) {
    TORCH_CHECK_VALUE(combine == KGE_SP_PO && out.size(2) == 2 && out.size(3) == m && (m <= 1 || out.stride(3) == 1) &&
                    out.stride(2) >= m,
                "kge_amd: a 4-D `out` is [L, n, 2, m] with unit inner stride");
    b2 = out.stride(2);
  } else {
    TORCH_CHECK_VALUE(out.dim() == 3 && out.size(2) == width && (width <= 1 || out.stride(2) == 1),
                "kge_amd: `out` must be [L, n, ", width, "] with unit inner stride");
  }
  const int64_t need = b2 > 0 ? b2 + m : width;
  if (n == 1 && ldo < need) ldo = need;
  TORCH_CHECK_VALUE(ldo >= need, "kge_amd: the rows of `out` overlap");
  kge_index all{nullptr, KGE_I64, 0, 1};
  kge_next_queries nx{};
  const kge_next_queries* nxp = nullptr;
  std::vector<at::Tensor> keep;
  if (next_queries.has_value() && next_queries->defined() && next_n > 0) {
    const at::Tensor& nq = *next_queries;
    TORCH_CHECK_VALUE(nq.is_cuda() && nq.is_contiguous() && nq.scalar_type() == at::kByte, "kge_amd: next_queries buffer");
    int64_t len = -1;
    nx.s = index_of(next_s, ent, keep, &len);
    nx.p = index_of(next_p, ent, keep, &len);
    nx.o = index_of(next_o, ent, keep, &len);
    TORCH_CHECK_VALUE(len == next_n * num_batches, "kge_amd: the next group needs ", next_n * num_batches, " index entries, got ",
                      len);
    nx.n = next_n;
    nx.queries = nq.data_ptr();
    nx.queries_bytes = nq.numel();
    nxp = &nx;
  }
  check(kge_score_queries_multi(&t, (int)combine, queries.data_ptr(), stride, n, num_batches, all, m, out.data_ptr<float>(),
                                num_batches > 1 ? out.stride(0) : 0, ldo, b2, nxp, nxp ? next_stride : 0, stream_of(ent)),
        "kge_score_queries_multi");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
  mod.doc() = "kge_amd: PyTorch-ROCm binding of libkge_amd.so (include/kge_amd.h)";
  mod.def("abi_version", []() { return kge_abi_version(); });
  mod.def("score_spo", &score_spo, "KgeModel.score_spo: [n] scores");
  mod.def("score_pairs", &score_pairs, "KgeModel.score_sp / score_po / score_sp_po: [n, m] / [n, 2 m] scores");
  mod.def("queries_bytes", &queries_bytes);
  mod.def("build_queries_group", &build_queries_group);
  mod.def("score_queries_group", &score_queries_group);
}
