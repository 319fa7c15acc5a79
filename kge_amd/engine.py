"""Tensor-level front end of the gfx950 scoring kernels (C ABI in include/kge_amd.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and
dtype/stride bookkeeping.  Every function launches hand-written HIP kernels from
libkge_amd.so; there is no torch/CPU fallback -- tensors that are not on a GPU raise.

Function <-> reference map (paths relative to the reference tree):
  score_spo / score_sp / score_po / score_sp_po   KgeModel.score_*     kge/model/kge_model.py:663-789
  score_emb                                        RelationalScorer.score_emb  kge_model.py:151-213
  score_neg                                        BatchNegativeSample.score   kge/util/sampler.py:263-306
  rank_counts                                      EntityRankingJob._filter_and_rank  kge/job/eval_entity_ranking.py:533-596
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (BF16, F32, FLAG_BF16_V3, FLAG_EXACT, FLAG_NO_MFMA, FLAG_SPLIT_QUERY, I32, I64, PO_,
                   SCORERS, SP_, SP_PO, SPO, KgeIndex, KgeNextQueries, KgeTables)

__all__ = ["Tables", "score_spo", "score_sp", "score_po", "score_sp_po", "score_neg",
           "score_emb", "embed", "shard_gather", "shard_pick", "ns_bce_loss", "rank_counts", "score_pitch", "eval_batch", "FLAG_EXACT", "FLAG_NO_MFMA", "FLAG_BF16_V3",
           "FLAG_SPLIT_QUERY", "reserve_cus", "Queries", "build_queries", "score_queries", "ScorePipeline"]


PADDED_BLOCKS = True  # score_emb_sp_po(pad_pitch=True) exists (kge_amd.sharded asks its backend)


def reserve_cus(n: int) -> int:
    """flags value: leave `n` compute units free for kernels on other streams (KGE_FLAG_RESERVE_CUS)."""
    return (int(n) & 255) << 8


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"kge_amd: {what} is on '{t.device}'. The MI355X engine has no CPU path "
            "(use the reference implementation for job.device=cpu).")


def _empty(shape, device, dtype=torch.float32):
    """Allocate through torch's caching allocator.  LibKGE's sub-batch auto-tuner string-
    matches 'CUDA out of memory' (kge/job/train.py:384-391); ROCm says 'HIP out of memory'."""
    try:
        return torch.empty(shape, device=device, dtype=dtype)
    except torch.OutOfMemoryError as e:  # pragma: no cover - needs a full GPU
        raise RuntimeError("CUDA out of memory (kge_amd: " + str(e) + ")") from e


_raw_stream_fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle(device) -> int:
    """Raw hipStream_t of torch's current stream on `device` (the fast C entry when torch has it:
    torch.cuda.current_stream() builds a Python Stream object, ~4 us per call)."""
    if _raw_stream_fn is not None:
        return _raw_stream_fn(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _stream(device):
    return ctypes.c_void_p(_stream_handle(device))


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the context manager
    costs ~3 us, the scoring calls are ~15 us of GPU time)."""

    def __init__(self, device):
        self.ctx = None if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


_WORKSPACES = {}
_WS_NEED = {}


def _workspace(tc, n, device, enable=True, stream=None):
    """(ptr, bytes) of the per-(device, stream) scratch buffer the bf16 path uses to build the
    query vectors once (kge_score_workspace_bytes); (None, 0) when the call needs none.
    Stream-ordered reuse: one buffer per stream, grown on demand through torch's allocator."""
    if not enable:
        return None, 0
    nk = (tc.dtype, tc.scorer, tc.dim, n, tc.flags & FLAG_SPLIT_QUERY)  # (split queries: about twice the bytes)
    need = _WS_NEED.get(nk)
    if need is None:
        need = _WS_NEED[nk] = _lib.lib().kge_score_workspace_bytes(ctypes.byref(tc), n)
    if need <= 0:
        return None, 0
    key = (device.index, _stream_handle(device) if stream is None else stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < need:
        # zero-initialised ONCE (include/kge_amd.h): besides scratch it holds the builders' flag lines
        # and the "degraded" word of the cooperative query build, maintained by the kernels from then on
        try:
            buf = torch.zeros((max(need, 1 << 20),), device=device, dtype=torch.uint8)
        except torch.OutOfMemoryError as e:  # pragma: no cover
            raise RuntimeError("CUDA out of memory (kge_amd: " + str(e) + ")") from e
        _WORKSPACES[key] = buf
    return buf.data_ptr(), buf.numel()


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"kge_amd: embedding tables must be float32 or bfloat16, got {t.dtype}")


def _index(ix, device, keep):
    """1-D int32/int64 tensor (any stride, e.g. triples[:, 0]) -> KgeIndex.  None -> identity."""
    if ix is None:
        return KgeIndex(None, I64, 0, 1)
    if not torch.is_tensor(ix):
        ix = torch.as_tensor(ix, device=device)
    if ix.device != device:
        raise RuntimeError(f"kge_amd: index tensor on {ix.device}, tables on {device}")
    if ix.dim() != 1:
        ix = ix.reshape(-1)
    if ix.dtype not in (torch.int32, torch.int64):
        ix = ix.long()
    stride = ix.stride(0) if ix.numel() > 1 else 1
    if stride < 1:
        ix = ix.contiguous()
        stride = 1
    keep.append(ix)
    return KgeIndex(ix.data_ptr(), I32 if ix.dtype == torch.int32 else I64, 0, stride)


def _index3(triples, device, keep):
    """An [n, 3] int32/int64 triples tensor (the reference's batch["triples"]: s, p, o = its columns,
    kge/job/eval_entity_ranking.py:196-199, train_1vsAll.py:53) -> three KgeIndex on the one allocation."""
    if triples.device != device:
        raise RuntimeError(f"kge_amd: index tensor on {triples.device}, tables on {device}")
    if triples.dim() != 2 or triples.shape[1] != 3:
        raise ValueError("kge_amd: a batch is (s, p, o) or an [n, 3] triples tensor")
    if triples.dtype not in (torch.int32, torch.int64):
        triples = triples.long()
    st0, st1 = triples.stride()
    if st0 < 1 or st1 < 1:
        triples = triples.contiguous()
        st0, st1 = 3, 1
    keep.append(triples)
    it, w = (I32, 4) if triples.dtype == torch.int32 else (I64, 8)
    base = triples.data_ptr()
    n = triples.shape[0]
    st = st0 if n > 1 else 1
    return KgeIndex(base, it, 0, st), KgeIndex(base + st1 * w, it, 0, st), KgeIndex(base + 2 * st1 * w, it, 0, st), n


def _targets(targets, t, keep):
    """The `targets` operand of score_sp / score_po / score_sp_po -> (KgeIndex, m): None = all entities, a `range`
    (step 1) = that contiguous chunk of the table -- kge_index.start, include/kge_amd.h: no index is read, the
    all-entities kernels stream rows [start, stop) --, anything else = a listed subset."""
    if isinstance(targets, range):
        if targets.step != 1 or targets.start < 0 or targets.stop > t.num_ent or targets.stop < targets.start:
            raise ValueError(f"kge_amd: a target range must be a step-1 range inside [0, {t.num_ent}); got {targets}")
        return KgeIndex(None, I64, targets.start, 1), len(targets)
    ti = _index(targets, t.device, keep)
    return ti, (t.num_ent if targets is None else keep[-1].numel())


def _same_len(ixs, what):
    """All index operands of one call must have the same length (the reference would raise a shape
    error; a shorter vector here would be an out-of-bounds read on the device)."""
    n = ixs[0].numel()
    for x in ixs[1:]:
        if x.numel() != n:
            raise ValueError(f"kge_amd: {what}: index vectors of different lengths "
                             f"({[int(y.numel()) for y in ixs]})")
    return n


class Tables:
    """Entity and relation lookup tables of one model (LookupEmbedder weights,
    kge/model/embedder/lookup_embedder.py:44-46) as the kernels see them."""

    def __init__(self, scorer, ent: torch.Tensor, rel: torch.Tensor, l_norm: float = 1.0,
                 flags: int = 0, use_workspace: bool = True, pad_pitch: bool = False):
        self.scorer = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
        _require_gpu(ent, "entity table")
        _require_gpu(rel, "relation table")
        if ent.dim() != 2 or rel.dim() != 2 or ent.stride(1) != 1 or rel.stride(1) != 1:
            raise ValueError("kge_amd: tables must be 2-D with unit inner stride")
        if ent.dtype != rel.dtype:
            raise TypeError("kge_amd: entity and relation tables must share a dtype")
        self.ent, self.rel = ent, rel
        self.l_norm, self.flags = float(l_norm), int(flags)
        # True (default): the bf16 ComplEx/DistMult kernel gets a per-(device, stream) scratch
        # buffer and builds the query vectors once, cooperatively, instead of once per
        # workgroup (same bits, one launch either way).  False: no scratch buffer.
        self.use_workspace = bool(use_workspace)
        # True (opt-in): score_sp / score_po return a [:, :m] VIEW of a matrix whose row pitch is
        # rounded up to 32 floats, so every row starts on a 128-byte line and the kernel's 16-byte
        # stores never straddle a 32-byte sector (E = 14,541: rows of a contiguous matrix start at
        # 4-byte granularity).  The reference returns a contiguous tensor, hence not the default.
        self.pad_pitch = bool(pad_pitch)
        self.device = ent.device
        self._c_cache = {}

    def c(self, flags=None) -> KgeTables:
        """The C view of the tables; cached per flags value (rebuilt if a table was re-allocated)."""
        e, r = self.ent, self.rel
        flags = self.flags if flags is None else flags
        hit = self._c_cache.get(flags)
        if hit is not None and hit[1] == e.data_ptr() and hit[2] == r.data_ptr():
            return hit[0]
        tc = KgeTables(e.data_ptr(), r.data_ptr(), _dtype_code(e), self.scorer, e.shape[0],
                       r.shape[0], e.shape[1], r.shape[1], e.stride(0), r.stride(0),
                       self.l_norm, flags)
        self._c_cache[flags] = (tc, e.data_ptr(), r.data_ptr())
        return tc

    @property
    def num_ent(self):
        return self.ent.shape[0]


def _ext():
    """The torch C++ extension kge_amd._C (csrc/torch_ext.cpp: one C++ call per scoring call, ~2 us of host time
    against ~9 through ctypes) -- the binding of the index-level scoring calls; KGE_AMD_BINDING=ctypes keeps them on
    ctypes (the binding of everything else).  Both end in the same C entry points of libkge_amd.so."""
    global _EXT
    if _EXT is None:
        if os.environ.get("KGE_AMD_BINDING", "ext") == "ctypes" or not os.path.exists(_lib.EXT_PATH):
            _EXT = False
        else:
            try:
                _EXT = _lib.ext()
            except (ImportError, OSError, RuntimeError) as exc:
                # a stale _C.so (built against another torch) or an ABI mismatch: both bindings end in the same C
                # entry points of libkge_amd.so, so say it once and keep scoring through ctypes
                import warnings
                warnings.warn(f"kge_amd: the torch extension kge_amd._C does not load ({type(exc).__name__}: {exc}); "
                              "scoring calls go through the ctypes binding")
                _EXT = False
    return _EXT


_EXT = None


def _workspace_tensor(tc, n, device, enable, st):
    ws, _ = _workspace(tc, n, device, enable, st)
    return None if ws is None else _WORKSPACES[(device.index, st)]


def score_spo(t: Tables, s, p, o, flags=None) -> torch.Tensor:
    ex = _ext()
    if ex and torch.is_tensor(s) and torch.is_tensor(p) and torch.is_tensor(o):
        with _on_device(t.device):
            return ex.score_spo(t.ent, t.rel, t.scorer, t.l_norm, t.flags if flags is None else flags, s, p, o)
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_spo")
    out = _empty((n,), t.device)
    with torch.cuda.device(t.device):
        tc = t.c(flags)
        _lib.check(_lib.lib().kge_score_spo(ctypes.byref(tc), si, pi, oi, n, out.data_ptr(),
                                            _stream(t.device)), "kge_score_spo")
    return out


def _pairs(fn_name, t: Tables, a, p, targets, flags, out=None, ldo=None, padded=None):
    ex = _ext()
    padded = t.pad_pitch if padded is None else bool(padded)
    # (small batches gain nothing from aligned rows and lose the extension's shorter host path: 12.7 -> 15.3 us at
    # n = 100, FB15k-237 shape; from n = 512 on the padded rows win: profiles/r6_bench_mid.json one_call_entry)
    if padded and torch.is_tensor(a) and a.numel() < 512:
        padded = False
    if ex and out is None and not padded and torch.is_tensor(a) and torch.is_tensor(p) and \
            (targets is None or torch.is_tensor(targets)):
        with _on_device(t.device):
            fl = t.flags if flags is None else flags
            st = _stream_handle(t.device)
            ws = _workspace_tensor(t.c(fl), a.numel(), t.device, t.use_workspace, st)
            return ex.score_pairs(t.ent, t.rel, t.scorer, t.l_norm, fl, SP_ if fn_name == "kge_score_sp" else PO_, a, p,
                                  None, targets, ws)
    keep = []
    ai, pi = _index(a, t.device, keep), _index(p, t.device, keep)
    n = _same_len(keep[:2], "_pairs")
    ti, m = _targets(targets, t, keep)
    ret = None
    if out is None:
        if padded and m > 0:
            # whole 256-byte lines per row, an odd number of them (score_pitch): every 16-byte lane store of the kernels
            # stays inside a 32-byte sector and the rows spread over the memory channels; the caller gets the [:, :m] view
            ldo = score_pitch(m)
            out = _empty((n, ldo), t.device)
            ret = out[:, :m]
        else:
            out = _empty((n, m), t.device)
            ldo = m
    with _on_device(t.device):
        tc = t.c(flags)
        fn = getattr(_lib.lib(), fn_name)
        # C signatures: kge_score_sp(t, s, p, ...) but kge_score_po(t, p, o, ...)
        first, second = (ai, pi) if fn_name == "kge_score_sp" else (pi, ai)
        st = _stream_handle(t.device)
        ws, wsb = _workspace(tc, n, t.device, t.use_workspace, st)
        rc = fn(ctypes.byref(tc), first, second, n, ti, m, out.data_ptr(), ldo, ws, wsb, st)
        if rc:
            _lib.check(rc, fn_name)
    return out if ret is None else ret


def score_sp(t: Tables, s, p, o=None, flags=None, padded=None) -> torch.Tensor:
    """[n, E|m] scores of (s_i, p_i, ·) against all / the listed objects / the objects of a step-1 `range` (a chunk of
    the table streamed without an index: the all-entities kernels).  padded=True (default: the tables'
    `pad_pitch`): the [:, :m] view of a matrix on the row pitch `score_pitch(m)` -- sector-aligned rows, what the store
    kernels are measured on (E = 14,541 is odd: rows of a contiguous matrix start at 4-byte granularity); the reference
    returns a contiguous tensor, so this is an option."""
    return _pairs("kge_score_sp", t, s, p, o, flags, padded=padded)


def score_po(t: Tables, p, o, s=None, flags=None, padded=None) -> torch.Tensor:
    """[n, E|m] scores of (·, p_i, o_i) against all / the listed subjects; `padded` as for score_sp."""
    return _pairs("kge_score_po", t, o, p, s, flags, padded=padded)


def score_sp_po(t: Tables, s, p, o, entity_subset=None, flags=None) -> torch.Tensor:
    """[n, 2m]: score_sp and score_po against one shared entity subset, written directly
    into the two halves of the output (no torch.cat copy, kge_model.py:789)."""
    ex = _ext()
    if ex and torch.is_tensor(s) and torch.is_tensor(p) and torch.is_tensor(o) and \
            (entity_subset is None or torch.is_tensor(entity_subset)):
        with _on_device(t.device):
            fl = t.flags if flags is None else flags
            st = _stream_handle(t.device)
            ws = _workspace_tensor(t.c(fl), s.numel(), t.device, t.use_workspace, st)
            return ex.score_pairs(t.ent, t.rel, t.scorer, t.l_norm, fl, SP_PO, s, p, o, entity_subset, ws)
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_sp_po")
    ti, m = _targets(entity_subset, t, keep)
    out = _empty((n, 2 * m), t.device)
    with _on_device(t.device):
        tc = t.c(flags)
        st = _stream_handle(t.device)
        ws, wsb = _workspace(tc, n, t.device, t.use_workspace, st)
        rc = _lib.lib().kge_score_sp_po(ctypes.byref(tc), si, pi, oi, n, ti, m, out.data_ptr(), 2 * m,
                                        ws, wsb, st)
        if rc:
            _lib.check(rc, "kge_score_sp_po")
    return out


def score_pitch(m: int) -> int:
    """Row pitch (floats) for a score block of m columns that the store path likes: whole 256-byte lines per row
    segment (sector-aligned 16-byte lane stores, write-through) and an ODD number of lines per row -- at the
    FB15k-237 shape a two-sided block on a pitch of 228 lines (14,592 floats, 2 x 228 per row pair) takes 21.2 us,
    on 229 lines 19.0 us (tools/pitch_probe.py: rows 228 lines apart land on few memory channels)."""
    lines = (m + 63) // 64
    return (lines | 1) * 64


# ---- prepared queries (kge_build_queries / kge_score_queries) ----------------------------------------------------
_COMBINE = {"sp_": SP_, "_po": PO_, "sp_po": SP_PO}


class Queries:
    """The prepared query vectors of one batch (opaque device buffer, include/kge_amd.h): built by
    `build_queries` or by the previous batch's `score_queries(..., next=...)`."""

    __slots__ = ("buf", "combine", "n", "flags")

    def __init__(self, t: Tables, combine: str, n: int, flags=None):
        tc = t.c(flags)
        need = _lib.lib().kge_queries_bytes(ctypes.byref(tc), _COMBINE[combine], n)
        if need <= 0:
            raise RuntimeError("kge_amd: prepared queries need bf16 ComplEx / DistMult tables of dim 256 / 512")
        self.buf = _empty((need,), t.device, torch.uint8)
        self.combine, self.n, self.flags = combine, n, flags


def build_queries(t: Tables, combine: str, s, p=None, o=None, flags=None, out: Queries = None, stream=None) -> Queries:
    """Query vectors of the batch (s, p, o) for `score_queries` (combine "sp_": s, p; "_po": p, o; "sp_po": all).
    stream: a raw hipStream_t to launch on (default: torch's current stream)."""
    keep = []
    if o is None and p is None and torch.is_tensor(s) and s.dim() == 2:  # [n, 3] triples
        si, pi, oi, n = _index3(s, t.device, keep)
    else:
        si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
        n = _same_len([k for k in keep], "build_queries")
    q = out if out is not None else Queries(t, combine, n, flags)
    if q.n != n or q.combine != combine:
        raise ValueError("kge_amd: build_queries: the Queries buffer was sized for another batch shape")
    with _on_device(t.device):
        tc = t.c(q.flags)
        rc = _lib.lib().kge_build_queries(ctypes.byref(tc), _COMBINE[combine], si, pi, oi, n, q.buf.data_ptr(),
                                          q.buf.numel(), _stream_handle(t.device) if stream is None else stream)
        if rc:
            _lib.check(rc, "kge_build_queries")
    return q


def _check_score_out(t, out, n, m, combine, what):
    """(ldo, block2_offset) of a caller's score buffer after checking it is one the kernels may write through its
    raw pointer: float32, on the tables' GPU, `[n, m]` / `[n, 2m]` with unit inner stride, or -- both blocks of
    "sp_po" on their own aligned columns -- `[n, 2, m]` with stride (ldo, block2_offset, 1)."""
    _require_gpu(out, "score buffer")
    if out.device != t.device:
        raise ValueError(f"kge_amd: {what}: `out` is on {out.device}, the tables on {t.device}")
    if out.dtype != torch.float32:
        raise ValueError(f"kge_amd: {what}: `out` must be float32, got {out.dtype}")
    width = 2 * m if combine == "sp_po" else m
    b2 = 0
    if out.dim() == 3:
        if combine != "sp_po" or tuple(out.shape) != (n, 2, m) or (m > 1 and out.stride(2) != 1):
            raise ValueError(f"kge_amd: {what}: a 3-D `out` is [n, 2, m] with unit inner stride for combine 'sp_po'")
        b2 = out.stride(1)
        if b2 < m:
            raise ValueError(f"kge_amd: {what}: the two blocks of `out` overlap")
        need = b2 + m
    elif out.dim() == 2:
        if tuple(out.shape) != (n, width) or (width > 1 and out.stride(1) != 1):
            raise ValueError(f"kge_amd: {what}: `out` must be [{n}, {width}] with unit inner stride, "
                             f"got {tuple(out.shape)} strides {out.stride()}")
        need = width
    else:
        raise ValueError(f"kge_amd: {what}: `out` must be 2-D (or [n, 2, m])")
    ldo = out.stride(0)
    if n == 1:
        ldo = max(ldo, need)
    if ldo < need:
        raise ValueError(f"kge_amd: {what}: the rows of `out` overlap (row stride {ldo} < {need})")
    return ldo, b2


def score_queries(t: Tables, q: Queries, targets=None, out=None, next_batch=None, next_queries: Queries = None,
                  stream=None):
    """[n, m] ("sp_", "_po") or [n, 2m] ("sp_po") scores of the prepared batch `q` against all / the listed
    entities -- the bits of score_sp / score_po / score_sp_po.  next_batch = (s, p, o) + next_queries: the NEXT
    batch's queries are built by idle workgroups of the same launch (one launch per batch, no start-up chain).
    stream: a raw hipStream_t to launch on (default: torch's current stream; the caller orders it against the
    producers of the operands and the consumers of `out`)."""
    keep = []
    ti = _index(targets, t.device, keep)
    m = t.num_ent if targets is None else keep[-1].numel()
    width = 2 * m if q.combine == "sp_po" else m
    if out is None:
        out = _empty((q.n, width), t.device)
    ldo, b2 = _check_score_out(t, out, q.n, m, q.combine, "score_queries")
    nxt = None
    if next_batch is not None:
        if next_queries is None:
            raise ValueError("kge_amd: score_queries: next_batch needs next_queries (the buffer its queries go to)")
        nkeep = []
        if torch.is_tensor(next_batch):  # [n, 3] triples: one allocation, three strided index vectors
            si, pi, oi, nn = _index3(next_batch, t.device, nkeep)
        else:
            si, pi, oi = (_index(x, t.device, nkeep) for x in next_batch)
            nn = _same_len(nkeep, "score_queries(next_batch)")
        keep += nkeep
        if next_queries.n != nn or next_queries.combine != q.combine or next_queries.flags != q.flags:
            raise ValueError("kge_amd: score_queries: next_queries does not match the next batch")
        nxt = KgeNextQueries(si, pi, oi, nn, next_queries.buf.data_ptr(), next_queries.buf.numel())
    with _on_device(t.device):
        tc = t.c(q.flags)
        rc = _lib.lib().kge_score_queries(ctypes.byref(tc), _COMBINE[q.combine], q.buf.data_ptr(), q.n, ti, m,
                                          out.data_ptr(), ldo, b2, ctypes.byref(nxt) if nxt is not None else None,
                                          _stream_handle(t.device) if stream is None else stream)
        if rc:
            _lib.check(rc, "kge_score_queries")
    return out


class QueriesGroup:
    """The prepared query vectors of a GROUP of `num_batches` equally shaped batches (kge_build_queries_multi): batch l
    of the group at byte offset l * stride of one device buffer."""

    __slots__ = ("buf", "combine", "n", "num_batches", "stride", "flags")

    def __init__(self, t: Tables, combine: str, n: int, num_batches: int, flags=None):
        tc = t.c(flags)
        per = _lib.lib().kge_queries_bytes(ctypes.byref(tc), _COMBINE[combine], n)
        if per <= 0:
            raise RuntimeError("kge_amd: prepared queries need bf16 ComplEx / DistMult tables of dim 256 / 512")
        self.stride = (per + 255) // 256 * 256
        self.buf = _empty((self.stride * num_batches,), t.device, torch.uint8)
        self.combine, self.n, self.num_batches, self.flags = combine, n, num_batches, flags


def _group_index(batch, n, num_batches, device, keep, what):
    """(s, p, o) index vectors of num_batches * n entries, or one [num_batches * n, 3] triples tensor."""
    if torch.is_tensor(batch):
        si, pi, oi, nn = _index3(batch, device, keep)
    else:
        k0 = len(keep)
        si, pi, oi = (_index(x, device, keep) for x in batch)
        nn = _same_len(keep[k0:], what)
    if nn != n * num_batches:
        raise ValueError(f"kge_amd: {what}: a group of {num_batches} batches of {n} rows needs {n * num_batches} "
                         f"index entries, got {nn}")
    return si, pi, oi


def build_queries_group(t: Tables, combine: str, batch, n: int, num_batches: int, flags=None, out: QueriesGroup = None,
                        stream=None) -> QueriesGroup:
    """Query vectors of `num_batches` batches of `n` rows in one launch: batch l = rows [l n, (l + 1) n) of the index
    vectors `batch` = (s, p, o) (or one [num_batches n, 3] triples tensor)."""
    keep = []
    si, pi, oi = _group_index(batch, n, num_batches, t.device, keep, "build_queries_group")
    q = out if out is not None else QueriesGroup(t, combine, n, num_batches, flags)
    if q.n != n or q.num_batches != num_batches or q.combine != combine:
        raise ValueError("kge_amd: build_queries_group: the QueriesGroup buffer was sized for another shape")
    with _on_device(t.device):
        tc = t.c(q.flags)
        rc = _lib.lib().kge_build_queries_multi(ctypes.byref(tc), _COMBINE[combine], si, pi, oi, n, num_batches,
                                                q.buf.data_ptr(), q.stride, q.buf.numel(),
                                                _stream_handle(t.device) if stream is None else stream)
        if rc:
            _lib.check(rc, "kge_build_queries_multi")
    return q


def score_pitch_group(n: int, m: int, combine: str = "sp_po"):
    """(row pitch, batch stride) in floats of a group's score buffer on the pitch `score_pitch` recommends."""
    pitch = score_pitch(m) * (2 if combine == "sp_po" else 1)
    return pitch, n * pitch


def score_queries_group(t: Tables, q: QueriesGroup, out: torch.Tensor, next_batch=None, next_queries: QueriesGroup = None,
                        stream=None):
    """Scores of a whole group in ONE persistent launch (kge_score_queries_multi): out[l] receives what
    score_queries writes for batch l -- `out` is [L, n, m] / [L, n, 2m] (any row pitch, unit inner stride) or
    [L, n, 2, m].  next_batch + next_queries: the NEXT group's query vectors are built by the same launch."""
    m = t.num_ent
    L, n = q.num_batches, q.n
    if out.dim() < 3 or out.shape[0] != L:
        raise ValueError(f"kge_amd: score_queries_group: `out` is [{L}, n, ...], one block per batch")
    if next_batch is not None and (next_queries is None or next_queries.num_batches != L or
                                   next_queries.combine != q.combine or next_queries.flags != q.flags):
        raise ValueError("kge_amd: score_queries_group: next_queries does not match the next group")
    ex = _ext()
    if ex and stream is None and (next_batch is None or torch.is_tensor(next_batch) or
                                  all(torch.is_tensor(x) for x in next_batch)):
        # the torch extension (csrc/torch_ext.cpp): one C++ call, `out` validated there (ValueError), the next group's
        # index vectors taken as tensors (a [L n, 3] triples tensor as its three strided columns)
        if next_batch is None:
            ns = np_ = no = nq = None
            nn = nstride = 0
        else:
            if torch.is_tensor(next_batch):
                if next_batch.dim() != 2 or next_batch.shape[1] != 3:
                    raise ValueError("kge_amd: score_queries_group(next_batch): a triples tensor is [rows, 3]")
                ns, np_, no = next_batch[:, 0], next_batch[:, 1], next_batch[:, 2]
            else:
                ns, np_, no = next_batch
            nq, nn, nstride = next_queries.buf, next_queries.n, next_queries.stride
        with _on_device(t.device):
            ex.score_queries_group(t.ent, t.rel, t.scorer, t.flags if q.flags is None else q.flags, _COMBINE[q.combine],
                                   q.buf, q.stride, n, L, out, ns, np_, no, nn, nq, nstride)
        return out
    ldo, b2 = _check_score_out(t, out[0], n, m, q.combine, "score_queries_group")
    ostride = out.stride(0) if L > 1 else 0
    nxt, keep = None, []
    if next_batch is not None:
        si, pi, oi = _group_index(next_batch, next_queries.n, L, t.device, keep, "score_queries_group(next_batch)")
        nxt = KgeNextQueries(si, pi, oi, next_queries.n, next_queries.buf.data_ptr(), next_queries.buf.numel())
    with _on_device(t.device):
        tc = t.c(q.flags)
        rc = _lib.lib().kge_score_queries_multi(
            ctypes.byref(tc), _COMBINE[q.combine], q.buf.data_ptr(), q.stride, n, L, KgeIndex(None, I64, 0, 1), m,
            out.data_ptr(), ostride, ldo, b2, ctypes.byref(nxt) if nxt is not None else None,
            next_queries.stride if nxt is not None else 0, _stream_handle(t.device) if stream is None else stream)
        if rc:
            _lib.check(rc, "kge_score_queries_multi")
    return out


class ScorePipeline:
    """A stream of equally shaped batches through `score_queries`: batch k is scored while a later batch's queries
    are built inside the same launch (two alternating Queries buffers per lane).

        pipe = ScorePipeline(T, "sp_po", n); pipe.start(s0, p0, o0)
        for k in ...: scores = pipe.step(next_batch=(s, p, o) of batch k + 1 or None)

    streams = L > 1: L batches in flight, batch k on HIP stream k % L (lane k % L).  One scoring launch fills the
    chip for ~10-20 us, of which the first ~2 us (launch gap, cold first tiles) and the last ~1 us (the last stores'
    acknowledgements) leave compute units idle, as do the 256 - 228 units a [512 x 14,541] grid cannot use; the
    launch of the next batch on ANOTHER stream runs in those holes (measured: 13.5 -> 10.3 us per one-sided batch,
    22.0 -> 18.9 us two-sided, tools/dual_stream_probe.py).  A lane's launch builds the queries of that lane's next
    batch, so `next_batch` is the batch L steps ahead:

        pipe = ScorePipeline(T, "sp_po", n, streams=2); pipe.start([(s0, p0, o0), (s1, p1, o1)])
        for k in ...: scores_k = pipe.step(next_batch=batch k + 2, out=buffers[k % 2])
        pipe.join()   # torch's current stream waits for the lanes (consumers on other streams: pipe.event(k % 2))

    fork() (called by start()) orders the lanes behind the current stream: the producers of the tables and of the
    batches' index tensors, and the READERS of a score buffer that a later step overwrites -- call it again after
    consuming scores on the current stream and before handing the same buffer to step() once more.
    """

    def __init__(self, t: Tables, combine: str, n: int, flags=None, streams: int = 1):
        self.t = t
        self.lanes = max(1, int(streams))
        self.q = [[Queries(t, combine, n, flags), Queries(t, combine, n, flags)] for _ in range(self.lanes)]
        self.cur = [0] * self.lanes
        self.k = 0
        self._streams = None
        if self.lanes > 1:
            self._streams = [torch.cuda.Stream(device=t.device) for _ in range(self.lanes)]
            self._handles = [st.cuda_stream for st in self._streams]

    def _handle(self, lane):
        return None if self._streams is None else self._handles[lane]

    def fork(self):
        """The lanes wait for everything issued to torch's current stream so far."""
        if self._streams is not None:
            ev = torch.cuda.Event()
            ev.record()
            for st in self._streams:
                st.wait_event(ev)

    def event(self, lane: int):
        """An event recorded behind the last launch of `lane` (None with a single lane: the current stream)."""
        if self._streams is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self._streams[lane])
        return ev

    def join(self):
        """Torch's current stream waits for every lane."""
        if self._streams is not None:
            cur = torch.cuda.current_stream(self.t.device)
            for lane in range(self.lanes):
                cur.wait_event(self.event(lane))

    def start(self, s, p=None, o=None):
        """The first batch (one lane) or the list of the first `streams` batches."""
        first = [s if p is None and o is None and torch.is_tensor(s) and s.dim() == 2 else (s, p, o)] \
            if self._streams is None else list(s)
        if len(first) != self.lanes:
            raise ValueError("kge_amd: ScorePipeline.start takes the first `streams` batches")
        self.fork()
        self.k = 0
        for lane, b in enumerate(first):
            self.cur[lane] = 0
            b = (b, None, None) if torch.is_tensor(b) else b
            build_queries(self.t, self.q[lane][0].combine, *b, out=self.q[lane][0], stream=self._handle(lane))

    def step(self, next_batch=None, targets=None, out=None):
        lane = self.k % self.lanes
        self.k += 1
        c = self.cur[lane]
        q, nq = self.q[lane][c], self.q[lane][1 - c]
        res = score_queries(self.t, q, targets, out, next_batch, nq if next_batch is not None else None,
                            stream=self._handle(lane))
        if out is None and self._streams is not None:
            res.record_stream(self._streams[lane])
        if next_batch is not None:
            self.cur[lane] = 1 - c
        return res


def score_neg(t: Tables, s, p, o, slot: int, neg: torch.Tensor, flags=None) -> torch.Tensor:
    """[n, K] scores of triple i with `slot` (0 = s, 2 = o) replaced by neg[i, k]."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_neg")
    _require_gpu(neg, "negative samples")
    if neg.dtype not in (torch.int32, torch.int64):
        neg = neg.long()
    if neg.dim() != 2 or neg.shape[0] != n:
        raise ValueError("kge_amd: neg must be [n, K]")
    if neg.stride(1) != 1:
        neg = neg.contiguous()
    K = neg.shape[1]
    out = _empty((n, K), t.device)
    with torch.cuda.device(t.device):
        tc = t.c(flags)
        _lib.check(_lib.lib().kge_score_neg(
            ctypes.byref(tc), si, pi, oi, n, int(slot), neg.data_ptr(),
            I32 if neg.dtype == torch.int32 else I64, neg.stride(0) if n > 1 else K, K,
            out.data_ptr(), K, _stream(t.device)), "kge_score_neg")
    return out


# kge_score_neg_bwd_accum_sorted (occurrences sorted by corrupted entity: one gradient-row flush per run of equal ids
# instead of one float atomic per element and occurrence) from this many occurrences on, when every entity is drawn
# several times on average; NEG_BWD_SORTED = True / False forces either (tests, bench.py's A/B leg).
NEG_BWD_SORTED_MIN = 1 << 17
NEG_BWD_SORTED = None


def _neg_bwd_sorted(n, K, num_ent) -> bool:
    if NEG_BWD_SORTED is not None:
        return bool(NEG_BWD_SORTED)
    return n * K >= NEG_BWD_SORTED_MIN and n * K >= 4 * num_ent


def score_neg_bwd_accum(t: Tables, s, p, o, slot: int, neg: torch.Tensor, gout, scores, grad_ent, grad_rel):
    """Backward of score_neg accumulated straight into the dense table gradients `grad_ent` [E, d] and
    `grad_rel` [R, d_r] (f32, modified in place); False if the kernel does not take this shape.  Many occurrences per
    entity (negative sampling with hundreds of negatives per positive): sorted by entity first (torch.sort on the device:
    plumbing), then kge_score_neg_bwd_accum_sorted."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_neg_bwd_accum")
    if neg.dtype not in (torch.int32, torch.int64):
        neg = neg.long()
    if neg.dim() != 2 or neg.shape[0] != n:
        raise ValueError("kge_amd: neg must be [n, K]")
    if neg.stride(1) != 1:
        neg = neg.contiguous()
    K = neg.shape[1]
    gout = gout.to(device=t.device, dtype=torch.float32)
    if gout.dim() != 2 or gout.stride(1) != 1:
        gout = gout.contiguous().view(n, K)
    sc = None
    if scores is not None:
        sc = scores if (scores.dim() == 2 and scores.stride(1) == 1) else scores.contiguous().view(n, K)
    if n * K > 0 and _neg_bwd_sorted(n, K, t.num_ent):
        flat = neg if neg.is_contiguous() else neg.contiguous()
        # counting sort by entity id: histogram + exclusive prefix sums (torch: plumbing), then one cursor bump per sample
        # (kge_neg_order twice around a cumsum -- torch.bincount would wait for the host: it reads the largest id back)
        counts = torch.zeros(t.num_ent, dtype=torch.int64, device=t.device)
        order = torch.empty(n * K, dtype=torch.int64, device=t.device)
        rot = (torch.empty(t.rel.shape[0], 2 * t.rel.shape[1], dtype=torch.float32, device=t.device)
               if t.scorer == SCORERS["rotate"] else None)  # scratch for the relations' cos / sin
        it = I32 if flat.dtype == torch.int32 else I64
        with _on_device(t.device):
            tc = t.c()
            st = _stream_handle(t.device)
            rc = _lib.lib().kge_neg_order(flat.data_ptr(), it, max(K, 1), n, K, t.num_ent, counts.data_ptr(), None, st)
            if rc:
                _lib.check(rc, "kge_neg_order (histogram)")
            cursor = torch.cumsum(counts, 0).sub_(counts)
            rc = _lib.lib().kge_neg_order(flat.data_ptr(), it, max(K, 1), n, K, t.num_ent, cursor.data_ptr(),
                                          order.data_ptr(), st)
            if rc:
                _lib.check(rc, "kge_neg_order")
            rc = _lib.lib().kge_score_neg_bwd_accum_sorted(
                ctypes.byref(tc), si, pi, oi, n, int(slot), flat.data_ptr(), I32 if flat.dtype == torch.int32 else I64,
                max(K, 1), K, order.data_ptr(), gout.data_ptr(), gout.stride(0) if n > 1 else max(K, 1),
                None if sc is None else sc.data_ptr(), 0 if sc is None else (sc.stride(0) if n > 1 else max(K, 1)),
                grad_ent.data_ptr(), grad_ent.stride(0), grad_rel.data_ptr(), grad_rel.stride(0),
                None if rot is None else rot.data_ptr(), _stream_handle(t.device))
        if rc == _lib.KGE_ERR_UNSUPPORTED:
            return False
        if rc:
            _lib.check(rc, "kge_score_neg_bwd_accum_sorted")
        return True
    with _on_device(t.device):
        tc = t.c()
        rc = _lib.lib().kge_score_neg_bwd_accum(
            ctypes.byref(tc), si, pi, oi, n, int(slot), neg.data_ptr(),
            I32 if neg.dtype == torch.int32 else I64, neg.stride(0) if n > 1 else max(K, 1), K,
            gout.data_ptr(), gout.stride(0) if n > 1 else max(K, 1), None if sc is None else sc.data_ptr(),
            0 if sc is None else (sc.stride(0) if n > 1 else max(K, 1)), grad_ent.data_ptr(), grad_ent.stride(0),
            grad_rel.data_ptr(), grad_rel.stride(0), _stream_handle(t.device))
    if rc == _lib.KGE_ERR_UNSUPPORTED:
        return False
    if rc:
        _lib.check(rc, "kge_score_neg_bwd_accum")
    return True


_EMB_TC = {}


def score_emb(scorer, s_emb, p_emb, o_emb, combine: str, l_norm: float = 1.0, flags: int = 0,
              pad_pitch: bool = False):
    """RelationalScorer.score_emb on dense embeddings (no gather).  pad_pitch: return a [:, :m] view
    of a matrix whose row pitch is rounded up to 32 floats (every row 128-byte aligned: the kernel's
    16-byte stores then never straddle a sector -- 28 % faster on a 574,311-column shard)."""
    code = {"spo": SPO, "sp_": SP_, "_po": PO_}.get(combine)
    if code is None:
        raise ValueError('cannot handle combine="{}"'.format(combine))
    for x in (s_emb, p_emb, o_emb):
        _require_gpu(x, "embedding")
    dt = {s_emb.dtype, p_emb.dtype, o_emb.dtype}
    if len(dt) != 1:
        raise TypeError("kge_amd: embeddings must share a dtype")
    s_emb, p_emb, o_emb = (x if x.stride(-1) == 1 else x.contiguous() for x in (s_emb, p_emb, o_emb))
    sc = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
    n = p_emb.shape[0]
    d, dr = s_emb.shape[1], p_emb.shape[1]
    if code == SPO:
        m, out = 0, _empty((n,), s_emb.device)
    else:
        m = o_emb.shape[0] if code == SP_ else s_emb.shape[0]
        out = _empty((n, (m + 31) // 32 * 32 if pad_pitch else m), s_emb.device)
    ldo = out.stride(0) if (out.dim() == 2 and n > 1) else max(m, 1)
    key = (s_emb.dtype, sc, d, dr, float(l_norm), int(flags))
    tc = _EMB_TC.get(key)
    if tc is None:
        tc = _EMB_TC[key] = KgeTables(None, None, _dtype_code(s_emb), sc, 0, 0, d, dr, d, dr, float(l_norm),
                                      int(flags))
    with _on_device(s_emb.device):
        st = _stream_handle(s_emb.device)
        ws, wsb = _workspace(tc, n, s_emb.device, True, st)
        rc = _lib.lib().kge_score_emb(
            ctypes.byref(tc), code, s_emb.data_ptr(), s_emb.stride(0), p_emb.data_ptr(),
            p_emb.stride(0), o_emb.data_ptr(), o_emb.stride(0), n, m, out.data_ptr(), max(ldo, 1),
            ws, wsb, st)
        if rc:
            _lib.check(rc, "kge_score_emb")
    if code == SPO:
        return out.view(n, -1)
    return out[:, :m] if out.shape[1] != m else out


def score_emb_sp_po(scorer, s_emb, p_emb, o_emb, targets, l_norm: float = 1.0, flags: int = 0, out=None,
                    pad_pitch: bool = False):
    """[n, 2m]: score_emb(.., "sp_") against `targets` followed by score_emb(.., "_po") against
    `targets`, for dense query rows (one two-sided launch on the bf16 matrix-core path).
    pad_pitch=True (no `out`): the two blocks as the [n, 2, m] view of an [n, 2, score_pitch(m)] buffer -- both on whole
    256-byte lines, the layout the direct-store kernel writes fastest (kge_score_emb_sp_po_blocks)."""
    for x in (s_emb, p_emb, o_emb, targets):
        _require_gpu(x, "embedding")
    if len({s_emb.dtype, p_emb.dtype, o_emb.dtype, targets.dtype}) != 1:
        raise TypeError("kge_amd: embeddings must share a dtype")
    s_emb, p_emb, o_emb, targets = (x if x.stride(-1) == 1 else x.contiguous() for x in (s_emb, p_emb, o_emb, targets))
    sc = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
    n, m = p_emb.shape[0], targets.shape[0]
    d, dr = s_emb.shape[1], p_emb.shape[1]
    b2 = m
    if out is None:
        if pad_pitch and m > 0:
            b2 = score_pitch(m)
            out = _empty((n, 2 * b2), s_emb.device)
        else:
            out = _empty((n, 2 * m), s_emb.device)
    key = (s_emb.dtype, sc, d, dr, float(l_norm), int(flags))
    tc = _EMB_TC.get(key)
    if tc is None:
        tc = _EMB_TC[key] = KgeTables(None, None, _dtype_code(s_emb), sc, 0, 0, d, dr, d, dr, float(l_norm),
                                      int(flags))
    with _on_device(s_emb.device):
        st = _stream_handle(s_emb.device)
        ws, wsb = _workspace(tc, n, s_emb.device, True, st)
        rc = _lib.lib().kge_score_emb_sp_po_blocks(
            ctypes.byref(tc), s_emb.data_ptr(), s_emb.stride(0), p_emb.data_ptr(), p_emb.stride(0),
            o_emb.data_ptr(), o_emb.stride(0), n, targets.data_ptr(), targets.stride(0), m, out.data_ptr(),
            out.stride(0) if n > 1 else max(2 * b2, 1), b2, ws, wsb, st)
        if rc:
            _lib.check(rc, "kge_score_emb_sp_po_blocks")
    return out.view(n, 2, b2)[:, :, :m] if b2 != m else out


def embed(t: Tables, ent_idx=None, rel_idx=None, ent_out=None, rel_out=None):
    """LookupEmbedder.embed for both tables in ONE launch: (ent[ent_idx], rel[rel_idx]); outputs
    may be preallocated (row stride free, e.g. a slice of an exchange buffer)."""
    keep = []
    tc = t.c()
    ne = nr = 0
    ei = ri = KgeIndex(None, I64, 0, 1)
    if ent_idx is not None:
        ei = _index(ent_idx, t.device, keep)
        ne = keep[-1].numel()
        if ent_out is None:
            ent_out = _empty((ne, t.ent.shape[1]), t.device, t.ent.dtype)
    if rel_idx is not None:
        ri = _index(rel_idx, t.device, keep)
        nr = keep[-1].numel()
        if rel_out is None:
            rel_out = _empty((nr, t.rel.shape[1]), t.device, t.rel.dtype)
    with _on_device(t.device):
        rc = _lib.lib().kge_embed(ctypes.byref(tc), ei, ne, ent_out.data_ptr() if ne else None,
                                  ent_out.stride(0) if ne else 0, ri, nr, rel_out.data_ptr() if nr else None,
                                  rel_out.stride(0) if nr else 0, _stream_handle(t.device))
        if rc:
            _lib.check(rc, "kge_embed")
    return ent_out, rel_out


NS_BCE_KINDS = {"bce": 0, "bce_mean": 1, "bce_self_adversarial": 2}


def ns_bce_loss(scores: torch.Tensor, kind: str, offset: float = 0.0, temperature: float = 1.0, want_grad: bool = True):
    """(loss_rows [n], grad [n, c] or None) of BCEWithLogitsKgeLoss over a negative-sampling score block [n, 1 + K]
    with the positives in column 0 (kge_ns_bce_loss): the loss value is loss_rows.sum()."""
    _require_gpu(scores, "scores")
    if scores.dim() != 2 or scores.dtype != torch.float32:
        raise ValueError("kge_amd: ns_bce_loss takes a float32 [n, 1 + K] score block")
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    n, c = scores.shape
    rows = _empty((n,), scores.device)
    grad = _empty((n, c), scores.device) if want_grad else None
    with _on_device(scores.device):
        _lib.check(_lib.lib().kge_ns_bce_loss(
            scores.data_ptr(), scores.stride(0) if n > 1 else c, n, c, NS_BCE_KINDS[kind], float(offset),
            float(temperature), rows.data_ptr(), None if grad is None else grad.data_ptr(), c,
            _stream_handle(scores.device)), "kge_ns_bce_loss")
    return rows, grad


def shard_gather(t: Tables, lo: int, ids, rel_idx, send: torch.Tensor, rel_out=None):
    """Row move 1 of the sharded exchange (kge_shard_gather): `t.ent` = this rank's rows [lo, lo + rows) of the entity
    table; send[j*n + i] = the local row of GLOBAL id ids[j][i] (any local row for an id of another rank), and
    rel_out[i] = rel[rel_idx[i]] -- one launch, the id arithmetic inside the kernel."""
    keep = []
    arr = (KgeIndex * len(ids))(*[_index(x, t.device, keep) for x in ids])
    n = _same_len(keep, "shard_gather")
    ri = _index(rel_idx, t.device, keep) if rel_out is not None else KgeIndex(None, I64, 0, 1)
    with _on_device(t.device):
        tc = t.c()
        rc = _lib.lib().kge_shard_gather(ctypes.byref(tc), int(lo), arr, len(ids), n, send.data_ptr(), send.stride(0),
                                         ri, None if rel_out is None else rel_out.data_ptr(),
                                         0 if rel_out is None else rel_out.stride(0), _stream_handle(t.device))
        if rc:
            _lib.check(rc, "kge_shard_gather")


def shard_pick(gathered: torch.Tensor, shard_rows: int, world: int, ids, rows: torch.Tensor):
    """Row move 2 (kge_shard_pick): rows[j*n + i] = the owner's copy of row (j, i) in the all-gathered blocks."""
    keep = []
    dev = gathered.device
    arr = (KgeIndex * len(ids))(*[_index(x, dev, keep) for x in ids])
    n = _same_len(keep, "shard_pick")
    with _on_device(dev):
        rc = _lib.lib().kge_shard_pick(gathered.data_ptr(), gathered.stride(0), _dtype_code(gathered),
                                       gathered.shape[1], int(shard_rows), int(world), arr, len(ids), n,
                                       rows.data_ptr(), rows.stride(0), _stream_handle(dev))
        if rc:
            _lib.check(rc, "kge_shard_pick")


def rank_counts(scores, true_scores, lbl_rowptr=None, lbl_col=None, col_offset=0, true_col=None,
                atol=1e-5, rtol=1e-4, rank=None, ties=None):
    """Accumulate (rank, ties) int64 counts of each row's true score within `scores`
    [n, c]; filtered columns given as CSR (lbl_rowptr int64 [n+1], lbl_col int64 [nnz],
    global ids minus col_offset; the entry equal to true_col[i] is kept)."""
    _require_gpu(scores, "scores")
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise ValueError("kge_amd: scores must be float32 [n, c] with unit inner stride")
    dev = scores.device
    n, c = scores.shape
    true_scores = true_scores.to(device=dev, dtype=torch.float32).contiguous()
    if rank is None:
        rank = torch.zeros(n, dtype=torch.int64, device=dev)
    if ties is None:
        ties = torch.zeros(n, dtype=torch.int64, device=dev)

    def p64(x):
        if x is None:
            return None
        x = x.to(device=dev, dtype=torch.int64).contiguous()
        keep.append(x)
        return x.data_ptr()

    keep = []
    rp, cl, tc = p64(lbl_rowptr), p64(lbl_col), p64(true_col)
    if rp is not None and not cl:  # empty filter set: the ABI still wants a non-NULL array
        cl = p64(torch.zeros(1, dtype=torch.int64, device=dev))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().kge_rank_counts(
            scores.data_ptr(), scores.stride(0) if n > 1 else max(c, 1), n, c,
            true_scores.data_ptr(), rp, cl, int(col_offset), tc, float(atol), float(rtol),
            rank.data_ptr(), ties.data_ptr(), _stream(dev)), "kge_rank_counts")
    return rank, ties


def filter_lookup(sorted_keys, starts, a, b, mult: int, begin, end):
    """begin[i], end[i] = range of key a[i] * mult + b[i] in a device-resident filter index
    (sorted unique int64 keys + starts[len(keys) + 1]); (0, 0) for unknown keys."""
    dev = begin.device
    keep = []
    ai, bi = _index(a, dev, keep), _index(b, dev, keep)
    n = _same_len(keep[:2], "filter_lookup")
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().kge_filter_lookup(
            sorted_keys.data_ptr(), sorted_keys.numel(), starts.data_ptr(), ai, bi, int(mult), n,
            begin.data_ptr(), end.data_ptr(), _stream(dev)), "kge_filter_lookup")
    return begin, end


def filter_lookup_multi(queries):
    """Several filter_lookup calls in ONE launch: queries = [(sorted_keys, starts, a, b, mult, begin, end), ...]
    (at most 4; all index vectors of the same length)."""
    if not queries:
        return
    dev = queries[0][5].device
    keep, arr = [], (_lib.KgeFilterQuery * len(queries))()
    n = None
    for q, (keys, starts, a, b, mult, begin, end) in enumerate(queries):
        k0 = len(keep)
        ai, bi = _index(a, dev, keep), _index(b, dev, keep)
        nq = _same_len(keep[k0:k0 + 2], "filter_lookup_multi")
        if n is not None and nq != n:
            raise ValueError("filter_lookup_multi: the queries differ in length")
        n = nq
        arr[q] = _lib.KgeFilterQuery(keys.data_ptr(), keys.numel(), starts.data_ptr(), ai, bi, int(mult),
                                     begin.data_ptr(), end.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().kge_filter_lookup_multi(arr, len(queries), n, _stream(dev)), "kge_filter_lookup_multi")


def rank_counts_multi(scores, true_scores, filters, col_offset, true_col, atol, rtol, rank, ties):
    """Raw + len(filters) filtered (rank, ties) counts from one scan of `scores` [n, c]:
    filters = [(begin [n], end [n], col [nnz]), ...] int64 device tensors; rank / ties are
    int64 [len(filters) + 1, n], accumulated."""
    _require_gpu(scores, "scores")
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise ValueError("kge_amd: scores must be float32 [n, c] with unit inner stride")
    dev = scores.device
    n, c = scores.shape
    K = len(filters)
    assert rank.shape == (K + 1, n) and ties.shape == (K + 1, n) and rank.is_contiguous() and ties.is_contiguous()
    arr = ctypes.c_void_p * max(K, 1)
    pb, pe, pc = (arr(*[f[j].data_ptr() for f in filters]) if K else None for j in range(3))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().kge_rank_counts_multi(
            scores.data_ptr(), scores.stride(0) if n > 1 else max(c, 1), n, c, true_scores.data_ptr(), K,
            pb, pe, pc, int(col_offset), None if true_col is None else true_col.data_ptr(), float(atol),
            float(rtol), rank.data_ptr(), ties.data_ptr(), _stream(dev)), "kge_rank_counts_multi")
    return rank, ties


_RANK_BITS = {}  # (device index, stream) -> all-zero buffer of kge_score_rank_sp_po's filter bits


def _rank_bits(need, device, stream):
    if need <= 0:
        return None, 0
    key = (device.index, stream)
    buf = _RANK_BITS.get(key)
    if buf is None or buf.numel() < need:
        # zeroed ONCE: every call clears the bits it set (include/kge_amd.h)
        buf = _RANK_BITS[key] = torch.zeros((max(need, 1 << 20),), device=device, dtype=torch.uint8)
    return buf.data_ptr(), buf.numel()


def _rank_args(n, true_sp, true_po, filters_sp, filters_po, rank_sp, ties_sp, rank_po, ties_po):
    """Checked (K, ld, six host arrays of device pointers) of the score_rank_* entries."""
    K = len(filters_sp)
    if len(filters_po) != K:
        raise ValueError("kge_amd: one filter list per direction and filter set")
    for x in (true_sp, true_po):
        _require_gpu(x, "true scores")
        if x.dtype != torch.float32 or x.numel() != n or not x.is_contiguous():
            raise ValueError("kge_amd: true scores must be contiguous float32 [n]")
    ld = None
    for x in (rank_sp, ties_sp, rank_po, ties_po):
        if x.dtype != torch.int64 or x.shape != (K + 1, n) or (n > 1 and x.stride(1) != 1):
            raise ValueError("kge_amd: rank / ties must be int64 [filters + 1, n] with unit inner stride")
        l = x.stride(0) if K > 0 and n > 0 else max(n, 1)
        if ld not in (None, l):
            raise ValueError("kge_amd: rank / ties must share one row stride")
        ld = l
    arr = ctypes.c_void_p * max(K, 1)
    lists = [(arr(*[f[j].data_ptr() for f in fl]) if K else None) for fl in (filters_sp, filters_po) for j in range(3)]
    return K, ld, lists


class RankBand:
    """Band-and-rescore state of one table slice (include/kge_amd.h: kge_rank_band): the slice's largest
    row norm (one launch, at construction and at every `refresh`: the bound must be the CURRENT table's), the waves'
    pair lists for batches of up to `n_max` rows and the status words.  Handed to score_rank_sp_po / eval_batch together
    with split-query flags, the counts are the split kernel's, from a single-pass launch plus a small launch over the
    listed pairs -- PROVIDED no pair was dropped.

    `status()` -> (pairs listed, pairs dropped) since the last `reset()` -- a host read, i.e. a wait: once per
    evaluation run, not per batch.  dropped != 0: some call's counts are incomplete (a wave's list was full); the caller
    counts those batches again without the band."""

    def __init__(self, t: "Tables", n_max: int, col_begin: int = 0, col_end=None):
        col_end = t.num_ent if col_end is None else int(col_end)
        self.col_begin, self.m, self.n_max = int(col_begin), col_end - int(col_begin), int(n_max)
        dev = t.device
        self.tmax = torch.zeros(1, dtype=torch.float32, device=dev)
        self.status_words = torch.zeros(4, dtype=torch.int32, device=dev)
        need = int(_lib.lib().kge_rank_band_list_bytes(self.n_max))
        self.list = torch.zeros(max(need, 16), dtype=torch.uint8, device=dev)  # zeroed ONCE: calls leave the lists empty
        self.c = _lib.KgeRankBand(self.tmax.data_ptr(), self.list.data_ptr(), self.list.numel(),
                                  self.status_words.data_ptr())
        self.refresh(t)

    def refresh(self, t: "Tables"):
        """The row-norm bound again (the table's values changed: a training step between two validations), on the
        current stream; the status words cleared."""
        with _on_device(t.device):
            tc = t.c(None)
            _lib.check(_lib.lib().kge_table_max_row_norm(ctypes.byref(tc), self.col_begin, self.m, self.tmax.data_ptr(),
                                                         _stream_handle(t.device)), "kge_table_max_row_norm")
        self.status_words.zero_()

    def pairs_of(self, n: int) -> int:
        """(row, column) pairs of a two-sided batch of n rows against this slice."""
        return 2 * int(n) * self.m

    def status(self):
        w = self.status_words.cpu()
        return int(w[0]) & 0xffffffff, int(w[1]) & 0xffffffff

    def reset(self):
        self.status_words.zero_()


def score_rank_sp_po(t: Tables, s, p, o, true_sp, true_po, filters_sp, filters_po, atol, rtol, rank_sp, ties_sp,
                     rank_po, ties_po, col_begin: int = 0, col_end=None, flags=None, band: "RankBand" = None) -> bool:
    """Raw + filtered (rank, ties) counts of both directions of a batch against the entity rows
    [col_begin, col_end), counted inside the scoring kernel: what score_sp_po + two rank_counts_multi calls
    give, without the [n, 2m] score matrix.  true_sp / true_po: float32 [n] scores of the triples
    themselves; filters_*: [(begin [n], end [n], col [nnz]), ...] (at most two); rank_* / ties_*: int64
    [len(filters) + 1, n] (a row stride >= n is fine), accumulated.  Counting kernels: float32 tables of every
    scorer, TransE / RotatE, bf16 ComplEx / DistMult with dim 256 / 512.  False: the library declines this
    configuration (other bf16 shapes, split queries) -- nothing was counted."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_rank_sp_po")
    col_end = t.num_ent if col_end is None else int(col_end)
    m = col_end - int(col_begin)
    K, ld, lists = _rank_args(n, true_sp, true_po, filters_sp, filters_po, rank_sp, ties_sp, rank_po, ties_po)
    with _on_device(t.device):
        tc = t.c(flags)
        st = _stream_handle(t.device)
        ws, wsb = _workspace(tc, n, t.device, True, st)
        bits, bits_bytes = _rank_bits(_lib.lib().kge_score_rank_bits_bytes(n, m, K), t.device, st)
        if band is not None and ((band.col_begin, band.m) != (int(col_begin), m) or n > band.n_max):
            raise ValueError("kge_amd: this RankBand was made for another column range or smaller batches")
        args = (ctypes.byref(tc), si, pi, oi, n, int(col_begin), m, true_sp.data_ptr(), true_po.data_ptr(), K,
                *lists, float(atol), float(rtol), rank_sp.data_ptr(), ties_sp.data_ptr(), rank_po.data_ptr(),
                ties_po.data_ptr(), ld, bits, bits_bytes, ws, wsb, st)
        rc = _lib.lib().kge_score_rank_sp_po(*args) if band is None else \
            _lib.lib().kge_score_rank_sp_po_band(*args, ctypes.byref(band.c))
        if rc == _lib.KGE_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, "kge_score_rank_sp_po")
    return True


def score_rank_emb_sp_po(scorer, s_emb, p_emb, o_emb, s_ids, o_ids, targets, col_begin, true_sp, true_po, filters_sp,
                         filters_po, atol, rtol, rank_sp, ties_sp, rank_po, ties_po, l_norm: float = 1.0) -> bool:
    """score_rank_sp_po for dense query rows against the entity rows `targets` [m, d] whose global ids start at
    col_begin (a rank's shard): s_ids / o_ids = the global ids of the true subjects / objects.  Counts of these
    m columns only (the sharded caller all-reduces).  False: declined, nothing counted."""
    for x in (s_emb, p_emb, o_emb, targets):
        _require_gpu(x, "embedding")
    if len({s_emb.dtype, p_emb.dtype, o_emb.dtype, targets.dtype}) != 1:
        raise TypeError("kge_amd: embeddings must share a dtype")
    s_emb, p_emb, o_emb, targets = (x if x.stride(-1) == 1 else x.contiguous() for x in (s_emb, p_emb, o_emb, targets))
    sc = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
    n, m = p_emb.shape[0], targets.shape[0]
    d, dr = s_emb.shape[1], p_emb.shape[1]
    dev = s_emb.device
    keep = []
    si, oi = _index(s_ids, dev, keep), _index(o_ids, dev, keep)
    if _same_len(keep[:2], "score_rank_emb_sp_po") != n:
        raise ValueError("kge_amd: one true id per query row")
    K, ld, lists = _rank_args(n, true_sp, true_po, filters_sp, filters_po, rank_sp, ties_sp, rank_po, ties_po)
    key = (s_emb.dtype, sc, d, dr, float(l_norm), 0)
    tc = _EMB_TC.get(key)
    if tc is None:
        tc = _EMB_TC[key] = KgeTables(None, None, _dtype_code(s_emb), sc, 0, 0, d, dr, d, dr, float(l_norm), 0)
    with _on_device(dev):
        st = _stream_handle(dev)
        ws, wsb = _workspace(tc, n, dev, True, st)
        bits, bits_bytes = _rank_bits(_lib.lib().kge_score_rank_bits_bytes(n, m, K), dev, st)
        rc = _lib.lib().kge_score_rank_emb_sp_po(
            ctypes.byref(tc), s_emb.data_ptr(), s_emb.stride(0), p_emb.data_ptr(), p_emb.stride(0), o_emb.data_ptr(),
            o_emb.stride(0), si, oi, n, targets.data_ptr(), targets.stride(0), int(col_begin), m, true_sp.data_ptr(),
            true_po.data_ptr(), K, *lists, float(atol), float(rtol), rank_sp.data_ptr(), ties_sp.data_ptr(),
            rank_po.data_ptr(), ties_po.data_ptr(), ld, bits, bits_bytes, ws, wsb, st)
        if rc == _lib.KGE_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, "kge_score_rank_emb_sp_po")
    return True


TIE_POLICIES = {"rounded_mean_rank": 0, "best_rank": 1, "worst_rank": 2}

_EVAL_SCRATCH = {}  # (device index, stream) -> scratch of kge_eval_batch (ranges, target list, true-score block)


def eval_batch(t: Tables, s, p, o, filters, atol, rtol, tie_policy, counts, hist, ranks_o=None, ranks_s=None,
               flags=None, band: "RankBand" = None) -> bool:
    """One evaluation batch against all entities in four launches (kge_eval_batch): filter lookup + filter bits,
    true scores, scoring + counting, bits cleared + tie policy + histograms.  filters = [((sp_keys, sp_starts,
    sp_values), (po_keys, po_starts, po_values)), ...] (at most two: the device-resident filter indexes of
    FilterIndex); counts: int64 [2, 2, len(filters) + 1, n], ALL-ZERO on entry and on return; hist: float32
    [len(filters) + 1, E] accumulated; ranks_o / ranks_s: int64 [len(filters) + 1, n] or None.  False: the library
    has no counting kernel for these tables -- nothing was counted."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "eval_batch")
    K = len(filters)
    M = K + 1
    if counts.dtype != torch.int64 or tuple(counts.shape) != (2, 2, M, n) or not counts.is_contiguous():
        raise ValueError("kge_amd: counts must be contiguous int64 [2, 2, filters + 1, n]")
    if hist.dtype != torch.float32 or hist.shape[0] != M or hist.shape[1] < t.num_ent or hist.stride(1) != 1:
        raise ValueError("kge_amd: hist must be float32 [filters + 1, >= num_entities]")
    for r in (ranks_o, ranks_s):
        if r is not None and (r.dtype != torch.int64 or tuple(r.shape) != (M, n) or not r.is_contiguous()):
            raise ValueError("kge_amd: ranks must be contiguous int64 [filters + 1, n]")
    arr = (_lib.KgeEvalFilter * max(K, 1))()
    for k, (sp, po) in enumerate(filters):
        arr[k] = _lib.KgeEvalFilter(sp[0].data_ptr(), sp[0].numel(), sp[1].data_ptr(), sp[2].data_ptr(),
                                    po[0].data_ptr(), po[0].numel(), po[1].data_ptr(), po[2].data_ptr())
    with _on_device(t.device):
        tc = t.c(flags)
        st = _stream_handle(t.device)
        ws, wsb = _workspace(tc, n, t.device, True, st)
        bits, bits_bytes = _rank_bits(_lib.lib().kge_score_rank_bits_bytes(n, t.num_ent, K), t.device, st)
        need = _lib.lib().kge_eval_batch_scratch_bytes(ctypes.byref(tc), n, K)
        key = (t.device.index, st)
        buf = _EVAL_SCRATCH.get(key)
        if buf is None or buf.numel() < need:
            buf = _EVAL_SCRATCH[key] = torch.empty((max(need, 1 << 20),), device=t.device, dtype=torch.uint8)
        if band is not None and ((band.col_begin, band.m) != (0, t.num_ent) or n > band.n_max):
            raise ValueError("kge_amd: eval_batch scores all entities; this RankBand was made for another range or smaller batches")
        args = (ctypes.byref(tc), si, pi, oi, n, K, arr, float(atol), float(rtol), TIE_POLICIES.get(tie_policy, tie_policy)
                if isinstance(tie_policy, str) else int(tie_policy), counts.data_ptr(), hist.data_ptr(), hist.stride(0),
                None if ranks_o is None else ranks_o.data_ptr(), None if ranks_s is None else ranks_s.data_ptr(),
                bits, bits_bytes, buf.data_ptr(), buf.numel(), ws, wsb, st)
        rc = _lib.lib().kge_eval_batch(*args) if band is None else \
            _lib.lib().kge_eval_batch_band(*args, ctypes.byref(band.c))
        if rc == _lib.KGE_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, "kge_eval_batch")
    return True


def rank_hist(rank, ties, tie_handling: str, hist, ranks_out=None):
    """hist[m, rank_of(rank[m, i], ties[m, i])] += 1 (float32 [M, E]) for int64 [M, n] counts."""
    if tie_handling not in TIE_POLICIES:
        raise NotImplementedError(tie_handling)
    M, n = rank.shape
    with torch.cuda.device(rank.device):
        _lib.check(_lib.lib().kge_rank_hist(
            rank.data_ptr(), ties.data_ptr(), M, n, TIE_POLICIES[tie_handling], hist.data_ptr(), hist.stride(0),
            hist.shape[1], None if ranks_out is None else ranks_out.data_ptr(), _stream(rank.device)),
            "kge_rank_hist")
    return hist


# ---- backward twins (csrc/bwd.hip) ---------------------------------------------------------
def _f32c(x, dev):
    return x.to(device=dev, dtype=torch.float32).contiguous()


def score_spo_bwd(t: Tables, s, p, o, gout, scores=None):
    """Gradients of sum_i gout[i]*score_spo_i w.r.t. the gathered rows:
    (g_s [n,d], g_p [n,d_r], g_o [n,d])."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_spo_bwd")
    d, dr = t.ent.shape[1], t.rel.shape[1]
    gout = _f32c(gout, t.device)
    sc = None if scores is None else _f32c(scores, t.device)
    g_s, g_p, g_o = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((n, d), t.device)
    with torch.cuda.device(t.device):
        tc = t.c()
        _lib.check(_lib.lib().kge_score_spo_bwd(
            ctypes.byref(tc), si, pi, oi, n, gout.data_ptr(), None if sc is None else sc.data_ptr(),
            g_s.data_ptr(), g_p.data_ptr(), g_o.data_ptr(), _stream(t.device)), "kge_score_spo_bwd")
    return g_s, g_p, g_o


def score_spo_bwd_accum(t: Tables, s, p, o, gout, scores, grad_ent, grad_rel):
    """score_spo backward accumulated straight into the dense table gradients `grad_ent` [E, d] and
    `grad_rel` [R, d_r] (f32, modified in place); returns False if the kernel does not take this
    shape (the caller then uses score_spo_bwd + index_add)."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "score_spo_bwd_accum")
    gout = _f32c(gout, t.device)
    sc = None if scores is None else _f32c(scores, t.device)
    with _on_device(t.device):
        tc = t.c()
        rc = _lib.lib().kge_score_spo_bwd_accum(
            ctypes.byref(tc), si, pi, oi, n, gout.data_ptr(), None if sc is None else sc.data_ptr(),
            grad_ent.data_ptr(), grad_ent.stride(0), grad_rel.data_ptr(), grad_rel.stride(0),
            _stream_handle(t.device))
    if rc == _lib.KGE_ERR_UNSUPPORTED:
        return False
    if rc:
        _lib.check(rc, "kge_score_spo_bwd_accum")
    return True


def score_pairs_bwd(t: Tables, direction: str, a, p, targets, gout, scores=None):
    """Backward of score_sp (direction 'sp', a = s) / score_po ('po', a = o):
    (g_a [n,d], g_p [n,d_r], g_targets [m,d])."""
    keep = []
    ai, pi = _index(a, t.device, keep), _index(p, t.device, keep)
    n = _same_len(keep[:2], "score_pairs_bwd")
    ti = _index(targets, t.device, keep)
    m = t.num_ent if targets is None else keep[-1].numel()
    d, dr = t.ent.shape[1], t.rel.shape[1]
    gout = gout.to(device=t.device, dtype=torch.float32)
    if gout.dim() != 2 or gout.stride(1) != 1:
        gout = gout.contiguous().view(n, m)
    sc = None
    if scores is not None:
        sc = scores if scores.stride(-1) == 1 else scores.contiguous()
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((m, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        need = _lib.lib().kge_score_bwd_workspace_bytes(ctypes.byref(tc), n, m)
        ws, wsb = (None, 0)
        if need > 0:  # bf16 tables: scratch for the bf16 copies of gout and of the query matrix
            key = (t.device.index, st, "bwd")
            buf = _WORKSPACES.get(key)
            if buf is None or buf.numel() < need:
                buf = _WORKSPACES[key] = _empty((need,), t.device, torch.uint8)
            ws, wsb = buf.data_ptr(), buf.numel()
        _lib.check(_lib.lib().kge_score_pairs_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, ai, pi, n, ti, m, gout.data_ptr(),
            gout.stride(0) if n > 1 else max(m, 1), None if sc is None else sc.data_ptr(),
            0 if sc is None else (sc.stride(0) if n > 1 else max(m, 1)), g_a.data_ptr(),
            g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_score_pairs_bwd")
    return g_a, g_p, g_t


def _ce_workspace(tc, n, device, st):
    need = _lib.lib().kge_ce_workspace_bytes(ctypes.byref(tc), n)
    if need <= 0:
        raise RuntimeError("kge_ce_fwd/kge_ce_bwd: bf16 ComplEx/DistMult tables with dim in {128, 256, 512} only")
    key = (device.index, st, "ce")
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < need:
        buf = _WORKSPACES[key] = torch.zeros((need,), device=device, dtype=torch.uint8)  # zeroed once, see _workspace
    return buf.data_ptr(), buf.numel()


def ce_supported(t: Tables) -> bool:
    """Can the fused 1vsAll loss run on these tables (kge_ce_workspace_bytes > 0)?"""
    if not t.ent.is_cuda:
        return False
    return _lib.lib().kge_ce_workspace_bytes(ctypes.byref(t.c()), 1) > 0


def ce_fwd(t: Tables, direction: str, a, p, label):
    """Fused score_sp ('sp': a = s, label = o) / score_po ('po': a = o, label = s) + cross entropy
    against all entities: (loss_rows [n], lse [n]); sum(loss_rows) = the reference's
    KLDivWithSoftmaxKgeLoss (kge/util/loss.py:192-207) on the [n, E] scores, never written here."""
    keep = []
    ai, pi, li = (_index(x, t.device, keep) for x in (a, p, label))
    n = _same_len(keep[:3], "ce_fwd")
    loss_rows, lse = _empty((n,), t.device), _empty((n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_fwd(ctypes.byref(tc), SP_ if direction == "sp" else PO_, ai, pi, li, n,
                                         loss_rows.data_ptr(), lse.data_ptr(), ws, wsb, st), "kge_ce_fwd")
    return loss_rows, lse


def ce_bwd(t: Tables, direction: str, a, p, label, lse, g_rows=None, g_scalar: float = 1.0):
    """Backward of ce_fwd: gradients of sum_i g_i * loss_rows[i] w.r.t. the gathered query rows and
    all entity rows: (g_a [n, d], g_p [n, d], g_entities [E, d])."""
    keep = []
    ai, pi, li = (_index(x, t.device, keep) for x in (a, p, label))
    n = _same_len(keep[:3], "ce_bwd")
    d, dr = t.ent.shape[1], t.rel.shape[1]
    lse = _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, ai, pi, li, n, lse.data_ptr(),
            None if gr is None else gr.data_ptr(), float(g_scalar), g_a.data_ptr(), g_p.data_ptr(),
            g_t.data_ptr(), ws, wsb, st), "kge_ce_bwd")
    return g_a, g_p, g_t


def ce_emb_fwd(t: Tables, direction: str, a_rows, p_rows, label):
    """ce_fwd with dense bf16 query rows against ALL rows of t.ent (the per-shard step of entity-sharded
    1vsAll training): (loss_rows [n] -- NaN where `label` (local row ids) is outside [0, num_ent) --, lse [n])."""
    keep = []
    n = a_rows.shape[0]
    li = _index(label, t.device, keep)
    loss_rows, lse = _empty((n,), t.device), _empty((n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_emb_fwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0),
            p_rows.data_ptr(), p_rows.stride(0), li, n, loss_rows.data_ptr(), lse.data_ptr(), ws, wsb, st),
            "kge_ce_emb_fwd")
    return loss_rows, lse


def ce_emb_bwd(t: Tables, direction: str, a_rows, p_rows, label, lse, g_rows=None, g_scalar: float = 1.0):
    """Backward of ce_emb_fwd given the (global) lse: (g_a [n, d], g_p [n, d_r] -- this shard's part of the
    query-row gradients --, g_targets [num_ent, d])."""
    keep = []
    n = a_rows.shape[0]
    li = _index(label, t.device, keep)
    lse = _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    d, dr = t.ent.shape[1], t.rel.shape[1]
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_emb_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0),
            p_rows.data_ptr(), p_rows.stride(0), li, n, lse.data_ptr(), None if gr is None else gr.data_ptr(),
            float(g_scalar), g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_ce_emb_bwd")
    return g_a, g_p, g_t


FLAG_CE_KEEP_QUERIES = 1 << 16  # include/kge_amd.h: the backward starts from the forward's query fragments
_CE2_GENERATION = {}  # (device index, stream) -> how often the two-sided loss workspace was handed to a call


def ce2_generation(device, stream=None) -> int:
    """How often the workspace of the kge_ce_sp_po_* entries (of this device and stream) has been handed to a call.  A
    forward that was the LAST call on it may be followed by a backward under FLAG_CE_KEEP_QUERIES (kge_amd.model's
    _FusedCE2Sum compares the count it saw after its forward with the count at its backward)."""
    with _on_device(device):
        st = _stream_handle(device) if stream is None else stream
    return _CE2_GENERATION.get((device.index, st), 0)


def _ce2_workspace(tc, n, device, st):
    need = _lib.lib().kge_ce_sp_po_workspace_bytes(ctypes.byref(tc), n)
    if need <= 0:
        raise RuntimeError("kge_ce_sp_po_*: bf16 ComplEx/DistMult tables with dim in {128, 256, 512} only")
    key = (device.index, st, "ce2")
    _CE2_GENERATION[(device.index, st)] = _CE2_GENERATION.get((device.index, st), 0) + 1
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < need:
        buf = _WORKSPACES[key] = torch.zeros((need,), device=device, dtype=torch.uint8)
    return buf.data_ptr(), buf.numel()


def ce_sp_po_fwd(t: Tables, s, p, o):
    """Both directions of a 1vsAll batch in one pass: (loss_rows [2n], lse [2n]); rows [0, n) =
    cross entropy of score_sp(s, p) against o, rows [n, 2n) = of score_po(p, o) against s."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "ce_sp_po_fwd")
    loss_rows, lse = _empty((2 * n,), t.device), _empty((2 * n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce2_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_sp_po_fwd(ctypes.byref(tc), si, pi, oi, n, loss_rows.data_ptr(),
                                               lse.data_ptr(), ws, wsb, st), "kge_ce_sp_po_fwd")
    return loss_rows, lse


def ce_sp_po_bwd(t: Tables, s, p, o, lse, g_rows=None, g_scalar: float = 1.0):
    """Backward of ce_sp_po_fwd: (g_a [2n, d]: rows [0, n) for the s rows, [n, 2n) for the o rows;
    g_p [2n, d]; g_entities [E, d])."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "ce_sp_po_bwd")
    d, dr = t.ent.shape[1], t.rel.shape[1]
    lse = _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    g_a, g_p, g_t = _empty((2 * n, d), t.device), _empty((2 * n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce2_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_sp_po_bwd(
            ctypes.byref(tc), si, pi, oi, n, lse.data_ptr(), None if gr is None else gr.data_ptr(),
            float(g_scalar), g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_ce_sp_po_bwd")
    return g_a, g_p, g_t


def ce_sp_po_bwd_accum(t: Tables, s, p, o, lse, g_rows=None, g_scalar: float = 1.0):
    """Backward of ce_sp_po_fwd with the row scatter-adds done by the library: the complete
    (grad_entities [E, d], grad_relations [R, d]) of the two tables."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "ce_sp_po_bwd_accum")
    lse = _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    ge, grel = _empty(tuple(t.ent.shape), t.device), _empty(tuple(t.rel.shape), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce2_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_sp_po_bwd_accum(
            ctypes.byref(tc), si, pi, oi, n, lse.data_ptr(), None if gr is None else gr.data_ptr(),
            float(g_scalar), ge.data_ptr(), grel.data_ptr(), ws, wsb, st), "kge_ce_sp_po_bwd_accum")
    return ge, grel


def _dev_scalar(x, device, what):
    """A float32 device scalar (0-d or one element) as the kernels read it, or None."""
    if x is None:
        return None
    if not (torch.is_tensor(x) and x.numel() == 1 and x.dtype == torch.float32 and x.device == device):
        raise ValueError(f"{what}: a float32 tensor of one element on {device} expected")
    return x


def ce_sp_po_fwd_sum(t: Tables, s, p, o, scale=None, keep_queries: bool = False):
    """ce_sp_po_fwd and, from the same launches, the batch loss as a device scalar:
    (loss_sum 0-d = scale * sum(loss_rows), loss_rows [2n], lse [2n]).  `scale`: None, a Python float, or a float32
    device scalar (read by the kernel at run time: a captured step follows it).  keep_queries: also leave the gradient
    products' query matrix in the workspace (FLAG_CE_KEEP_QUERIES) for a backward called with keep_queries=True -- which
    the caller may only do if no other call used the workspace in between (ce2_generation)."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "ce_sp_po_fwd_sum")
    sd = _dev_scalar(scale, t.device, "ce_sp_po_fwd_sum(scale)") if torch.is_tensor(scale) else None
    sh = 1.0 if scale is None or sd is not None else float(scale)
    loss_rows, lse, total = _empty((2 * n,), t.device), _empty((2 * n,), t.device), _empty((), t.device)
    with _on_device(t.device):
        tc = t.c(int(t.flags) | FLAG_CE_KEEP_QUERIES) if keep_queries else t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce2_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_sp_po_fwd_sum(
            ctypes.byref(tc), si, pi, oi, n, loss_rows.data_ptr(), lse.data_ptr(),
            None if sd is None else sd.data_ptr(), sh, total.data_ptr(), ws, wsb, st), "kge_ce_sp_po_fwd_sum")
    return total, loss_rows, lse


def ce_sp_po_bwd_accum_sum(t: Tables, s, p, o, lse, g=None, scale=None, keep_queries: bool = False):
    """Backward of ce_sp_po_fwd_sum: the complete (grad_entities, grad_relations) for the upstream gradient `g` of
    loss_sum (a float32 device scalar, or None = 1) and the forward's `scale`; neither is read by the host.
    keep_queries: the forward was called with keep_queries=True, on the same tables / indexes, and was the last call on
    the workspace: its query fragments are used instead of building them again."""
    keep = []
    si, pi, oi = (_index(x, t.device, keep) for x in (s, p, o))
    n = _same_len(keep[:3], "ce_sp_po_bwd_accum_sum")
    lse = _f32c(lse, t.device)
    gd = _dev_scalar(g, t.device, "ce_sp_po_bwd_accum_sum(g)")
    sd = _dev_scalar(scale, t.device, "ce_sp_po_bwd_accum_sum(scale)") if torch.is_tensor(scale) else None
    sh = 1.0 if scale is None or sd is not None else float(scale)
    ge, grel = _empty(tuple(t.ent.shape), t.device), _empty(tuple(t.rel.shape), t.device)
    with _on_device(t.device):
        tc = t.c(int(t.flags) | FLAG_CE_KEEP_QUERIES) if keep_queries else t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce2_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_ce_sp_po_bwd_accum_sum(
            ctypes.byref(tc), si, pi, oi, n, lse.data_ptr(), None if gd is None else gd.data_ptr(),
            None if sd is None else sd.data_ptr(), sh, ge.data_ptr(), grel.data_ptr(), ws, wsb, st),
            "kge_ce_sp_po_bwd_accum_sum")
    return ge, grel


def multilabel2_bwd_accum(t: Tables, kind: str, offset: float, sp, po, g=None, scale: float = 1.0):
    """Backward of BOTH query types of a KvsAll batch with complete table gradients (kge_multilabel2_bwd_accum):
    `sp` = (s, p, lbl_rowptr, lbl_col, lse, g_rows) of the sp_ queries, `po` = (o, p, lbl_rowptr, lbl_col, lse, g_rows) of
    the _po queries (lse: kl_fwd's, None for kind "bce"; g_rows: upstream gradients of the loss rows, or None: every
    row's gradient is `scale` x the float32 device scalar `g` -- the backward of scale * (sum of all loss rows));
    returns (grad_entities [E, d], grad_relations [R, d_r]).  Either side may have no rows."""
    keep, sides, n = [], [], []
    gd = _dev_scalar(g, t.device, "multilabel2_bwd_accum(g)")
    for name, (a, p, rowptr, col, lse, g_rows) in (("sp", sp), ("po", po)):
        k0 = len(keep)
        ai, pi = (_index(x, t.device, keep) for x in (a, p))
        nk = _same_len(keep[k0:k0 + 2], "multilabel2_bwd_accum")
        rp, cl = _csr64(rowptr, col, t.device)
        ls = None if lse is None else _f32c(lse, t.device)
        gr = None if g_rows is None else _f32c(g_rows, t.device)
        if kind == "kl" and nk > 0 and ls is None:
            raise ValueError("multilabel2_bwd_accum: the kl loss needs kl_fwd's lse")
        keep += [rp, cl, ls, gr]
        sides.append(_lib.KgeLabelQueries(ai, pi, nk, rp.data_ptr(), cl.data_ptr(), None if ls is None else ls.data_ptr(),
                                          None if gr is None else gr.data_ptr(), float(scale) if gr is None else 1.0,
                                          None if gd is None or gr is not None else gd.data_ptr()))
        n.append(nk)
    ge, grel = _empty(tuple(t.ent.shape), t.device), _empty(tuple(t.rel.shape), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        need = _lib.lib().kge_multilabel2_workspace_bytes(ctypes.byref(tc), n[0], n[1])
        if need <= 0:
            raise RuntimeError("kge_multilabel2_bwd_accum: bf16 ComplEx/DistMult tables with dim in {128, 256, 512} only")
        key = (t.device.index, st, "ml2")
        buf = _WORKSPACES.get(key)
        if buf is None or buf.numel() < need:
            buf = _WORKSPACES[key] = torch.zeros((need,), device=t.device, dtype=torch.uint8)  # (zeroed once)
        _lib.check(_lib.lib().kge_multilabel2_bwd_accum(
            ctypes.byref(tc), _lib.LOSS_KL if kind == "kl" else _lib.LOSS_BCE, float(offset), ctypes.byref(sides[0]),
            ctypes.byref(sides[1]), ge.data_ptr(), grel.data_ptr(), buf.data_ptr(), buf.numel(), st),
            "kge_multilabel2_bwd_accum")
    return ge, grel


def _csr64(rowptr, col, dev):
    rp = rowptr.to(device=dev, dtype=torch.int64).contiguous()
    cl = col.to(device=dev, dtype=torch.int64).contiguous()
    if cl.numel() == 0:
        cl = torch.zeros(1, dtype=torch.int64, device=dev)
    return rp, cl


def kl_fwd(t: Tables, direction: str, a, p, lbl_rowptr, lbl_col, label_weight=None):
    """KvsAll: fused score_sp / score_po + KL divergence from the rows' normalised multi-hot
    labels (int64 CSR): (loss_rows [n], lse [n]); kge/util/loss.py:208-213 without smoothing.
    With `label_weight` [n] (label smoothing; include/kge_amd.h kge_kl_weighted_fwd):
    loss_rows[i] = lse[i] - label_weight[i] * (sum of row i's label scores)."""
    keep = []
    ai, pi = (_index(x, t.device, keep) for x in (a, p))
    n = _same_len(keep[:2], "kl_fwd")
    rp, cl = _csr64(lbl_rowptr, lbl_col, t.device)
    loss_rows, lse = _empty((n,), t.device), _empty((n,), t.device)
    lw = None if label_weight is None else _f32c(label_weight, t.device)
    if lw is not None and lw.numel() != n:
        raise ValueError(f"kl_fwd: label_weight has {lw.numel()} entries for {n} rows")
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        dirc = SP_ if direction == "sp" else PO_
        if lw is None:
            _lib.check(_lib.lib().kge_kl_fwd(ctypes.byref(tc), dirc, ai, pi, n, rp.data_ptr(), cl.data_ptr(),
                                             loss_rows.data_ptr(), lse.data_ptr(), ws, wsb, st), "kge_kl_fwd")
        else:
            _lib.check(_lib.lib().kge_kl_weighted_fwd(
                ctypes.byref(tc), dirc, ai, pi, n, rp.data_ptr(), cl.data_ptr(), lw.data_ptr(),
                loss_rows.data_ptr(), lse.data_ptr(), ws, wsb, st), "kge_kl_weighted_fwd")
    return loss_rows, lse


def kl_bwd(t: Tables, direction: str, a, p, lbl_rowptr, lbl_col, lse, g_rows=None, g_scalar: float = 1.0,
           label_weight=None, label_bias=None):
    """Backward of kl_fwd: (g_a [n, d], g_p [n, d], g_entities [E, d])."""
    keep = []
    ai, pi = (_index(x, t.device, keep) for x in (a, p))
    n = _same_len(keep[:2], "kl_bwd")
    rp, cl = _csr64(lbl_rowptr, lbl_col, t.device)
    d, dr = t.ent.shape[1], t.rel.shape[1]
    lse = _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    lw = None if label_weight is None else _f32c(label_weight, t.device)
    if lw is not None and lw.numel() != n:
        raise ValueError(f"kl_bwd: label_weight has {lw.numel()} entries for {n} rows")
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        dirc = SP_ if direction == "sp" else PO_
        grp = None if gr is None else gr.data_ptr()
        if lw is None:
            _lib.check(_lib.lib().kge_kl_bwd(
                ctypes.byref(tc), dirc, ai, pi, n, rp.data_ptr(), cl.data_ptr(), lse.data_ptr(), grp,
                float(g_scalar), g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_kl_bwd")
        else:
            lb = None if label_bias is None else _f32c(label_bias, t.device)
            if lb is not None and lb.numel() != n:
                raise ValueError(f"kl_bwd: label_bias has {lb.numel()} entries for {n} rows")
            _lib.check(_lib.lib().kge_kl_weighted_bwd(
                ctypes.byref(tc), dirc, ai, pi, n, rp.data_ptr(), cl.data_ptr(), lw.data_ptr(),
                None if lb is None else lb.data_ptr(), lse.data_ptr(),
                grp, float(g_scalar), g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st),
                "kge_kl_weighted_bwd")
    return g_a, g_p, g_t


# ---- KvsAll losses on dense query rows against ONE shard of the entity table (kge_amd.sharded) -----------------------
def _emb_loss_args(t, a_rows, p_rows, lbl_rowptr, lbl_col):
    n = a_rows.shape[0]
    if a_rows.dim() != 2 or p_rows.dim() != 2 or p_rows.shape[0] != n or a_rows.stride(1) != 1 or p_rows.stride(1) != 1:
        raise ValueError("kge_amd: dense query rows must be [n, dim] / [n, rel_dim] with unit inner stride")
    if a_rows.dtype != t.ent.dtype or p_rows.dtype != t.rel.dtype:
        raise ValueError("kge_amd: dense query rows must have the tables' dtype")
    rp, cl = _csr64(lbl_rowptr, lbl_col, t.device)
    if rp.numel() != n + 1:
        raise ValueError(f"kge_amd: label rowptr has {rp.numel()} entries for {n} rows")
    return n, rp, cl


def kl_emb_fwd(t: Tables, direction: str, a_rows, p_rows, lbl_rowptr, lbl_col, col_lo: int, label_weight):
    """(loss_rows [n], lse [n]) of kge_kl_weighted_emb_fwd: this shard's log-sum-exp and lse - w_i * (sum of the
    scores of row i's labels INSIDE the shard [col_lo, col_lo + num_ent); lbl_col holds global ids)."""
    n, rp, cl = _emb_loss_args(t, a_rows, p_rows, lbl_rowptr, lbl_col)
    lw = _f32c(label_weight, t.device)
    loss_rows, lse = _empty((n,), t.device), _empty((n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_kl_weighted_emb_fwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0), p_rows.data_ptr(),
            p_rows.stride(0), n, rp.data_ptr(), cl.data_ptr(), int(col_lo), lw.data_ptr(), loss_rows.data_ptr(),
            lse.data_ptr(), ws, wsb, st), "kge_kl_weighted_emb_fwd")
    return loss_rows, lse


def kl_emb_bwd(t: Tables, direction: str, a_rows, p_rows, lbl_rowptr, lbl_col, col_lo: int, label_weight, lse,
               g_rows=None, g_scalar: float = 1.0, label_bias=None):
    """Backward of kl_emb_fwd with the GLOBAL log-sum-exp: (g_a [n, d], g_p [n, d_r], g_shard [num_ent, d])."""
    n, rp, cl = _emb_loss_args(t, a_rows, p_rows, lbl_rowptr, lbl_col)
    lw, lse = _f32c(label_weight, t.device), _f32c(lse, t.device)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    lb = None if label_bias is None else _f32c(label_bias, t.device)
    d, dr = t.ent.shape[1], t.rel.shape[1]
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_kl_weighted_emb_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0), p_rows.data_ptr(),
            p_rows.stride(0), n, rp.data_ptr(), cl.data_ptr(), int(col_lo), lw.data_ptr(),
            None if lb is None else lb.data_ptr(), lse.data_ptr(), None if gr is None else gr.data_ptr(), float(g_scalar),
            g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_kl_weighted_emb_bwd")
    return g_a, g_p, g_t


def bce_emb_fwd(t: Tables, direction: str, a_rows, p_rows, lbl_rowptr, lbl_col, col_lo: int, offset: float = 0.0):
    """loss_rows [n]: this shard's part of sum_j BCEWithLogits(score(i, j) + offset, y_ij) (kge_bce_emb_fwd)."""
    n, rp, cl = _emb_loss_args(t, a_rows, p_rows, lbl_rowptr, lbl_col)
    loss_rows = _empty((n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_bce_emb_fwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0), p_rows.data_ptr(),
            p_rows.stride(0), n, rp.data_ptr(), cl.data_ptr(), int(col_lo), float(offset), loss_rows.data_ptr(), ws, wsb,
            st), "kge_bce_emb_fwd")
    return loss_rows


def bce_emb_bwd(t: Tables, direction: str, a_rows, p_rows, lbl_rowptr, lbl_col, col_lo: int, offset: float = 0.0,
                g_rows=None, g_scalar: float = 1.0):
    n, rp, cl = _emb_loss_args(t, a_rows, p_rows, lbl_rowptr, lbl_col)
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    d, dr = t.ent.shape[1], t.rel.shape[1]
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_bce_emb_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, a_rows.data_ptr(), a_rows.stride(0), p_rows.data_ptr(),
            p_rows.stride(0), n, rp.data_ptr(), cl.data_ptr(), int(col_lo), float(offset),
            None if gr is None else gr.data_ptr(), float(g_scalar), g_a.data_ptr(), g_p.data_ptr(), g_t.data_ptr(), ws,
            wsb, st), "kge_bce_emb_bwd")
    return g_a, g_p, g_t


def bce_fwd(t: Tables, direction: str, a, p, lbl_rowptr, lbl_col, offset: float = 0.0):
    """Fused score_sp / score_po + BCEWithLogits (summed over all entities) against the rows'
    multi-hot labels (int64 CSR): loss_rows [n]; kge/util/loss.py:137-159, bce_type None."""
    keep = []
    ai, pi = (_index(x, t.device, keep) for x in (a, p))
    n = _same_len(keep[:2], "bce_fwd")
    rp, cl = _csr64(lbl_rowptr, lbl_col, t.device)
    loss_rows = _empty((n,), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_bce_fwd(ctypes.byref(tc), SP_ if direction == "sp" else PO_, ai, pi, n,
                                          rp.data_ptr(), cl.data_ptr(), float(offset), loss_rows.data_ptr(),
                                          ws, wsb, st), "kge_bce_fwd")
    return loss_rows


def bce_bwd(t: Tables, direction: str, a, p, lbl_rowptr, lbl_col, offset: float = 0.0, g_rows=None,
            g_scalar: float = 1.0):
    """Backward of bce_fwd: (g_a [n, d], g_p [n, d], g_entities [E, d])."""
    keep = []
    ai, pi = (_index(x, t.device, keep) for x in (a, p))
    n = _same_len(keep[:2], "bce_bwd")
    rp, cl = _csr64(lbl_rowptr, lbl_col, t.device)
    d, dr = t.ent.shape[1], t.rel.shape[1]
    gr = None if g_rows is None else _f32c(g_rows, t.device)
    g_a, g_p, g_t = _empty((n, d), t.device), _empty((n, dr), t.device), _empty((t.num_ent, d), t.device)
    with _on_device(t.device):
        tc = t.c()
        st = _stream_handle(t.device)
        ws, wsb = _ce_workspace(tc, max(n, 1), t.device, st)
        _lib.check(_lib.lib().kge_bce_bwd(
            ctypes.byref(tc), SP_ if direction == "sp" else PO_, ai, pi, n, rp.data_ptr(), cl.data_ptr(),
            float(offset), None if gr is None else gr.data_ptr(), float(g_scalar), g_a.data_ptr(),
            g_p.data_ptr(), g_t.data_ptr(), ws, wsb, st), "kge_bce_bwd")
    return g_a, g_p, g_t


def score_emb_bwd(scorer, s_emb, p_emb, o_emb, combine: str, l_norm, gout, scores=None):
    """Backward of score_emb: gradients w.r.t. (s_emb, p_emb, o_emb)."""
    code = {"spo": SPO, "sp_": SP_, "_po": PO_}[combine]
    dev = s_emb.device
    s_emb, p_emb, o_emb = (_f32c(x.detach(), dev) for x in (s_emb, p_emb, o_emb))
    sc_code = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
    n, d, dr = p_emb.shape[0], s_emb.shape[1], p_emb.shape[1]
    m = 0 if code == SPO else (o_emb.shape[0] if code == SP_ else s_emb.shape[0])
    gout = _f32c(gout, dev).view(n, -1) if code != SPO else _f32c(gout, dev).view(-1)
    sc = None if scores is None else _f32c(scores, dev)
    g_s, g_p, g_o = _empty(tuple(s_emb.shape), dev), _empty(tuple(p_emb.shape), dev), _empty(tuple(o_emb.shape), dev)
    tc = KgeTables(None, None, F32, sc_code, 0, 0, d, dr, d, dr, float(l_norm), 0)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().kge_score_emb_bwd(
            ctypes.byref(tc), code, s_emb.data_ptr(), s_emb.stride(0), p_emb.data_ptr(),
            p_emb.stride(0), o_emb.data_ptr(), o_emb.stride(0), n, m, gout.data_ptr(), max(m, 1),
            None if sc is None else sc.data_ptr(), max(m, 1), g_s.data_ptr(), g_p.data_ptr(),
            g_o.data_ptr(), _stream(dev)), "kge_score_emb_bwd")
    return g_s, g_p, g_o
