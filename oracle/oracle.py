"""numpy front end of the CPU oracle (oracle/kge_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of kge_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Function names mirror the reference API they restate:
  score_spo / score_sp / score_po / score_sp_po  -> KgeModel.score_*  (kge/model/kge_model.py:663-789)
  score_neg                                        -> BatchNegativeSample.score, impl "triple" (kge/util/sampler.py:291-306)
  rank_counts                                      -> EntityRankingJob._filter_and_rank (kge/job/eval_entity_ranking.py:533-596)
  evaluate_ranks                                   -> EntityRankingJob._evaluate inner loop (eval_entity_ranking.py:163-333)
  labels_csr                                       -> _collate / get_sp_po_coords_from_spo_batch (eval_entity_ranking.py:77-101, job/util.py:6-29)
  compute_metrics                                  -> hist_all + _compute_metrics (eval_entity_ranking.py:620-649,665-687)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkge_oracle.so")

COMPLEX, DISTMULT, TRANSE, ROTATE = 0, 1, 2, 3
SCORERS = {"complex": COMPLEX, "distmult": DISTMULT, "transe": TRANSE, "rotate": ROTATE}
F32, BF16 = 0, 1
SP, PO = 1, 2


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kge_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkge_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Tables(ctypes.Structure):
    _fields_ = [
        ("ent", ctypes.c_void_p), ("rel", ctypes.c_void_p),
        ("dtype", ctypes.c_int32), ("scorer", ctypes.c_int32),
        ("num_ent", ctypes.c_int64), ("num_rel", ctypes.c_int64),
        ("dim", ctypes.c_int64), ("rel_dim", ctypes.c_int64),
        ("ent_ld", ctypes.c_int64), ("rel_ld", ctypes.c_int64),
        ("l_norm", ctypes.c_float), ("reserved", ctypes.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ko_bf16_to_f32.restype = ctypes.c_float
        _lib.ko_bf16_to_f32.argtypes = [ctypes.c_uint16]
        _lib.ko_f32_to_bf16.restype = ctypes.c_uint16
        _lib.ko_f32_to_bf16.argtypes = [ctypes.c_float]
    return _lib


# ---- bf16 helpers (numpy has no bf16: carried as uint16) ------------------------
def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 bit patterns (uint16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + (0x7FFF + ((u >> 16) & 1))) >> 16
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def bf16_to_f32(h: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


class Tables:
    """Entity/relation lookup tables (LookupEmbedder weights) for the oracle."""

    def __init__(self, scorer, ent, rel, l_norm=1.0, split_query=False):
        """split_query: bf16 ComplEx / DistMult sp_ / _po scores with q = q_hi + q_lo (KGE_FLAG_SPLIT_QUERY)."""
        self.scorer = SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
        ent = np.ascontiguousarray(ent)
        rel = np.ascontiguousarray(rel)
        if ent.dtype == np.uint16:
            assert rel.dtype == np.uint16
            self.dtype = BF16
        else:
            ent = ent.astype(np.float32, copy=False)
            rel = rel.astype(np.float32, copy=False)
            self.dtype = F32
        self.ent, self.rel = ent, rel
        self.l_norm = float(l_norm)
        self.c = _Tables(ent.ctypes.data, rel.ctypes.data, self.dtype, self.scorer,
                         ent.shape[0], rel.shape[0], ent.shape[1], rel.shape[1],
                         ent.shape[1], rel.shape[1], self.l_norm, 1 if split_query else 0)

    @property
    def num_ent(self):
        return self.ent.shape[0]

    @property
    def dim(self):
        return self.ent.shape[1]


def _idx(a):
    """-> (keepalive array, void*, itype, stride).  None = identity."""
    if a is None:
        return None, None, 1, 1
    a = np.asarray(a)
    if a.dtype == np.int32:
        it = 0
    else:
        a = a.astype(np.int64, copy=False)
        it = 1
    assert a.ndim == 1
    stride = a.strides[0] // a.itemsize if a.size > 1 else 1
    return a, ctypes.c_void_p(a.ctypes.data), it, stride


def _pairs(t: Tables, direction, a, p, targets):
    a_k, a_p, a_t, a_s = _idx(a)
    p_k, p_p, p_t, p_s = _idx(p)
    n = len(a_k)
    t_k, t_p, t_t, t_s = _idx(targets)
    m = t.num_ent if targets is None else len(t_k)
    out = np.empty((n, m), dtype=np.float32)
    rc = lib().ko_score_pairs(
        ctypes.byref(t.c), direction, a_p, a_t, ctypes.c_int64(a_s), p_p, p_t,
        ctypes.c_int64(p_s), ctypes.c_int64(n), t_p, t_t, ctypes.c_int64(t_s),
        ctypes.c_int64(m), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(m))
    assert rc == 0
    return out


def score_sp(t: Tables, s, p, o=None):
    """KgeModel.score_sp (kge_model.py:682-702)."""
    return _pairs(t, SP, s, p, o)


def score_po(t: Tables, p, o, s=None):
    """KgeModel.score_po (kge_model.py:704-725)."""
    return _pairs(t, PO, o, p, s)


def score_sp_po(t: Tables, s, p, o, entity_subset=None):
    """KgeModel.score_sp_po (kge_model.py:749-789)."""
    return np.concatenate(
        (score_sp(t, s, p, entity_subset), score_po(t, p, o, entity_subset)), axis=1)


def score_spo(t: Tables, s, p, o):
    """KgeModel.score_spo (kge_model.py:663-680)."""
    s_k, s_p, s_t, s_s = _idx(s)
    p_k, p_p, p_t, p_s = _idx(p)
    o_k, o_p, o_t, o_s = _idx(o)
    n = len(s_k)
    out = np.empty((n,), dtype=np.float32)
    rc = lib().ko_score_spo(
        ctypes.byref(t.c), s_p, s_t, ctypes.c_int64(s_s), p_p, p_t, ctypes.c_int64(p_s),
        o_p, o_t, ctypes.c_int64(o_s), ctypes.c_int64(n),
        out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def score_neg(t: Tables, s, p, o, slot, neg):
    """BatchNegativeSample.score, implementation 'triple' (sampler.py:291-306)."""
    s_k, s_p, s_t, s_s = _idx(s)
    p_k, p_p, p_t, p_s = _idx(p)
    o_k, o_p, o_t, o_s = _idx(o)
    neg = np.ascontiguousarray(neg)
    if neg.dtype != np.int32:
        neg = neg.astype(np.int64, copy=False)
    n, K = neg.shape
    out = np.empty((n, K), dtype=np.float32)
    rc = lib().ko_score_neg(
        ctypes.byref(t.c), s_p, s_t, ctypes.c_int64(s_s), p_p, p_t, ctypes.c_int64(p_s),
        o_p, o_t, ctypes.c_int64(o_s), ctypes.c_int64(n), int(slot),
        ctypes.c_void_p(neg.ctypes.data), 0 if neg.dtype == np.int32 else 1,
        ctypes.c_int64(K), ctypes.c_int64(K), out.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(K))
    assert rc == 0
    return out


def sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.empty_like(x)
    c = np.empty_like(x)
    f = lib().ko_sincosf
    f.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    sf, cf = ctypes.c_float(), ctypes.c_float()
    xs, ss, cc = x.ravel(), s.ravel(), c.ravel()
    for i in range(xs.size):
        f(float(xs[i]), ctypes.byref(sf), ctypes.byref(cf))
        ss[i], cc[i] = sf.value, cf.value
    return s, c


def rank_counts(scores, true_scores, lbl_rowptr=None, lbl_col=None, col_offset=0,
                true_col=None, atol=1e-5, rtol=1e-4, rank=None, ties=None):
    """EntityRankingJob._filter_and_rank/_get_ranks_and_num_ties
    (eval_entity_ranking.py:533-596); accumulates into rank/ties (int64)."""
    scores = np.asarray(scores, dtype=np.float32)
    assert scores.ndim == 2 and scores.strides[1] == 4
    n, c = scores.shape
    lds = scores.strides[0] // 4
    true_scores = np.ascontiguousarray(true_scores, dtype=np.float32)
    if rank is None:
        rank = np.zeros(n, dtype=np.int64)
    if ties is None:
        ties = np.zeros(n, dtype=np.int64)
    rp = cl = tc = None
    if lbl_rowptr is not None:
        rp = np.ascontiguousarray(lbl_rowptr, dtype=np.int64)
        cl = np.ascontiguousarray(lbl_col, dtype=np.int64)
    if true_col is not None:
        tc = np.ascontiguousarray(true_col, dtype=np.int64)

    def ptr(a):
        return None if a is None else ctypes.c_void_p(a.ctypes.data)

    rc = lib().ko_rank_counts(
        ctypes.c_void_p(scores.ctypes.data), ctypes.c_int64(lds), ctypes.c_int64(n),
        ctypes.c_int64(c), ptr(true_scores), ptr(rp), ptr(cl), ctypes.c_int64(col_offset),
        ptr(tc), ctypes.c_float(atol), ctypes.c_float(rtol), ptr(rank), ptr(ties))
    assert rc == 0
    return rank, ties


# ---- host-side restatements around the rank core ------------------------------
def build_index(triples, key_cols, value_col):
    """KvsAllIndex (kge/indexing.py:10-56): (key) -> sorted unique values."""
    idx = {}
    for row in np.asarray(triples):
        idx.setdefault((int(row[key_cols[0]]), int(row[key_cols[1]])), set()).add(int(row[value_col]))
    return idx


def labels_csr(batch, indexes):
    """CSR of filtered entity ids per batch row for one direction.

    `indexes` is a list of dicts (one per filter split) as built by build_index;
    the union over splits is what the reference's sparse label tensor holds after
    densification (duplicate coordinates sum to inf, eval_entity_ranking.py:77-101,
    489-531).  Returns (rowptr int64 [n+1], col int64 [nnz]) with unique sorted cols.
    """
    rowptr = [0]
    cols = []
    for key in batch:
        u = set()
        for ix in indexes:
            u |= ix.get((int(key[0]), int(key[1])), set())
        cols.extend(sorted(u))
        rowptr.append(len(cols))
    return np.asarray(rowptr, dtype=np.int64), np.asarray(cols, dtype=np.int64)


def evaluate_ranks(t: Tables, triples, filter_index_sp=None, filter_index_po=None,
                   chunk_size=-1, atol=1e-5, rtol=1e-4, tie_handling="rounded_mean_rank"):
    """EntityRankingJob._evaluate for one batch and one ranking
    (eval_entity_ranking.py:163-333).  filter_index_* = lists of build_index dicts
    (None -> raw ranking).  Returns (s_ranks, o_ranks) int64, 0-based."""
    triples = np.asarray(triples)
    s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
    n, E = len(s), t.num_ent
    # true scores through the subset path (:192-203)
    uo, uo_inv = np.unique(o, return_inverse=True)
    o_true = score_sp(t, s, p, uo)[np.arange(n), uo_inv]
    us, us_inv = np.unique(s, return_inverse=True)
    s_true = score_po(t, p, o, us)[np.arange(n), us_inv]
    if filter_index_sp is not None:
        sp_rp, sp_col = labels_csr(triples[:, [0, 1]], filter_index_sp)
        po_rp, po_col = labels_csr(triples[:, [1, 2]], filter_index_po)
    else:
        sp_rp = sp_col = po_rp = po_col = None
    cs = E if chunk_size < 0 else chunk_size
    o_rank = np.zeros(n, np.int64); o_ties = np.zeros(n, np.int64)
    s_rank = np.zeros(n, np.int64); s_ties = np.zeros(n, np.int64)
    for start in range(0, E, cs):
        end = min(start + cs, E)
        sub = np.arange(start, end, dtype=np.int64)
        sc = score_sp_po(t, s, p, o, sub)
        c = end - start
        rank_counts(sc[:, :c], o_true, sp_rp, sp_col, start, o, atol, rtol, o_rank, o_ties)
        rank_counts(sc[:, c:], s_true, po_rp, po_col, start, s, atol, rtol, s_rank, s_ties)
    return get_ranks(s_rank, s_ties, tie_handling), get_ranks(o_rank, o_ties, tie_handling)


def get_ranks(rank, ties, tie_handling="rounded_mean_rank"):
    """EntityRankingJob._get_ranks (eval_entity_ranking.py:598-618)."""
    if tie_handling == "rounded_mean_rank":
        return rank + ties // 2
    if tie_handling == "best_rank":
        return rank
    if tie_handling == "worst_rank":
        return rank + ties - 1
    raise NotImplementedError(tie_handling)


def compute_metrics(s_ranks, o_ranks, num_entities, hits_at_k=(1, 3, 10, 50, 100, 200, 300, 400, 500, 1000)):
    """hist_all + _compute_metrics (eval_entity_ranking.py:620-649,665-687);
    the histogram and the MRR sum are float32 like the reference's."""
    hist = np.zeros(num_entities, dtype=np.float32)
    for r in (o_ranks, s_ranks):
        u, cnt = np.unique(r, return_counts=True)
        np.add.at(hist, u, cnt.astype(np.float32))
    n = float(hist.sum(dtype=np.float32))
    ranks = np.arange(1, num_entities + 1, dtype=np.float32)
    out = {
        "mean_rank": float((hist * ranks).sum(dtype=np.float32)) / n if n > 0 else 0.0,
        "mean_reciprocal_rank": float((hist * (np.float32(1.0) / ranks)).sum(dtype=np.float32)) / n if n > 0 else 0.0,
    }
    kmax = min(max(hits_at_k), num_entities)
    cum = np.cumsum(hist[:kmax].astype(np.float64)) / n if n > 0 else np.zeros(kmax)
    for k in hits_at_k:
        if k <= kmax:
            out[f"hits_at_{k}"] = float(cum[k - 1])
    return out
