"""Entity-ranking evaluation on the fused kernels: host-side mirror of the reference's
EntityRankingJob (kge/job/eval_entity_ranking.py) for the part of it that is on the hot path.

What is kept from the reference and what is replaced:
  * batching, the true-score-through-subset rule (:192-203), the entity chunk loop
    (:216-229), tie policy (:598-618), float32 rank histogram and MRR / Hits@k
    (:620-649, 665-687) -- same semantics, same names;
  * `_collate` + sparse label tensors + `_densify_chunk_of_labels` (:77-101, 179-181,
    489-531) become ranges into a device-resident filter index (FilterIndex: sorted
    (key, value) arrays instead of the numba KvsAllIndex dict, kge/indexing.py:10-194),
    looked up per batch by kge_filter_lookup -- no host work, no host -> device copies;
  * `_filter_and_rank` / `_get_ranks_and_num_ties` (:533-596), run once per ranking by the
    reference, become ONE kge_rank_counts_multi scan per direction for raw + filtered +
    filtered-with-test; `_get_ranks` + `hist_all` become kge_rank_hist.  The loop never waits
    for the device (the reference's torch.unique / .item() calls do).

Rankings follow the reference: "_raw", "_filt" (filter_splits + the eval split) and, if
filter_with_test, "_filt_test" (additionally the test split).  Because the reference
applies the filters cumulatively (:305-307), "_filt_test" filters with the union.
"""
from typing import Dict, List


import numpy as np
import torch

from . import engine


class FilterIndex:
    """(s,p)->{o} and (p,o)->{s} over a union of splits, as sorted arrays (CSR by key).

    Replaces KvsAllIndex (kge/indexing.py:10-194) + get_sp_po_coords_from_spo_batch
    (kge/job/util.py:6-29).  Values are unique and sorted per key, so the union over
    splits needs no de-duplication at lookup time."""

    def __init__(self, triples_list: List[np.ndarray], num_entities: int, num_relations: int):
        tri = np.concatenate([np.asarray(t).reshape(-1, 3) for t in triples_list]).astype(np.int64)
        self.E, self.R = int(num_entities), int(num_relations)
        self._sp = self._build(tri[:, 0] * self.R + tri[:, 1], tri[:, 2])
        self._po = self._build(tri[:, 1] * self.E + tri[:, 2], tri[:, 0])

    def _build(self, keys, vals):
        kv = np.unique(keys * self.E + vals)  # sorted unique (key, value) pairs
        k, v = kv // self.E, kv % self.E
        uk, start = np.unique(k, return_index=True)
        return uk, np.append(start, len(k)).astype(np.int64), v

    @staticmethod
    def _lookup(index, keys):
        uk, start, v = index
        pos = np.searchsorted(uk, keys)
        pos_c = np.minimum(pos, len(uk) - 1) if len(uk) else pos
        hit = (pos < len(uk)) & (uk[pos_c] == keys) if len(uk) else np.zeros(len(keys), bool)
        lo = np.where(hit, start[pos_c], 0)
        hi = np.where(hit, start[pos_c + 1], 0) if len(uk) else lo
        cnt = hi - lo
        rowptr = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(cnt, out=rowptr[1:])
        # gather the segments
        idx = np.repeat(lo - rowptr[:-1], cnt) + np.arange(rowptr[-1])
        return rowptr, v[idx] if rowptr[-1] else np.zeros(0, dtype=np.int64)

    @staticmethod
    def _ranges(index, keys):
        uk, start, _ = index
        if not len(uk):
            z = np.zeros(len(keys), dtype=np.int64)
            return z, z.copy()
        pos = np.minimum(np.searchsorted(uk, keys), len(uk) - 1)
        hit = uk[pos] == keys
        return np.where(hit, start[pos], 0).astype(np.int64), np.where(hit, start[pos + 1], 0).astype(np.int64)

    def ranges(self, batch: np.ndarray):
        """-> (sp_begin, sp_end, po_begin, po_end): per row the range of its known answers in
        `sp_values` / `po_values` -- what kge_filter_lookup computes on the device."""
        b = np.asarray(batch).astype(np.int64)
        return self._ranges(self._sp, b[:, 0] * self.R + b[:, 1]) + self._ranges(self._po, b[:, 1] * self.E + b[:, 2])

    @property
    def sp_values(self):
        return self._sp[2]

    @property
    def po_values(self):
        return self._po[2]

    def labels(self, batch: np.ndarray):
        """-> (sp_rowptr, sp_col, po_rowptr, po_col) for a batch of (s,p,o) triples."""
        b = np.asarray(batch).astype(np.int64)
        sp = self._lookup(self._sp, b[:, 0] * self.R + b[:, 1])
        po = self._lookup(self._po, b[:, 1] * self.E + b[:, 2])
        return sp + po


def get_ranks(rank, ties, tie_handling="rounded_mean_rank"):
    """EntityRankingJob._get_ranks (eval_entity_ranking.py:598-618)."""
    if tie_handling == "rounded_mean_rank":
        return rank + ties // 2
    if tie_handling == "best_rank":
        return rank
    if tie_handling == "worst_rank":
        return rank + ties - 1
    raise NotImplementedError


def compute_metrics(rank_hist: torch.Tensor, hits_at_k_s, suffix="") -> Dict[str, float]:
    """EntityRankingJob._compute_metrics (eval_entity_ranking.py:620-649), same arithmetic:
    float32 histogram, float32 reciprocal ranks, python-float division."""
    metrics = {}
    n = torch.sum(rank_hist).item()
    E = rank_hist.numel()
    ranks = torch.arange(1, E + 1, device=rank_hist.device).float()
    metrics["mean_rank" + suffix] = (torch.sum(rank_hist * ranks).item() / n) if n > 0.0 else 0.0
    metrics["mean_reciprocal_rank" + suffix] = (
        (torch.sum(rank_hist * (1.0 / ranks)).item() / n) if n > 0.0 else 0.0)
    hits = [k for k in hits_at_k_s if k <= E]
    if hits:
        cum = (torch.cumsum(rank_hist[: max(hits)], dim=0, dtype=torch.float64) / n).tolist() \
            if n > 0.0 else [0.0] * max(hits)
        for k in hits:
            metrics["hits_at_{}{}".format(k, suffix)] = cum[k - 1]
    return metrics


class EntityRankingEvaluator:
    """Mirror of EntityRankingJob._evaluate (eval_entity_ranking.py:103-481)."""

    # score-matrix bytes (both directions of a batch) from which float32 / TransE / RotatE tables are counted inside
    # the exact kernels instead of scored and scanned (see run())
    FUSED_EXACT_MIN_BYTES = 1 << 30
    TWO_STEP_LANE_BYTES = 128 << 20  # two-step settings: lanes of captured batches only below this score-matrix size
    # band-and-rescore (engine.RankBand): taken for a run when its first batch lists at most this share
    # of its (row, column) pairs -- a trained model: ~5e-5; random tables list 1.5e-2 and keep the split kernel
    BAND_MAX_LISTED = 5e-4
    # ... and, under "auto", only from this many entities on: the form costs two launches and a read of every row's q_lo
    # block per (side, chunk) -- ~14 us at the FB15k-237 shape, where the split kernel's whole second chain is 16 us
    BAND_MIN_ENTITIES = 100000
    # measurement options and their defaults (constructor keywords): two_step = never count inside the scoring kernel;
    # launch_by_launch = the fused loop as separate engine calls; reserve_cus = compute units the persistent kernels leave
    # free; hip_graph = full batches as graph replays; lanes = captured batches in flight; fused_exact = the exact
    # kernels' counting epilogue (None: from FUSED_EXACT_MIN_BYTES of score matrix on)
    OPTIONS = {"two_step": False, "launch_by_launch": False, "reserve_cus": 0, "hip_graph": True, "lanes": 3,
               "fused_exact": None}

    def __init__(self, model, splits: Dict[str, np.ndarray], num_entities: int, num_relations: int,
                 eval_split: str = "valid", filter_splits=("train", "valid"),
                 filter_with_test: bool = True, batch_size: int = 100, chunk_size: int = -1,
                 tie_handling: str = "rounded_mean_rank", tie_atol: float = 1e-5,
                 tie_rtol: float = 1e-4,
                 hits_at_k_s=(1, 3, 10, 50, 100, 200, 300, 400, 500, 1000), band_rescore="auto", **options):
        self.model = model
        # measurement options (OPTIONS below; keyword arguments override the class defaults -- nothing here is read
        # from the environment: tools/ map their old KGE_EVAL_* variables onto OPTIONS themselves, tools/_eval_env.py)
        unknown = set(options) - set(self.OPTIONS)
        if unknown:
            raise TypeError(f"EntityRankingEvaluator: unknown option(s) {sorted(unknown)}")
        opt = dict(self.OPTIONS, **options)
        # split-query evaluation of bf16 ComplEx / DistMult tables through band-and-rescore: "auto" (probe the first
        # batch of every run), True (no probe; a run that dropped pairs still falls back), False (the split kernel)
        self.band_rescore = band_rescore
        self.band_runs = 0       # runs that counted through band-and-rescore (complete: no dropped pair)
        self.band_listed = None  # (pairs listed, pairs) of the last run's probe batch
        self._band_off = False
        self.E, self.R = num_entities, num_relations
        self.triples = np.asarray(splits[eval_split]).reshape(-1, 3)
        fs = list(filter_splits)
        if eval_split not in fs:
            fs.append(eval_split)  # eval_entity_ranking.py:28-29
        self.filter_with_test = filter_with_test and "test" not in fs
        self.index_filt = FilterIndex([splits[s] for s in fs], num_entities, num_relations)
        self.index_filt_test = (FilterIndex([splits[s] for s in fs + ["test"]], num_entities,
                                            num_relations) if self.filter_with_test else None)
        self.batch_size, self.chunk_size = batch_size, chunk_size
        self.tie_handling, self.tie_atol, self.tie_rtol = tie_handling, tie_atol, tie_rtol
        self.hits_at_k_s = [k for k in hits_at_k_s if k <= min(num_entities, max(hits_at_k_s))]
        # count inside the scoring kernel where the library offers it (set False to force the two-step path)
        self._fused = not opt["two_step"]
        self._declined = set()  # batch sizes kge_score_rank_sp_po declined for these tables
        self._declined_for = None
        # launch_by_launch: the fused loop as separate engine calls (a dozen launches per batch) instead of
        # kge_eval_batch's four
        self.four_launches = not opt["launch_by_launch"]
        self.reserve_cus = int(opt["reserve_cus"])
        # replay the fused loop's full batches as one hipGraph (hip_graph=False: issue every launch from Python)
        self.hip_graph = bool(opt["hip_graph"])
        self.graph_batches = 0  # batches that ran as graph replays (all runs)
        self.lanes = int(opt["lanes"])  # captured batches in flight (one HIP stream each)
        self.fused_exact = opt["fused_exact"]  # None: by score-matrix size (FUSED_EXACT_MIN_BYTES); True / False
        self._graph = None

    @staticmethod
    def _band_for(st, tables, which):
        """The RankBand of the probe / of the eager (ragged last) batches, made at first use."""
        b = st.setdefault("bands", {}).get(which)
        if b is None:
            b = st["bands"][which] = engine.RankBand(tables, st["ranges"].shape[-1])
        return b

    def _device_state(self, dev):
        """Everything the loop needs, resident on `dev` (built once): the eval triples, the filter
        indexes as sorted arrays, per-batch scratch."""
        st = getattr(self, "_dev_state", None)
        if st is not None and st["dev"] == dev:
            return st
        idx = [self.index_filt] + ([self.index_filt_test] if self.filter_with_test else [])
        K, bs = len(idx), self.batch_size
        st = {
            "dev": dev,
            "triples": torch.from_numpy(np.ascontiguousarray(self.triples.astype(np.int64))).to(dev),
            "sp": [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in i._sp) for i in idx],
            "po": [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in i._po) for i in idx],
            # [direction (o: sp-filters, s: po-filters)][filter][begin | end][row]
            "ranges": torch.zeros(2, K, 2, bs, dtype=torch.int64, device=dev),
            # [direction][rank | ties][ranking][row]
            "counts": torch.zeros(2, 2, K + 1, bs, dtype=torch.int64, device=dev),
            "diag": {},
            "counts4": {},  # batch size -> all-zero counters of kge_eval_batch (it returns them zeroed)
        }
        for k in range(K):  # an empty value array still needs an address
            for side in ("sp", "po"):
                uk, start, v = st[side][k]
                if v.numel() == 0:
                    st[side][k] = (uk, start, torch.zeros(1, dtype=torch.int64, device=dev))
        self._dev_state = st
        return st

    @torch.no_grad()
    def _run_sharded(self, sh, return_ranks: bool):
        """The same loop over an entity-sharded table (kge_amd.sharded.ShardedEntityTable; every rank calls this
        with the same evaluator arguments): the filter index is replicated, each rank counts over its own entity
        rows -- inside the scoring kernel where the backend offers it --, ONE int64 all-reduce per batch makes the
        counts global (ShardedEntityTable.rank_batch_multi), and every rank ends with the full histograms and the
        same metrics.  No score matrix, no device -> host synchronisation inside the loop."""
        dev = sh.ent_local.device
        E, R = self.E, self.R
        rankings = ["_raw", "_filt"] + (["_filt_test"] if self.filter_with_test else [])
        M = len(rankings)
        st = self._device_state(dev)
        hist = torch.zeros(M, E, dtype=torch.float, device=dev)
        all_ranks = {f"{d}{r}": [] for r in rankings for d in "so"}
        triples, bs = st["triples"], self.batch_size
        for b0 in range(0, len(self.triples), bs):
            batch = triples[b0:b0 + bs]
            n = batch.shape[0]
            s, p, o = batch[:, 0], batch[:, 1], batch[:, 2]
            rng = st["ranges"][:, :, :, :n].contiguous() if n != bs else st["ranges"]
            filt_o, filt_s, lookups = [], [], []
            for k in range(M - 1):
                uk, start, v = st["sp"][k]
                lookups.append((uk, start, s, p, R, rng[0, k, 0], rng[0, k, 1]))
                filt_o.append((rng[0, k, 0], rng[0, k, 1], v))
                uk, start, v = st["po"][k]
                lookups.append((uk, start, p, o, E, rng[1, k, 0], rng[1, k, 1]))
                filt_s.append((rng[1, k, 0], rng[1, k, 1], v))
            engine.filter_lookup_multi(lookups)
            cnt = sh.rank_batch_multi(batch, filt_o, filt_s, self.tie_atol, self.tie_rtol)
            ro = torch.empty(M, n, dtype=torch.int64, device=dev) if return_ranks else None
            rs = torch.empty(M, n, dtype=torch.int64, device=dev) if return_ranks else None
            engine.rank_hist(cnt[0, 0], cnt[0, 1], self.tie_handling, hist, ro)
            engine.rank_hist(cnt[1, 0], cnt[1, 1], self.tie_handling, hist, rs)
            if return_ranks:
                for m_, r in enumerate(rankings):
                    all_ranks["o" + r].append(ro[m_])
                    all_ranks["s" + r].append(rs[m_])
        suffix = {"_raw": "", "_filt": "_filtered", "_filt_test": "_filtered_with_test"}
        metrics = {}
        for m_, r in enumerate(rankings):
            metrics.update(compute_metrics(hist[m_], self.hits_at_k_s, suffix[r]))
        if return_ranks:
            return metrics, {k: torch.cat(v).cpu().numpy() if v else np.zeros(0, np.int64)
                             for k, v in all_ranks.items()}
        return metrics

    @torch.no_grad()
    def run(self, return_ranks: bool = False):
        """No device -> host synchronisation inside the loop: the batch's filter ranges come from
        kge_filter_lookup on the device-resident index, all rankings of a direction from one
        kge_rank_counts_multi scan, tie policy + histogram from kge_rank_hist."""
        model = self.model
        tables = model.tables() if hasattr(model, "tables") else model
        from .sharded import ShardedEntityTable
        if isinstance(tables, ShardedEntityTable):
            return self._run_sharded(tables, return_ranks)
        dev = tables.device
        E, R = self.E, self.R
        rankings = ["_raw", "_filt"] + (["_filt_test"] if self.filter_with_test else [])
        M = len(rankings)
        st = self._device_state(dev)
        # a captured batch (below) survives from run to run while the tables stay where they are
        # (everything a capture freezes into its kernel arguments is part of the key: table flags -- split queries,
        # exact chain --, the tie policy and tolerances, which of the fused / four-launch paths is taken)
        gkey = ((tables.ent.data_ptr(), tables.rel.data_ptr(), tuple(tables.ent.shape), tuple(tables.rel.shape),
                 tables.ent.stride(0), tables.rel.stride(0), tables.scorer, bool(return_ranks), M, int(tables.flags),
                 bool(self._fused), bool(self.four_launches), self.tie_handling, float(self.tie_atol),
                 float(self.tie_rtol), self.fused_exact, int(self.chunk_size), int(self.reserve_cus))
                if isinstance(tables, engine.Tables) else None)
        # ---- band-and-rescore for this run?
        chunk0 = E if self.chunk_size < 0 else self.chunk_size
        use_band = (self.band_rescore in ("auto", True) and not self._band_off and self._fused and self.four_launches
                    and isinstance(tables, engine.Tables) and M <= 3 and chunk0 >= E and dev.type == "cuda"
                    and bool(int(tables.flags) & engine.FLAG_SPLIT_QUERY) and tables.ent.dtype == torch.bfloat16
                    and tables.ent.shape[1] in (256, 512) and len(self.triples) > 0
                    and tables.scorer in (engine.SCORERS["complex"], engine.SCORERS["distmult"]))
        if use_band and self.band_rescore == "auto" and E < self.BAND_MIN_ENTITIES:
            use_band = False
        if use_band:
            for b in st.setdefault("bands", {}).values():
                b.refresh(tables)
            if self.band_rescore == "auto":
                n0 = min(self.batch_size, len(self.triples))
                pb = self._band_for(st, tables, "probe")
                t0 = st["triples"][:n0]
                c0 = torch.zeros(2, 2, M, n0, dtype=torch.int64, device=dev)
                ok = engine.eval_batch(tables, t0[:, 0], t0[:, 1], t0[:, 2],
                                       [(st["sp"][k], st["po"][k]) for k in range(M - 1)], self.tie_atol, self.tie_rtol,
                                       self.tie_handling, c0, torch.zeros(M, E, dtype=torch.float, device=dev), None,
                                       None, band=pb)
                listed, dropped = pb.status()  # (the run's one host wait besides its result)
                self.band_listed = (listed, pb.pairs_of(n0))
                use_band = bool(ok) and dropped == 0 and listed <= self.BAND_MAX_LISTED * pb.pairs_of(n0)
                pb.reset()
        if gkey is not None:
            gkey = gkey + (bool(use_band),)
        held = self._graph if (self._graph is not None and self._graph["key"] == gkey) else None
        hist = held["hist"].zero_() if held is not None else torch.zeros(M, E, dtype=torch.float, device=dev)
        all_ranks = {f"{d}{r}": [] for r in rankings for d in "so"}
        chunk = E if self.chunk_size < 0 else self.chunk_size
        triples = st["triples"]
        # float32 tables of every scorer and bf16 ComplEx / DistMult at d 256 / 512 -- one rounded query vector or
        # split queries (engine.FLAG_SPLIT_QUERY, the rank-parity setting of bf16 tables: pairs_bf16_v8_rank_kernel
        # adds the two partial scores inside a lane) -- have a counting kernel; what the library declines (other bf16
        # shapes) is remembered per batch size and scored in two steps
        fused = self._fused and isinstance(tables, engine.Tables) and M <= 3
        # the exact kernels' counting epilogue (float32 tables, TransE / RotatE) saves the [n, 2E] score matrix, not
        # time: their scoring is compute-bound and the true scores cost a launch pair of their own (C4 shape,
        # float32 DistMult: 0.35 ms per batch against 0.24 for score + scan, tools/eval_f32_probe.py) -- taken when
        # the matrix would be large (FUSED_EXACT_MIN_BYTES) or asked for (fused_exact=True)
        if fused and (tables.ent.dtype != torch.bfloat16 or
                      tables.scorer not in (engine.SCORERS["complex"], engine.SCORERS["distmult"])):
            want = self.fused_exact
            fused = bool(want) if want is not None else 8 * self.batch_size * min(chunk, E) >= self.FUSED_EXACT_MIN_BYTES
        if self._declined_for != gkey:  # other tables: ask again
            self._declined_for, self._declined = gkey, set()
        declined = self._declined

        # Several batches in flight: the persistent counting kernel fills every compute unit's register file, so the
        # small launches of the OTHER lanes (filter lookup, true scores, histograms: latency-bound chains) cannot start
        # beside it unless it leaves some compute units free (KGE_FLAG_RESERVE_CUS): a few per cent of its rate for
        # the ~20 us per batch those launches otherwise add to it
        lane_flags = None
        if isinstance(tables, engine.Tables) and self.lanes > 1 and self.hip_graph and self.reserve_cus > 0:
            lane_flags = int(tables.flags) | engine.reserve_cus(self.reserve_cus)

        bands_used = []

        def do_batch(batch, rng, cnt, ro, rs, band=None):
            """One batch: filter ranges, counts (in place in `cnt`), tie policy + histogram; launches only."""
            s, p, o = batch[:, 0], batch[:, 1], batch[:, 2]
            n = batch.shape[0]
            if fused and chunk >= E and M <= 3 and n not in declined and self.four_launches:
                # the whole batch in four launches (kge_eval_batch): filter lookup + filter bits | true scores |
                # scoring + counting | bits cleared + tie policy + histograms + counters back to zero
                ck = (n, engine._stream_handle(dev))  # per stream: two batches may be in flight (lanes, below)
                c4 = st["counts4"].get(ck)
                if c4 is None:  # zero once; every call leaves it zero
                    c4 = st["counts4"][ck] = torch.zeros(2, 2, M, n, dtype=torch.int64, device=dev)
                if engine.eval_batch(tables, s, p, o, [(st["sp"][k], st["po"][k]) for k in range(M - 1)],
                                     self.tie_atol, self.tie_rtol, self.tie_handling, c4, hist, ro, rs,
                                     flags=lane_flags, band=band):
                    return
                declined.add(n)
            sc_, oc_ = s.contiguous(), o.contiguous()  # true_col of the po / sp rankings
            cnt.zero_()
            filt_o, filt_s, lookups = [], [], []
            for k in range(M - 1):
                uk, start, v = st["sp"][k]
                lookups.append((uk, start, s, p, R, rng[0, k, 0], rng[0, k, 1]))
                filt_o.append((rng[0, k, 0], rng[0, k, 1], v))
                uk, start, v = st["po"][k]
                lookups.append((uk, start, p, o, E, rng[1, k, 0], rng[1, k, 1]))
                filt_s.append((rng[1, k, 0], rng[1, k, 1], v))
            engine.filter_lookup_multi(lookups)  # all lookups of the batch in one launch

            o_true = s_true = None
            use_fused = fused and n not in declined
            if use_fused:
                # the counting kernel needs the true scores up front: the batch against its own targets in
                # one two-sided launch ([n, 4n]: sp_ scores of (o | s), then _po scores of (o | s)), the
                # two diagonals kept -- elements of the score matrix bit for bit, as in the chunked case
                both = engine.score_sp_po(tables, s, p, o, torch.cat([oc_, sc_]))
                diag = st["diag"].get(n)
                if diag is None:  # flat positions of (i, i) and (i, 3n + i) in the [n, 4n] block
                    ar = torch.arange(n, device=dev)
                    diag = st["diag"][n] = torch.cat([ar * (4 * n + 1), ar * (4 * n + 1) + 3 * n])
                true = both.view(-1).index_select(0, diag)
                o_true, s_true = true[:n], true[n:]
            elif chunk < E:
                # true scores through the subset path (:192-203), without torch.unique (a host
                # sync): score every row against the batch's own targets, keep the diagonal --
                # each score is its own chain, so the bits do not depend on the subset
                o_true = engine.score_sp(tables, s, p, o).diagonal().contiguous()
                s_true = engine.score_po(tables, p, o, s).diagonal().contiguous()
            for start in range(0, E, chunk):
                end = min(start + chunk, E)
                if use_fused:
                    # scoring + counting in one kernel: no [n, 2c] score matrix (kge_score_rank_sp_po)
                    if engine.score_rank_sp_po(tables, s, p, o, o_true, s_true, filt_o, filt_s, self.tie_atol,
                                               self.tie_rtol, cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1], start, end):
                        continue
                    # declined, before anything was counted: two steps for batches of this size (a shape the
                    # library has no counting kernel for declines every size; a device with fewer compute units than
                    # the launch needs only some)
                    use_fused = False
                    declined.add(n)
                sub = None if (start == 0 and end == E) else torch.arange(start, end, device=dev)
                scores = engine.score_sp_po(tables, s, p, o, sub)
                c = end - start
                sc_sp, sc_po = scores[:, :c], scores[:, c:]
                if o_true is None:  # unchunked: the true scores are elements of the matrix
                    o_true = sc_sp.gather(1, oc_.view(-1, 1)).view(-1)
                    s_true = sc_po.gather(1, sc_.view(-1, 1)).view(-1)
                engine.rank_counts_multi(sc_sp, o_true, filt_o, start, oc_, self.tie_atol, self.tie_rtol,
                                         cnt[0, 0], cnt[0, 1])
                engine.rank_counts_multi(sc_po, s_true, filt_s, start, sc_, self.tie_atol, self.tie_rtol,
                                         cnt[1, 0], cnt[1, 1])
            # hist_all (:665-687): object ranks and subject ranks into the same histograms
            engine.rank_hist(cnt[0, 0], cnt[0, 1], self.tie_handling, hist, ro)
            engine.rank_hist(cnt[1, 0], cnt[1, 1], self.tie_handling, hist, rs)

        def keep_ranks(ro, rs, copy):
            for m_, r in enumerate(rankings):
                all_ranks["o" + r].append(ro[m_].clone() if copy else ro[m_])
                all_ranks["s" + r].append(rs[m_].clone() if copy else rs[m_])

        N, bs = len(self.triples), self.batch_size
        # Full batches of the fused, unchunked loop have one shape and touch only resident buffers: ONE batch is
        # captured into a hipGraph (its triples in a static buffer) and replayed -- a copy + a graph launch per
        # batch instead of a dozen launches issued from Python (the loop is host-bound at FB15k-237 size).
        # `lanes` such graphs, each with its own static buffers on its own stream, batch k replayed on lane k % lanes:
        # the launches of a batch depend on each other and each of them leaves compute units idle at its ends (and a
        # whole batch is four of them, ~60 us at FB15k-237 size) -- the next batch's launches on the other stream run
        # there.  The histograms are shared: float atomic adds of 1.0, exact in any order.
        lanes = held["lanes"] if held is not None else []
        # (the two-step settings too -- float32 tables, split queries: score matrix + scan + histogram of a batch are
        # launches on resident buffers just the same; their score matrix lives in the graph's memory pool)
        use_graph = (self.hip_graph and isinstance(tables, engine.Tables) and chunk >= E and dev.type == "cuda"
                     and N // bs >= 4)
        L = max(1, self.lanes) if use_graph else 1
        if use_graph and not fused and 8 * bs * E > self.TWO_STEP_LANE_BYTES:
            L = 1  # every lane's graph pool keeps its own [bs, 2E] score matrix alive: one lane beyond ~128 MiB
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        used = set()
        kb = 0
        for b0 in range(0, N, bs):
            n = min(bs, N - b0)
            if use_graph and n == bs and bs not in declined:
                li = kb % L
                kb += 1
                if li >= len(lanes):
                    lanes.append({"batch": torch.empty(bs, 3, dtype=torch.int64, device=dev),
                                  "ro": torch.empty(M, bs, dtype=torch.int64, device=dev) if return_ranks else None,
                                  "rs": torch.empty(M, bs, dtype=torch.int64, device=dev) if return_ranks else None,
                                  "ranges": torch.empty_like(st["ranges"]), "counts": torch.zeros_like(st["counts"]),
                                  "stream": torch.cuda.Stream(dev), "graph": None,
                                  "band": engine.RankBand(tables, bs) if use_band else None, "band_fresh": True})
                ln = lanes[li]
                if ln["band"] is not None and li not in used:
                    if not ln["band_fresh"]:
                        ln["band"].refresh(tables)  # (a held lane: the table's values may have moved since its capture)
                    ln["band_fresh"] = False
                    bands_used.append(ln["band"])
                if li not in used:  # the lane waits for whatever produced the tables / zeroed the histograms
                    ln["stream"].wait_stream(cur)
                    used.add(li)
                args = (ln["batch"], ln["ranges"], ln["counts"], ln["ro"], ln["rs"], ln["band"])
                with torch.cuda.stream(ln["stream"]):
                    ln["batch"].copy_(triples[b0:b0 + bs])
                    if ln["graph"] is None:
                        # this batch eagerly on the lane's stream (scratch buffers are per stream: they get allocated
                        # outside the capture), then the capture itself (records, does not run)
                        do_batch(*args)
                        if bs not in declined:
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, stream=ln["stream"]):
                                do_batch(*args)
                            ln["graph"] = g
                            self._graph = {"key": gkey, "lanes": lanes, "hist": hist}
                    else:
                        ln["graph"].replay()
                        self.graph_batches += 1
                    if return_ranks:
                        keep_ranks(ln["ro"], ln["rs"], True)
                        for r_ in rankings:
                            for d_ in "so":
                                all_ranks[d_ + r_][-1].record_stream(cur)
                continue
            if used:  # a ragged last batch on the current stream: behind the lanes (shared scratch of the eager path)
                for li in used:
                    cur.wait_stream(lanes[li]["stream"])
                used = set()
            batch = triples[b0:b0 + bs]
            rng = st["ranges"][:, :, :, :n].contiguous() if n != bs else st["ranges"]
            cnt = st["counts"][:, :, :, :n].contiguous() if n != bs else st["counts"]
            ro = torch.empty(M, n, dtype=torch.int64, device=dev) if return_ranks else None
            rs = torch.empty(M, n, dtype=torch.int64, device=dev) if return_ranks else None
            eb = self._band_for(st, tables, "eager") if use_band else None
            if eb is not None and not any(eb is b for b in bands_used):
                bands_used.append(eb)
            do_batch(batch, rng, cnt, ro, rs, eb)
            if return_ranks:
                keep_ranks(ro, rs, False)

        for li in used:
            cur.wait_stream(lanes[li]["stream"])
        if use_band:
            # a pair dropped anywhere (a wave's list was full): some batch's counts are incomplete -- the whole run again
            # on the split kernel, and no band for this evaluator from here on
            if any(b.status()[1] != 0 for b in bands_used):
                self._band_off = True
                return self.run(return_ranks)
            self.band_runs += 1
        suffix = {"_raw": "", "_filt": "_filtered", "_filt_test": "_filtered_with_test"}
        metrics = {}
        for m_, r in enumerate(rankings):
            metrics.update(compute_metrics(hist[m_], self.hits_at_k_s, suffix[r]))
        if return_ranks:
            return metrics, {k: torch.cat(v).cpu().numpy() if v else np.zeros(0, np.int64)
                             for k, v in all_ranks.items()}
        return metrics
