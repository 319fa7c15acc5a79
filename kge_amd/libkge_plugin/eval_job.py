"""EntityRankingJob on the fused kernels (eval.type: hip_entity_ranking)."""
import math
import time

import numpy as np
import torch

from kge.job import EvaluationJob
from kge.job.eval_entity_ranking import EntityRankingJob, hist_all

from .. import engine
from ..eval import FilterIndex


class _SliceLoader:
    """len() and iteration of DataLoader(triples, batch_size=b, shuffle=False) with the fast path's collate:
    one-element tuples of consecutive [<= b, 3] slices."""

    def __init__(self, triples, batch_size):
        self.triples, self.batch_size = triples, int(batch_size)

    def __len__(self):
        return (len(self.triples) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b0 in range(0, len(self.triples), self.batch_size):
            yield (self.triples[b0:b0 + self.batch_size],)


class HipEntityRankingJob(EntityRankingJob):
    """`EntityRankingJob._evaluate` (eval_entity_ranking.py:103-481) without its per-batch host work.

    With a hip_* model on a GPU (`_fast_path()`), a batch never leaves the device:

      * `_collate`'s numba dictionary lookups + `coord_to_sparse_tensor` + `_densify_chunk_of_labels`
        (:77-101, 179-181, 489-531; kge/job/util.py:6-60) become ranges into a device-resident
        sorted filter index (kge_amd.eval.FilterIndex, built once in `_prepare` from the same
        splits) looked up by kge_filter_lookup -- no `[n, 2E]` dense 0/inf label matrix;
      * one `_filter_and_rank` per ranking (:533-596, ~10 passes over `[n, 2E]` each) becomes ONE
        kge_rank_counts_multi scan per direction for raw + filtered + filtered-with-test;
      * the true scores are elements of the score matrix (unchunked) or come from scoring every row
        against the batch's own targets (chunked) -- the same kernel chain as the matrix entries, so
        the reference's tie-consistency check (:254-274) holds by construction and the
        `torch.unique` waits (:192-203) disappear;
      * `hist_all` (:665-687) becomes kge_rank_hist when it is the only histogram hook.

    Everything observable stays the reference's: hooks, trace entries and their keys, per-batch and
    final metrics (`_compute_metrics`, `_get_ranks` are inherited), console output.  Any other model
    or device runs the reference's `_evaluate`, with `_get_ranks_and_num_ties` (:571-596) replaced
    by the streaming rank-count kernel when the scores are on a GPU."""

    def _get_ranks_and_num_ties(self, scores: torch.Tensor, true_scores: torch.Tensor):
        if not scores.is_cuda:
            return super()._get_ranks_and_num_ties(scores, true_scores)
        if scores.stride(-1) != 1:
            scores = scores.contiguous()
        return engine.rank_counts(scores, true_scores.view(-1), atol=self.tie_atol, rtol=self.tie_rtol)

    # ------------------------------------------------------------------------------------------
    def _fast_path(self) -> bool:
        m = self.model
        if not (hasattr(m, "_fused") and hasattr(m, "_w")):
            return False
        if not str(self.device).startswith("cuda"):
            return False
        was_training = m.training
        m.eval()  # _evaluate runs in eval mode (eval.py:58-95): dropout is inactive then
        ok = bool(m._fused()) and m._w()[0].is_cuda
        m.train(was_training)
        return ok

    def _prepare(self):
        self._hip_fast = self._fast_path()
        if not self._hip_fast:
            return super()._prepare()
        EvaluationJob._prepare(self)  # the base class's part; EntityRankingJob's builds the numba indexes
        self.triples = self.dataset.split(self.config.get("eval.split"))
        E, R = self.dataset.num_entities(), self.dataset.num_relations()
        dev = torch.device(self.device)
        splits = [self.dataset.split(s).numpy() for s in self.filter_splits]
        idx = [FilterIndex(splits, E, R)]
        self._hip_filter_with_test = "test" not in self.filter_splits and self.filter_with_test
        if self._hip_filter_with_test:
            idx.append(FilterIndex(splits + [self.dataset.split("test").numpy()], E, R))

        def dev_arrays(t):
            uk, start, v = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in t)
            if v.numel() == 0:  # an empty value array still needs an address
                v = torch.zeros(1, dtype=torch.int64, device=dev)
            return uk, start, v

        self._hip_sp = [dev_arrays(i._sp) for i in idx]
        self._hip_po = [dev_arrays(i._po) for i in idx]
        if self.config.get("eval.num_workers") == 0 and not self.config.get("eval.pin_memory"):
            # same batches in the same order as the DataLoader below (shuffle=False), without indexing the
            # triple tensor row by row and concatenating 512 one-row tensors per batch (0.45 ms per batch)
            self.loader = _SliceLoader(self.triples, self.batch_size)
        else:
            self.loader = torch.utils.data.DataLoader(
                self.triples,
                collate_fn=lambda batch: (torch.cat(batch).reshape((-1, 3)),),
                shuffle=False,
                batch_size=self.batch_size,
                num_workers=self.config.get("eval.num_workers"),
                pin_memory=self.config.get("eval.pin_memory"),
            )

    def _option(self, name, default):
        """hip_entity_ranking.<name> of the job's configuration (a config written before the option existed: `default`)."""
        try:
            return self.config.get(f"hip_entity_ranking.{name}")
        except KeyError:
            return default

    def _eval_begin(self, M, chunk_size):
        """What one evaluation decides once: which tables the counting kernel runs on, split queries or not."""
        # A plain hip_* model (fused gather): scoring and counting in one kernel -- float32 tables of every scorer
        # (the exact kernels' counting epilogue), bf16 ComplEx / DistMult tables with dim 256 / 512 (the loader /
        # consumer kernel's); what the library declines falls back below.  hip_entity_ranking.two_step: true = never.
        E = self.dataset.num_entities()
        fused_tables = getattr(self.model, "_rank_tables", None)
        if self._option("two_step", False) or M > 3 or fused_tables is None or fused_tables() is None:
            fused_tables = None
        # (the exact kernels' counting saves the score matrix, not time -- kge_amd.eval.EntityRankingEvaluator.run:
        # taken from FUSED_EXACT_MIN_BYTES of score matrix on, or with hip_entity_ranking.fused_exact: always)
        if fused_tables is not None and (fused_tables().ent.dtype != torch.bfloat16
                                         or self.model._scorer.name not in ("complex", "distmult")):
            from ..eval import EntityRankingEvaluator as _Ev
            want = self._option("fused_exact", "auto")
            if not ((want == "always") if want in ("always", "never")
                    else 8 * self.batch_size * min(chunk_size, E) >= _Ev.FUSED_EXACT_MIN_BYTES):
                fused_tables = None
        # hip_entity_ranking.bf16_queries (hip_entity_ranking.yaml): "split" (default) scores bf16 tables with split
        # queries -- rank parity with float32 arithmetic on those tables; "single": one rounded query vector per row.
        # Both are counted inside the scoring kernel (pairs_bf16_v8_rank_kernel: no [n, 2E] score matrix); should the
        # library decline a shape, the two-step path below keeps the split queries
        try:
            split = self.config.get("hip_entity_ranking.bf16_queries") != "single"
        except KeyError:
            split = True
        split_tables = fused_tables if (split and fused_tables is not None
                                        and fused_tables().ent.dtype == torch.bfloat16
                                        and self.model._scorer.name in ("complex", "distmult")) else None
        rank_flags = engine.FLAG_SPLIT_QUERY if split_tables is not None else None

        self._ev = {"fused_tables": fused_tables, "split_tables": split_tables, "rank_flags": rank_flags, "band": None}
        # hip_entity_ranking.band_rescore (hip_entity_ranking.yaml): the split counts from a single-pass launch + a
        # small launch over the pairs it could not decide (engine.RankBand).  Unchunked evaluation only.
        # The status words are read after every batch (this loop waits for every batch's ranks anyway): a batch that
        # dropped pairs is counted again by the split kernel, and under "auto" a batch that lists more than
        # BAND_MAX_LISTED of its pairs -- an untrained model -- ends the band for this evaluation.
        try:
            want_band = self.config.get("hip_entity_ranking.band_rescore")
        except KeyError:
            want_band = "auto"
        from ..eval import EntityRankingEvaluator as _Ev
        if (split_tables is not None and rank_flags and chunk_size >= E and want_band in ("auto", "always")
                and (want_band != "auto" or E >= _Ev.BAND_MIN_ENTITIES)
                and fused_tables().ent.is_cuda and fused_tables().ent.shape[1] in (256, 512)):
            band = getattr(self, "_rank_band", None)
            ft = fused_tables()
            if band is None or band.m != E or band.n_max < self.batch_size or band.tmax.device != ft.ent.device:
                band = self._rank_band = engine.RankBand(ft, self.batch_size)
            else:
                band.refresh(ft)  # (the tables moved since the last validation)
            self._ev.update(band=band, band_auto=want_band == "auto", band_batches=0)

    def _batch_counts(self, batch, M, chunk_size):
        """int64 counts [o | s][rank | ties][ranking][row] of one batch (on the device; launches only).  The sharded
        job (sharded_job.HipShardedEntityRankingJob) overrides this one method."""
        ev = self._ev
        fused_tables, split_tables, rank_flags = ev["fused_tables"], ev["split_tables"], ev["rank_flags"]
        E, R = self.dataset.num_entities(), self.dataset.num_relations()
        dev = batch.device
        s, p, o = batch[:, 0], batch[:, 1], batch[:, 2]
        n = batch.shape[0]
        s64, o64 = s.long().contiguous(), o.long().contiguous()  # true_col of the po / sp rankings
        rng = torch.empty(2, M - 1, 2, n, dtype=torch.int64, device=dev)
        cnt = torch.zeros(2, 2, M, n, dtype=torch.int64, device=dev)  # [o|s][rank|ties][ranking][row]
        filt_o, filt_s, lookups = [], [], []
        for k in range(M - 1):
            uk, start, v = self._hip_sp[k]
            lookups.append((uk, start, s, p, R, rng[0, k, 0], rng[0, k, 1]))
            filt_o.append((rng[0, k, 0], rng[0, k, 1], v))
            uk, start, v = self._hip_po[k]
            lookups.append((uk, start, p, o, E, rng[1, k, 0], rng[1, k, 1]))
            filt_s.append((rng[1, k, 0], rng[1, k, 1], v))
        engine.filter_lookup_multi(lookups)  # kge_filter_lookup_multi: the batch's four lookups, one launch

        o_true = s_true = None
        ft = fused_tables() if fused_tables is not None else None
        if ft is not None:
            # counts straight from the scoring kernel (kge_score_rank_sp_po): the true scores up front, as
            # the two diagonals of ONE two-sided call against the batch's own targets (o | s)
            both = engine.score_sp_po(ft, s, p, o, torch.cat([o64, s64]), flags=rank_flags)
            o_true = both.as_strided((n,), (4 * n + 1,)).contiguous()
            s_true = both.as_strided((n,), (4 * n + 1,), 3 * n).contiguous()
        elif chunk_size < E and split_tables is not None:
            both = engine.score_sp_po(split_tables(), s, p, o, torch.cat([o64, s64]), flags=engine.FLAG_SPLIT_QUERY)
            o_true = both.as_strided((n,), (4 * n + 1,)).contiguous()
            s_true = both.as_strided((n,), (4 * n + 1,), 3 * n).contiguous()
        elif chunk_size < E:
            # the subset path of :192-203 without torch.unique: every row against the batch's
            # own targets, diagonal kept (each score is its own kernel chain)
            o_true = self.model.score_sp(s, p, o64).diagonal().contiguous()
            s_true = self.model.score_po(p, o, s64).diagonal().contiguous()
        for chunk_number in range(math.ceil(E / chunk_size)):
            chunk_start = chunk_size * chunk_number
            chunk_end = min(chunk_size * (chunk_number + 1), E)
            c = chunk_end - chunk_start
            band = ev.get("band") if ft is not None and c == E else None
            if band is not None:
                from ..eval import EntityRankingEvaluator as _Ev
                cb = torch.zeros_like(cnt)
                listed0, dropped0 = band.status()
                ok = engine.score_rank_sp_po(ft, s, p, o, o_true, s_true, filt_o, filt_s, self.tie_atol, self.tie_rtol,
                                             cb[0, 0], cb[0, 1], cb[1, 0], cb[1, 1], chunk_start, chunk_end,
                                             flags=rank_flags, band=band)
                listed, dropped = band.status()
                if ok and dropped == dropped0:
                    cnt += cb
                    ev["band_batches"] += 1
                    if ev["band_auto"] and listed - listed0 > _Ev.BAND_MAX_LISTED * band.pairs_of(n):
                        ev["band"] = None  # complete, but no gain on these tables: the split kernel from here on
                    continue
                ev["band"] = None  # pairs were dropped (or the library declined): this batch again, without the band
            if ft is not None:
                if engine.score_rank_sp_po(ft, s, p, o, o_true, s_true, filt_o, filt_s, self.tie_atol,
                                           self.tie_rtol, cnt[0, 0], cnt[0, 1], cnt[1, 0], cnt[1, 1],
                                           chunk_start, chunk_end, flags=rank_flags):
                    continue
                # declined: the two-step path from here on (o_true / s_true stay)
                ft = ev["fused_tables"] = fused_tables = None
            if split_tables is not None:
                # (the chunk as a RANGE of the table -- kge_index.start --: the all-entities kernels, not a listed subset)
                scores = engine.score_sp_po(split_tables(), s, p, o, None if c == E else range(chunk_start, chunk_end),
                                            flags=engine.FLAG_SPLIT_QUERY)
            else:
                sub = None if c == E else torch.arange(chunk_start, chunk_end, device=dev)
                scores = self.model.score_sp_po(s, p, o, sub)
            scores_sp, scores_po = scores[:, :c], scores[:, c:]
            if o_true is None:
                o_true = scores_sp.gather(1, o64.view(-1, 1)).view(-1)
                s_true = scores_po.gather(1, s64.view(-1, 1)).view(-1)
            engine.rank_counts_multi(scores_sp, o_true, filt_o, chunk_start, o64, self.tie_atol,
                                     self.tie_rtol, cnt[0, 0], cnt[0, 1])
            engine.rank_counts_multi(scores_po, s_true, filt_s, chunk_start, s64, self.tie_atol,
                                     self.tie_rtol, cnt[1, 0], cnt[1, 1])

        return cnt

    @torch.no_grad()
    def _evaluate(self):
        if not getattr(self, "_hip_fast", False):
            return super()._evaluate()
        E, R = self.dataset.num_entities(), self.dataset.num_relations()
        dev = torch.device(self.device)
        filter_with_test = self._hip_filter_with_test
        rankings = ["_raw", "_filt", "_filt_test"] if filter_with_test else ["_raw", "_filt"]
        M = len(rankings)
        suffixes = ["", "_filtered", "_filtered_with_test"][:M]
        kernel_hist = self.hist_hooks == [hist_all] and not self.head_and_tail and dev.type == "cuda"
        hists = [dict() for _ in range(M)]  # raw, filt, filt_test: key -> histogram
        chunk_size = self.config.get("entity_ranking.chunk_size")
        if chunk_size <= -1:
            chunk_size = E

        self.current_trace["epoch"] = dict(
            type="entity_ranking", scope="epoch", split=self.eval_split, filter_splits=self.filter_splits,
            epoch=self.epoch, batches=len(self.loader), size=len(self.triples))
        for f in self.pre_epoch_hooks:
            f(self)

        self._eval_begin(M, chunk_size)

        metrics = {}
        epoch_time = -time.time()
        for batch_number, batch_coords in enumerate(self.loader):
            self.current_trace["batch"] = dict(
                type="entity_ranking", scope="batch", split=self.eval_split, filter_splits=self.filter_splits,
                epoch=self.epoch, batch=batch_number, size=len(batch_coords[0]), batches=len(self.loader))
            for f in self.pre_batch_hooks:
                f(self)

            batch = batch_coords[0].to(dev)
            s, p, o = batch[:, 0], batch[:, 1], batch[:, 2]
            n = batch.shape[0]
            s64, o64 = s.long().contiguous(), o.long().contiguous()  # true_col of the po / sp rankings
            cnt = self._batch_counts(batch, M, chunk_size)

            # final ranks from the counts (inherited tie policy) + histograms
            o_ranks = [self._get_ranks(cnt[0, 0, m], cnt[0, 1, m]) for m in range(M)]
            s_ranks = [self._get_ranks(cnt[1, 0, m], cnt[1, 1, m]) for m in range(M)]
            batch_hists = [dict() for _ in range(M)]
            if kernel_hist:
                h = torch.zeros(M, E, dtype=torch.float, device=dev)
                engine.rank_hist(cnt[0, 0], cnt[0, 1], self.tie_handling, h)
                engine.rank_hist(cnt[1, 0], cnt[1, 1], self.tie_handling, h)
                for m in range(M):
                    batch_hists[m]["all"] = h[m]
            else:
                for f in self.hist_hooks:
                    for m in range(M):
                        f(batch_hists[m], s, p, o, s_ranks[m], o_ranks[m], job=self)

            if self.trace_examples:
                entry = {"type": "entity_ranking", "scope": "example", "split": self.eval_split,
                         "filter_splits": self.filter_splits, "size": len(batch), "batches": len(self.loader),
                         "epoch": self.epoch}
                sl, pl, ol = s.tolist(), p.tolist(), o.tolist()
                o_r = [r.tolist() for r in o_ranks]
                s_r = [r.tolist() for r in s_ranks]
                for i in range(len(batch)):
                    entry["batch"] = i
                    entry["s"], entry["p"], entry["o"] = sl[i], pl[i], ol[i]
                    if filter_with_test:
                        entry["rank_filtered_with_test"] = o_r[2][i] + 1
                    self.trace(event="example_rank", task="sp", rank=o_r[0][i] + 1,
                               rank_filtered=o_r[1][i] + 1, **entry)
                    if filter_with_test:
                        entry["rank_filtered_with_test"] = s_r[2][i] + 1
                    self.trace(event="example_rank", task="po", rank=s_r[0][i] + 1,
                               rank_filtered=s_r[1][i] + 1, **entry)

            metrics = {}
            for m in range(M):
                metrics.update(self._compute_metrics(batch_hists[m]["all"], suffix=suffixes[m]))
            self.current_trace["batch"].update(metrics)
            for f in self.post_batch_hooks:
                f(self)
            if self.trace_batch:
                self.trace(**self.current_trace["batch"])
            self.current_trace["batch"] = None

            self.config.print(
                ("\r" + "{}  batch:{: " + str(1 + int(math.ceil(math.log10(len(self.loader))))) + "d}/{}, "
                 + "mrr (filt.): {:4.3f} ({:4.3f}), hits@1: {:4.3f} ({:4.3f}), hits@{}: {:4.3f} ({:4.3f})"
                 + "\033[K").format(
                    self.config.log_prefix, batch_number, len(self.loader) - 1,
                    metrics["mean_reciprocal_rank"], metrics["mean_reciprocal_rank_filtered"],
                    metrics["hits_at_1"], metrics["hits_at_1_filtered"], self.hits_at_k_s[-1],
                    metrics["hits_at_{}".format(self.hits_at_k_s[-1])],
                    metrics["hits_at_{}_filtered".format(self.hits_at_k_s[-1])]),
                end="", flush=True)

            for m in range(M):
                for key, hist in batch_hists[m].items():
                    hists[m][key] = hists[m][key] + hist if key in hists[m] else hist

        self.config.print("\033[2K\r", end="", flush=True)
        for key in hists[0]:
            name = "_" + key if key != "all" else ""
            for m in range(M):
                metrics.update(self._compute_metrics(hists[m][key], suffix=suffixes[m] + name))
        epoch_time += time.time()
        self.current_trace["epoch"].update(dict(epoch_time=epoch_time, event="eval_completed", **metrics))
