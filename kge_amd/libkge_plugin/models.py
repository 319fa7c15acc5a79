"""KgeModel / RelationalScorer subclasses routing the hot path to libkge_amd.so."""
import torch
from torch import Tensor

from kge import Config, Dataset
from kge.model.kge_model import KgeModel, RelationalScorer
from kge.model.rotate import RotatE as _RefRotatE
from kge.model.transe import TransE as _RefTransE

from .. import engine
from ..model import (BF16Shadow, _FusedCE, _FusedCE2, _FusedCE2Sum, _FusedMultiLabel2, _ScoreEmb, _ScoreNeg,
                     _ScoreNegBlocks, _ScorePairs, _ScoreSPO, bce_fused, ce_fused_dropout, kl_fused, neg_blocks_fusable)


class _HipScorer(RelationalScorer):
    """score_emb of the reference scorers (complex.py:18-43, distmult.py:13-25,
    transe.py:15-37, rotate.py:20-69) on dense embeddings; `s_o` keeps the reference's
    generic fallback (kge_model.py:202-209), which lands in score_emb(..., "spo") here."""

    name = None

    def __init__(self, config: Config, dataset: Dataset, configuration_key=None):
        super().__init__(config, dataset, configuration_key)
        self._norm = float(self.get_option("l_norm")) if self.name in ("transe", "rotate") else 1.0

    def score_emb(self, s_emb, p_emb, o_emb, combine: str):
        if not p_emb.is_cuda and combine in ("spo", "sp_", "_po"):
            # `job.device: cpu` (BASELINE configs[0], examples/toy-complex-train.yaml on CPU): there is no HIP
            # device to score on, and the class this one stands in for is right there -- the REFERENCE scorer's own
            # score_emb (complex.py:18-43, ...) on this object (it reads nothing but `_norm`).  Not the oracle.
            return self._reference_scorer().score_emb(self, s_emb, p_emb, o_emb, combine)
        if combine in ("spo", "sp_", "_po"):
            n = p_emb.size(0)
            return _ScoreEmb.apply(self.name, combine, self._norm, s_emb, p_emb, o_emb).view(n, -1)
        return super().score_emb(s_emb, p_emb, o_emb, combine)

    def _reference_scorer(self):
        from kge.model.complex import ComplExScorer
        from kge.model.distmult import DistMultScorer
        from kge.model.rotate import RotatEScorer
        from kge.model.transe import TransEScorer
        return {"complex": ComplExScorer, "distmult": DistMultScorer, "transe": TransEScorer,
                "rotate": RotatEScorer}[self.name]


class HipComplExScorer(_HipScorer):
    name = "complex"


class HipDistMultScorer(_HipScorer):
    name = "distmult"


class HipTransEScorer(_HipScorer):
    name = "transe"


class HipRotatEScorer(_HipScorer):
    name = "rotate"


class _FusedScoring:
    """Overrides of KgeModel.score_* (kge_model.py:663-789): fused gather + score when both
    embedders are plain lookup tables (no active dropout, shared entity embedder)."""

    def _fused(self) -> bool:
        from kge.model import LookupEmbedder
        se, oe, pe = self.get_s_embedder(), self.get_o_embedder(), self.get_p_embedder()
        if se is not oe or type(se) is not LookupEmbedder or type(pe) is not LookupEmbedder:
            return False
        if not se._embeddings.weight.is_cuda:  # job.device: cpu -> KgeModel.score_* with the reference scorer's arithmetic
            return False
        return not (self.training and (se.dropout.p > 0 or pe.dropout.p > 0))

    def _dropout_only(self):
        """(p_entity, p_relation) if the ONLY thing that keeps `_fused()` from holding is embedder dropout in training
        mode on float32 ComplEx / DistMult tables scored in bfloat16 -- the fused 1vsAll loss then applies the masks
        itself (kge_amd.model.ce_fused_dropout: lookup_embedder.py:64-69, 102-105) --, else None."""
        from kge.model import LookupEmbedder
        se, oe, pe = self.get_s_embedder(), self.get_o_embedder(), self.get_p_embedder()
        if se is not oe or type(se) is not LookupEmbedder or type(pe) is not LookupEmbedder:
            return None
        ent, rel = self._w()
        if not (self.training and (se.dropout.p > 0 or pe.dropout.p > 0)) or not ent.is_cuda:
            return None
        if self._scorer.name not in ("complex", "distmult") or ent.dtype != torch.float32 or ent.shape[1] not in (128, 256, 512):
            return None
        try:
            if self.get_option("score_dtype") not in ("bfloat16", "bf16"):
                return None
        except KeyError:
            return None
        return float(se.dropout.p), float(pe.dropout.p)

    def _w(self):
        return (self.get_s_embedder()._embeddings.weight, self.get_p_embedder()._embeddings.weight)

    def penalty(self, **kwargs):
        """KgeModel.penalty (kge_model.py:603-640) first copies the batch's triples to the device -- a blocking
        host-to-device copy in every batch -- and then asks the embedders, which answer with nothing when their
        `regularize_weight` is 0 (the default; lookup_embedder.py:122-126).  In exactly that case (plain
        LookupEmbedders, no penalty configured) the result is the empty list either way; the copy is skipped
        (0.1 ms of the 0.5 ms a fused 1vsAll batch takes through the trainer).  Anything else: the reference's code."""
        from kge.model import LookupEmbedder
        for e in {id(x): x for x in (self.get_s_embedder(), self.get_o_embedder(), self.get_p_embedder())}.values():
            if type(e) is not LookupEmbedder or not (e.regularize == "" or e.get_option("regularize_weight") == 0.0):
                return super().penalty(**kwargs)
        return []

    def _fwd_tables(self):
        """`score_dtype: bfloat16` (hip_complex.yaml / hip_distmult.yaml) with float32 parameters:
        sp_/_po scores come from the bf16 matrix-core kernel on bf16 copies of the tables (re-cast
        after every optimizer step), gradients are taken w.r.t. the float32 masters."""
        ent, rel = self._w()
        if self._scorer.name not in ("complex", "distmult") or ent.dtype != torch.float32:
            return None
        try:
            want = self.get_option("score_dtype")
        except KeyError:
            return None
        if want not in ("bfloat16", "bf16"):
            return None
        if getattr(self, "_bf16_shadow", None) is None:
            self._bf16_shadow = BF16Shadow()
        return self._bf16_shadow.tables(self._scorer.name, ent, rel, self._scorer._norm)

    def score_spo(self, s: Tensor, p: Tensor, o: Tensor, direction=None) -> Tensor:
        if not self._fused():
            return super().score_spo(s, p, o, direction)
        ent, rel = self._w()
        return _ScoreSPO.apply(self._scorer.name, self._scorer._norm, ent, rel, s, p, o)

    def score_neg(self, s: Tensor, p: Tensor, o: Tensor, slot: int, neg: Tensor) -> Tensor:
        """[n, K] scores of the positives with slot (0 = s, 2 = o) replaced by neg[i, k]: what
        BatchNegativeSample.score returns (kge/util/sampler.py:263-306), from kge_score_neg -- the
        positives' fixed rows are read once, only the corrupted rows stream; None if the fused
        path does not apply (HipTrainingJobNegativeSampling then calls the sampler's own score)."""
        if not self._fused() or slot not in (0, 2):
            return None
        ent, rel = self._w()
        if not ent.is_cuda:
            return None
        return _ScoreNeg.apply(self._scorer.name, self._scorer._norm, ent, rel, s, p, o, int(slot), neg)

    def score_neg_blocks(self, s: Tensor, p: Tensor, o: Tensor, neg_s: Tensor = None, neg_o: Tensor = None):
        """(positives [n], subject-slot block [n, K_s] or None, object-slot block [n, K_o] or None) as ONE autograd node
        whose backward fills one pair of table gradients (kge_amd.model._ScoreNegBlocks); None if the fused gather does
        not apply (the caller composes score_spo + score_neg)."""
        if not self._fused():
            return None
        ent, rel = self._w()
        if not neg_blocks_fusable(ent, rel):
            return None
        return _ScoreNegBlocks.apply(self._scorer.name, self._scorer._norm, ent, rel, s, p, o, neg_s, neg_o)

    def _padded(self) -> bool:
        """`padded_scores` (hip_*.yaml, default true): where no gradient is recorded -- evaluation, scoring under
        torch.no_grad() -- score_sp / score_po return the [:, :E] view of a matrix whose rows start on whole 256-byte
        lines (engine.score_pitch): what the store kernels are measured on (E is odd on most datasets: the rows of a
        contiguous [n, E] float matrix start at 4-byte granularity and every 16-byte lane store straddles sectors).
        The values are the same; a caller that needs contiguous memory calls .contiguous()."""
        if torch.is_grad_enabled():
            return False
        try:
            return bool(self.get_option("padded_scores"))
        except KeyError:
            return False

    # ---- scoring without a recorded gradient: what an UNMODIFIED EntityRankingJob (eval.type: entity_ranking) calls
    # -- score_sp_po(s, p, o, torch.arange(chunk_start, chunk_end)) per chunk, score_sp / score_po against the batch's
    # unique true answers (kge/job/eval_entity_ranking.py:143-229).
    RANGE_MIN = 1024  # subsets from this length on are checked for being a contiguous range (one host read)

    def _no_grad_call(self):
        """(tables, flags) of a scoring call under torch.no_grad(), None where a gradient is recorded.  On bf16 tables
        (ComplEx / DistMult, `score_dtype: bfloat16` or bf16 parameters) the flags carry KGE_FLAG_SPLIT_QUERY unless
        `no_grad_queries: single`: the query vector as q_hi + q_lo -- the ranks of float32 arithmetic on those tables
        (what `hip_entity_ranking.bf16_queries: split` counts), where a single rounded query vector moves 4 % of them."""
        if torch.is_grad_enabled():
            return None
        ent, rel = self._w()
        t = self._fwd_tables() or engine.Tables(self._scorer.name, ent.detach(), rel.detach(), self._scorer._norm)
        flags = None
        if t.ent.dtype == torch.bfloat16 and self._scorer.name in ("complex", "distmult"):
            try:
                single = self.get_option("no_grad_queries") == "single"
            except KeyError:
                single = False
            if not single:
                flags = int(t.flags) | engine.FLAG_SPLIT_QUERY
        return t, flags

    def _targets(self, subset):
        """The reference hands an entity chunk over as torch.arange(chunk_start, chunk_end) -- also the whole table when
        nothing is chunked.  A LISTED subset keeps the kernels that gather their target rows through the index (and, with
        split queries, the f32 chain); recognised as a contiguous range (strictly increasing, last - first == len - 1:
        one small reduction + one host read, in a loop that waits for the device several times per batch anyway) it is
        scored by the all-entities kernels on rows [start, stop) of the table (kge_index.start, include/kge_amd.h)."""
        if subset is None or not torch.is_tensor(subset) or subset.dim() != 1 or subset.numel() < self.RANGE_MIN \
                or subset.dtype not in (torch.int32, torch.int64):
            return subset
        if subset.is_cuda and torch.cuda.is_current_stream_capturing():
            return subset  # (no host read inside a capture: the listed-subset kernels)
        m = subset.numel()
        ok = ((subset[-1] - subset[0]) == m - 1) & (subset[1:] > subset[:-1]).all()
        ok, first = torch.stack((ok.to(torch.int64), subset[0].to(torch.int64))).tolist()
        E = self._w()[0].shape[0]
        if not ok or first < 0 or first + m > E:
            return subset
        return None if (first == 0 and m == E) else range(first, first + m)

    def score_sp(self, s: Tensor, p: Tensor, o: Tensor = None) -> Tensor:
        if not self._fused():
            return super().score_sp(s, p, o)
        ng = self._no_grad_call()
        if ng is not None:
            return engine.score_sp(ng[0], s, p, self._targets(o), flags=ng[1], padded=self._padded())
        ent, rel = self._w()
        return _ScorePairs.apply(self._scorer.name, self._scorer._norm, "sp", ent, rel, s, p, o,
                                 self._fwd_tables(), self._padded())

    def score_po(self, p: Tensor, o: Tensor, s: Tensor = None) -> Tensor:
        if not self._fused():
            return super().score_po(p, o, s)
        ng = self._no_grad_call()
        if ng is not None:
            return engine.score_po(ng[0], p, o, self._targets(s), flags=ng[1], padded=self._padded())
        ent, rel = self._w()
        return _ScorePairs.apply(self._scorer.name, self._scorer._norm, "po", ent, rel, o, p, s,
                                 self._fwd_tables(), self._padded())

    # 1vsAll loss fused with the scoring (HipTrainingJob1vsAll; kge_ce_fwd / kge_ce_bwd)
    def _ce_tables(self):
        if not self._fused() or self._scorer.name not in ("complex", "distmult"):
            return None
        ent, rel = self._w()
        t = self._fwd_tables()
        if t is None and ent.dtype == torch.bfloat16:
            t = engine.Tables(self._scorer.name, ent.detach(), rel.detach(), self._scorer._norm)
        return t if (t is not None and ent.is_cuda and engine.ce_supported(t)) else None

    def _rank_tables(self):
        """The tables HipEntityRankingJob counts on (kge_score_rank_sp_po: scoring + _filter_and_rank counts in one
        kernel) -- the ones score_sp_po scores on under no_grad, so that fused and two-step counts agree bit for bit:
        the bf16 copies under `score_dtype: bfloat16`, else the parameters themselves (float32: every scorer).  None:
        no fused gather (dropout active, embedders that are not plain lookup tables, job.device cpu)."""
        if not self._fused():
            return None
        ent, rel = self._w()
        return self._fwd_tables() or engine.Tables(self._scorer.name, ent.detach(), rel.detach(), self._scorer._norm)

    def loss_sp(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        """[n] cross entropy of score_sp(s, p) against o; None if the fused path does not apply."""
        t = self._ce_tables()
        dp = self._dropout_only() if t is None else None
        if t is None and dp is None:
            return None
        ent, rel = self._w()
        if dp is not None:  # embedder dropout in training: the masks applied here, the fused kernels on dense rows
            return ce_fused_dropout(self._scorer.name, self._scorer._norm, "sp", ent, rel, s, p, o, dp[0], dp[1])
        return _FusedCE.apply("sp", ent, rel, s, p, o, t)

    def loss_po(self, p: Tensor, o: Tensor, s: Tensor) -> Tensor:
        t = self._ce_tables()
        dp = self._dropout_only() if t is None else None
        if t is None and dp is None:
            return None
        ent, rel = self._w()
        if dp is not None:  # embedder dropout in training: the masks applied here, the fused kernels on dense rows
            return ce_fused_dropout(self._scorer.name, self._scorer._norm, "po", ent, rel, o, p, s, dp[0], dp[1])
        return _FusedCE.apply("po", ent, rel, o, p, s, t)

    def loss_sp_po(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        """[2n] loss_sp rows then loss_po rows from one pass over the batch; None if not applicable."""
        t = self._ce_tables()
        if t is None:
            if self._dropout_only() is None:
                return None
            # independent masks per direction, as the reference's two score_* calls draw them
            return torch.cat((self.loss_sp(s, p, o), self.loss_po(p, o, s)))
        ent, rel = self._w()
        return _FusedCE2.apply(ent, rel, s, p, o, t)

    def loss_sp_po_sum(self, s: Tensor, p: Tensor, o: Tensor, scale=None) -> Tensor:
        """0-d: scale * loss_sp_po(s, p, o).sum(), summed and back-propagated inside the library's launches
        (kge_amd.model._FusedCE2Sum; the step HipTrainingJob1vsAll captures); None if not applicable."""
        t = self._ce_tables()
        if t is None:
            rows = self.loss_sp_po(s, p, o)
            if rows is None:
                return None
            return rows.sum() if scale is None else rows.sum() * scale
        ent, rel = self._w()
        return _FusedCE2Sum.apply(ent, rel, s, p, o, t, scale)

    def kl_loss_sp(self, s: Tensor, p: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor,
                   label_smoothing: float = 0.0) -> Tensor:
        """[n] KL divergence of softmax(score_sp(s, p)) from the rows' normalised (and, with
        `label_smoothing`, smoothed: train_KvsAll.py:260-266) multi-hot labels (int64 CSR); None if the
        fused path does not apply (HipTrainingJobKvsAll; kge_kl_fwd / kge_kl_weighted_fwd)."""
        t = self._ce_tables()
        if t is None:
            return None
        ent, rel = self._w()
        return kl_fused(self._scorer.name, self._scorer._norm, "sp", ent, rel, s, p, lbl_rowptr, lbl_col,
                        float(label_smoothing), t)

    def kl_loss_po(self, p: Tensor, o: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor,
                   label_smoothing: float = 0.0) -> Tensor:
        t = self._ce_tables()
        if t is None:
            return None
        ent, rel = self._w()
        return kl_fused(self._scorer.name, self._scorer._norm, "po", ent, rel, o, p, lbl_rowptr, lbl_col,
                        float(label_smoothing), t)

    def bce_loss_sp(self, s: Tensor, p: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor, offset: float = 0.0,
                    label_smoothing: float = 0.0):
        """[n] sum over all entities of BCEWithLogits(score_sp(s, p) + offset, (smoothed) multi-hot labels);
        None if the fused path does not apply (kge_bce_fwd)."""
        t = self._ce_tables()
        if t is None:
            return None
        ent, rel = self._w()
        return bce_fused(self._scorer.name, self._scorer._norm, "sp", ent, rel, s, p, lbl_rowptr, lbl_col,
                         float(offset), float(label_smoothing), t)

    def bce_loss_po(self, p: Tensor, o: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor, offset: float = 0.0,
                    label_smoothing: float = 0.0):
        t = self._ce_tables()
        if t is None:
            return None
        ent, rel = self._w()
        return bce_fused(self._scorer.name, self._scorer._norm, "po", ent, rel, o, p, lbl_rowptr, lbl_col,
                         float(offset), float(label_smoothing), t)

    def multilabel_loss_sp_po(self, kind: str, s: Tensor, p_sp: Tensor, rowptr_sp: Tensor, col_sp: Tensor, o: Tensor,
                              p_po: Tensor, rowptr_po: Tensor, col_po: Tensor, offset: float = 0.0,
                              sum_scale: float = None):
        """(loss rows of the sp_ queries, loss rows of the _po queries) of a KvsAll subbatch -- or, with `sum_scale`,
        the 0-d sum_scale * (sum of all rows) -- with ONE backward for both types (kge_amd.model._FusedMultiLabel2: no
        label smoothing); None if the fused path does not apply."""
        t = self._ce_tables()
        if t is None:
            return None
        ent, rel = self._w()
        return _FusedMultiLabel2.apply(kind, float(offset), ent, rel, s, p_sp, rowptr_sp, col_sp, o, p_po, rowptr_po,
                                       col_po, t, sum_scale)

    def score_sp_po(self, s: Tensor, p: Tensor, o: Tensor, entity_subset: Tensor = None) -> Tensor:
        if not self._fused():
            return super().score_sp_po(s, p, o, entity_subset)
        ng = self._no_grad_call()
        if ng is not None:
            return engine.score_sp_po(ng[0], s, p, o, self._targets(entity_subset), flags=ng[1])
        return torch.cat((self.score_sp(s, p, entity_subset), self.score_po(p, o, entity_subset)), dim=1)


def _init(self, scorer, config, dataset, configuration_key, init_for_load_only):
    KgeModel.__init__(self, config=config, dataset=dataset, scorer=scorer,
                      configuration_key=configuration_key, init_for_load_only=init_for_load_only)


class HipComplEx(_FusedScoring, KgeModel):
    def __init__(self, config: Config, dataset: Dataset, configuration_key=None, init_for_load_only=False):
        _init(self, HipComplExScorer, config, dataset, configuration_key, init_for_load_only)


class HipDistMult(_FusedScoring, KgeModel):
    def __init__(self, config: Config, dataset: Dataset, configuration_key=None, init_for_load_only=False):
        _init(self, HipDistMultScorer, config, dataset, configuration_key, init_for_load_only)


class HipTransE(_FusedScoring, _RefTransE):
    """Inherits prepare_job (forces negative_sampling.implementation=triple, transe.py:58-68)."""

    def __init__(self, config: Config, dataset: Dataset, configuration_key=None, init_for_load_only=False):
        _init(self, HipTransEScorer, config, dataset, configuration_key, init_for_load_only)


class HipRotatE(_FusedScoring, _RefRotatE):
    """Inherits normalize_phases / prepare_job (rotate.py:103-143); the constructor repeats
    rotate.py:72-101 with the HIP scorer."""

    def __init__(self, config: Config, dataset: Dataset, configuration_key=None, init_for_load_only=False):
        self._init_configuration(config, configuration_key)
        if self.get_option("entity_embedder.dim") % 2 != 0:
            raise ValueError("RotatE requires embeddings of even dimensionality"
                             " (got {})".format(self.get_option("entity_embedder.dim")))
        if self.get_option("relation_embedder.dim") < 0:
            self.set_option("relation_embedder.dim", self.get_option("entity_embedder.dim") // 2, log=True)
        _init(self, HipRotatEScorer, config, dataset, self.configuration_key, init_for_load_only)
        self._normalize_phases = self.get_option("normalize_phases")


from kge.model.reciprocal_relations_model import ReciprocalRelationsModel as _RefReciprocal  # noqa: E402


class HipReciprocalRelationsModel(_RefReciprocal):
    """`model: hip_reciprocal_relations_model` over a hip_* base model -- the reference's wrapper
    (kge/model/reciprocal_relations_model.py: separate relation embeddings p and p + R for predicting objects and
    subjects, used by most of LibKGE's tuned 1vsAll / KvsAll configurations) with its scoring routed to the base model's
    fused INDEX-level calls.  The reference wrapper embeds the batch's rows and the whole entity table and calls
    `scorer.score_emb` on the dense rows (reciprocal_relations_model.py:84-131) -- correct over a hip_* base too, through
    the dense-row kernels and a full-table gather per call --; here
        score_po(p, o, s)        = base.score_sp(o, p + R, s)                       (reciprocal_relations_model.py:84-91)
        score_sp_po(s, p, o, E') = [base.score_sp(s, p, E') | base.score_sp(o, p + R, E')]          (:96-131)
    and the fused-loss hooks of the hip_* training jobs (loss_sp / loss_po / loss_sp_po, kl_loss_*, bce_loss_*) exist with
    the same translation: BOTH directions of a 1vsAll batch are sp_ queries of the base model -- rows (s, p | label o)
    and (o, p + R | label s) -- so `loss_sp_po` is ONE fused one-sided launch over 2 n rows, and `train.type: hip_1vsAll`
    (graph_step included) / `hip_KvsAll` run their fused paths on reciprocal models.  Where the base model's fused path
    does not apply (job.device cpu, other embedders, dropout) every call is the reference wrapper's own code.
    `eval.type: hip_entity_ranking` takes its two-step path (score_sp_po above + kge_rank_counts_multi): the counting
    kernel's second side is a _po query of ONE relation table, which a reciprocal model does not have."""

    def _R(self) -> int:
        return self.dataset.num_relations()

    def _base_fused(self) -> bool:
        b = self._base_model
        return isinstance(b, _FusedScoring) and b._fused()

    # ---- scores
    def score_sp(self, s: Tensor, p: Tensor, o: Tensor = None) -> Tensor:
        if not self._base_fused():
            return super().score_sp(s, p, o)
        return self._base_model.score_sp(s, p, o)

    def score_po(self, p: Tensor, o: Tensor, s: Tensor = None) -> Tensor:
        if not self._base_fused():
            return super().score_po(p, o, s)
        return self._base_model.score_sp(o, p + self._R(), s)

    def score_sp_po(self, s: Tensor, p: Tensor, o: Tensor, entity_subset: Tensor = None) -> Tensor:
        if not self._base_fused():
            return super().score_sp_po(s, p, o, entity_subset)
        b = self._base_model
        # (where no gradient is recorded: ONE range check for both calls -- a Python range passes through the base
        # model's own check untouched)
        sub = b._targets(entity_subset) if not torch.is_grad_enabled() else entity_subset
        return torch.cat((b.score_sp(s, p, sub), b.score_sp(o, p + self._R(), sub)), dim=1)

    def score_spo(self, s: Tensor, p: Tensor, o: Tensor, direction=None) -> Tensor:
        # (reciprocal_relations_model.py:74-82: "o" scores the triple as it stands, "s" the reversed triple with p + R)
        if not self._base_fused() or direction not in ("o", "s"):
            return super().score_spo(s, p, o, direction)
        if direction == "o":
            return self._base_model.score_spo(s, p, o, "o")
        return self._base_model.score_spo(o, p + self._R(), s, "o")

    def score_neg(self, s: Tensor, p: Tensor, o: Tensor, slot: int, neg: Tensor):
        """[n, K] scores of the positives with slot (0 = s, 2 = o) replaced by neg[i, k] -- the hook
        HipTrainingJobNegativeSampling hands a slot's samples to (BatchNegativeSample.score, kge/util/sampler.py:263-306,
        which asks score_spo(..., direction) of this wrapper): a corrupted SUBJECT is the corrupted object of the
        reversed triple (o, p + R, s').  None: the sampler's own code."""
        b = self._base_model
        if not self._base_fused() or not hasattr(b, "score_neg"):
            return None
        if slot == 2:
            return b.score_neg(s, p, o, 2, neg)
        if slot == 0:
            return b.score_neg(o, p + self._R(), s, 2, neg)
        return None

    # ---- the fused-loss hooks of HipTrainingJob1vsAll / HipTrainingJobKvsAll (train_job.py)
    def _ce_tables(self):
        f = getattr(self._base_model, "_ce_tables", None)
        return f() if f is not None else None

    def _dropout_only(self):
        f = getattr(self._base_model, "_dropout_only", None)
        return f() if f is not None else None

    def loss_sp(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        return self._base_model.loss_sp(s, p, o)

    def loss_po(self, p: Tensor, o: Tensor, s: Tensor) -> Tensor:
        return self._base_model.loss_sp(o, p + self._R(), s)

    def loss_sp_po(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        """[2n] loss rows, the sp_ queries first: one fused launch over the 2 n sp_ queries of the base model."""
        return self._base_model.loss_sp(torch.cat((s, o)), torch.cat((p, p + self._R())), torch.cat((o, s)))

    def kl_loss_sp(self, s, p, lbl_rowptr, lbl_col, label_smoothing: float = 0.0):
        return self._base_model.kl_loss_sp(s, p, lbl_rowptr, lbl_col, label_smoothing)

    def kl_loss_po(self, p, o, lbl_rowptr, lbl_col, label_smoothing: float = 0.0):
        return self._base_model.kl_loss_sp(o, p + self._R(), lbl_rowptr, lbl_col, label_smoothing)

    def bce_loss_sp(self, s, p, lbl_rowptr, lbl_col, offset: float = 0.0, label_smoothing: float = 0.0):
        return self._base_model.bce_loss_sp(s, p, lbl_rowptr, lbl_col, offset, label_smoothing)

    def bce_loss_po(self, p, o, lbl_rowptr, lbl_col, offset: float = 0.0, label_smoothing: float = 0.0):
        return self._base_model.bce_loss_sp(o, p + self._R(), lbl_rowptr, lbl_col, offset, label_smoothing)
