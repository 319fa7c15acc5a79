"""Host-side mirror of the reference's model API for the scoring hot path.

Same class names, method names, argument meaning and error behaviour as the reference
(kge/model/kge_model.py, kge/model/embedder/lookup_embedder.py, complex.py, distmult.py,
transe.py, rotate.py), with every score_* routed to the fused HIP kernels.  The reference
package itself cannot travel to the GPU box, so these classes are self-contained
torch.nn.Modules; `kge_amd.libkge_plugin` wraps the same kernels in subclasses of the
*reference's* KgeModel for use inside LibKGE (modules: [..., kge_amd.libkge_plugin]).

Parameter names/shapes equal the reference's (`_entity_embedder._embeddings.weight` [E,d],
`_relation_embedder._embeddings.weight` [R,d_r]) so state_dicts / checkpoints interoperate.
"""
import math

import torch
from torch import Tensor

from . import engine


class LookupEmbedder(torch.nn.Module):
    """kge/model/embedder/lookup_embedder.py:13-112 (table owner; dropout; normalize)."""

    def __init__(self, vocab_size: int, dim: int, dropout: float = 0.0, normalize_p: float = -1.0,
                 initialize: str = "normal_", initialize_args=None, sparse: bool = False,
                 dtype=torch.float32, device=None):
        super().__init__()
        self.vocab_size, self.dim = vocab_size, dim
        self.normalize_p = normalize_p
        self._embeddings = torch.nn.Embedding(vocab_size, dim, sparse=sparse, device=device, dtype=dtype)
        getattr(torch.nn.init, initialize)(self._embeddings.weight.data, **(initialize_args or {}))
        self._normalize_embeddings()
        self.dropout = torch.nn.Dropout(max(0.0, dropout))

    def _normalize_embeddings(self):
        if self.normalize_p > 0:
            with torch.no_grad():
                self._embeddings.weight.data = torch.nn.functional.normalize(
                    self._embeddings.weight.data, p=self.normalize_p, dim=-1)

    def embed(self, indexes: Tensor) -> Tensor:
        return self._postprocess(self._embeddings(indexes.long()))

    def embed_all(self) -> Tensor:
        # the reference gathers arange(E): a full-table copy (lookup_embedder.py:107-112);
        # the weight itself is returned here, the fused kernels read it in place.
        return self._postprocess(self._embeddings.weight)

    def _postprocess(self, embeddings: Tensor) -> Tensor:
        if self.dropout.p > 0:
            embeddings = self.dropout(embeddings)
        return embeddings

    @property
    def weight(self) -> Tensor:
        return self._embeddings.weight

    def fused_ok(self) -> bool:
        """Fused gather is valid when the embedder is a pure lookup (no active dropout)."""
        return not (self.training and self.dropout.p > 0)


class RelationalScorer(torch.nn.Module):
    """kge/model/kge_model.py:111-213: embedding-level scoring, combine in spo/sp_/_po/s_o."""

    name = None

    def __init__(self, l_norm: float = 1.0):
        super().__init__()
        self._norm = float(l_norm)

    def score_emb_spo(self, s_emb: Tensor, p_emb: Tensor, o_emb: Tensor) -> Tensor:
        return self.score_emb(s_emb, p_emb, o_emb, "spo")

    def score_emb(self, s_emb: Tensor, p_emb: Tensor, o_emb: Tensor, combine: str) -> Tensor:
        n = p_emb.size(0)
        if combine in ("spo", "sp_", "_po"):
            if combine == "spo":
                assert s_emb.size(0) == n and o_emb.size(0) == n
            return _ScoreEmb.apply(self.name, combine, self._norm, s_emb, p_emb, o_emb).view(n, -1)
        if combine == "s_o":  # generic fallback of the reference (kge_model.py:202-209)
            n = s_emb.size(0)
            assert o_emb.size(0) == n
            n_p = p_emb.size(0)
            s_embs = s_emb.repeat_interleave(n_p, 0)
            p_embs = p_emb.repeat((n, 1))
            o_embs = o_emb.repeat_interleave(n_p, 0)
            return _ScoreEmb.apply(self.name, "spo", self._norm, s_embs, p_embs, o_embs).view(n, -1)
        raise ValueError('cannot handle combine="{}"'.format(combine))


class ComplExScorer(RelationalScorer):
    name = "complex"


class DistMultScorer(RelationalScorer):
    name = "distmult"


class TransEScorer(RelationalScorer):
    name = "transe"


class RotatEScorer(RelationalScorer):
    name = "rotate"


class KgeModel(torch.nn.Module):
    """Index-level API of kge/model/kge_model.py:587-789 on the fused kernels."""

    scorer_cls = None

    def __init__(self, num_entities: int, num_relations: int, dim: int, l_norm: float = 1.0,
                 dtype=torch.float32, device=None, entity_args=None, relation_args=None,
                 score_dtype=None):
        super().__init__()
        # score_dtype=torch.bfloat16 with f32 parameters (ComplEx / DistMult): sp_/_po scores from
        # the bf16 matrix-core kernel on bf16 copies of the tables, gradients w.r.t. the f32 masters
        self._shadow = BF16Shadow() if (score_dtype == torch.bfloat16 and dtype == torch.float32 and
                                        self.scorer_cls in (ComplExScorer, DistMultScorer)) else None
        rel_dim = dim // 2 if self.scorer_cls is RotatEScorer else dim
        if self.scorer_cls in (RotatEScorer, ComplExScorer) and dim % 2:
            raise ValueError("{} requires embeddings of even dimensionality (got {})".format(
                type(self).__name__, dim))
        relation_args = dict(relation_args or {})
        if self.scorer_cls is RotatEScorer:  # rotate.yaml:22-26
            relation_args.setdefault("initialize", "uniform_")
            relation_args.setdefault("initialize_args", {"a": -math.pi, "b": math.pi})
        self._entity_embedder = LookupEmbedder(num_entities, dim, dtype=dtype, device=device,
                                               **(entity_args or {}))
        self._relation_embedder = LookupEmbedder(num_relations, rel_dim, dtype=dtype, device=device,
                                                 **relation_args)
        self._scorer = self.scorer_cls(l_norm)

    # -- accessors (kge_model.py:649-661)
    def get_s_embedder(self):
        return self._entity_embedder

    def get_o_embedder(self):
        return self._entity_embedder

    def get_p_embedder(self):
        return self._relation_embedder

    def get_scorer(self):
        return self._scorer

    def tables(self, flags: int = 0) -> engine.Tables:
        """The tables the pair scores are computed from (the bf16 copies in mixed precision)."""
        if self._shadow is not None and flags == 0:
            return self._fwd_tables()
        return engine.Tables(self._scorer.name, self._entity_embedder.weight.detach(),
                             self._relation_embedder.weight.detach(), self._scorer._norm, flags)

    def _fwd_tables(self):
        if self._shadow is None:
            return None
        return self._shadow.tables(self._scorer.name, self._entity_embedder.weight,
                                   self._relation_embedder.weight, self._scorer._norm)

    def _fused(self) -> bool:
        return self._entity_embedder.fused_ok() and self._relation_embedder.fused_ok()

    # -- scoring
    def score_spo(self, s: Tensor, p: Tensor, o: Tensor, direction=None) -> Tensor:
        if self._fused():
            return _ScoreSPO.apply(self._scorer.name, self._scorer._norm, self._entity_embedder.weight,
                                   self._relation_embedder.weight, s, p, o)
        se, pe, oe = self._entity_embedder.embed(s), self._relation_embedder.embed(p), self._entity_embedder.embed(o)
        return self._scorer.score_emb(se, pe, oe, combine="spo").view(-1)

    def score_neg(self, s: Tensor, p: Tensor, o: Tensor, slot: int, neg: Tensor) -> Tensor:
        """[n, K]: score of triple i with slot (0 = s, 2 = o) replaced by neg[i, k]; equals
        BatchNegativeSample.score with implementation "triple" (kge/util/sampler.py:291-306)."""
        if self._fused() and slot in (0, 2):
            return _ScoreNeg.apply(self._scorer.name, self._scorer._norm, self._entity_embedder.weight,
                                   self._relation_embedder.weight, s, p, o, int(slot), neg)
        K = neg.shape[1]
        tr = [x.reshape(-1).long().repeat_interleave(K) for x in (s, p, o)]
        tr[slot] = neg.reshape(-1).long()
        return self.score_spo(tr[0], tr[1], tr[2]).view(-1, K)

    def score_neg_blocks(self, s: Tensor, p: Tensor, o: Tensor, neg_s: Tensor = None, neg_o: Tensor = None):
        """(positives [n], [n, K_s] scores with the subject replaced by neg_s[i, k] or None, [n, K_o] with the object
        replaced or None): score_spo + score_neg per slot as ONE autograd node (one pair of table gradients in the
        backward); composed from the single calls where the fused node does not apply."""
        ent, rel = self._entity_embedder.weight, self._relation_embedder.weight
        if self._fused() and neg_blocks_fusable(ent, rel):
            return _ScoreNegBlocks.apply(self._scorer.name, self._scorer._norm, ent, rel, s, p, o, neg_s, neg_o)
        pos = self.score_spo(s, p, o)
        return (pos, None if neg_s is None else self.score_neg(s, p, o, 0, neg_s),
                None if neg_o is None else self.score_neg(s, p, o, 2, neg_o))

    # score_sp / score_po without a recorded gradient return the [:, :E] view of a matrix with sector-aligned rows
    # (engine.score_pitch) instead of a contiguous tensor: see the plugin's `padded_scores` option
    padded_scores = True

    def score_sp(self, s: Tensor, p: Tensor, o: Tensor = None) -> Tensor:
        if self._fused():
            return _ScorePairs.apply(self._scorer.name, self._scorer._norm, "sp",
                                     self._entity_embedder.weight,
                                     self._relation_embedder.weight, s, p, o, self._fwd_tables(),
                                     self.padded_scores and not torch.is_grad_enabled())
        se, pe = self._entity_embedder.embed(s), self._relation_embedder.embed(p)
        oe = self._entity_embedder.embed_all() if o is None else self._entity_embedder.embed(o)
        return self._scorer.score_emb(se, pe, oe, combine="sp_")

    def score_po(self, p: Tensor, o: Tensor, s: Tensor = None) -> Tensor:
        if self._fused():
            return _ScorePairs.apply(self._scorer.name, self._scorer._norm, "po",
                                     self._entity_embedder.weight,
                                     self._relation_embedder.weight, o, p, s, self._fwd_tables(),
                                     self.padded_scores and not torch.is_grad_enabled())
        se = self._entity_embedder.embed_all() if s is None else self._entity_embedder.embed(s)
        oe, pe = self._entity_embedder.embed(o), self._relation_embedder.embed(p)
        return self._scorer.score_emb(se, pe, oe, combine="_po")

    # -- 1vsAll loss: train_1vsAll.py:64-65 / 75-76 in one call
    def _ce_tables(self):
        """bf16 tables for the fused loss, or None (then the loss is composed from score_sp/_po)."""
        if not self._fused() or self._scorer.name not in ("complex", "distmult"):
            return None
        w = self._entity_embedder.weight
        t = self._fwd_tables()
        if t is None and w.dtype == torch.bfloat16:
            t = engine.Tables(self._scorer.name, w.detach(), self._relation_embedder.weight.detach())
        return t if (t is not None and w.is_cuda and engine.ce_supported(t)) else None

    def loss_sp(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        """Per-row cross entropy of score_sp(s, p) against the true objects o ([n]); its sum is the
        reference's `self.loss(scores_sp, triples[:, 2])` with train.loss=kl (loss.py:192-207)."""
        t = self._ce_tables()
        if t is not None:
            return _FusedCE.apply("sp", self._entity_embedder.weight, self._relation_embedder.weight, s, p, o, t)
        return torch.nn.functional.cross_entropy(self.score_sp(s, p), o.long(), reduction="none")

    def loss_sp_po(self, s: Tensor, p: Tensor, o: Tensor) -> Tensor:
        """[2n]: loss_sp(s, p, o) followed by loss_po(p, o, s) -- both directions of a 1vsAll batch
        from one scoring launch and one pair of gradient products (train_1vsAll.py:64-81 sums and
        back-propagates the two separately)."""
        t = self._ce_tables()
        if t is not None:
            return _FusedCE2.apply(self._entity_embedder.weight, self._relation_embedder.weight, s, p, o, t)
        return torch.cat([self.loss_sp(s, p, o), self.loss_po(p, o, s)])

    def loss_sp_po_sum(self, s: Tensor, p: Tensor, o: Tensor, scale=None) -> Tensor:
        """0-d: scale * loss_sp_po(s, p, o).sum() -- the batch loss of TrainingJob1vsAll (train_1vsAll.py:64-82 with
        scale = 1 / batch_size) -- summed inside the loss kernels' own launches, its backward taking the upstream
        gradient and `scale` (None, a float, or a float32 device scalar) as device scalars: three launches and three
        autograd nodes fewer per step than `.sum() * scale` around loss_sp_po."""
        t = self._ce_tables()
        if t is not None:
            return _FusedCE2Sum.apply(self._entity_embedder.weight, self._relation_embedder.weight, s, p, o, t, scale)
        total = self.loss_sp_po(s, p, o).sum()
        return total if scale is None else total * scale

    def loss_po(self, p: Tensor, o: Tensor, s: Tensor) -> Tensor:
        """Per-row cross entropy of score_po(p, o) against the true subjects s."""
        t = self._ce_tables()
        if t is not None:
            return _FusedCE.apply("po", self._entity_embedder.weight, self._relation_embedder.weight, o, p, s, t)
        return torch.nn.functional.cross_entropy(self.score_po(p, o), s.long(), reduction="none")

    # -- KvsAll loss: train_KvsAll.py:274-294 with train.loss=kl
    @staticmethod
    def _kl_composed(scores: Tensor, rowptr: Tensor, col: Tensor, label_smoothing: float = 0.0) -> Tensor:
        """loss.py:208-213 row by row: KLDivLoss(log_softmax(scores), normalize(labels, p=1)); labels
        smoothed as train_KvsAll.py:260-266 does."""
        n = scores.shape[0]
        cnt = (rowptr[1:] - rowptr[:-1]).to(scores.device)
        rows = torch.repeat_interleave(torch.arange(n, device=scores.device), cnt)
        labels = torch.zeros_like(scores)
        labels[rows, col.to(scores.device).long()] = 1.0
        if label_smoothing > 0.0:
            labels = (1.0 - label_smoothing) * labels + 1.0 / labels.size(1)
        y = torch.nn.functional.normalize(labels, p=1, dim=1)
        return torch.nn.functional.kl_div(torch.log_softmax(scores, dim=1), y, reduction="none").sum(dim=1)

    def _kl_fused(self, direction: str, a: Tensor, p: Tensor, rowptr: Tensor, col: Tensor, eps: float, t) -> Tensor:
        return kl_fused(self._scorer.name, self._scorer._norm, direction, self._entity_embedder.weight,
                        self._relation_embedder.weight, a, p, rowptr, col, eps, t)

    def kl_loss_sp(self, s: Tensor, p: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor,
                   label_smoothing: float = 0.0) -> Tensor:
        """Per-row KL divergence of softmax(score_sp(s, p)) from the normalised multi-hot labels
        given as an int64 CSR over the rows (the known objects of each (s, p) query)."""
        t = self._ce_tables()
        if t is not None:
            return self._kl_fused("sp", s, p, lbl_rowptr, lbl_col, float(label_smoothing), t)
        return self._kl_composed(self.score_sp(s, p), lbl_rowptr, lbl_col, label_smoothing)

    def kl_loss_po(self, p: Tensor, o: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor,
                   label_smoothing: float = 0.0) -> Tensor:
        t = self._ce_tables()
        if t is not None:
            return self._kl_fused("po", o, p, lbl_rowptr, lbl_col, float(label_smoothing), t)
        return self._kl_composed(self.score_po(p, o), lbl_rowptr, lbl_col, label_smoothing)

    def multilabel_loss_sp_po(self, kind: str, s: Tensor, p_sp: Tensor, rowptr_sp: Tensor, col_sp: Tensor, o: Tensor,
                              p_po: Tensor, rowptr_po: Tensor, col_po: Tensor, offset: float = 0.0,
                              label_smoothing: float = 0.0, sum_scale: float = None):
        """(loss rows of the sp_ queries (s, p_sp), loss rows of the _po queries (p_po, o)) of a KvsAll batch, `kind`
        "kl" or "bce": kl_loss_sp + kl_loss_po (bce_loss_sp + bce_loss_po) with ONE backward for both types where the
        fused path applies (_FusedMultiLabel2; no label smoothing), else the two per-type calls.  `sum_scale` given:
        the 0-d batch loss sum_scale * (sum of all rows) instead of the two row vectors."""
        t = self._ce_tables()
        if t is not None and label_smoothing == 0.0:
            return _FusedMultiLabel2.apply(kind, float(offset), self._entity_embedder.weight, self._relation_embedder.weight,
                                           s, p_sp, rowptr_sp, col_sp, o, p_po, rowptr_po, col_po, t, sum_scale)
        if kind == "kl":
            both = (self.kl_loss_sp(s, p_sp, rowptr_sp, col_sp, label_smoothing),
                    self.kl_loss_po(p_po, o, rowptr_po, col_po, label_smoothing))
        else:
            both = (self.bce_loss_sp(s, p_sp, rowptr_sp, col_sp, offset, label_smoothing),
                    self.bce_loss_po(p_po, o, rowptr_po, col_po, offset, label_smoothing))
        return both if sum_scale is None else (both[0].sum() + both[1].sum()) * float(sum_scale)

    # -- bce loss (train.loss: bce, loss.py:137-159 with bce_type None) on multi-hot labels
    @staticmethod
    def _bce_composed(scores: Tensor, rowptr: Tensor, col: Tensor, offset: float = 0.0,
                      label_smoothing: float = 0.0) -> Tensor:
        n = scores.shape[0]
        cnt = (rowptr[1:] - rowptr[:-1]).to(scores.device)
        rows = torch.repeat_interleave(torch.arange(n, device=scores.device), cnt)
        labels = torch.zeros_like(scores)
        labels[rows, col.to(scores.device).long()] = 1.0
        if label_smoothing > 0.0:
            labels = (1.0 - label_smoothing) * labels + 1.0 / labels.size(1)
        return torch.nn.functional.binary_cross_entropy_with_logits(scores + offset, labels,
                                                                     reduction="none").sum(dim=1)

    def bce_loss_sp(self, s: Tensor, p: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor, offset: float = 0.0,
                    label_smoothing: float = 0.0) -> Tensor:
        """Per-row sum over all entities of BCEWithLogits(score_sp(s, p) + offset, multi-hot labels)."""
        t = self._ce_tables()
        if t is not None:
            return bce_fused(self._scorer.name, self._scorer._norm, "sp", self._entity_embedder.weight,
                             self._relation_embedder.weight, s, p, lbl_rowptr, lbl_col, float(offset),
                             float(label_smoothing), t)
        return self._bce_composed(self.score_sp(s, p), lbl_rowptr, lbl_col, offset, label_smoothing)

    def bce_loss_po(self, p: Tensor, o: Tensor, lbl_rowptr: Tensor, lbl_col: Tensor, offset: float = 0.0,
                    label_smoothing: float = 0.0) -> Tensor:
        t = self._ce_tables()
        if t is not None:
            return bce_fused(self._scorer.name, self._scorer._norm, "po", self._entity_embedder.weight,
                             self._relation_embedder.weight, o, p, lbl_rowptr, lbl_col, float(offset),
                             float(label_smoothing), t)
        return self._bce_composed(self.score_po(p, o), lbl_rowptr, lbl_col, offset, label_smoothing)

    def score_so(self, s: Tensor, o: Tensor, p: Tensor = None) -> Tensor:
        se, oe = self._entity_embedder.embed(s), self._entity_embedder.embed(o)
        pe = self._relation_embedder.embed_all() if p is None else self._relation_embedder.embed(p)
        return self._scorer.score_emb(se, pe, oe, combine="s_o")

    def score_sp_po(self, s: Tensor, p: Tensor, o: Tensor, entity_subset: Tensor = None) -> Tensor:
        if self._fused() and not torch.is_grad_enabled():
            return engine.score_sp_po(self.tables(), s, p, o, entity_subset)
        return torch.cat((self.score_sp(s, p, entity_subset), self.score_po(p, o, entity_subset)), dim=1)


class ComplEx(KgeModel):
    scorer_cls = ComplExScorer


class DistMult(KgeModel):
    scorer_cls = DistMultScorer


class TransE(KgeModel):
    scorer_cls = TransEScorer


class RotatE(KgeModel):
    scorer_cls = RotatEScorer

    @torch.no_grad()
    def normalize_phases(self):
        """rotate.py:103-118: wrap phases to [-pi, pi)."""
        w = self._relation_embedder.weight.data
        w[:] = torch.remainder(w + math.pi, 2.0 * math.pi) - math.pi


MODELS = {"complex": ComplEx, "distmult": DistMult, "transe": TransE, "rotate": RotatE}


def create(model: str, num_entities: int, num_relations: int, dim: int, **kw) -> KgeModel:
    return MODELS[model](num_entities, num_relations, dim, **kw)


# ---- autograd glue (the reference gets backward from torch autograd; here the twins in
# ---- csrc/bwd.hip are called explicitly) --------------------------------------------------
def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t.requires_grad for t in ts)


def _scatter_rows(grad_table, idx, rows):
    grad_table.index_add_(0, idx.reshape(-1).long(), rows)


class _ScoreSPO(torch.autograd.Function):
    @staticmethod
    def forward(ctx, name, l_norm, ent, rel, s, p, o):
        t = engine.Tables(name, ent.detach(), rel.detach(), l_norm)
        ctx.t, ctx.idx = t, (s, p, o)
        out = engine.score_spo(t, s, p, o)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        s, p, o = ctx.idx
        (scores,) = ctx.saved_tensors
        ge, gr = torch.zeros_like(ctx.t.ent), torch.zeros_like(ctx.t.rel)
        # one kernel: row gradients accumulated into the table gradients (runs of equal s / p
        # summed in registers first); fallback: row gradients + three scatter-adds
        if not engine.score_spo_bwd_accum(ctx.t, s, p, o, gout.contiguous(), scores, ge, gr):
            g_s, g_p, g_o = engine.score_spo_bwd(ctx.t, s, p, o, gout.contiguous(), scores)
            _scatter_rows(ge, s, g_s)
            _scatter_rows(ge, o, g_o)
            _scatter_rows(gr, p, g_p)
        return None, None, ge, gr, None, None, None


class _ScoreNeg(torch.autograd.Function):
    """[n, K] scores of the positives with `slot` (0 = s, 2 = o) replaced by neg[i, k] -- what
    BatchNegativeSample.score computes with implementation "triple" (sampler.py:291-306) -- from
    kge_score_neg; backward = kge_score_neg_bwd_accum (complete table gradients, no [n*K, 3] index
    tensor and no [n*K, d] gathered rows in either pass)."""

    @staticmethod
    def forward(ctx, name, l_norm, ent, rel, s, p, o, slot, neg):
        t = engine.Tables(name, ent.detach(), rel.detach(), l_norm)
        out = engine.score_neg(t, s, p, o, slot, neg)
        ctx.t, ctx.idx, ctx.slot = t, (s, p, o, neg), slot
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        s, p, o, neg = ctx.idx
        (scores,) = ctx.saved_tensors
        ge, gr = torch.zeros_like(ctx.t.ent), torch.zeros_like(ctx.t.rel)
        if not engine.score_neg_bwd_accum(ctx.t, s, p, o, ctx.slot, neg, gout, scores, ge, gr):
            # shapes the accumulate kernel does not take (bf16 parameters, d > 1024): expanded triples
            K = neg.shape[1]
            tr = [x.reshape(-1).long().repeat_interleave(K) for x in (s, p, o)]
            tr[ctx.slot] = neg.reshape(-1).long()
            g_s, g_p, g_o = engine.score_spo_bwd(ctx.t, tr[0], tr[1], tr[2], gout.reshape(-1).contiguous(),
                                                 scores.reshape(-1))
            ge, gr = ge.float(), gr.float()
            _scatter_rows(ge, tr[0], g_s)
            _scatter_rows(ge, tr[2], g_o)
            _scatter_rows(gr, tr[1], g_p)
            ge, gr = ge.to(ctx.t.ent.dtype), gr.to(ctx.t.rel.dtype)
        return None, None, ge, gr, None, None, None, None, None


class _ScoreNegBlocks(torch.autograd.Function):
    """What TrainingJobNegativeSampling._process_subbatch scores for a subbatch (train_negative_sampling.py:120-151) in
    ONE autograd node: the positives (score_spo) and the negative blocks of the subject and the object slot
    (BatchNegativeSample.score, sampler.py:263-306).  Its backward accumulates all three into ONE pair of dense table
    gradients (kge_score_spo_bwd_accum + kge_score_neg_bwd_accum per slot on the same buffers) -- composed from
    _ScoreSPO / _ScoreNeg every node brings its own zero-filled [E, d] gradient and autograd adds them up: six fills and
    six adds of the entity table's size per step (0.2 of a 1.6 ms step at the WN18RR shape).  float32 tables, dim <= 1024
    (the accumulate kernels' domain; the caller checks)."""

    @staticmethod
    def forward(ctx, name, l_norm, ent, rel, s, p, o, neg_s, neg_o):
        t = engine.Tables(name, ent.detach(), rel.detach(), l_norm)
        pos = engine.score_spo(t, s, p, o)
        sc_s = engine.score_neg(t, s, p, o, 0, neg_s) if neg_s is not None else None
        sc_o = engine.score_neg(t, s, p, o, 2, neg_o) if neg_o is not None else None
        ctx.t, ctx.idx = t, (s, p, o, neg_s, neg_o)
        ctx.save_for_backward(pos, sc_s, sc_o)
        return pos, sc_s, sc_o

    @staticmethod
    def backward(ctx, g_pos, g_s, g_o):
        s, p, o, neg_s, neg_o = ctx.idx
        pos, sc_s, sc_o = ctx.saved_tensors
        ge, gr = torch.zeros_like(ctx.t.ent), torch.zeros_like(ctx.t.rel)
        if g_pos is not None and not engine.score_spo_bwd_accum(ctx.t, s, p, o, g_pos.contiguous(), pos, ge, gr):
            raise RuntimeError("kge_amd: kge_score_spo_bwd_accum declined a shape score_neg_blocks admitted")
        for slot, neg, g, sc in ((0, neg_s, g_s, sc_s), (2, neg_o, g_o, sc_o)):
            if neg is not None and g is not None:
                if not engine.score_neg_bwd_accum(ctx.t, s, p, o, slot, neg, g, sc, ge, gr):
                    raise RuntimeError("kge_amd: kge_score_neg_bwd_accum declined a shape score_neg_blocks admitted")
        return None, None, ge, gr, None, None, None, None, None


def neg_blocks_fusable(ent: torch.Tensor, rel: torch.Tensor) -> bool:
    return (ent.is_cuda and ent.dtype == torch.float32 and rel.dtype == torch.float32 and ent.shape[1] <= 1024
            and ent.is_contiguous() and rel.is_contiguous())


def _bf16_copy_of(param):
    from .optim import bf16_copy_of  # (copy, version, data_ptr of the master) must all still match
    return bf16_copy_of(param)


class BF16Shadow:
    """bf16 copies of f32 master tables for mixed-precision scoring (`score_dtype: bfloat16`):
    the forward runs the bf16 matrix-core kernel on the copies, the backward differentiates the
    f32 masters.  With autograd enabled (training) the copies are re-cast on every call (two
    15 MB writes at the FB15k-237 shape, ~10 us); without (evaluation) only when a master's
    storage or version counter changed -- in-place edits through `.data` do not bump the
    counter, hence no caching while training."""

    def __init__(self):
        self._key, self._tables = None, None

    def tables(self, name, ent, rel, l_norm) -> "engine.Tables":
        # copies written by the optimizer's own pass over the tables (kge_amd.optim.Adagrad with
        # bf16_copies=True) are fresh by construction: no cast at all
        e16, r16 = _bf16_copy_of(ent), _bf16_copy_of(rel)
        if e16 is not None and r16 is not None:
            key = (e16.data_ptr(), ent._version, ent.data_ptr(), r16.data_ptr(), rel._version, rel.data_ptr(), "opt")
            if key != self._key:
                self._tables, self._key = engine.Tables(name, e16, r16, l_norm), key
            return self._tables
        key = (ent.data_ptr(), ent._version, rel.data_ptr(), rel._version)
        if key != self._key or torch.is_grad_enabled():
            self._tables = engine.Tables(name, ent.detach().to(torch.bfloat16), rel.detach().to(torch.bfloat16),
                                         l_norm)
            self._key = key
        return self._tables


class _ScorePairs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, name, l_norm, direction, ent, rel, a, p, targets, fwd_tables=None, padded=False):
        t = engine.Tables(name, ent.detach(), rel.detach(), l_norm)
        tf = t if fwd_tables is None else fwd_tables  # mixed precision: bf16 copies, forward only
        # padded (the models' `padded_scores` option, honoured where no gradient is recorded): the scores as the [:, :m]
        # view of a matrix with sector-aligned rows (engine.score_pitch)
        out = (engine.score_sp if direction == "sp" else engine.score_po)(tf, *((a, p) if direction == "sp" else (p, a)), targets,
                                                                          padded=bool(padded))
        ctx.t, ctx.direction, ctx.idx = t, direction, (a, p, targets)
        ctx.tf = fwd_tables
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, p, targets = ctx.idx
        (scores,) = ctx.saved_tensors
        # mixed precision: both gradient products on the bf16 matrix cores too (bf16 copies of the
        # tables, gout rounded to bf16, f32 accumulation) -- autocast semantics
        g_a, g_p, g_t = engine.score_pairs_bwd(ctx.tf if ctx.tf is not None else ctx.t, ctx.direction, a, p,
                                               targets, gout, scores)
        gr = torch.zeros_like(ctx.t.rel)
        _scatter_rows(gr, p, g_p)
        if targets is None:
            ge = g_t  # [E, d], fresh: the dense target gradient IS the table gradient, plus the query rows
        else:
            ge = torch.zeros_like(ctx.t.ent)
            _scatter_rows(ge, targets, g_t)
        _scatter_rows(ge, a, g_a)
        return None, None, None, ge, gr, None, None, None, None, None


class _FusedCE(torch.autograd.Function):
    """Per-row 1vsAll cross entropy fused with the sp_/_po scoring (kge_ce_fwd / kge_ce_bwd):
    `tables16` are the bf16 tables the scores come from (the parameters themselves, or their
    bf16 copies in mixed precision); gradients go to `ent` / `rel`."""

    @staticmethod
    def forward(ctx, direction, ent, rel, a, p, label, tables16):
        loss_rows, lse = engine.ce_fwd(tables16, direction, a, p, label)
        ctx.t16, ctx.direction, ctx.idx = tables16, direction, (a, p, label)
        ctx.rel_shape = rel.shape
        ctx.save_for_backward(lse)
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        a, p, label = ctx.idx
        (lse,) = ctx.saved_tensors
        g_a, g_p, ge = engine.ce_bwd(ctx.t16, ctx.direction, a, p, label, lse, g_rows=g_rows.contiguous())
        gr = torch.zeros(ctx.rel_shape, dtype=torch.float32, device=ge.device)
        _scatter_rows(gr, p, g_p)
        _scatter_rows(ge, a, g_a)  # ge [E, d] is fresh: the dense target gradient + the query rows
        return None, ge, gr, None, None, None, None


class _FusedCE2(torch.autograd.Function):
    """Both directions of a 1vsAll batch (kge_ce_sp_po_fwd / _bwd): [2n] loss rows, sp_ queries
    first; one scoring launch and one pair of gradient products for the whole batch."""

    @staticmethod
    def forward(ctx, ent, rel, s, p, o, tables16):
        loss_rows, lse = engine.ce_sp_po_fwd(tables16, s, p, o)
        ctx.t16, ctx.idx, ctx.rel_shape = tables16, (s, p, o), rel.shape
        ctx.save_for_backward(lse)
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        s, p, o = ctx.idx
        (lse,) = ctx.saved_tensors
        # complete table gradients from the library (the scatter-add of the gathered rows included)
        ge, gr = engine.ce_sp_po_bwd_accum(ctx.t16, s, p, o, lse, g_rows=g_rows.contiguous())
        return ge, gr, None, None, None, None


class _FusedCE2Sum(torch.autograd.Function):
    """scale * sum of _FusedCE2's rows as a 0-d tensor (kge_ce_sp_po_fwd_sum / kge_ce_sp_po_bwd_accum_sum): the sum
    leaves the combine launch, the backward reads the upstream gradient and the scale from device memory.  `scale`
    is a constant of the graph (no gradient flows into it)."""

    @staticmethod
    def forward(ctx, ent, rel, s, p, o, tables16, scale):
        # keep_queries: the forward's build launch also leaves the gradient products' query matrix in the workspace, and
        # the backward -- if it is the NEXT call on that workspace (the generation count says so: a training step's
        # loss.backward() right behind the loss; always true inside a captured step) -- starts from the forward's query
        # fragments instead of building them again: one launch less per step
        total, _rows, lse = engine.ce_sp_po_fwd_sum(tables16, s, p, o, scale, keep_queries=True)
        ctx.t16, ctx.idx, ctx.scale = tables16, (s, p, o), scale
        ctx.generation = engine.ce2_generation(tables16.device)
        ctx.versions = tuple(x._version for x in (s, p, o) if torch.is_tensor(x))
        ctx.save_for_backward(lse)
        return total

    @staticmethod
    def backward(ctx, gout):
        s, p, o = ctx.idx
        (lse,) = ctx.saved_tensors
        g = gout if gout.dtype == torch.float32 and gout.is_contiguous() else gout.float().contiguous()
        kept = (engine.ce2_generation(ctx.t16.device) == ctx.generation
                and ctx.versions == tuple(x._version for x in (s, p, o) if torch.is_tensor(x)))
        ge, gr = engine.ce_sp_po_bwd_accum_sum(ctx.t16, s, p, o, lse, g=g, scale=ctx.scale, keep_queries=kept)
        return ge, gr, None, None, None, None, None


# ---- embedder dropout inside the fused 1vsAll loss ------------------------------------------------------------------
# LookupEmbedder._postprocess (kge/model/embedder/lookup_embedder.py:64-69, 102-105) applies torch.nn.Dropout to the
# rows it returns: independently to the gathered query rows (embed), to the gathered relation rows and to ALL entity
# rows (embed_all) -- score_sp(s, p) of a model in training mode scores dropout(E[s]) (x) dropout(R[p]) against
# dropout(E).  Every tuned LibKGE config sets entity / relation dropout, so a fused loss that declines it trains those
# configs on the unfused path.  Here the three masks are applied BEFORE the fused kernels -- on the float32 rows and
# on a masked copy of the table (one elementwise pass over [E, d]: ~15 MB against the [n, E] score matrix the fused loss
# does not write) --, the kernels take the dense rows (kge_ce_emb_fwd / _bwd) and torch's autograd carries the kernel's
# gradients back through the masks and the gathers.  The masks are torch's (Philox on the device) unless handed in.
def embedder_dropout(x: Tensor, p: float, mask: Tensor = None) -> Tensor:
    """torch.nn.functional.dropout(x, p, training=True) with the mask optionally given: mask (same shape, 0 / 1) keeps
    the elements where it is 1 and scales them by 1 / (1 - p)."""
    if p <= 0.0:
        return x
    if mask is None:
        return torch.nn.functional.dropout(x, p, True)
    return x * (mask.to(x.dtype) * (1.0 / (1.0 - p)))


class _FusedCEEmb(torch.autograd.Function):
    """Per-row 1vsAll cross entropy of dense query rows against all rows of a dense table (kge_ce_emb_fwd / _bwd): the
    inputs are the float32 (dropped-out) rows; they are rounded to bf16 for the matrix-core kernels inside, gradients
    come back in float32 w.r.t. the inputs."""

    @staticmethod
    def forward(ctx, scorer, l_norm, direction, table, a_rows, p_rows, label):
        t16 = engine.Tables(scorer, table.detach().to(torch.bfloat16), p_rows.detach().to(torch.bfloat16), l_norm)
        a16 = a_rows.detach().to(torch.bfloat16).contiguous()
        loss_rows, lse = engine.ce_emb_fwd(t16, direction, a16, t16.rel, label)
        ctx.t16, ctx.direction, ctx.label = t16, direction, label
        ctx.save_for_backward(a16, lse)
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        a16, lse = ctx.saved_tensors
        g_a, g_p, g_t = engine.ce_emb_bwd(ctx.t16, ctx.direction, a16, ctx.t16.rel, ctx.label, lse,
                                          g_rows=g_rows.contiguous())
        return None, None, None, g_t, g_a, g_p, None


def ce_fused_dropout(scorer: str, l_norm: float, direction: str, ent: Tensor, rel: Tensor, a: Tensor, p: Tensor,
                     label: Tensor, p_ent: float, p_rel: float, masks: dict = None) -> Tensor:
    """[n] cross entropy of score_sp(a, p) ("sp") / score_po(p, a) ("po") over all entities with the embedders'
    dropout applied as the reference applies it (three independent masks: "a" [n, d], "p" [n, d_r], "all" [E, d];
    `masks` may hand any of them in).  Differentiable w.r.t. the float32 tables `ent` / `rel`."""
    masks = masks or {}
    a_rows = embedder_dropout(ent[a.reshape(-1).long()], p_ent, masks.get("a"))
    p_rows = embedder_dropout(rel[p.reshape(-1).long()], p_rel, masks.get("p"))
    table = embedder_dropout(ent, p_ent, masks.get("all"))
    return _FusedCEEmb.apply(scorer, l_norm, direction, table, a_rows, p_rows, label)


class _FusedKL(torch.autograd.Function):
    """Per-row KvsAll KL loss fused with the sp_/_po scoring (kge_kl_fwd / kge_kl_bwd); labels as an
    int64 CSR (rowptr [n + 1], col [nnz]) of the rows' known answers."""

    @staticmethod
    def forward(ctx, direction, ent, rel, a, p, rowptr, col, label_weight, label_bias, tables16):
        loss_rows, lse = engine.kl_fwd(tables16, direction, a, p, rowptr, col, label_weight)
        ctx.t16, ctx.direction, ctx.idx = tables16, direction, (a, p, rowptr, col, label_weight, label_bias)
        ctx.rel_shape = rel.shape
        ctx.save_for_backward(lse)
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        a, p, rowptr, col, label_weight, label_bias = ctx.idx
        (lse,) = ctx.saved_tensors
        g_a, g_p, ge = engine.kl_bwd(ctx.t16, ctx.direction, a, p, rowptr, col, lse, g_rows=g_rows.contiguous(),
                                     label_weight=label_weight, label_bias=label_bias)
        gr = torch.zeros(ctx.rel_shape, dtype=torch.float32, device=ge.device)
        _scatter_rows(gr, p, g_p)
        _scatter_rows(ge, a, g_a)
        return None, ge, gr, None, None, None, None, None, None, None


class _FusedMultiLabel2(torch.autograd.Function):
    """Both query types of a KvsAll batch: (loss rows of the sp_ queries, loss rows of the _po queries), each from its
    fused forward (kge_kl_fwd / kge_bce_fwd), and ONE backward for both (kge_multilabel2_bwd_accum): two d loss / d score
    passes into one gradient matrix, the two gradient products once over all rows, the gathered rows' gradients
    scattered by the library -- where two _FusedKL nodes run four products and leave two index_add passes and a
    gradient accumulation over [E, d] to autograd (train_KvsAll.py:274-294 back-propagates the two losses separately:
    the same gradients, accumulated).  No label smoothing (its extra terms are torch ops around the per-type nodes)."""

    @staticmethod
    def forward(ctx, kind, offset, ent, rel, s, p_sp, rowptr_sp, col_sp, o, p_po, rowptr_po, col_po, tables16,
                sum_scale=None):
        # sum_scale (a float): ONE output, sum_scale * (sum of all loss rows) -- the batch loss of the training job; its
        # backward hands the upstream gradient to the kernels as a device scalar (no expand / copy / scale launches)
        lse = (None, None)
        if kind == "kl":
            rows_sp, lse_sp = engine.kl_fwd(tables16, "sp", s, p_sp, rowptr_sp, col_sp, None)
            rows_po, lse_po = engine.kl_fwd(tables16, "po", o, p_po, rowptr_po, col_po, None)
            lse = (lse_sp, lse_po)
        else:
            rows_sp = engine.bce_fwd(tables16, "sp", s, p_sp, rowptr_sp, col_sp, offset)
            rows_po = engine.bce_fwd(tables16, "po", o, p_po, rowptr_po, col_po, offset)
        ctx.t16, ctx.kind, ctx.offset, ctx.lse, ctx.sum_scale = tables16, kind, offset, lse, sum_scale
        ctx.idx = ((s, p_sp, rowptr_sp, col_sp), (o, p_po, rowptr_po, col_po))
        if sum_scale is not None:
            return (rows_sp.sum() + rows_po.sum()) * float(sum_scale)
        return rows_sp, rows_po

    @staticmethod
    def backward(ctx, *gout):
        sp, po = ctx.idx
        if ctx.sum_scale is not None:
            g = gout[0] if gout[0].dtype == torch.float32 and gout[0].is_contiguous() else gout[0].float().contiguous()
            ge, gr = engine.multilabel2_bwd_accum(ctx.t16, ctx.kind, ctx.offset, sp + (ctx.lse[0], None),
                                                  po + (ctx.lse[1], None), g=g, scale=float(ctx.sum_scale))
        else:
            ge, gr = engine.multilabel2_bwd_accum(ctx.t16, ctx.kind, ctx.offset, sp + (ctx.lse[0], gout[0].contiguous()),
                                                  po + (ctx.lse[1], gout[1].contiguous()))
        return (None, None, ge, gr) + (None,) * 10


def kl_fused(name: str, l_norm, direction: str, ent: Tensor, rel: Tensor, a: Tensor, p: Tensor, rowptr: Tensor,
             col: Tensor, eps: float, t) -> Tensor:
    """The fused KvsAll KL loss with label smoothing `eps` (train_KvsAll.py:260-266: labels =
    (1 - eps) * multi_hot + 1/E before loss.py:208-213 normalises them).  The smoothed label row is a_i on
    row i's k_i labels and b_i elsewhere (Z_i = (1 - eps) k_i + 1, a_i = (1 - eps + 1/E) / Z_i,
    b_i = (1/E) / Z_i), so
        KL_i = lse_i - (a_i - b_i) * sum_{labels} score_ij       <- kge_kl_weighted_fwd, fused
               - b_i * sum_j score_ij                            <- linear in the entity table (ComplEx and
                                                                    DistMult, the fused path's scorers): ONE
                                                                    [n, 1] score against the table's column sum
               + k_i a_i log a_i + (E - k_i) b_i log b_i         <- constant."""
    if eps == 0.0:
        return _FusedKL.apply(direction, ent, rel, a, p, rowptr, col, None, None, t)
    E = ent.shape[0]
    k = (rowptr[1:] - rowptr[:-1]).to(device=ent.device, dtype=torch.float32)
    Z = (1.0 - eps) * k + 1.0
    a_w, b_w = (1.0 - eps + 1.0 / E) / Z, (1.0 / E) / Z
    # the GRADIENT of the uniform term is taken inside the gradient kernel (label_bias = b_i: b_i is subtracted at
    # every column next to the softmax), so only its VALUE is computed here
    fused = _FusedKL.apply(direction, ent, rel, a, p, rowptr, col, (a_w - b_w).contiguous(), b_w.contiguous(), t)
    with torch.no_grad():
        s_all = _sum_of_scores_value(name, direction, ent, rel, a, p)
    const = k * a_w * torch.log(a_w) + (E - k) * b_w * torch.log(b_w)
    return fused - b_w * s_all + const


class _FusedBCE(torch.autograd.Function):
    """Per-row BCE-with-logits (summed over all entities) fused with the sp_/_po scoring
    (kge_bce_fwd / kge_bce_bwd); labels as an int64 CSR of the rows' positives."""

    @staticmethod
    def forward(ctx, direction, ent, rel, a, p, rowptr, col, offset, tables16):
        loss_rows = engine.bce_fwd(tables16, direction, a, p, rowptr, col, offset)
        ctx.t16, ctx.direction, ctx.idx, ctx.offset = tables16, direction, (a, p, rowptr, col), offset
        ctx.rel_shape = rel.shape
        return loss_rows

    @staticmethod
    def backward(ctx, g_rows):
        a, p, rowptr, col = ctx.idx
        g_a, g_p, ge = engine.bce_bwd(ctx.t16, ctx.direction, a, p, rowptr, col, ctx.offset,
                                      g_rows=g_rows.contiguous())
        gr = torch.zeros(ctx.rel_shape, dtype=torch.float32, device=ge.device)
        _scatter_rows(gr, p, g_p)
        _scatter_rows(ge, a, g_a)
        return None, ge, gr, None, None, None, None, None, None


def _score_against_column_sum(name: str, l_norm, direction: str, ent: Tensor, rel: Tensor, a: Tensor,
                              p: Tensor) -> Tensor:
    """[n]: sum_j score(i, j) over ALL entities j.  ComplEx and DistMult are linear in the target row, so
    this is one score of each query against the entity table's column sum."""
    colsum = ent.float().sum(dim=0, keepdim=True)
    rows, prow = ent[a.long()].float(), rel[p.long()].float()
    if direction == "sp":
        return _ScoreEmb.apply(name, "sp_", l_norm, rows, prow, colsum).view(-1)
    return _ScoreEmb.apply(name, "_po", l_norm, colsum, prow, rows).view(-1)


def _sum_of_scores_value(name: str, direction: str, ent: Tensor, rel: Tensor, a: Tensor, p: Tensor) -> Tensor:
    """[n] values of sum_j score(i, j) (no autograd) from a handful of elementwise ops on [n, d]: the query
    against the table's column sum c.  ComplEx (complex.py:30-39): Re<s, r, conj(o)> =
    (s_re r_re - s_im r_im) o_re + (s_re r_im + s_im r_re) o_im, linear in o ("sp": o = c) and in s ("po": s = c)."""
    c = ent.float().sum(dim=0)
    x, r = ent[a.long()].float(), rel[p.long()].float()
    if name == "distmult":
        return (x * r * c).sum(dim=1)
    h = x.shape[1] // 2
    x_re, x_im, r_re, r_im, c_re, c_im = x[:, :h], x[:, h:], r[:, :h], r[:, h:], c[:h], c[h:]
    if direction == "sp":
        return ((x_re * r_re - x_im * r_im) * c_re + (x_re * r_im + x_im * r_re) * c_im).sum(dim=1)
    return ((c_re * r_re - c_im * r_im) * x_re + (c_re * r_im + c_im * r_re) * x_im).sum(dim=1)


def bce_fused(name: str, l_norm, direction: str, ent: Tensor, rel: Tensor, a: Tensor, p: Tensor, rowptr: Tensor,
              col: Tensor, offset: float, eps: float, t) -> Tensor:
    """The fused KvsAll BCE loss with label smoothing `eps` (train_KvsAll.py:260-266; loss.py:137-159 with
    bce_type None).  With x_ij = score_ij + offset and y_ij = (1 - eps) [j in labels_i] + 1/E,
        sum_j softplus(x_ij) - y_ij x_ij
          = [sum_j softplus(x_ij) - sum_{labels} x_ij]      <- kge_bce_fwd, fused, scores never written
            + eps * sum_{labels} x_ij                       <- nnz(labels) spo scores (kge_score_spo)
            - (1/E) * sum_j x_ij                            <- one [n, 1] score against the column sum."""
    fused = _FusedBCE.apply(direction, ent, rel, a, p, rowptr, col, offset, t)
    if eps == 0.0:
        return fused
    n, E = a.shape[0], ent.shape[0]
    cnt = (rowptr[1:] - rowptr[:-1]).to(ent.device)
    col = col.to(ent.device).long()
    rows = torch.repeat_interleave(torch.arange(n, device=ent.device), cnt, output_size=col.numel())
    ar, pr = a.long()[rows], p.long()[rows]
    ent32, rel32 = ent.float(), rel.float()
    lab = (_ScoreSPO.apply(name, l_norm, ent32, rel32, ar, pr, col) if direction == "sp"
           else _ScoreSPO.apply(name, l_norm, ent32, rel32, col, pr, ar))
    s_lab = torch.zeros(n, dtype=torch.float32, device=ent.device).index_add(0, rows, lab.view(-1) + offset)
    s_all = _score_against_column_sum(name, l_norm, direction, ent, rel, a, p) + E * offset
    return fused + eps * s_lab - s_all / E


class _ScoreEmb(torch.autograd.Function):
    """Dense-embedding scoring (RelationalScorer.score_emb)."""

    @staticmethod
    def forward(ctx, name, combine, l_norm, s_emb, p_emb, o_emb):
        out = engine.score_emb(name, s_emb.detach(), p_emb.detach(), o_emb.detach(), combine, l_norm)
        ctx.meta = (name, combine, l_norm)
        ctx.save_for_backward(s_emb, p_emb, o_emb, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        name, combine, l_norm = ctx.meta
        s_emb, p_emb, o_emb, out = ctx.saved_tensors
        g_s, g_p, g_o = engine.score_emb_bwd(name, s_emb, p_emb, o_emb, combine, l_norm, gout, out)
        return None, None, None, g_s, g_p, g_o
