"""Literal torch restatement of the reference's CPU scoring path (TEST INFRASTRUCTURE).

Used only by bench.py's `cpu_baseline` leg (kind "port") and by tests: it issues the SAME
torch op sequence as the reference -- embedding gather, cat/chunk, elementwise mul, `mm`,
`cdist`, `pairwise_distance`, cos/sin -- so on a given host it costs what the reference
costs and returns what the reference returns (tests/test_oracle_vs_reference.py checks
bit-equality against the live reference in the build container).  The reference tree
itself cannot travel to the GPU box.

  score_emb  <- ComplExScorer.score_emb   kge/model/complex.py:18-43
                DistMultScorer.score_emb  kge/model/distmult.py:13-25
                TransEScorer.score_emb    kge/model/transe.py:15-37
                RotatEScorer.score_emb    kge/model/rotate.py:20-69 (+ helpers :146-213)
  score_sp/po/spo <- KgeModel.score_*     kge/model/kge_model.py:663-725
                     (LookupEmbedder.embed / embed_all, lookup_embedder.py:96-112)
  ns_bce_loss     <- BCEWithLogitsKgeLoss.__call__  kge/util/loss.py:153-186 on the label matrix of
                     TrainingJobNegativeSampling (column 0 = 1, train_negative_sampling.py:128-137)
  kl_loss         <- KLDivWithSoftmaxKgeLoss.__call__  kge/util/loss.py:192-213 (index labels: TrainingJob1vsAll;
                     label matrix: TrainingJobKvsAll)
  bce_loss        <- BCEWithLogitsKgeLoss.__call__ with bce_type None  kge/util/loss.py:137-159
  smooth_labels   <- TrainingJobKvsAll's label smoothing  kge/job/train_KvsAll.py:260-266
"""
import torch
import torch.nn.functional as F


def _abs_complex(re, im):
    return torch.norm(torch.stack((re, im), dim=0), dim=0)


def _norm_nonneg(x, dim, p):
    return torch.sum(x, dim=dim) if p == 1.0 else torch.norm(x, dim=dim, p=p)


def score_emb(model, s, p, o, combine, l_norm=1.0):
    n = p.size(0)
    if model == "distmult":
        if combine == "spo":
            out = (s * p * o).sum(dim=1)
        elif combine == "sp_":
            out = (s * p).mm(o.transpose(0, 1))
        else:
            out = (o * p).mm(s.transpose(0, 1))
    elif model == "complex":
        p_re, p_im = (t.contiguous() for t in p.chunk(2, dim=1))
        o_re, o_im = (t.contiguous() for t in o.chunk(2, dim=1))
        s_all = torch.cat((s, s), dim=1)
        r_all = torch.cat((p_re, p, -p_im), dim=1)
        o_all = torch.cat((o, o_im, o_re), dim=1)
        if combine == "spo":
            out = (s_all * o_all * r_all).sum(dim=1)
        elif combine == "sp_":
            out = (s_all * r_all).mm(o_all.transpose(0, 1))
        else:
            out = (r_all * o_all).mm(s_all.transpose(0, 1))
    elif model == "transe":
        if combine == "spo":
            out = -F.pairwise_distance(s + p, o, p=l_norm)
        elif combine == "sp_":
            out = -torch.cdist(s + p, o, p=l_norm, compute_mode="donot_use_mm_for_euclid_dist")
        else:
            out = -torch.cdist(o - p, s, p=l_norm, compute_mode="donot_use_mm_for_euclid_dist")
    elif model == "rotate":
        s_re, s_im = torch.chunk(s, 2, dim=1)
        o_re, o_im = torch.chunk(o, 2, dim=1)
        p_re, p_im = torch.cos(p), torch.sin(p)
        if combine == "spo":
            sp_re = s_re * p_re - s_im * p_im
            sp_im = s_re * p_im + s_im * p_re
            out = -_norm_nonneg(_abs_complex(sp_re - o_re, sp_im - o_im), 1, l_norm)
        elif combine == "sp_":
            sp_re = s_re * p_re - s_im * p_im
            sp_im = s_re * p_im + s_im * p_re
            d_re = sp_re.unsqueeze(1) - o_re
            d_im = sp_im.unsqueeze(1) - o_im
            out = -_norm_nonneg(_abs_complex(d_re, d_im), 2, l_norm)
        else:
            p_im = -p_im
            po_re = p_re * o_re - p_im * o_im
            po_im = p_re * o_im + p_im * o_re
            d_re = po_re.unsqueeze(1) - s_re
            d_im = po_im.unsqueeze(1) - s_im
            out = -_norm_nonneg(_abs_complex(d_re, d_im), 2, l_norm)
    else:
        raise ValueError(model)
    return out.view(n, -1)


def _embed(table, idx):
    return torch.nn.functional.embedding(idx.long(), table)


def _embed_all(table):
    return _embed(table, torch.arange(table.size(0), dtype=torch.long))


def score_spo(model, ent, rel, s, p, o, l_norm=1.0):
    return score_emb(model, _embed(ent, s), _embed(rel, p), _embed(ent, o), "spo", l_norm).view(-1)


def score_sp(model, ent, rel, s, p, o=None, l_norm=1.0):
    tg = _embed_all(ent) if o is None else _embed(ent, o)
    return score_emb(model, _embed(ent, s), _embed(rel, p), tg, "sp_", l_norm)


def score_po(model, ent, rel, p, o, s=None, l_norm=1.0):
    tg = _embed_all(ent) if s is None else _embed(ent, s)
    return score_emb(model, tg, _embed(rel, p), _embed(ent, o), "_po", l_norm)


def ns_bce_loss(scores, kind, offset=0.0, temperature=1.0):
    """BCEWithLogitsKgeLoss (kge/util/loss.py:153-186) on a negative-sampling score block [n, 1 + K] whose label matrix
    has ones in column 0: kind "bce" (reduction "sum"), "bce_mean", "bce_self_adversarial".  The reference's op
    sequence: offset, BCEWithLogitsLoss over the flattened block, the positive column picked by index, the negatives by
    sum - positive (mean) or by a mask (self-adversarial: softmax of the detached scores * temperature)."""
    n, c = scores.shape
    labels = torch.zeros(scores.shape, device=scores.device, dtype=torch.float)
    labels[:, 0] = 1
    if offset != 0.0:
        scores = scores + offset
    reduction = "sum" if kind == "bce" else "none"
    losses = torch.nn.BCEWithLogitsLoss(reduction=reduction)(scores.view(-1), labels.view(-1))
    if kind == "bce":
        return losses
    pos_idx = torch.zeros(n, dtype=torch.long, device=scores.device)
    losses = losses.view(scores.shape)
    losses_positives = losses[range(n), pos_idx]
    if kind == "bce_mean":
        losses_negatives = losses.sum(dim=1) - losses_positives
        return (losses_positives.sum() + losses_negatives.sum() / (c - 1)) / 2.0
    negative_indexes = torch.nonzero(labels.view(-1) == 0.0)
    scores_negatives = scores.detach().view(-1)[negative_indexes].view((n, c - 1))
    losses_negatives = losses.view(-1)[negative_indexes].view((n, c - 1))
    losses_negatives = (F.softmax(scores_negatives * temperature, dim=1) * losses_negatives).sum(dim=1)
    return (losses_positives.sum() + losses_negatives.sum()) / 2.0


def kl_loss(scores, labels, reduction="sum"):
    """KLDivWithSoftmaxKgeLoss (kge/util/loss.py:192-213), the loss of TrainingJob1vsAll (index labels, [n]:
    CrossEntropyLoss, loss.py:196-207) and of TrainingJobKvsAll (a label matrix, [n, E]: KLDivLoss of log_softmax against
    the L1-normalised labels, loss.py:208-213).  reduction "sum" is what the jobs use; "rows" gives the per-row terms the
    fused kernels return (kge_ce_fwd / kge_kl_fwd: loss_rows) -- the same torch functions with reduction "none"."""
    red = "none" if reduction == "rows" else reduction
    if labels.dim() == 1:
        return torch.nn.CrossEntropyLoss(reduction=red)(scores, labels)
    out = torch.nn.KLDivLoss(reduction=red)(F.log_softmax(scores, dim=1), F.normalize(labels.float(), p=1, dim=1))
    return out.sum(dim=1) if reduction == "rows" else out


def bce_loss(scores, labels, offset=0.0, reduction="sum"):
    """BCEWithLogitsKgeLoss with bce_type None (kge/util/loss.py:137-159) on a label matrix [n, E] (TrainingJobKvsAll;
    index labels [n] become a one-hot matrix first, loss.py:105-116: TrainingJob1vsAll): offset, then BCEWithLogitsLoss
    over the flattened matrices.  reduction "rows": the per-row sums kge_bce_fwd returns."""
    if labels.dim() == 1:
        x = torch.zeros(scores.shape, device=scores.device, dtype=torch.float)
        x[range(len(scores)), labels] = 1.0
        labels = x
    if offset != 0.0:
        scores = scores + offset
    if reduction == "rows":
        return torch.nn.BCEWithLogitsLoss(reduction="none")(scores.view(-1), labels.view(-1)).view(scores.shape).sum(dim=1)
    return torch.nn.BCEWithLogitsLoss(reduction=reduction)(scores.view(-1), labels.view(-1))


def smooth_labels(labels, label_smoothing):
    """TrainingJobKvsAll's label smoothing (kge/job/train_KvsAll.py:260-266, as in ConvE): applied to the multi-hot
    label matrix before the loss."""
    if label_smoothing > 0.0:
        return (1.0 - label_smoothing) * labels + 1.0 / labels.size(1)
    return labels

