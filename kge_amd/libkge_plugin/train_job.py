"""TrainingJob1vsAll / TrainingJobKvsAll with the kl or bce loss fused into the scoring kernel
(train.type: hip_1vsAll / hip_KvsAll)."""
import time

import torch


def _plain_bce(loss):
    """BCEWithLogitsKgeLoss as the fused kernel computes it: bce_type None (sum over all entities),
    default BCEWithLogitsLoss arguments; returns the score offset (train.loss_arg) or None."""
    from kge.util.loss import BCEWithLogitsKgeLoss
    if not isinstance(loss, BCEWithLogitsKgeLoss) or loss._bce_type is not None:
        return None
    inner = loss._loss
    if getattr(inner, "weight", None) is not None or getattr(inner, "pos_weight", None) is not None:
        return None
    return float(loss._offset)

from kge.job import Job
from kge.job.train_1vsAll import TrainingJob1vsAll
from kge.job.train_KvsAll import TrainingJobKvsAll
from kge.util.loss import BCEWithLogitsKgeLoss, KLDivWithSoftmaxKgeLoss


class HipTrainingJob1vsAll(TrainingJob1vsAll):
    """Overrides only `_process_subbatch` (train_1vsAll.py:48-92).  With `train.loss: kl` and a
    model that offers `loss_sp` / `loss_po` (HipComplEx / HipDistMult scoring in bfloat16), the
    [n, E] score matrix of each direction is never written: one kernel produces the per-row
    cross entropy (kge_ce_fwd), its backward recomputes the tiles (kge_ce_bwd).  Anything else
    (other losses, float32 scoring, dropout, other models) runs the reference's code."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        if self.__class__ == HipTrainingJob1vsAll:
            for f in Job.job_created_hooks:
                f(self)

    def _process_subbatch_bce(self, batch_index, batch, subbatch_slice, result, offset):
        """train.loss: bce -- every triple's (s, p) row has the single label o (and (p, o) the label s):
        kge_bce_fwd with a one-entry-per-row CSR."""
        batch_size = result.size
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(self.device)
        rowptr = torch.arange(len(triples) + 1, dtype=torch.long, device=triples.device)
        result.prepare_time += time.time()
        for rows_fn in (lambda: self.model.bce_loss_sp(triples[:, 0], triples[:, 1], rowptr, triples[:, 2], offset),
                        lambda: self.model.bce_loss_po(triples[:, 1], triples[:, 2], rowptr, triples[:, 0], offset)):
            result.forward_time -= time.time()
            rows = rows_fn()
            if rows is None:
                return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        offset = _plain_bce(self.loss)
        if offset is not None and hasattr(self.model, "bce_loss_sp"):
            return self._process_subbatch_bce(batch_index, batch, subbatch_slice, result, offset)
        fused = isinstance(self.loss, KLDivWithSoftmaxKgeLoss) and hasattr(self.model, "loss_sp")
        if not fused:
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        batch_size = result.size
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(self.device)
        result.prepare_time += time.time()
        if hasattr(self.model, "loss_sp_po"):
            # both directions from one scoring launch and one pair of gradient products; the sum of
            # the two losses is back-propagated once (the reference does it in two passes:
            # the same gradients, accumulated)
            result.forward_time -= time.time()
            rows = self.model.loss_sp_po(triples[:, 0], triples[:, 1], triples[:, 2])
            if rows is not None:
                loss_value = rows.sum() / batch_size
                result.avg_loss += loss_value.item()
                result.forward_time += time.time()
                result.backward_time -= time.time()
                if not self.is_forward_only:
                    loss_value.backward()
                result.backward_time += time.time()
                return
            result.forward_time += time.time()
        for loss_rows_fn in (lambda: self.model.loss_sp(triples[:, 0], triples[:, 1], triples[:, 2]),
                             lambda: self.model.loss_po(triples[:, 1], triples[:, 2], triples[:, 0])):
            result.forward_time -= time.time()
            rows = loss_rows_fn()
            if rows is None:  # the model declined (tables / options changed): reference path for all
                return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()


class HipTrainingJobKvsAll(TrainingJobKvsAll):
    """Overrides only `_process_subbatch` (train_KvsAll.py:216-294).  With `train.loss: kl` or `bce`
    (plain: bce_type None), no label smoothing and a model that offers `kl_loss_sp` / `kl_loss_po`
    (`bce_loss_sp` / `bce_loss_po`), the sp_ and _po queries of a subbatch get their loss from one
    fused kernel each (kge_kl_fwd / kge_bce_fwd: scores never written; labels as a CSR cut out of the
    batch's `label_coords`); s_o queries and every other configuration run the reference's code."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        if self.__class__ == HipTrainingJobKvsAll:
            for f in Job.job_created_hooks:
                f(self)

    def _fused_ok(self) -> bool:
        if self.label_smoothing != 0.0 or "s_o" in self.query_types:
            return False
        if isinstance(self.loss, KLDivWithSoftmaxKgeLoss):
            return hasattr(self.model, "kl_loss_sp")
        return _plain_bce(self.loss) is not None and hasattr(self.model, "bce_loss_sp")

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        if not self._fused_ok():
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        batch_size = result.size
        result.prepare_time -= time.time()
        queries = batch["queries"][subbatch_slice].to(self.device)
        coords = batch["label_coords"]  # [nnz, 2] (batch row, label), rows ascending; on the device
        qtype = batch["query_type_indexes"][subbatch_slice].to(self.device)
        row0 = subbatch_slice.start or 0
        # label ranges of all batch rows: counts -> offsets (no host sync)
        counts = torch.bincount(coords[:, 0].long(), minlength=batch_size)
        offsets = torch.zeros(batch_size + 1, dtype=torch.long, device=coords.device)
        torch.cumsum(counts, 0, out=offsets[1:])
        result.prepare_time += time.time()
        for query_type_index, query_type in enumerate(self.query_types):
            examples = (qtype == query_type_index).nonzero(as_tuple=False).view(-1)
            if len(examples) == 0:
                continue
            result.forward_time -= time.time()
            rows = examples + row0
            cnt = counts[rows]
            rowptr = torch.zeros(len(rows) + 1, dtype=torch.long, device=cnt.device)
            torch.cumsum(cnt, 0, out=rowptr[1:])
            total = int(rowptr[-1])
            idx = torch.repeat_interleave(offsets[rows] - rowptr[:-1], cnt, output_size=total) \
                + torch.arange(total, device=cnt.device)
            col = coords[idx, 1].long()
            q0, q1 = queries[examples, 0], queries[examples, 1]
            offset = _plain_bce(self.loss)
            if offset is None:
                loss_rows = (self.model.kl_loss_sp(q0, q1, rowptr, col) if query_type == "sp_"
                             else self.model.kl_loss_po(q0, q1, rowptr, col))
            else:
                loss_rows = (self.model.bce_loss_sp(q0, q1, rowptr, col, offset) if query_type == "sp_"
                             else self.model.bce_loss_po(q0, q1, rowptr, col, offset))
            if loss_rows is None:  # the model declined: reference path for the whole subbatch
                result.forward_time += time.time()
                return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
            loss_value = loss_rows.sum() / batch_size  # averaged over the batch, not the subbatch
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()
