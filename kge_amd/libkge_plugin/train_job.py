"""TrainingJob1vsAll with the loss fused into the scoring kernel (train.type: hip_1vsAll)."""
import time

from kge.job import Job
from kge.job.train_1vsAll import TrainingJob1vsAll
from kge.util.loss import KLDivWithSoftmaxKgeLoss


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

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        fused = isinstance(self.loss, KLDivWithSoftmaxKgeLoss) and hasattr(self.model, "loss_sp")
        if not fused:
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        batch_size = result.size
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(self.device)
        result.prepare_time += time.time()
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
