"""TrainingJob1vsAll / TrainingJobKvsAll with the kl or bce loss fused into the scoring kernel
(train.type: hip_1vsAll / hip_KvsAll) and TrainingJobNegativeSampling with the negatives of the
per-triple sampler scored by the fused gather + score kernel (train.type: hip_negative_sampling)."""
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



def _model_takes_fused_loss(model, with_dropout: bool = False) -> bool:
    """Decided ONCE per subbatch, before any backward: a decline after the first direction's
    loss.backward() would make the reference path back-propagate that direction a second time.
    with_dropout (the 1vsAll kl loss): embedder dropout in training does not decline -- the fused loss applies the
    masks itself (_FusedScoring._dropout_only)."""
    f = getattr(model, "_ce_tables", None)
    if f is not None and f() is not None:
        return True
    g = getattr(model, "_dropout_only", None)
    return bool(with_dropout and g is not None and g() is not None)


def _declined_late(what):
    raise RuntimeError(f"kge_amd: {what} declined after part of the subbatch was already back-propagated "
                       "(tables or options changed inside a subbatch)")


from kge.job import Job
from kge.job.train_1vsAll import TrainingJob1vsAll
from kge.job.train_KvsAll import TrainingJobKvsAll
from kge.job.train_negative_sampling import TrainingJobNegativeSampling, S, P, O
from kge.util.loss import BCEWithLogitsKgeLoss, KLDivWithSoftmaxKgeLoss


def _optimizer_is_capturable(opt) -> bool:
    """Only optimizers whose captured step() replays correctly: kge_amd.optim.Adagrad says so itself (without lr_decay),
    HipAdam says no (host-computed bias correction would be frozen at the capture step), of torch's own only SGD."""
    return (all(g.get("lr_decay", 0) == 0 for g in opt.param_groups)
            and bool(getattr(opt, "graph_capturable", type(opt) is torch.optim.SGD)))


def _no_penalty(job, batch_index, batch) -> bool:
    """A penalty term back-propagates between the batch and the optimizer's step: such a job stays eager -- unless the
    terms are folded into the optimizer's pass (_fold_penalties)."""
    if getattr(job, "_folded_penalties", None):
        return True
    try:
        return len(job.model.penalty(epoch=job.epoch, batch_index=batch_index, num_batches=len(job.loader),
                                     batch=batch)) == 0
    except Exception:
        return False


def _fold_penalties(job) -> bool:
    """The embedders' UNWEIGHTED Lp / N3 penalty terms (lookup_embedder.py:122-147 through KgeModel.penalty,
    kge_model.py:603-649) folded into HipAdagrad's pass: their gradient is a function of the element alone, so the
    optimizer adds it in registers and sums the term's value on the way (kge_adagrad_step_multi_penalty).  A step taken
    inside `_process_batch` (a GraphedStep, replayed or eager) then carries the penalty; TrainingJob.run_epoch still
    calls `model.penalty()` afterwards, back-propagates what it gets and writes the values to the trace
    (train.py:417-436): it gets leaf scalars holding the values the step computed -- their backward is a no-op.  A
    batch whose step was NOT taken yet (subbatches, a declined fused loss) gets the reference's terms and the
    optimizer skips its folded ones for that step.

    -> True if every term of this model is folded (the job may capture its step), False if the model has no foldable
    shape (weighted terms, other embedders, other regularizers, another optimizer): nothing is changed then."""
    from kge.model import KgeModel, LookupEmbedder
    from ..optim import Adagrad as HipAdagrad
    model, opt = job.model, job.optimizer
    if getattr(job, "_folded_penalties", None) is not None:
        return bool(job._folded_penalties)
    job._folded_penalties = []
    from .models import _FusedScoring  # (its penalty() is KgeModel.penalty behind a shortcut for "no terms")
    if not isinstance(opt, HipAdagrad) or type(model).penalty not in (KgeModel.penalty, _FusedScoring.penalty):
        return False
    pe, se, oe = model.get_p_embedder(), model.get_s_embedder(), model.get_o_embedder()
    plan = []
    for emb, times in ((pe, 1.0),) + (((se, 2.0),) if se is oe else ((se, 1.0), (oe, 1.0))):
        if type(emb).penalty is not LookupEmbedder.penalty:
            return False
        if emb.regularize == "" or emb.get_option("regularize_weight") == 0.0:
            continue
        if emb.regularize not in ("lp", "n3") or emb.get_option("regularize_args.weighted"):
            return False
        if emb.regularize == "n3":
            p, kind = 3, ("n3_complex" if emb.space == "complex" else "lp")
        else:
            p = emb.get_option("regularize_args.p") if emb.has_option("regularize_args.p") else 2
            kind = "lp"
        if p not in (1, 2, 3) or float(p) != int(p):
            return False
        w = emb._embeddings.weight
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            return False
        if kind == "n3_complex" and w.shape[1] % 8 != 0:
            return False
        if not any(w is q for g in opt.param_groups for q in g["params"]):
            return False
        plan.append((emb, w, kind, int(p), float(emb._get_regularize_weight()), times))
    if not plan:
        return False  # (nothing to fold: _no_penalty's own check decides)
    for emb, w, kind, p, weight, times in plan:
        opt.set_penalty(w, kind, p, weight, times)
    job._folded_penalties = plan
    reference_penalty = model.penalty

    def penalty(**kwargs):
        if job._skip_optimizer_step:  # the step -- with the terms' gradient in it -- is taken: hand over its values
            return [(f"{emb.configuration_key}.L{p}_penalty", opt.penalty_value(w).detach().requires_grad_(True))
                    for emb, w, kind, p, weight, times in plan]
        opt.skip_penalties_once()
        return reference_penalty(**kwargs)
    model.penalty = penalty
    return True


def _graphed_step_of(job, loss_fn):
    """kge_amd.train_graph.GraphedStep over `loss_fn` and the job's optimizer.  TrainingJob.run_epoch
    (kge/job/train.py:452-474) calls optimizer.step() itself after the batch: a batch that went through the GraphedStep
    (replayed or eager) has taken that step already, so the job's optimizer skips exactly that one call
    (`job._skip_optimizer_step`)."""
    from kge_amd.train_graph import GraphedStep
    opt = job.optimizer
    # (a job may build a second GraphedStep -- hip_KvsAll when its capacities grow --: always around the optimizer's OWN step)
    real_step = getattr(job, "_real_optimizer_step", None) or opt.step
    job._real_optimizer_step = real_step

    class _Opt:  # what GraphedStep needs of the optimizer, with the REAL step
        param_groups = opt.param_groups
        zero_grad = staticmethod(opt.zero_grad)
        step = staticmethod(real_step)
        graph_capturable = True  # (the caller checked the real optimizer: _optimizer_is_capturable)
        after_graph_replay = staticmethod(getattr(opt, "after_graph_replay", lambda params: None))

    def step_or_skip(*a, **k):
        if job._skip_optimizer_step:
            job._skip_optimizer_step = False
            opt._opt_called = True  # (what a learning-rate scheduler's wrapper of step() records)
            return None
        return real_step(*a, **k)
    if hasattr(real_step, "_wrapped_by_lr_sched"):  # torch's schedulers look for their mark on step()
        step_or_skip._wrapped_by_lr_sched = True
    opt.step = step_or_skip
    return GraphedStep(loss_fn, _Opt, warmup=2)


class _CudaOomText:
    """TrainingJob.run_epoch halves `train.subbatch_size` when a batch fails with a RuntimeError whose text contains
    "CUDA out of memory" (train.subbatch_auto_tune, kge/job/train.py:384-413).  On ROCm torch's allocator says "HIP out
    of memory" -- for its own allocations as for ours -- and the tuner never fires.  The hip_* jobs re-raise any
    out-of-memory error of a batch under the text the trainer matches (error conventions)."""

    def _process_batch(self, batch_index, batch):
        try:
            return super()._process_batch(batch_index, batch)
        except torch.OutOfMemoryError as e:
            if "CUDA out of memory" in str(e):
                raise
            raise RuntimeError("CUDA out of memory (ROCm: " + str(e) + ")") from e


class HipTrainingJob1vsAll(_CudaOomText, TrainingJob1vsAll):
    """Overrides only `_process_subbatch` (train_1vsAll.py:48-92).  With `train.loss: kl` and a
    model that offers `loss_sp` / `loss_po` (HipComplEx / HipDistMult scoring in bfloat16), the
    [n, E] score matrix of each direction is never written: one kernel produces the per-row
    cross entropy (kge_ce_fwd), its backward recomputes the tiles (kge_ce_bwd).  Anything else
    (other losses, float32 scoring, dropout, other models) runs the reference's code."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        self._graph_step = None       # kge_amd.train_graph.GraphedStep, built at the first batch that qualifies
        self._graph_step_ok = None    # decided at the first batch (None: not yet)
        self._skip_optimizer_step = False
        if self.__class__ == HipTrainingJob1vsAll:
            for f in Job.job_created_hooks:
                f(self)

    # ---- hip_1vsAll.graph_step: forward + backward + optimizer.step of a full batch as one hipGraph replay.
    # TrainingJob.run_epoch (kge/job/train.py:452-474) calls optimizer.step() itself after the batch: the replay has
    # taken that step already, so the job's optimizer skips exactly that one call.
    def _graph_step_for(self, batch_index, batch, subbatch_slice):
        if self._graph_step_ok is False or self.is_forward_only:
            return None
        if self._graph_step_ok is None:
            ok = bool(self.config.get_default("hip_1vsAll.graph_step")) and str(self.device).startswith("cuda")
            ok = ok and _model_takes_fused_loss(self.model) and hasattr(self.model, "loss_sp_po")
            ok = ok and _optimizer_is_capturable(self.optimizer)
            ok = ok and (_fold_penalties(self) or _no_penalty(self, batch_index, batch))
            self._graph_step_ok = ok
            if ok:
                # the batch goes in as ONE tensor (one copy into the static buffer per replay, not three); the captured
                # batches all have the size of the first (another size runs eagerly), so 1 / batch size is a constant
                self._graph_step = _graphed_step_of(
                    self, (lambda t: self.model.loss_sp_po_sum(t[:, 0], t[:, 1], t[:, 2], 1.0 / len(t)))
                    if hasattr(self.model, "loss_sp_po_sum")
                    else (lambda t: self.model.loss_sp_po(t[:, 0], t[:, 1], t[:, 2]).sum() / len(t)))
        if not self._graph_step_ok:
            return None
        n = len(batch["triples"])
        sl = subbatch_slice
        whole = sl.start in (0, None) and (sl.stop is None or sl.stop >= n) and sl.step in (1, None)
        return self._graph_step if whole and _model_takes_fused_loss(self.model) else None

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
                _declined_late("bce_loss_sp / bce_loss_po")
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        kl = isinstance(self.loss, KLDivWithSoftmaxKgeLoss) and hasattr(self.model, "loss_sp")
        if not _model_takes_fused_loss(self.model, with_dropout=kl):
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        offset = _plain_bce(self.loss)
        if offset is not None and hasattr(self.model, "bce_loss_sp"):
            if not _model_takes_fused_loss(self.model):  # (the bce loss has no dropout form)
                return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
            return self._process_subbatch_bce(batch_index, batch, subbatch_slice, result, offset)
        fused = kl
        if not fused:
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        batch_size = result.size
        if hasattr(self.model, "loss_sp_po"):
            gs = self._graph_step_for(batch_index, batch, subbatch_slice)
            if gs is not None and gs.enabled:
                result.forward_time -= time.time()
                # the batch as the loader made it (a host tensor): GraphedStep copies it host-to-device straight into the
                # captured step's input buffer -- no device-side hop in front of the replay
                loss_value = gs(batch["triples"][subbatch_slice])  # (whole batch: len(triples) == batch_size)
                self._skip_optimizer_step = True  # (eager or replayed: the step is taken)
                result.avg_loss += loss_value.item()
                result.forward_time += time.time()
                return
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(self.device)
        result.prepare_time += time.time()
        if hasattr(self.model, "loss_sp_po"):
            # both directions from one scoring launch and one pair of gradient products; the sum of
            # the two losses is back-propagated once (the reference does it in two passes:
            # the same gradients, accumulated)
            result.forward_time -= time.time()
            if hasattr(self.model, "loss_sp_po_sum"):  # the sum (and its gradient) inside the loss kernels' launches
                loss_value = self.model.loss_sp_po_sum(triples[:, 0], triples[:, 1], triples[:, 2], 1.0 / batch_size)
                if loss_value is None:
                    _declined_late("loss_sp_po_sum")
            else:
                rows = self.model.loss_sp_po(triples[:, 0], triples[:, 1], triples[:, 2])
                if rows is None:
                    _declined_late("loss_sp_po")
                loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()
            return
        for loss_rows_fn in (lambda: self.model.loss_sp(triples[:, 0], triples[:, 1], triples[:, 2]),
                             lambda: self.model.loss_po(triples[:, 1], triples[:, 2], triples[:, 0])):
            result.forward_time -= time.time()
            rows = loss_rows_fn()
            if rows is None:
                _declined_late("loss_sp / loss_po")
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()


class HipTrainingJobKvsAll(_CudaOomText, TrainingJobKvsAll):
    """Overrides only `_process_subbatch` (train_KvsAll.py:216-294).  With `train.loss: kl` or `bce`
    (plain: bce_type None), with or without `KvsAll.label_smoothing`, and a model
    that offers `kl_loss_sp` / `kl_loss_po` (`bce_loss_sp` / `bce_loss_po`), the sp_ and _po queries of a
    subbatch get their loss from one fused kernel each (kge_kl_fwd / kge_kl_weighted_fwd /
    kge_bce_fwd: scores never written; labels as a CSR cut out of the batch's `label_coords`); s_o
    queries and every other configuration run the reference's code."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        self._graph_step = None       # kge_amd.train_graph.GraphedStep over padded inputs (hip_KvsAll.graph_step)
        self._graph_step_ok = None    # decided at the first batch that could take it (None: not yet)
        self._graph_caps = None       # (rows per query type, label entries per query type) the captured step holds
        self._graph_overflows = 0     # batches in a row that did not fit the capacities
        self.graph_batches = 0        # batches that went through the GraphedStep (replayed or eager)
        self._skip_optimizer_step = False
        if self.__class__ == HipTrainingJobKvsAll:
            for f in Job.job_created_hooks:
                f(self)

    # ---- hip_KvsAll.graph_step: forward + ONE backward for both query types + optimizer.step of a whole batch as one
    # hipGraph replay (train_KvsAll.py:216-294 issues ~40 launches per batch from ~0.3 ms of Python; the kernel-level
    # step replays in 0.2 ms: bench.py roofline_train.kvsall_step).  A batch's shapes vary -- how many of its queries
    # are sp_ / _po, how many labels each has -- so the captured step works on PADDED inputs: `rows` queries per type
    # and `labels` label entries per type; a padding row is query (0, 0) with the single label 0 and weight 0 in the
    # batch loss (its loss row is finite, its gradient row exactly zero).  A batch that does not fit runs as before;
    # three such batches in a row grow the capacities and capture again.
    def _graph_inputs(self, per_type, batch_size, dev):
        """-> the eight padded input tensors of the captured step, or None if the batch does not fit."""
        rows_cap, lab_cap = self._graph_caps
        out = []
        for qt in ("sp_", "_po"):
            if qt in per_type:
                q0, q1, rowptr, col = per_type[qt]
                n, nnz = len(q0), int(col.numel())
            else:
                q0 = q1 = col = torch.zeros(0, dtype=torch.long, device=dev)
                rowptr = torch.zeros(1, dtype=torch.long, device=dev)
                n, nnz = 0, 0
            pad = rows_cap - n
            if pad < 0 or nnz + pad > lab_cap:
                return None
            ids = torch.zeros(rows_cap, 2, dtype=torch.long, device=dev)
            ids[:n, 0], ids[:n, 1] = q0, q1
            rp = torch.empty(rows_cap + 1, dtype=torch.long, device=dev)
            rp[:n + 1] = rowptr
            if pad:
                rp[n + 1:] = nnz + torch.arange(1, pad + 1, device=dev)
            cl = torch.zeros(lab_cap, dtype=torch.long, device=dev)
            cl[:nnz] = col
            w = torch.zeros(rows_cap, dtype=torch.float32, device=dev)
            w[:n] = 1.0 / batch_size
            out += [ids, rp, cl, w]
        return out

    def _graph_loss(self, ids_sp, rp_sp, cl_sp, w_sp, ids_po, rp_po, cl_po, w_po):
        offset = _plain_bce(self.loss)
        rows_sp, rows_po = self.model.multilabel_loss_sp_po(
            "kl" if offset is None else "bce", ids_sp[:, 0], ids_sp[:, 1], rp_sp, cl_sp, ids_po[:, 1], ids_po[:, 0], rp_po,
            cl_po, 0.0 if offset is None else offset)
        return (rows_sp * w_sp).sum() + (rows_po * w_po).sum()

    def _graph_step_for(self, batch_index, batch, subbatch_slice, per_type, totals):
        """The GraphedStep for this batch (made or re-made when the capacities change), or None."""
        if self._graph_step_ok is False or self.is_forward_only:
            return None
        n = len(batch["queries"])
        sl = subbatch_slice
        whole = sl.start in (0, None) and (sl.stop is None or sl.stop >= n) and sl.step in (1, None)
        if not whole:
            return None
        if self._graph_step_ok is None:
            try:
                want = bool(self.config.get_default("hip_KvsAll.graph_step"))
            except KeyError:
                want = False
            ok = want and str(self.device).startswith("cuda") and hasattr(self.model, "multilabel_loss_sp_po")
            ok = ok and float(self.label_smoothing) == 0.0 and _optimizer_is_capturable(self.optimizer)
            ok = ok and (_fold_penalties(self) or _no_penalty(self, batch_index, batch))
            self._graph_step_ok = ok
        if not self._graph_step_ok:
            return None
        caps = self._graph_caps
        if caps is None or self._graph_overflows >= 3:
            # capacities from this batch with head room (the split of a batch into sp_ / _po queries is binomial, the
            # label counts follow the data's degree distribution); never below the previous ones
            need_rows = max([len(v[0]) for v in per_type.values()] + [1])
            need_lab = max([totals[k] for k in per_type] + [1])
            up = lambda x, m: (int(x) + m - 1) // m * m
            rows_cap = min(up(max(need_rows * 1.15 + 16, (caps or (0, 0))[0]), 32), up(n, 32))
            rows_cap = max(rows_cap, up(need_rows, 32))
            lab_cap = up(max((need_lab + rows_cap) * 1.5, (caps or (0, 0))[1]), 256)
            self._graph_caps = (rows_cap, lab_cap)
            self._graph_step = _graphed_step_of(self, self._graph_loss)
            self._graph_overflows = 0
        return self._graph_step

    def _fused_ok(self) -> bool:
        # (s_o queries -- relation targets, off by default: config-default.yaml:322-325 -- ride along: their few hundred
        # columns go through the reference's own ops below, the entity-target queries keep the fused loss)
        if isinstance(self.loss, KLDivWithSoftmaxKgeLoss):
            return hasattr(self.model, "kl_loss_sp")
        return _plain_bce(self.loss) is not None and hasattr(self.model, "bce_loss_sp")

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        if not self._fused_ok() or not _model_takes_fused_loss(self.model):
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
        offset, ls = _plain_bce(self.loss), float(self.label_smoothing)
        per_type, totals = {}, {}
        for query_type_index, query_type in enumerate(self.query_types):
            examples = (qtype == query_type_index).nonzero(as_tuple=False).view(-1)
            if len(examples) == 0:
                continue
            rows = examples + row0
            cnt = counts[rows]
            rowptr = torch.zeros(len(rows) + 1, dtype=torch.long, device=cnt.device)
            torch.cumsum(cnt, 0, out=rowptr[1:])
            total = int(rowptr[-1])
            idx = torch.repeat_interleave(offsets[rows] - rowptr[:-1], cnt, output_size=total) \
                + torch.arange(total, device=cnt.device)
            per_type[query_type] = (queries[examples, 0], queries[examples, 1], rowptr, coords[idx, 1].long())
            totals[query_type] = total
        if "s_o" in per_type:
            # relation targets: score_so + the job's loss on the dense [n_q, R] label matrix, exactly as the reference does
            # it (train_KvsAll.py:255-258, 279-291; label smoothing is for entity targets only there, :262-266)
            q0, q1, rowptr, col = per_type.pop("s_o")
            totals.pop("s_o")
            result.forward_time -= time.time()
            nq, R = len(q0), self.dataset.num_relations()
            labels = torch.zeros(nq, R, device=q0.device)
            rws = torch.repeat_interleave(torch.arange(nq, device=q0.device), rowptr[1:] - rowptr[:-1],
                                          output_size=int(col.numel()))
            labels[rws, col] = 1.0
            loss_value = self.loss(self.model.score_so(q0, q1), labels) / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()
        # both query types of the subbatch: their loss rows with ONE backward (two d loss / d score passes, the gradient
        # products once over all rows, no index_add / accumulation passes of autograd in between); the reference
        # back-propagates the two losses one after the other -- the same gradients, accumulated
        if ls == 0.0 and len(per_type) >= 1 and hasattr(self.model, "multilabel_loss_sp_po") and "s_o" not in self.query_types:
            gs = self._graph_step_for(batch_index, batch, subbatch_slice, per_type, totals)
            if gs is not None and gs.enabled:
                result.prepare_time -= time.time()
                inputs = self._graph_inputs(per_type, batch_size, queries.device)
                result.prepare_time += time.time()
                if inputs is None:
                    self._graph_overflows += 1
                else:
                    self._graph_overflows = 0
                    result.forward_time -= time.time()
                    loss_value = gs(*inputs)
                    self._skip_optimizer_step = True  # (replayed or eager: GraphedStep has taken the optimizer's step)
                    self.graph_batches += 1
                    result.avg_loss += loss_value.item()
                    result.forward_time += time.time()
                    return
        if ls == 0.0 and len(per_type) == 2 and hasattr(self.model, "multilabel_loss_sp_po"):
            result.forward_time -= time.time()
            (s_, p_sp, rp_sp, cl_sp), (p_po, o_, rp_po, cl_po) = per_type["sp_"], per_type["_po"]
            # (averaged over the batch, not the subbatch)
            loss_value = self.model.multilabel_loss_sp_po("kl" if offset is None else "bce", s_, p_sp, rp_sp, cl_sp, o_, p_po,
                                                          rp_po, cl_po, 0.0 if offset is None else offset,
                                                          sum_scale=1.0 / batch_size)
            if loss_value is None:
                _declined_late("multilabel_loss_sp_po")
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()
            return
        for query_type, (q0, q1, rowptr, col) in per_type.items():
            result.forward_time -= time.time()
            if offset is None:
                loss_rows = (self.model.kl_loss_sp(q0, q1, rowptr, col, ls) if query_type == "sp_"
                             else self.model.kl_loss_po(q0, q1, rowptr, col, ls))
            else:
                loss_rows = (self.model.bce_loss_sp(q0, q1, rowptr, col, offset, ls) if query_type == "sp_"
                             else self.model.bce_loss_po(q0, q1, rowptr, col, offset, ls))
            if loss_rows is None:
                _declined_late("kl_loss_* / bce_loss_*")
            loss_value = loss_rows.sum() / batch_size  # averaged over the batch, not the subbatch
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            if not self.is_forward_only:
                loss_value.backward()
            result.backward_time += time.time()


class _FusedNegativeScore:
    """Stands in for `BatchNegativeSample.score` (sampler.py:263-344) of ONE slot's sample object for the
    duration of a subbatch: the same call, the same `forward_time` / `prepare_time` attributes -- the negatives
    go to the model's `score_neg` as (positives, neg [n, K]).  Declined by the model: the sampler's own code."""

    def __init__(self, sample, slot):
        self._sample, self._slot, self._score = sample, slot, sample.score

    def __call__(self, model, indexes=None):
        smp = self._sample
        smp.forward_time = smp.prepare_time = 0.0
        smp.prepare_time -= time.time()
        neg = smp.samples(indexes)
        tri = smp.positive_triples[indexes, :] if indexes else smp.positive_triples
        smp.prepare_time += time.time()
        smp.forward_time -= time.time()
        scores = model.score_neg(tri[:, S], tri[:, P], tri[:, O], self._slot, neg)
        smp.forward_time += time.time()
        return scores if scores is not None else self._score(model, indexes=indexes)


class _FusedNsBce(torch.autograd.Function):
    """loss = sum of kge_ns_bce_loss's per-row terms; the kernel writes d loss / d scores in the same pass."""

    @staticmethod
    def forward(ctx, scores, kind, offset, temperature):
        from .. import engine
        rows, grad = engine.ns_bce_loss(scores, kind, offset, temperature, want_grad=True)
        ctx.save_for_backward(grad)
        return rows.sum()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class _HipNsBceLoss:
    """Stands in for the job's BCEWithLogitsKgeLoss (`train.loss: bce | bce_mean | bce_self_adversarial`,
    kge/util/loss.py:136-189) on the [n, 1 + K] score block of a negative-sampling slot: one kernel for the loss and its
    gradient instead of ~15 launches -- and, for the self-adversarial form, instead of two torch.nonzero calls (two
    device -> host waits per slot and step).  Everything it does not recognise -- CPU tensors, index labels, a label
    matrix that is not "column 0 positive, the rest negative" (checked once per label tensor: the job builds it once
    per shape and reuses it, train_negative_sampling.py:128-137) -- goes to the reference loss it wraps."""

    def __init__(self, ref_loss):
        self.ref = ref_loss
        self.kind = {None: "bce", "mean": "bce_mean", "self_adversarial": "bce_self_adversarial"}[ref_loss._bce_type]
        self.offset = float(ref_loss._offset)
        self.temperature = float(getattr(ref_loss, "_temperature", 1.0))
        self._pattern_ok = {}
        self.fused_calls = 0

    def __getattr__(self, name):  # everything else (config, _loss, _offset, ...) is the wrapped loss's
        if name == "ref":  # not set yet (copy / pickle protocols probe attributes on a blank instance): no recursion
            raise AttributeError(name)
        return getattr(self.ref, name)

    def __call__(self, scores, labels, **kwargs):
        if not (torch.is_tensor(labels) and scores.is_cuda and scores.dim() == 2 and scores.dtype == torch.float32
                and labels.dim() == 2 and labels.shape == scores.shape and scores.shape[1] >= 2):
            return self.ref(scores, labels, **kwargs)
        key = (labels.data_ptr(), tuple(labels.shape))
        ok = self._pattern_ok.get(key)
        if ok is None:
            ok = self._pattern_ok[key] = bool((labels[:, 0] == 1).all()) and bool((labels[:, 1:] == 0).all())
        if not ok:
            return self.ref(scores, labels, **kwargs)
        self.fused_calls += 1
        return _FusedNsBce.apply(scores, self.kind, self.offset, self.temperature)


def _fusable_ns_loss(loss) -> bool:
    if not isinstance(loss, BCEWithLogitsKgeLoss) or loss._bce_type not in (None, "mean", "self_adversarial"):
        return False
    inner = loss._loss
    return getattr(inner, "weight", None) is None and getattr(inner, "pos_weight", None) is None


class HipTrainingJobNegativeSampling(_CudaOomText, TrainingJobNegativeSampling):
    """`_process_subbatch` is the reference's (train_negative_sampling.py:103-164), called unchanged: labels,
    positive scores, loss, averaging, backward and every timing key are its code.  What changes is what
    `batch["negative_samples"][slot].score` does for the subject and object slots of a model that offers
    `score_neg` (the hip_* models) when the samples are the per-triple kind (DefaultBatchNegativeSample,
    sampler.py:359-380): instead of an [n*K, 3] index tensor and three [n*K, d] gathers for score_spo
    ("triple"), or an [n, U] matrix against the unique samples ("batch" / "all"), the relation row and the
    uncorrupted entity row of each positive are read once and only the corrupted rows stream (kge_score_neg,
    SURVEY 8f N2); the backward accumulates straight into the table gradients (kge_score_neg_bwd_accum).
    Shared samples (NaiveShared / DefaultShared: their own `score`, which never materialises [n, K] samples)
    and the relation slot stay the sampler's code."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        self.type_str = "negative_sampling"
        # the bce family of losses on a GPU: loss + gradient of a slot's score block in one kernel
        # (hip_negative_sampling.fused_loss: false = the reference's loss object, untouched)
        try:
            fused_loss = bool(config.get("hip_negative_sampling.fused_loss"))
        except KeyError:
            fused_loss = True
        if str(self.device).startswith("cuda") and _fusable_ns_loss(self.loss) and fused_loss:
            self.loss = _HipNsBceLoss(self.loss)
        self._graph_step = None       # kge_amd.train_graph.GraphedStep (hip_negative_sampling.graph_step)
        self._graph_step_ok = None    # decided at the first batch
        self._graph_slots = ()
        self._graph_labels = {}
        self._skip_optimizer_step = False
        if self.__class__ == HipTrainingJobNegativeSampling:
            for f in Job.job_created_hooks:
                f(self)

    # ---- hip_negative_sampling.graph_step: the whole step of a full batch -- positives, the slots' negative blocks, the
    # loss of every slot, backward, optimizer.step -- as ONE hipGraph replay (kge_amd.train_graph.GraphedStep).  Through
    # the trainer a negative-sampling batch is ~50 launches of a few microseconds each, issued by ~1 ms of Python
    # (profiles/r5_libkge_plugin_gpu.jsonl, case b); its inputs have fixed shapes: triples [n, 3], negatives [n, K].
    def _graph_step_for(self, batch_index, batch, subbatch_slice):
        if self._graph_step_ok is False or self.is_forward_only:
            return None
        if self._graph_step_ok is None:
            from kge.util.sampler import DefaultBatchNegativeSample
            try:
                want = bool(self.config.get_default("hip_negative_sampling.graph_step"))
            except KeyError:
                want = False
            ok = want and str(self.device).startswith("cuda") and hasattr(self.model, "score_neg")
            ok = ok and bool(getattr(self.model, "_fused", lambda: False)())
            slots = tuple(slot for slot in (S, O) if self._sampler.num_samples[slot] > 0)
            ok = ok and len(slots) > 0 and self._sampler.num_samples[P] <= 0
            ok = ok and all(type(batch["negative_samples"][slot]) is DefaultBatchNegativeSample for slot in slots)
            # losses that are launches only: the kl loss (log_softmax + kl_div), plain bce, the one-kernel bce stand-ins
            ok = ok and (isinstance(self.loss, (KLDivWithSoftmaxKgeLoss, _HipNsBceLoss))
                         or (isinstance(self.loss, BCEWithLogitsKgeLoss) and self.loss._bce_type is None))
            ok = ok and _optimizer_is_capturable(self.optimizer)
            ok = ok and (_fold_penalties(self) or _no_penalty(self, batch_index, batch))
            self._graph_step_ok, self._graph_slots = ok, slots
            if ok:
                self._graph_step = _graphed_step_of(self, self._graph_loss)
        if not self._graph_step_ok:
            return None
        n = len(batch["triples"])
        sl = subbatch_slice
        whole = sl.start in (0, None) and (sl.stop is None or sl.stop >= n) and sl.step in (1, None)
        return self._graph_step if whole and self.model._fused() else None

    def _graph_loss(self, triples, *args):
        """TrainingJobNegativeSampling._process_subbatch (train_negative_sampling.py:120-163) as one function of the
        batch's tensors: per slot the [n, 1 + K] block (positives | negatives), the job's loss on it / batch size; the
        slots' losses summed (the reference back-propagates them one after the other: the same gradients, accumulated)."""
        negs, inv = args[:-1], args[-1]
        s, p, o = triples[:, S], triples[:, P], triples[:, O]
        n = triples.shape[0]
        total = None
        by_slot = dict(zip(self._graph_slots, negs))
        blocks = None
        if hasattr(self.model, "score_neg_blocks"):  # positives + both slots' blocks as one autograd node
            blocks = self.model.score_neg_blocks(s, p, o, by_slot.get(S), by_slot.get(O))
        for slot, neg in zip(self._graph_slots, negs):
            K = neg.shape[1]
            labels = self._graph_labels.get((slot, n, K))
            if labels is None:
                labels = self._graph_labels[(slot, n, K)] = torch.zeros((n, 1 + K), device=triples.device)
                labels[:, 0] = 1
            if blocks is not None:
                pos, sc = blocks[0], blocks[1 if slot == S else 2]
            else:
                pos = self.model.score_spo(s, p, o, direction="spo"[slot])
                sc = self.model.score_neg(s, p, o, slot, neg)
            if sc is None:
                _declined_late("score_neg")
            scores = torch.cat([pos.view(-1, 1), sc], dim=1)
            part = self.loss(scores, labels, num_negatives=K) * inv
            total = part if total is None else total + part
        return total

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        gs = self._graph_step_for(batch_index, batch, subbatch_slice)
        if gs is not None and gs.enabled:
            batch_size = result.size
            result.prepare_time -= time.time()
            triples = batch["triples"]
            negs = [batch["negative_samples"][slot].samples() for slot in self._graph_slots]
            inv = torch.full((), 1.0 / batch_size, device=triples.device)
            result.prepare_time += time.time()
            result.forward_time -= time.time()
            loss_value = gs(triples, *negs, inv)
            self._skip_optimizer_step = True  # (replayed or eager: GraphedStep has taken the optimizer's step)
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            return
        swapped = []
        if hasattr(self.model, "score_neg"):
            from kge.util.sampler import DefaultBatchNegativeSample
            for slot in (S, O):
                smp = batch["negative_samples"][slot]
                if self._sampler.num_samples[slot] > 0 and type(smp) is DefaultBatchNegativeSample:
                    smp.score = _FusedNegativeScore(smp, slot)  # instance attribute: shadows the method
                    swapped.append(smp)
        try:
            return super()._process_subbatch(batch_index, batch, subbatch_slice, result)
        finally:
            for smp in swapped:
                del smp.score
