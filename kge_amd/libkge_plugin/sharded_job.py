"""The entity-sharded engine behind LibKGE's job API (VERDICT r4 "e-plugin").

    modules: [kge.job, kge.model, kge.model.embedder, kge_amd.libkge_plugin]
    train.type: hip_sharded_1vsAll | hip_sharded_KvsAll | hip_sharded_negative_sampling
    eval.type:  hip_sharded_entity_ranking

LibKGE resolves a job by `class_name` from the packages in `modules` (kge/job/train.py:118-137, kge/job/eval.py:35-48,
config-default.yaml:301-302, 483-484); the classes below are what these names resolve to.  One process per GPU, started
as

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m kge_amd.libkge_plugin.launch start cfg.yaml

(`launch` = `kge start` with a per-rank output folder (<folder>, <folder>-rank1, ...) and `job.device: cuda:<LOCAL_RANK>`, nothing else; an unmodified
`kge.cli` underneath).  Every rank runs the SAME job on the SAME batches (the loader's shuffle and the negative sampler
are seeded at the start of every epoch from one number rank 0 broadcast at job creation + the epoch, and the first
batch of every epoch is compared across ranks by a checksum); what is split is the entity table:

  * rank g owns entity rows [g S, (g + 1) S), S = ceil(E / G), and their optimizer state; its float32 master IS that
    row range of the LibKGE model's own `_entity_embedder._embeddings.weight` (a view, no second copy; negative
    sampling keeps its shard at the head of a buffer with slack rows instead) and the relation master IS the model's
    relation parameter, so model hooks (RotatE's normalize_phases, rotate.py:103-143) act on what is trained;
  * the optimizer is the configured one (`train.optimizer.default`; parameter groups by regex are not supported) over
    [this rank's entity rows, the replicated relation table]; LibKGE's scheduler, warm-up, early stopping, tracing
    and checkpoint schedule run unchanged on top of it;
  * a subbatch is TrainingJob1vsAll / KvsAll / NegativeSampling._process_subbatch (train_1vsAll.py:48-82,
    train_KvsAll.py:216-294, train_negative_sampling.py:103-164) with the scores and the loss coming from
    kge_amd.sharded.ShardedEntityTable: fused score + loss kernels on the shard, the exchange steps of SURVEY 8e;
  * validation (`hip_sharded_entity_ranking`): every rank counts over its own entity rows inside the scoring kernel,
    ONE int64 all-reduce per batch, the reference's `_get_ranks` / `_compute_metrics` / hooks / traces on every rank;
  * a checkpoint is the reference's (train.py:284-298): the shards are gathered into the model's [E, d] parameter and
    the per-row optimizer state into torch's unsharded layout; rank 0 writes the file.  It resumes on any number of
    ranks, sharded or not (`kge resume`, `kge valid`, `kge test` of an unmodified LibKGE included).

Penalty terms (lookup_embedder.regularize lp / n3, weighted or not: lookup_embedder.py:122-177, kge_model.py:603-649)
are computed per shard -- every rank sums the terms of the entity rows it owns, ONE scalar all-reduce gives the value
the trace shows, the gradient lands on the owned rows; the relation embedder's term is the same on every rank -- and
embedder dropout (lookup_embedder.py:64-69, 102-105) is applied in front of the dense-row loss kernels (1vsAll and
KvsAll; kge_amd.sharded: _dropout_loss).  Every rank starts from rank 0's parameters (one broadcast at job creation).

What it declines, loudly (ValueError at job creation): embedder dropout under negative sampling,
entity and relation embedders that are not plain LookupEmbedders shared between the s and o slot, a reciprocal-relations
wrapper under negative sampling (1vsAll / KvsAll / evaluation take it: the table scores its subject direction as an sp_
query with relation p + R, kge_amd.sharded: _recip), a second optimizer TYPE in a parameter group (groups with their own
args are taken), `train.loss` other than kl (1vsAll, KvsAll, negative sampling) or plain bce
(KvsAll, negative sampling), KvsAll.label_smoothing under bce (kl takes it), s_o queries, negatives for the relation slot.

CPU / gloo: the CPU test (tests/test_libkge_sharded_plugin_cpu.py) runs these jobs on two gloo ranks with the test
suite's stand-in backend handed in through `SHARD_BACKEND`; the product default is kge_amd.engine (HIP kernels, no CPU
fallback).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from kge.job import Job
from kge.job.train_1vsAll import TrainingJob1vsAll
from kge.job.train_KvsAll import TrainingJobKvsAll
from kge.job.train_negative_sampling import TrainingJobNegativeSampling, S, P, O
from kge.util import KgeLRScheduler
from kge.util.loss import BCEWithLogitsKgeLoss, KLDivWithSoftmaxKgeLoss

from ..eval import FilterIndex
from ..sharded import ShardedEntityTable
from ..sharded_train import ENT_KEY, REL_KEY, _ShardedJob
from .eval_job import HipEntityRankingJob
from .train_job import _CudaOomText, _plain_bce

# The scoring backend of a job.device cpu job (None: kge_amd.engine, the HIP kernels -- which have no CPU path, so such a
# job fails loudly).  The CPU test suite assigns its oracle-backed stand-in to this attribute in the processes it starts
# (tests/_launch_with_oracle_backend.py for jobs under torchrun); nothing in the package reads the environment for it,
# and on a GPU it is never looked at.
SHARD_BACKEND = None


def _backend_for(device):
    if torch.device(device).type == "cuda":
        return None
    return SHARD_BACKEND

_SCORER_BY_CLASS = {"ComplExScorer": "complex", "DistMultScorer": "distmult", "TransEScorer": "transe",
                    "RotatEScorer": "rotate", "HipComplExScorer": "complex", "HipDistMultScorer": "distmult",
                    "HipTransEScorer": "transe", "HipRotatEScorer": "rotate"}


def init_process_group(device) -> None:
    """torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*), once: RCCL ("nccl") for a GPU
    job, gloo for job.device cpu.  Without that environment the job runs as ONE shard (no collectives)."""
    if dist.is_initialized() or "WORLD_SIZE" not in os.environ or "MASTER_ADDR" not in os.environ:
        return
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")


def _scorer_of(model):
    sc = model.get_scorer()
    name = getattr(sc, "name", None) or _SCORER_BY_CLASS.get(type(sc).__name__)
    if name not in ("complex", "distmult", "transe", "rotate"):
        raise ValueError(f"kge_amd: hip_sharded_* jobs score ComplEx / DistMult / TransE / RotatE; got {type(sc).__name__}")
    l_norm = float(sc.get_option("l_norm")) if name in ("transe", "rotate") else 1.0
    return name, l_norm


def _reciprocal_R(model) -> int:
    """The number of relations R of a reciprocal-relations wrapper (kge/model/reciprocal_relations_model.py: 2 R relation
    rows, the subject direction scores with relation p + R), 0 for any other model."""
    from kge.model.reciprocal_relations_model import ReciprocalRelationsModel
    return int(model.dataset.num_relations()) if isinstance(model, ReciprocalRelationsModel) else 0


def _check_model(model, allow_dropout=True, allow_reciprocal=True):
    """The model shapes the sharded table can stand in for (module docstring); -> (entity weight, relation weight)."""
    from kge.model import LookupEmbedder
    if _reciprocal_R(model) and not allow_reciprocal:
        # (its embedders ARE the base model's plain lookup tables, so the check below would pass -- and the job would then
        # score the subject direction with relation p instead of p + R: a different model, silently)
        raise ValueError("kge_amd: hip_sharded_negative_sampling does not take a reciprocal-relations wrapper (its subject "
                         "direction scores with relation p + R); hip_sharded_1vsAll / hip_sharded_KvsAll do")
    se, oe, pe = model.get_s_embedder(), model.get_o_embedder(), model.get_p_embedder()
    if se is not oe or type(se) is not LookupEmbedder or type(pe) is not LookupEmbedder:
        raise ValueError("kge_amd: hip_sharded_* jobs need plain lookup embedders, the entity embedder shared by the "
                         "subject and object slot (no reciprocal-relations wrapper)")
    for e, what in ((se, "entity"), (pe, "relation")):
        if e.dropout.p > 0 and not allow_dropout:
            raise ValueError(f"kge_amd: this hip_sharded_* job does not support {what} embedder dropout")
        if e.regularize not in ("", "lp", "n3"):
            raise ValueError(f"Invalid value regularize={e.regularize}")  # (the reference's text, at the first penalty)
    return se._embeddings.weight, pe._embeddings.weight


def _score_dtype(config, key, scorer, dim, device, ns=False):
    """`<train.type>.score_dtype`: float32 | bfloat16 | auto (bfloat16 where the fused score + loss kernels take it:
    ComplEx / DistMult, dim 128 / 256 / 512, on a GPU; negative sampling: always float32, the gather kernels' dtype)."""
    try:
        want = str(config.get(key + ".score_dtype"))
    except KeyError:
        want = "auto"
    if ns:
        return torch.float32
    if want in ("float32", "fp32", "f32"):
        return torch.float32
    if want in ("bfloat16", "bf16"):
        return torch.bfloat16
    if want != "auto":
        raise ValueError(f"kge_amd: {key}.score_dtype must be float32, bfloat16 or auto")
    ok = torch.device(device).type == "cuda" and scorer in ("complex", "distmult") and dim in (128, 256, 512)
    return torch.bfloat16 if ok else torch.float32


class _ShardState(_ShardedJob):
    """kge_amd.sharded_train._ShardedJob as the state behind a LibKGE job: this rank's masters (views of the model's
    parameters), the sharded scoring table, the optimizer on the owned rows, gather / scatter of checkpoints."""

    def __init__(self, job, slack_rows=0):
        cfg, model = job.config, job.model
        ent_w, rel_w = _check_model(model, allow_dropout=slack_rows <= 0, allow_reciprocal=slack_rows <= 0)
        scorer, l_norm = _scorer_of(model)
        dev = torch.device(job.device)
        if ent_w.device != dev and not (dev.type == "cuda" and dev.index is None and ent_w.is_cuda):
            raise ValueError(f"kge_amd: the model is on {ent_w.device}, job.device is {job.device}")
        dev = ent_w.device
        init_process_group(dev)
        # Every rank built its own model: LibKGE's default is random_seed -1, and nothing else makes the ranks' initial
        # parameters equal.  Rank 0's are everybody's (the relation table is replicated and its gradient is assumed
        # identical on every rank; only rank 0's copy reaches the checkpoint) -- ADVICE r5.
        if dist.is_initialized() and dist.get_world_size() > 1:
            with torch.no_grad():
                for w in (ent_w, rel_w):
                    dist.broadcast(w.data, src=0)
        E, d = ent_w.shape
        R, dr = rel_w.shape
        key = cfg.get("train.type")
        ns = slack_rows > 0
        sd = _score_dtype(cfg, key, scorer, d, dev, ns=ns)
        # train.optimizer.<group> (kge/util/optimizer.py:28-95): parameters grouped by a regex over their names, a group's
        # args on top of the default's; the groups first, in the configuration's order, then "default" -- the order torch
        # numbers the parameters in, i.e. the checkpoint's.  Here there are two parameters to place.
        import re
        name_of = {id(v): k for k, v in model.named_parameters()}
        names = {"ent": name_of[id(ent_w)], "rel": name_of[id(rel_w)]}
        self._opt_groups, taken = [], {}
        for gname, g in cfg.get("train.optimizer").items():
            if gname == "default":
                continue
            if "type" in g:
                raise NotImplementedError("Multiple optimizer types are not yet supported.")  # (the reference's text)
            pat = re.compile(g["regex"])
            members = [w for w in ("ent", "rel") if pat.match(names[w])]
            for w in members:
                if w in taken:
                    raise ValueError(f"The parameters {{'{names[w]}'}}, were matched by the optimizer group {taken[w]} "
                                     f"and ['{gname}']")
                taken[w] = gname
            self._opt_groups.append((gname, dict(g.get("args") or {}), members))
        if self._opt_groups:
            order = [w for _, _, ms in self._opt_groups for w in ms]
            self._param_order = tuple(order + [w for w in ("ent", "rel") if w not in order])
        args = dict(cfg.get("train.optimizer.default.args"))
        lr = args.pop("lr", None)
        state = {ENT_KEY: ent_w.data, REL_KEY: rel_w.data}
        self.model_ent, self.model_rel = ent_w, rel_w
        # (make_optimizer below needs these before the base constructor builds the optimizer)
        self._opt_type, self._opt_lr = cfg.get("train.optimizer.default.type"), lr
        super().__init__(scorer, E, R, d, rel_dim=dr, state_dict=state, lr=lr if lr is not None else 0.0,
                         optimizer=self._opt_type, optimizer_args=args, score_dtype=sd, device=dev,
                         backend=_backend_for(dev), l_norm=l_norm, slack_rows=slack_rows, config=cfg, alias_state=True)
        self.table.reciprocal_R = _reciprocal_R(model)  # (every "po" of the table becomes "sp" with p + R: sharded.py _recip)
        if dev.type == "cuda" and sd != torch.float32 and not ns:
            from .. import engine
            if _backend_for(dev) is None and not engine.ce_supported(self.table._tables(self.table.ent_local, "local")):
                raise ValueError(f"kge_amd: no fused score + loss kernel for {scorer} at dim {d} in {sd}")
        elif dev.type == "cuda" and not ns and _backend_for(dev) is None:
            raise ValueError(f"kge_amd: {key} on a GPU needs score_dtype bfloat16 (ComplEx / DistMult, dim 128 / 256 / "
                             f"512): the fused score + loss kernels read bf16 tables; got {scorer}, dim {d}, {sd}")

    def _make_optimizer(self, name, lr, args):
        params = [self.ent_master, self.rel_master]
        if getattr(self, "_opt_groups", None):
            master = {"ent": self.ent_master, "rel": self.rel_master}
            placed = [w for _, _, ms in self._opt_groups for w in ms]
            params = [dict(ga, params=[master[w] for w in ms], name=gname) for gname, ga, ms in self._opt_groups]
            params.append({"params": [master[w] for w in ("ent", "rel") if w not in placed], "name": "default"})
        kw = dict(args)
        if self._opt_lr is not None:
            kw["lr"] = self._opt_lr
        if name in ("Adagrad", "HipAdagrad") and self.ent_master.is_cuda:
            from ..optim import Adagrad
            return Adagrad(params, **kw)
        if name == "HipAdam" and self.ent_master.is_cuda:
            from ..optim import Adam
            return Adam(params, **kw)
        name = {"HipAdagrad": "Adagrad", "HipAdam": "Adam"}.get(name, name)
        kw.pop("bf16_copies", None)
        try:
            return getattr(torch.optim, name)(params, **kw)
        except AttributeError:
            raise ValueError(f"Could not create optimizer {name}. Please specify an optimizer provided in torch.optim")

    @torch.no_grad()
    def sync_model(self):
        """Collective: every rank's rows into every rank's [E, d] model parameter (own rows are already there when the
        master is a view of it; the relation parameter IS the master)."""
        if self.world == 1 and self.ent_master.data_ptr() == self.model_ent.data_ptr():
            return
        full = self._gather_rows(self.ent_master)
        self.model_ent.data.copy_(full.to(self.model_ent.device))

    @torch.no_grad()
    def load_from_model(self):
        """The inverse (after the model's parameters were replaced: a checkpoint loaded into the same model)."""
        if self.ent_master.data_ptr() != self.model_ent.data[self.lo:self.hi].data_ptr() and self.hi > self.lo:
            self.ent_master.copy_(self.model_ent.data[self.lo:self.hi])
        self._after_step()


def _lookup_penalty(emb, rows, lo, indexes, allreduce):
    """LookupEmbedder.penalty (kge/model/embedder/lookup_embedder.py:122-177) over `rows` = the rows [lo, lo + len(rows))
    of the embedder's table that THIS rank owns (a leaf the optimizer steps): the same torch operations on the owned
    rows only.  The Lp / N3 penalty is a sum of per-row terms, so the shards' sums add up to the reference's value:
    `allreduce` (None for the replicated relation table) makes the returned VALUE global while the gradient stays on
    the owned rows (value = local + (global - local).detach()).  -> [] or [(key, 0-d tensor)]."""
    if emb.regularize == "" or emb.get_option("regularize_weight") == 0.0:
        return []
    if emb.regularize not in ("lp", "n3"):
        raise ValueError(f"Invalid value regularize={emb.regularize}")
    if emb.regularize == "n3":
        p = 3
    else:
        p = emb.get_option("regularize_args.p") if emb.has_option("regularize_args.p") else 2
    weight = emb._get_regularize_weight()
    if not emb.get_option("regularize_args.weighted"):
        parameters = rows
        if emb.regularize == "n3" and emb.space == "complex":
            parameters = emb._abs_complex(parameters)
        local = (weight / p * parameters.norm(p=p) ** p).sum()
    else:
        if indexes is None:
            raise KeyError("indexes")  # (the reference's failure for a weighted penalty on a batch without triples)
        unique_indexes, counts = torch.unique(indexes, return_counts=True)
        mine = (unique_indexes >= lo) & (unique_indexes < lo + rows.shape[0])
        parameters = rows[(unique_indexes[mine] - lo).long()]
        counts = counts[mine]
        if emb.regularize == "n3" and emb.space == "complex":
            parameters = emb._abs_complex(parameters)
        if (p % 2 == 1) and (emb.regularize != "n3"):
            parameters = torch.abs(parameters)
        local = (weight / p * (parameters ** p * counts.float().view(-1, 1))).sum() / len(indexes)
    if allreduce is not None:
        total = local.detach().clone().view(1)
        allreduce(total)
        local = local + (total.view(()) - local.detach())
    return [(f"{emb.configuration_key}.L{p}_penalty", local)]


def sharded_model_penalty(model, sh, **kwargs):
    """KgeModel.penalty (kge/model/kge_model.py:603-649) for a model whose entity rows are trained as shards: the
    relation embedder's terms on the replicated table, the (shared) entity embedder's on this rank's rows -- weighted:
    the batch's s and o ids, unweighted: all rows, doubled as the reference doubles it."""
    from kge.job.train_negative_sampling import S as _S, P as _P, O as _O
    pe, se = model.get_p_embedder(), model.get_s_embedder()
    Eg = sh.hi - sh.lo
    ent_rows, rel_rows = sh.ent_master[:Eg], sh.rel_master

    def allreduce(t):
        if sh.world > 1:
            dev = t.device
            buf = t if dist.get_backend(sh.group) != "gloo" else t.cpu()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=sh.group)
            if buf is not t:
                t.copy_(buf.to(dev))
        return t
    triples = None
    if "batch" in kwargs and "triples" in kwargs["batch"]:
        triples = kwargs["batch"]["triples"].to(ent_rows.device)
    result = _lookup_penalty(pe, rel_rows, 0, None if triples is None else triples[:, _P], None)
    weighted = se.get_option("regularize_args.weighted")
    entity_indexes = None
    if triples is not None and weighted:
        entity_indexes = torch.cat((triples[:, _S].view(-1, 1), triples[:, _O].view(-1, 1)), dim=1)
    ent_pen = _lookup_penalty(se, ent_rows, sh.lo, entity_indexes, allreduce)
    if not (triples is not None and weighted):
        ent_pen = [(k, v * 2) for k, v in ent_pen]  # (kge_model.py:620-625 / 636-640: "backwards compatibility")
    result = result + ent_pen
    # a reciprocal-relations wrapper adds the weighted term of the reciprocal relation rows p + R
    # (reciprocal_relations_model.py:59-72), behind the base model's terms
    R = _reciprocal_R(model)
    if R and pe.get_option("regularize_args.weighted") and pe.regularize != "" and pe.get_option("regularize_weight") != 0.0:
        if triples is None:
            raise KeyError("batch")
        result = result + _lookup_penalty(pe, rel_rows, 0, triples[:, _P] + R, None)
    return result


def _batch_checksum(batch) -> torch.Tensor:
    acc = []
    for key in ("triples", "queries"):
        t = batch.get(key)
        if torch.is_tensor(t):
            acc.append(t.long().sum().cpu())
    for ns in batch.get("negative_samples") or []:
        smp = getattr(ns, "_samples", None)
        if torch.is_tensor(smp):
            acc.append(smp.long().sum().cpu())
    return torch.stack(acc).sum().view(1).double() if acc else torch.zeros(1, dtype=torch.double)


def epoch_seed(base: int, epoch: int) -> int:
    return (int(base) + 1000003 * int(epoch)) % (2 ** 31 - 1)


class _ShardedTrainMixin(_CudaOomText):
    """What the three sharded training jobs share (module docstring).  Mixed in FRONT of a reference TrainingJob*."""

    SLACK = False

    def _sharded_init(self):
        if self.is_forward_only:
            raise ValueError("kge_amd: hip_sharded_* training jobs are not available forward-only")
        slack = int(self.config.get("train.batch_size")) if self.SLACK else 0
        self._sh = _ShardState(self, slack_rows=slack)
        sh = self._sh
        # the job's optimizer and scheduler are the ones over this rank's rows (train.py:87-92)
        self.optimizer = sh.optimizer
        self.kge_lr_scheduler = KgeLRScheduler(self.config, self.optimizer)
        for group in self.optimizer.param_groups:
            group["initial_lr"] = group["lr"]
        self.optimizer.register_step_post_hook(lambda opt, a, k: sh._after_step())
        # TrainingJob.run_epoch calls self.model.penalty(...) and back-propagates every term (train.py:417-436): on this
        # job the terms are the shards' (sharded_model_penalty)
        model = self.model
        model.penalty = lambda **kw: sharded_model_penalty(model, sh, **kw)
        # embedder dropout of a training step (1vsAll, KvsAll: kge_amd.sharded._dropout_loss); the table's mask of a
        # multi-rank job from a generator of this rank's own (the default generators draw alike on every rank)
        self._dropout = (float(model.get_s_embedder().dropout.p), float(model.get_p_embedder().dropout.p))
        if max(self._dropout) > 0 and sh.world > 1:
            dev = sh.ent_master.device
            sh.table.table_generator = torch.Generator(device=dev)
        self._epoch_checked = -1
        self._seed_base = self._draw_seed_base()
        # validation: a sharded evaluation job finds the table through its parent; any other evaluation job reads the
        # model, which must hold every rank's rows first
        vj = getattr(self, "valid_job", None)
        if vj is not None and not isinstance(vj, HipShardedEntityRankingJob):
            run = vj.run

            def synced_run(*a, **k):
                sh.sync_model()
                return run(*a, **k)
            vj.run = synced_run

    # ---- the same batches on every rank -------------------------------------------------------------------------------
    def _draw_seed_base(self):
        """One number from rank 0's torch generator, at job creation (module docstring)."""
        sh = self._sh
        seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
        if sh.world > 1:
            dev = sh.ent_master.device if dist.get_backend(sh.group) == "nccl" else torch.device("cpu")
            seed = seed.to(dev)
            dist.broadcast(seed, src=0, group=sh.group)
            seed = seed.cpu()
        return int(seed)

    def _seed_epoch(self):
        """The loader's shuffle (torch) and the samplers (torch / numpy / random, kge/util/sampler.py:593-741) draw from
        the process-wide generators: every rank seeds them at the start of every epoch with a number that depends on
        nothing but the job's base seed and the epoch -- not on how many draws validation, checkpointing or a hook made
        on this rank in between."""
        import random
        v = epoch_seed(self._seed_base, self.epoch)
        torch.manual_seed(v)
        tg = getattr(self._sh.table, "table_generator", None)
        if tg is not None:
            tg.manual_seed((v * 131 + 7919 * (self._sh.rank + 1)) % (2 ** 31 - 1))
        np.random.seed(v % (2 ** 32))
        random.seed(v)

    def run_epoch(self):
        self._seed_epoch()
        return super().run_epoch()

    def _check_same_batch(self, batch):
        sh = self._sh
        if sh.world == 1 or self._epoch_checked == self.epoch:
            return
        self._epoch_checked = self.epoch
        c = _batch_checksum(batch)
        dev = sh.ent_master.device if dist.get_backend(sh.group) == "nccl" else torch.device("cpu")
        lo, hi = c.clone().to(dev), c.clone().to(dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=sh.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=sh.group)
        if float(lo) != float(hi):
            raise RuntimeError("kge_amd: the ranks of a hip_sharded_* job drew different batches (train.num_workers with "
                               "a worker_init_fn that seeds per process? a sampler that does not draw from the "
                               "process-wide generators?)")

    def _prepare_batch(self, batch_index, batch, result):
        super()._prepare_batch(batch_index, batch, result)
        self._check_same_batch(batch)

    # ---- checkpoints: the reference's layout, written by rank 0 ---------------------------------------------------------
    def save(self, filename) -> None:
        checkpoint = self.save_to({})  # collective
        if self._sh.rank == 0:
            self.config.log("Saving checkpoint to {}...".format(filename))
            torch.save(checkpoint, filename)
        if self._sh.world > 1:
            dist.barrier(group=self._sh.group)

    def save_to(self, checkpoint):
        sh = self._sh
        sh.sync_model()
        real = self.optimizer

        class _Gathered:  # TrainingJob.save_to calls optimizer.state_dict(): the unsharded layout
            @staticmethod
            def state_dict():
                return sh.optimizer_state_dict()
        self.optimizer = _Gathered
        try:
            return super().save_to(checkpoint)
        finally:
            self.optimizer = real

    def _delete_checkpoint(self, checkpoint_id):
        if self._sh.rank == 0:
            super()._delete_checkpoint(checkpoint_id)

    def _load(self, checkpoint):
        sh = self._sh
        real = self.optimizer

        class _Scatter:  # TrainingJob._load calls optimizer.load_state_dict(): onto this rank's rows
            @staticmethod
            def load_state_dict(sd):
                sh.load_optimizer_state_dict(sd)
        self.optimizer = _Scatter
        try:
            super()._load(checkpoint)
        finally:
            self.optimizer = real
        sh.load_from_model()  # (Job.create_from loaded the checkpoint's parameters into the model before this job was built)


class HipShardedTrainingJob1vsAll(_ShardedTrainMixin, TrainingJob1vsAll):
    """train.type: hip_sharded_1vsAll -- TrainingJob1vsAll._process_subbatch (train_1vsAll.py:48-82) over the sharded
    table: per direction the cross entropy over ALL entities (ShardedEntityTable.ce_loss), `sum` / batch size, one
    backward per direction."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        if not isinstance(self.loss, KLDivWithSoftmaxKgeLoss):
            raise ValueError("kge_amd: hip_sharded_1vsAll supports train.loss: kl")
        self._sharded_init()
        if self.__class__ == HipShardedTrainingJob1vsAll:
            for f in Job.job_created_hooks:
                f(self)

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        sh = self._sh
        batch_size = result.size
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(sh.ent_master.device)
        result.prepare_time += time.time()
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        for direction, ids, labels in (("sp", s, o), ("po", o, s)):
            result.forward_time -= time.time()
            rows = sh.table.ce_loss(direction, ids, p, labels, sh.ent_master, sh.rel_master,
                                    dropout=self._dropout if self.model.training else None)
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            loss_value.backward()
            result.backward_time += time.time()


def label_csr_of_queries(batch, subbatch_slice, batch_size, examples):
    """The label CSR (rowptr [k + 1], col [nnz]: GLOBAL entity ids) of the subbatch rows `examples` cut out of the
    batch's `label_coords` ([nnz, 2] = (batch row, label), rows ascending: train_KvsAll.py:178-214) -- no host sync."""
    coords = batch["label_coords"]
    row0 = subbatch_slice.start or 0
    counts = torch.bincount(coords[:, 0].long(), minlength=batch_size)
    offsets = torch.zeros(batch_size + 1, dtype=torch.long, device=coords.device)
    torch.cumsum(counts, 0, out=offsets[1:])
    rows = examples + row0
    cnt = counts[rows]
    rowptr = torch.zeros(len(rows) + 1, dtype=torch.long, device=cnt.device)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    total = int(rowptr[-1])
    idx = torch.repeat_interleave(offsets[rows] - rowptr[:-1], cnt, output_size=total) + torch.arange(total, device=cnt.device)
    return rowptr, coords[idx, 1].long()


class HipShardedTrainingJobKvsAll(_ShardedTrainMixin, TrainingJobKvsAll):
    """train.type: hip_sharded_KvsAll -- TrainingJobKvsAll._process_subbatch (train_KvsAll.py:216-294): sp_ and _po
    queries with their multi-hot labels (a CSR of global ids cut out of the batch's label_coords, the same on every
    rank: each shard's kernel takes the labels it owns), train.loss kl or plain bce."""

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        if self.config.get("KvsAll.query_types").get("s_o"):
            raise ValueError("kge_amd: hip_sharded_KvsAll scores sp_ and _po queries (KvsAll.query_types.s_o: false)")
        self._bce_offset = _plain_bce(self.loss)
        if float(self.label_smoothing) != 0.0 and self._bce_offset is not None:
            raise ValueError("kge_amd: hip_sharded_KvsAll supports KvsAll.label_smoothing with train.loss: kl (not bce)")
        if not isinstance(self.loss, KLDivWithSoftmaxKgeLoss) and self._bce_offset is None:
            raise ValueError("kge_amd: hip_sharded_KvsAll supports train.loss: kl and plain bce")
        self._sharded_init()
        if self.__class__ == HipShardedTrainingJobKvsAll:
            for f in Job.job_created_hooks:
                f(self)

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        sh = self._sh
        dev = sh.ent_master.device
        batch_size = result.size
        result.prepare_time -= time.time()
        queries = batch["queries"][subbatch_slice].to(dev)
        qtype = batch["query_type_indexes"][subbatch_slice].to(dev)
        result.prepare_time += time.time()
        for query_type_index, query_type in enumerate(self.query_types):
            examples = (qtype == query_type_index).nonzero(as_tuple=False).view(-1)
            if len(examples) == 0:
                continue
            result.forward_time -= time.time()
            rowptr, col = label_csr_of_queries(batch, subbatch_slice, batch_size, examples)
            q0, q1 = queries[examples, 0], queries[examples, 1]
            # sp_: (s, p) rows, labels = objects;  _po: (p, o) rows, labels = subjects
            direction, ids, p = ("sp", q0, q1) if query_type == "sp_" else ("po", q1, q0)
            if self._bce_offset is None:
                rows = sh.table.kl_loss(direction, ids, p, rowptr, col, sh.ent_master, sh.rel_master,
                                        dropout=self._dropout if self.model.training else None,
                                        label_smoothing=float(self.label_smoothing))
            else:
                rows = sh.table.bce_loss(direction, ids, p, rowptr, col, self._bce_offset, sh.ent_master, sh.rel_master,
                                         dropout=self._dropout if self.model.training else None)
            loss_value = rows.sum() / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            loss_value.backward()
            result.backward_time += time.time()


class HipShardedTrainingJobNegativeSampling(_ShardedTrainMixin, TrainingJobNegativeSampling):
    """train.type: hip_sharded_negative_sampling -- TrainingJobNegativeSampling._process_subbatch
    (train_negative_sampling.py:103-164) with the subject- and object-slot blocks [n, 1 + K] (positive, negatives) from
    ShardedEntityTable.neg_scores: every rank scores the negatives it owns (kge_score_neg on local row ids, the batch's
    own s / o rows exchanged into slack rows behind the shard), one all-reduce of [n, K] floats; the job's own loss
    object on the block, / batch size, one backward per slot.  float32 tables (the gather-bound kernels' dtype)."""

    SLACK = True

    def __init__(self, config, dataset, parent_job=None, model=None, forward_only=False):
        super().__init__(config, dataset, parent_job, model=model, forward_only=forward_only)
        self.type_str = "negative_sampling"
        if self._sampler.num_samples[P] > 0:
            raise ValueError("kge_amd: hip_sharded_negative_sampling corrupts subjects and objects "
                             "(negative_sampling.num_samples.p: 0)")
        if not isinstance(self.loss, (KLDivWithSoftmaxKgeLoss, BCEWithLogitsKgeLoss)):
            raise ValueError("kge_amd: hip_sharded_negative_sampling supports train.loss: kl and the bce family")
        self._sharded_init()
        if self.__class__ == HipShardedTrainingJobNegativeSampling:
            for f in Job.job_created_hooks:
                f(self)

    def _process_subbatch(self, batch_index, batch, subbatch_slice, result):
        sh = self._sh
        dev = sh.ent_master.device
        batch_size = result.size
        result.prepare_time -= time.time()
        triples = batch["triples"][subbatch_slice].to(dev)
        subbatch_size = len(triples)
        s, p, o = (triples[:, k].contiguous() for k in (S, P, O))
        labels = batch["labels"]  # reused between subbatches (train_negative_sampling.py:99, 123-137)
        result.prepare_time += time.time()
        for slot in (S, O):
            num_samples = self._sampler.num_samples[slot]
            if num_samples <= 0:
                continue
            if labels[slot] is None or labels[slot].shape != (subbatch_size, 1 + num_samples):
                result.prepare_time -= time.time()
                labels[slot] = torch.zeros((subbatch_size, 1 + num_samples), device=dev)
                labels[slot][:, 0] = 1
                result.prepare_time += time.time()
            result.prepare_time -= time.time()
            neg = batch["negative_samples"][slot].samples(subbatch_slice).to(dev).long().contiguous()
            result.prepare_time += time.time()
            result.forward_time -= time.time()
            pos, sc = sh.table.neg_scores(s, p, o, int(slot), neg, sh.ent_master, sh.rel_master)
            scores = torch.cat([pos.view(-1, 1), sc], dim=1)
            loss_value = self.loss(scores, labels[slot], num_negatives=num_samples) / batch_size
            result.avg_loss += loss_value.item()
            result.forward_time += time.time()
            result.backward_time -= time.time()
            loss_value.backward()
            result.backward_time += time.time()


class HipShardedEntityRankingJob(HipEntityRankingJob):
    """eval.type: hip_sharded_entity_ranking -- HipEntityRankingJob's loop (EntityRankingJob._evaluate,
    eval_entity_ranking.py:103-481: hooks, traces, tie policy, histograms and metrics are the reference's, on every
    rank) with the counts of a batch from the sharded table: each rank counts over its own entity rows -- inside the
    scoring kernel where the backend offers it --, ONE int64 all-reduce per batch makes them global
    (ShardedEntityTable.rank_batch_multi; the reference sums per-chunk counts the same way, :222-313).

    The table: the parent training job's (validation inside a hip_sharded_* training run: what is being trained, no
    gather), else this rank's rows of the model's parameters (`kge valid` / `kge test` of a checkpoint)."""

    def _sharded_table(self):
        parent = getattr(self, "parent_job", None)
        st = getattr(parent, "_sh", None)
        if st is not None:
            return st.table
        tb = getattr(self, "_own_table", None)
        if tb is None:
            ent_w, rel_w = _check_model(self.model)
            scorer, l_norm = _scorer_of(self.model)
            init_process_group(ent_w.device)
            E, d = ent_w.shape
            try:
                key = self.config.get("eval.type")
            except KeyError:
                key = "hip_sharded_entity_ranking"
            sd = _score_dtype(self.config, key, scorer, d, ent_w.device)
            world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
            lo, hi = ShardedEntityTable.partition(E, world, rank)
            tb = ShardedEntityTable(scorer, ent_w.detach()[lo:hi].to(sd).contiguous(), rel_w.detach().to(sd).contiguous(),
                                    E, l_norm=l_norm, backend=_backend_for(ent_w.device))
            tb.reciprocal_R = _reciprocal_R(self.model)
            self._own_table = tb
        return tb

    def _fast_path(self) -> bool:
        self._sharded_table()  # raises for a model the sharded table cannot stand in for
        return True

    def _prepare(self):
        super()._prepare()
        # job.device cpu (gloo tests): the ranges of a batch from the host-side index instead of kge_filter_lookup
        E, R = self.dataset.num_entities(), self.dataset.num_relations()
        splits = [self.dataset.split(s).numpy() for s in self.filter_splits]
        self._host_index = [FilterIndex(splits, E, R)]
        if self._hip_filter_with_test:
            self._host_index.append(FilterIndex(splits + [self.dataset.split("test").numpy()], E, R))

    def _eval_begin(self, M, chunk_size):
        # Without a sharded parent job the table is a SNAPSHOT of the model's rows (a bf16 copy under score_dtype auto
        # on a GPU): taken anew at every run -- a training job validating through this job moves the weights between
        # runs, and a table kept from the first validation froze every later metric.
        if getattr(getattr(self, "parent_job", None), "_sh", None) is None:
            self._own_table = None
        E = self.dataset.num_entities()
        if chunk_size < E:
            self.config.log("hip_sharded_entity_ranking: entity_ranking.chunk_size is ignored (a rank's chunk is its shard)")
        self._ev = {}

    def _batch_counts(self, batch, M, chunk_size):
        sh = self._sharded_table()
        E, R = self.dataset.num_entities(), self.dataset.num_relations()
        dev = batch.device
        s, p, o = batch[:, 0], batch[:, 1], batch[:, 2]
        n = batch.shape[0]
        filt_o, filt_s = [], []
        if dev.type == "cuda":
            from .. import engine
            rng = torch.empty(2, M - 1, 2, n, dtype=torch.int64, device=dev)
            lookups = []
            for k in range(M - 1):
                uk, start, v = self._hip_sp[k]
                lookups.append((uk, start, s, p, R, rng[0, k, 0], rng[0, k, 1]))
                filt_o.append((rng[0, k, 0], rng[0, k, 1], v))
                uk, start, v = self._hip_po[k]
                lookups.append((uk, start, p, o, E, rng[1, k, 0], rng[1, k, 1]))
                filt_s.append((rng[1, k, 0], rng[1, k, 1], v))
            engine.filter_lookup_multi(lookups)
        else:
            b = batch.cpu().numpy()
            for k in range(M - 1):
                sb, se, pb, pe = (torch.from_numpy(x) for x in self._host_index[k].ranges(b))
                filt_o.append((sb, se, self._hip_sp[k][2]))
                filt_s.append((pb, pe, self._hip_po[k][2]))
        return sh.rank_batch_multi(batch, filt_o, filt_s, self.tie_atol, self.tie_rtol)
