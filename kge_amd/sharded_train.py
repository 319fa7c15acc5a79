"""Training over an entity-sharded model: one process per GPU, `torch.distributed` (RCCL over xGMI; the CPU tests run
it on "gloo" with a fake backend).  Three jobs, the three training types of the reference:

  ShardedTrainingJob1vsAll            TrainingJob1vsAll._process_subbatch (kge/job/train_1vsAll.py:48-82): score_sp
                                      against all entities with labels o, score_po with labels s, KL / CE loss with
                                      reduction="sum" divided by the batch size, one backward per direction;
  ShardedTrainingJobKvsAll            TrainingJobKvsAll._process_subbatch (kge/job/train_KvsAll.py:216-294): sp_ and _po
                                      queries with multi-hot labels from the KvsAll index (a label CSR of GLOBAL entity
                                      ids, the same on every rank: each shard's kernel takes the labels it owns),
                                      train.loss kl (statistics merged as for 1vsAll) or bce (additive: no exchange);
  ShardedTrainingJobNegativeSampling  TrainingJobNegativeSampling._process_subbatch (train_negative_sampling.py:103-164)
                                      with BatchNegativeSample.score (kge/util/sampler.py:263-306): per slot the [n, 1 + K]
                                      block (positive, negatives) and the kl loss with label 0 (the reference default)
                                      or bce; a rank scores the negatives it owns, one all-reduce of [n, K] floats;

all inside TrainingJob's batch loop (kge/job/train.py:406-520: zero_grad -> subbatches -> optimizer.step) and with
TrainingJob.save_to's checkpoint layout (train.py:284-298, kge_model.py:106-108): `model` = (state_dict, meta) with ONE
`_entity_embedder._embeddings.weight` of shape [E, d] and one `_relation_embedder._embeddings.weight` whatever the number
of ranks that trained them, `optimizer_state_dict` in torch's state_dict layout (parameter 0 = entities, 1 = relations,
the per-row state gathered to [E, d]), `epoch`, `valid_trace`, `type`, and -- if the job was given a LibKGE Config --
`config`: such a checkpoint loads into an unsharded reference TrainingJob / KgeModel.create_from; without a Config it
carries everything else and `kge resume` cannot open it (the reference reads checkpoint["config"] first).

How the work is split (BASELINE configs[3] / [4]):
  * rank g owns entity rows [g*S, (g+1)*S) as a float32 master parameter and their optimizer state -- the optimizer
    (kge_amd.optim.Adagrad: one pass over parameter, gradient and accumulator; torch.optim.* by name otherwise) steps
    this rank's shard only: no E x d gradient or state ever crosses a link;
  * the relation table is replicated; its gradient comes out identical on every rank, so the replicas step in lock-step
    without an all-reduce of their own;
  * 1vsAll / KvsAll: scores, statistics and loss are ShardedEntityTable.ce_loss / kl_loss / bce_loss: the fused score +
    loss kernels per shard on the exchanged query rows, one all-gather of n floats (log-sum-exps: not for bce), one
    all-reduce of n floats, one all-reduce of [n, d + d_r] floats in the backward (query-row gradients); scoring tables
    are bf16 copies of the masters (`score_dtype`), re-cast after every step;
  * negative sampling: float32 tables (the gather-bound kernels' dtype); the shard is followed by slack rows that take the
    exchanged s / o rows of the batch, so that kge_score_neg and its backward run on local row ids.

Not overlapped: the row exchange of batch k + 1 with the scoring of batch k.
"""
from typing import Optional

import torch
import torch.distributed as dist

from .sharded import ShardedEntityTable

ENT_KEY = "_entity_embedder._embeddings.weight"
REL_KEY = "_relation_embedder._embeddings.weight"


class _ShardedJob:
    """What the three jobs share: the sharded parameters, the optimizer on this rank's rows, epochs, checkpoints."""

    TYPE = "sharded"

    def __init__(self, scorer: str, num_entities: int, num_relations: int, dim: int, *, rel_dim: Optional[int] = None,
                 state_dict: Optional[dict] = None, init_std: float = 0.1, seed: int = 0, lr: float = 0.1,
                 optimizer: str = "Adagrad", optimizer_args: Optional[dict] = None, score_dtype=torch.bfloat16,
                 device=None, group=None, backend=None, l_norm: float = 1.0, slack_rows: int = 0, config=None,
                 alias_state: bool = False):
        """`state_dict`: full tables under the reference's parameter names (every rank passes the same ones and keeps
        its rows), else normal_(0, init_std) drawn from `seed` for the FULL table on every rank (then sliced): the
        initial model does not depend on the number of ranks.  `config`: a LibKGE Config to carry in the checkpoints.
        `alias_state` (the LibKGE plugin's sharded jobs): the masters are VIEWS of the given float32 tables on this
        device -- the relation master is the caller's relation table, the entity master its rows [lo, hi) (except with
        slack rows, which need a buffer of their own) -- so the caller's model sees every update of what this rank
        owns and no second copy is held; without it the masters are private copies."""
        self.scorer, self.E, self.R, self.d = scorer, int(num_entities), int(num_relations), int(dim)
        self.dr = int(rel_dim) if rel_dim is not None else self.d
        self.group = group
        self.config = config
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        ShardedEntityTable.check_partition(self.E, self.world)
        self.lo, self.hi = ShardedEntityTable.partition(self.E, self.world, self.rank)
        if state_dict is not None:
            ent_full, rel_full = state_dict[ENT_KEY], state_dict[REL_KEY]
        else:
            g = torch.Generator().manual_seed(seed)
            ent_full = torch.empty(self.E, self.d).normal_(0.0, init_std, generator=g)
            rel_full = torch.empty(self.R, self.dr).normal_(0.0, init_std, generator=g)
        if tuple(ent_full.shape) != (self.E, self.d) or tuple(rel_full.shape) != (self.R, self.dr):
            raise ValueError("kge_amd: state_dict does not match the model's shape")
        mine = ent_full[self.lo:self.hi].to(torch.float32).to(self.device).contiguous()
        rel_mine = rel_full.to(torch.float32).to(self.device).contiguous()
        if alias_state:
            if rel_mine.data_ptr() != rel_full.data_ptr() or (slack_rows <= 0 and self.hi > self.lo
                                                             and mine.data_ptr() != ent_full[self.lo:self.hi].data_ptr()):
                raise ValueError("kge_amd: alias_state needs contiguous float32 tables on the job's device")
        else:  # (`.to` / `.contiguous` return their argument when there is nothing to convert: copy explicitly)
            if mine.numel() and mine.data_ptr() == ent_full[self.lo:self.hi].data_ptr():
                mine = mine.clone()
            if rel_mine.data_ptr() == rel_full.data_ptr():
                rel_mine = rel_mine.clone()
        self.ent_ext = None
        if slack_rows > 0:  # negative sampling: the master IS the head of a larger buffer (ShardedEntityTable.with_slack)
            self.ent_ext, mine = ShardedEntityTable.with_slack(mine, slack_rows)
        self.ent_master = torch.nn.Parameter(mine)
        self.rel_master = torch.nn.Parameter(rel_mine)
        self.score_dtype = score_dtype
        if score_dtype == torch.float32:  # the masters are the scoring tables
            self.table = ShardedEntityTable(scorer, self.ent_master.detach(), self.rel_master.detach(), self.E,
                                            l_norm=l_norm, group=group, backend=backend)
        else:
            self.table = ShardedEntityTable(scorer, self.ent_master.detach().to(score_dtype),
                                            self.rel_master.detach().to(score_dtype), self.E, l_norm=l_norm, group=group,
                                            backend=backend)
        self.table.ent_ext = self.ent_ext
        self.optimizer = self._make_optimizer(optimizer, lr, optimizer_args or {})
        self.epoch = 0
        self.trace = []
        self.valid_trace = []

    def _make_optimizer(self, name, lr, args):
        params = [self.ent_master, self.rel_master]
        if name in ("Adagrad", "HipAdagrad") and self.ent_master.is_cuda:
            from .optim import Adagrad
            return Adagrad(params, lr=lr, **args)
        if name == "HipAdagrad":
            name = "Adagrad"
        return getattr(torch.optim, name)(params, lr=lr, **args)

    def _after_step(self):
        if self.score_dtype != torch.float32:
            self.table.refresh_tables(self.ent_master.detach(), self.rel_master.detach())

    # ---- checkpoints: one [E, d] parameter, however many ranks ---------------------------------------------------
    def _gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """[E, cols] from every rank's [E_g, cols] (ragged last shard: padded to the shard size for the collective)."""
        if self.world == 1:
            return local.detach().clone()
        S = self.table.shard
        pad = torch.zeros(S, local.shape[1], dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local.detach()
        out = torch.empty(self.world * S, local.shape[1], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(-1), pad.view(-1), group=self.group)
        return out[:self.E].contiguous()

    def state_dict(self) -> dict:
        """The model as an unsharded LibKGE model holds it (collective: every rank calls it, every rank gets it)."""
        return {ENT_KEY: self._gather_rows(self.ent_master).cpu(), REL_KEY: self.rel_master.detach().cpu().clone()}

    def load_state_dict(self, sd: dict):
        with torch.no_grad():
            self.ent_master.copy_(sd[ENT_KEY][self.lo:self.hi].to(self.device))
            self.rel_master.copy_(sd[REL_KEY].to(self.device))
        self._after_step()

    def optimizer_state_dict(self) -> dict:
        """torch's optimizer.state_dict() of an UNSHARDED optimizer over [entities, relations]: parameter ids 0 / 1,
        every per-row state tensor of the entity parameter gathered to [E, ...] (collective)."""
        sd = self.optimizer.state_dict()
        state = {}
        # parameter ids in the optimizer's own order: [entities, relations] unless parameter groups (the LibKGE plugin's
        # `train.optimizer.<group>`, sharded_job._ShardState) put them another way round
        for pid, which in enumerate(getattr(self, "_param_order", ("ent", "rel"))):
            prm = self.ent_master if which == "ent" else self.rel_master
            conv = {}
            for k, v in sd["state"].get(pid, {}).items():
                if torch.is_tensor(v) and v.shape == prm.shape:
                    conv[k] = (self._gather_rows(v) if which == "ent" else v.detach().clone()).cpu()
                else:
                    conv[k] = v.detach().cpu().clone() if torch.is_tensor(v) else v
            if conv:
                state[pid] = conv
        return {"state": state, "param_groups": [dict(g) for g in sd["param_groups"]]}

    def load_optimizer_state_dict(self, sd: dict):
        """The inverse: an unsharded optimizer state (this job's, or a reference TrainingJob's) onto this rank's rows."""
        local = {"state": {}, "param_groups": [dict(g) for g in sd["param_groups"]]}
        for pid, which in enumerate(getattr(self, "_param_order", ("ent", "rel"))):
            conv = {}
            for k, v in sd["state"].get(pid, {}).items():
                if torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == (self.E if which == "ent" else self.R):
                    conv[k] = (v[self.lo:self.hi] if which == "ent" else v).to(self.device).clone()
                else:
                    conv[k] = v.clone() if torch.is_tensor(v) else v
            if conv:
                local["state"][pid] = conv
        self.optimizer.load_state_dict(local)

    def checkpoint(self) -> dict:
        """TrainingJob.save_to's dictionary (kge/job/train.py:284-298).  Collective; identical on every rank."""
        ck = {"type": "train", "epoch": self.epoch, "valid_trace": list(self.valid_trace),
              "model": (self.state_dict(), {"sharded_train": {"type": self.TYPE, "scorer": self.scorer,
                                                              "world_size": self.world, "trace": list(self.trace)}}),
              "optimizer_state_dict": self.optimizer_state_dict(), "lr_scheduler_state_dict": {}, "job_id": None}
        if self.config is not None:
            ck["config"] = self.config
        return ck

    def save_checkpoint(self, path: str):
        ck = self.checkpoint()
        if self.rank == 0:
            torch.save(ck, path)
        if self.world > 1:
            dist.barrier(group=self.group)

    def load_checkpoint(self, ck):
        if isinstance(ck, str):
            ck = torch.load(ck, map_location="cpu", weights_only=False)
        self.load_state_dict(ck["model"][0])
        if "optimizer_state_dict" in ck and ck["optimizer_state_dict"].get("state"):
            self.load_optimizer_state_dict(ck["optimizer_state_dict"])
        self.epoch = int(ck.get("epoch", 0))
        self.valid_trace = list(ck.get("valid_trace", []))
        meta = ck["model"][1] if len(ck["model"]) > 1 and isinstance(ck["model"][1], dict) else {}
        self.trace = list(meta.get("sharded_train", {}).get("trace", []))

    def _epoch_entry(self, nb, N, sum_loss):
        entry = {"type": self.TYPE, "scope": "epoch", "epoch": self.epoch, "batches": nb, "size": N,
                 "avg_loss": float(sum_loss) / max(N, 1), "world_size": self.world}
        self.trace.append(entry)
        return entry


class ShardedTrainingJob1vsAll(_ShardedJob):
    TYPE = "1vsAll_sharded"

    # ---- one batch (train.py:406-520 with train_1vsAll.py:48-82 as the only subbatch) ------------------------------
    def step(self, triples: torch.Tensor) -> torch.Tensor:
        """One optimizer step on a batch of (s, p, o) triples -- the SAME batch on every rank.  Returns the batch's
        avg_loss (a 0-d tensor on the device: no host wait here)."""
        triples = triples.to(self.device)
        n = triples.shape[0]
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        self.optimizer.zero_grad(set_to_none=True)
        total = torch.zeros((), device=self.device)
        for direction, ids, labels in (("sp", s, o), ("po", o, s)):
            rows = self.table.ce_loss(direction, ids, p, labels, self.ent_master, self.rel_master)
            loss = rows.sum() / n
            total = total + loss.detach()
            loss.backward()
        self.optimizer.step()
        self._after_step()
        return total

    def run_epoch(self, triples: torch.Tensor, batch_size: int, shuffle_seed: Optional[int] = None) -> dict:
        """All batches of `triples` ([N, 3], identical on every rank) in the order of a permutation drawn from
        `shuffle_seed` + epoch (None: as given); the entry appended to `trace` carries the reference's keys."""
        N = triples.shape[0]
        order = torch.arange(N)
        if shuffle_seed is not None:
            order = torch.randperm(N, generator=torch.Generator().manual_seed(shuffle_seed + self.epoch))
        self.epoch += 1
        sum_loss = torch.zeros((), device=self.device)
        nb = 0
        for b0 in range(0, N, batch_size):
            batch = triples[order[b0:b0 + batch_size]]
            sum_loss += self.step(batch) * batch.shape[0]
            nb += 1
        return self._epoch_entry(nb, N, sum_loss)


class ShardedTrainingJobKvsAll(_ShardedJob):
    """step(queries): `queries` = the batch's query groups as TrainingJobKvsAll builds them (train_KvsAll.py:216-260):
    a list of (direction, ids [n_q], p [n_q], label_rowptr [n_q + 1], label_col [nnz]) with direction "sp" (ids =
    subjects, labels = objects) or "po" (ids = objects, labels = subjects); label columns are GLOBAL entity ids (int64,
    unique per row).  loss: "kl" (KLDivWithSoftmaxKgeLoss on the normalised multi-hot labels) or "bce"
    (BCEWithLogitsKgeLoss; `loss_arg` = its offset).  The batch's loss = sum over all query rows / number of rows
    (train_KvsAll.py:288-294)."""
    TYPE = "KvsAll_sharded"

    def __init__(self, *a, loss: str = "kl", loss_arg: float = 0.0, **kw):
        super().__init__(*a, **kw)
        if loss not in ("kl", "bce"):
            raise ValueError("kge_amd: ShardedTrainingJobKvsAll: train.loss must be kl or bce")
        self.loss, self.loss_arg = loss, float(loss_arg)

    def step(self, queries) -> torch.Tensor:
        n_all = sum(int(q[1].numel()) for q in queries)
        self.optimizer.zero_grad(set_to_none=True)
        total = torch.zeros((), device=self.device)
        for direction, ids, p, rowptr, col in queries:
            ids, p = ids.to(self.device), p.to(self.device)
            rowptr, col = rowptr.to(self.device).long(), col.to(self.device).long()
            if self.loss == "kl":
                rows = self.table.kl_loss(direction, ids, p, rowptr, col, self.ent_master, self.rel_master)
            else:
                rows = self.table.bce_loss(direction, ids, p, rowptr, col, self.loss_arg, self.ent_master, self.rel_master)
            loss = rows.sum() / max(n_all, 1)
            total = total + loss.detach()
            loss.backward()
        self.optimizer.step()
        self._after_step()
        return total


class ShardedTrainingJobNegativeSampling(_ShardedJob):
    """step(triples, neg_s, neg_o): triples [n, 3]; neg_s / neg_o [n, K_s] / [n, K_o] GLOBAL entity ids drawn by the
    caller's sampler (the same on every rank; either may be None = num_samples 0 for that slot).  Per slot the
    [n, 1 + K] block (column 0 the positive) and `loss`: "kl" (the reference's default for negative sampling: cross
    entropy with label 0, loss.py:192-207) or "bce" (sum over the block); the slot's loss / n, one backward per slot
    (train_negative_sampling.py:120-163).  float32 tables; `n_max` = the largest batch (slack rows)."""
    TYPE = "negative_sampling_sharded"

    def __init__(self, *a, n_max: int = 1024, loss: str = "kl", loss_arg: float = 0.0, **kw):
        kw["score_dtype"] = torch.float32
        kw["slack_rows"] = int(n_max)
        super().__init__(*a, **kw)
        if loss not in ("kl", "bce"):
            raise ValueError("kge_amd: ShardedTrainingJobNegativeSampling: train.loss must be kl or bce")
        self.loss, self.loss_arg = loss, float(loss_arg)

    def step(self, triples: torch.Tensor, neg_s: Optional[torch.Tensor], neg_o: Optional[torch.Tensor]) -> torch.Tensor:
        triples = triples.to(self.device)
        n = triples.shape[0]
        s, p, o = triples[:, 0].contiguous(), triples[:, 1].contiguous(), triples[:, 2].contiguous()
        self.optimizer.zero_grad(set_to_none=True)
        total = torch.zeros((), device=self.device)
        for slot, neg in ((0, neg_s), (2, neg_o)):
            if neg is None or neg.numel() == 0:
                continue
            neg = neg.to(self.device).long().contiguous()
            pos, sc = self.table.neg_scores(s, p, o, slot, neg, self.ent_master, self.rel_master)
            block = torch.cat([pos.view(-1, 1), sc], dim=1)
            if self.loss == "kl":
                loss = torch.nn.functional.cross_entropy(block, torch.zeros(n, dtype=torch.long, device=self.device),
                                                         reduction="sum") / n
            else:
                labels = torch.zeros_like(block)
                labels[:, 0] = 1.0
                loss = torch.nn.functional.binary_cross_entropy_with_logits(block + self.loss_arg, labels,
                                                                            reduction="sum") / n
            total = total + loss.detach()
            loss.backward()
        self.optimizer.step()
        self._after_step()
        return total
