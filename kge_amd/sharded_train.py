"""1vsAll training over an entity-sharded model: one process per GPU, `torch.distributed` (RCCL over xGMI; the CPU
tests run it on "gloo" with a fake backend).

What it mirrors.  TrainingJob1vsAll._process_subbatch (kge/job/train_1vsAll.py:48-82: score_sp against all entities
with labels o, score_po with labels s, KL/CE loss with `reduction="sum"` divided by the batch size, one backward per
direction) inside TrainingJob's batch loop (kge/job/train.py:406-520: zero_grad -> subbatches -> optimizer.step), and
the checkpoint layout of TrainingJob.save / KgeModel.save (train.py:276-298, kge_model.py:431-447): ONE
`_entity_embedder._embeddings.weight` of shape [E, d] and one `_relation_embedder._embeddings.weight`, whatever the
number of ranks that trained them.

How the work is split (BASELINE configs[3] / [4]):
  * rank g owns entity rows [g*S, (g+1)*S) as a float32 master parameter and their optimizer state -- the optimizer
    (kge_amd.optim.Adagrad: one pass over parameter, gradient and accumulator; torch.optim.* by name otherwise) steps
    this rank's shard only: no E x d gradient or state ever crosses a link;
  * the relation table is replicated; its gradient comes out identical on every rank (ShardedEntityTable.ce_loss), so
    the replicas step in lock-step without an all-reduce;
  * scores, the softmax statistics and the loss are ShardedEntityTable.ce_loss's: the fused score + loss kernels per
    shard on the exchanged query rows, one all-gather of n floats (log-sum-exps), one all-reduce of n floats (label
    scores), one all-reduce of [n, d + d_r] floats in the backward (query-row gradients);
  * scoring tables are bf16 copies of the masters (`score_dtype`), re-cast after every step.

Not sharded here: KvsAll, BCE and negative sampling (hip_KvsAll / hip_negative_sampling run unsharded); the row
exchange of batch k + 1 is not overlapped with the scoring of batch k.
"""
import math
from typing import Optional

import torch
import torch.distributed as dist

from .sharded import ShardedEntityTable

ENT_KEY = "_entity_embedder._embeddings.weight"
REL_KEY = "_relation_embedder._embeddings.weight"


class ShardedTrainingJob1vsAll:
    def __init__(self, scorer: str, num_entities: int, num_relations: int, dim: int, *, rel_dim: Optional[int] = None,
                 state_dict: Optional[dict] = None, init_std: float = 0.1, seed: int = 0, lr: float = 0.1,
                 optimizer: str = "Adagrad", optimizer_args: Optional[dict] = None, score_dtype=torch.bfloat16,
                 device=None, group=None, backend=None, l_norm: float = 1.0):
        """`state_dict`: full tables under the reference's parameter names (every rank passes the same ones and keeps
        its rows), else normal_(0, init_std) drawn from `seed` for the FULL table on every rank (then sliced): the
        initial model does not depend on the number of ranks."""
        self.scorer, self.E, self.R, self.d = scorer, int(num_entities), int(num_relations), int(dim)
        self.dr = int(rel_dim) if rel_dim is not None else self.d
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.lo, self.hi = ShardedEntityTable.partition(self.E, self.world, self.rank)
        if state_dict is not None:
            ent_full, rel_full = state_dict[ENT_KEY], state_dict[REL_KEY]
        else:
            g = torch.Generator().manual_seed(seed)
            ent_full = torch.empty(self.E, self.d).normal_(0.0, init_std, generator=g)
            rel_full = torch.empty(self.R, self.dr).normal_(0.0, init_std, generator=g)
        if tuple(ent_full.shape) != (self.E, self.d) or tuple(rel_full.shape) != (self.R, self.dr):
            raise ValueError("kge_amd: state_dict does not match the model's shape")
        self.ent_master = torch.nn.Parameter(ent_full[self.lo:self.hi].to(torch.float32).to(self.device).contiguous())
        self.rel_master = torch.nn.Parameter(rel_full.to(torch.float32).to(self.device).contiguous())
        self.table = ShardedEntityTable(scorer, self.ent_master.detach().to(score_dtype),
                                        self.rel_master.detach().to(score_dtype), self.E, l_norm=l_norm, group=group,
                                        backend=backend)
        self.optimizer = self._make_optimizer(optimizer, lr, optimizer_args or {})
        self.epoch = 0
        self.trace = []

    def _make_optimizer(self, name, lr, args):
        params = [self.ent_master, self.rel_master]
        if name in ("Adagrad", "HipAdagrad") and self.ent_master.is_cuda:
            from .optim import Adagrad
            return Adagrad(params, lr=lr, **args)
        if name == "HipAdagrad":
            name = "Adagrad"
        return getattr(torch.optim, name)(params, lr=lr, **args)

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
        self.table.refresh_tables(self.ent_master.detach(), self.rel_master.detach())
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
        entry = {"type": "1vsAll_sharded", "scope": "epoch", "epoch": self.epoch, "batches": nb, "size": N,
                 "avg_loss": float(sum_loss) / max(N, 1), "world_size": self.world}
        self.trace.append(entry)
        return entry

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
        self.table.refresh_tables(self.ent_master.detach(), self.rel_master.detach())

    def optimizer_state(self) -> dict:
        """Per-parameter optimizer state with the entity rows gathered to [E, ...] (Adagrad: `sum`, `step`)."""
        out = {}
        for key, prm in ((ENT_KEY, self.ent_master), (REL_KEY, self.rel_master)):
            st, conv = self.optimizer.state.get(prm, {}), {}
            for k, v in st.items():
                if torch.is_tensor(v) and v.shape == prm.shape:
                    conv[k] = (self._gather_rows(v) if prm is self.ent_master else v.detach().clone()).cpu()
                else:
                    conv[k] = v.detach().cpu().clone() if torch.is_tensor(v) else v
            out[key] = conv
        return out

    def load_optimizer_state(self, state: dict):
        for key, prm in ((ENT_KEY, self.ent_master), (REL_KEY, self.rel_master)):
            st = self.optimizer.state[prm]
            for k, v in state.get(key, {}).items():
                if torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == (self.E if prm is self.ent_master else self.R):
                    v = v[self.lo:self.hi] if prm is self.ent_master else v
                    st[k] = v.to(self.device).clone()
                else:
                    st[k] = v.clone() if torch.is_tensor(v) else v

    def checkpoint(self) -> dict:
        """What TrainingJob.save writes, as far as this job has it: `model` = [config, state_dict] (kge_model.py:431-447),
        `epoch`, `valid_trace`-like `trace`, the gathered optimizer state.  Collective; identical on every rank."""
        return {"type": "train", "epoch": self.epoch, "trace": list(self.trace),
                "model": [{"scorer": self.scorer, "num_entities": self.E, "num_relations": self.R, "dim": self.d,
                           "rel_dim": self.dr}, self.state_dict()],
                "optimizer_state": self.optimizer_state()}

    def save_checkpoint(self, path: str):
        ck = self.checkpoint()
        if self.rank == 0:
            torch.save(ck, path)
        if self.world > 1:
            dist.barrier(group=self.group)

    def load_checkpoint(self, ck):
        if isinstance(ck, str):
            ck = torch.load(ck, map_location="cpu", weights_only=False)
        self.load_state_dict(ck["model"][1])
        if not self.optimizer.state:  # state is created lazily: a zero-gradient step materialises it
            for prm in (self.ent_master, self.rel_master):
                prm.grad = torch.zeros_like(prm)
            lrs = [g["lr"] for g in self.optimizer.param_groups]
            for g in self.optimizer.param_groups:
                g["lr"] = 0.0
            self.optimizer.step()
            for g, lr in zip(self.optimizer.param_groups, lrs):
                g["lr"] = lr
            self.optimizer.zero_grad(set_to_none=True)
        self.load_optimizer_state(ck.get("optimizer_state", {}))
        self.epoch = int(ck.get("epoch", 0))
        self.trace = list(ck.get("trace", []))
