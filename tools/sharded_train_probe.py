#!/usr/bin/env python3
"""One entity-sharded 1vsAll training step (ShardedEntityTable.ce_loss for both directions + backward + one-pass Adagrad
with bf16 copies + table refresh) on ONE rank's shard of the Wikidata5M shape: 574,311 entity rows (E / 8), R = 822,
d = 256, batch 512 -- what each of 8 ranks does per step, without the collectives (one process, no process group)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import optim as ko
from kge_amd.sharded import ShardedEntityTable

dev = torch.device("cuda", 0)
Eg, R, d, n = int(os.environ.get("ROWS", "574311")), 822, 256, 512
g = torch.Generator(device=dev).manual_seed(0)
ent_m = torch.empty(Eg, d, device=dev).normal_(0, 0.1, generator=g).requires_grad_(True)
rel_m = torch.empty(R, d, device=dev).normal_(0, 0.1, generator=g).requires_grad_(True)
s, o = (torch.randint(Eg, (n,), device=dev, generator=g) for _ in range(2))
p = torch.randint(R, (n,), device=dev, generator=g)
opt = ko.Adagrad([ent_m, rel_m], lr=0.1, bf16_copies=True)
sh = ShardedEntityTable("complex", ent_m.detach().bfloat16(), rel_m.detach().bfloat16(), Eg)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.cat([sh.ce_loss("sp", s, p, o, ent_m, rel_m), sh.ce_loss("po", o, p, s, ent_m, rel_m)]).sum() / n
    loss.backward()
    opt.step()
    ce, cr = ko.bf16_copy_of(ent_m), ko.bf16_copy_of(rel_m)  # written by the optimizer kernel in the same pass
    sh.refresh_tables(ent_m if ce is None else ce, rel_m if cr is None else cr)
    return loss


for _ in range(3):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    l = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
print(f"rows per rank {Eg}, d {d}, batch {n}: {el * 1e3:.2f} ms per training step (both directions, loss {float(l):.4f}); "
      f"{2 * n * Eg / el / 1e9:.1f} G scored triples/s per rank, forward + backward + optimizer")
