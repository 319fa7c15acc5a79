#!/usr/bin/env python3
"""Kernel-level breakdown of a negative-sampling training step as HipTrainingJobNegativeSampling issues it
(train_negative_sampling.py:103-164: per slot positives via score_spo, negatives via score_neg, KL loss, backward;
then Adagrad) at E = 14,541, d = 256, n = 512, K = 100 per slot: run under `rocprofv3 --kernel-trace --stats`."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import model as km, optim as ko

dev = torch.device("cuda", 0)
E, R, d, n, K = int(os.environ.get("E", "14541")), int(os.environ.get("R", "237")), int(os.environ.get("D", "256")), 512, int(os.environ.get("K", "100"))
name = os.environ.get("MODEL", "rotate")
g = torch.Generator().manual_seed(0)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
o = torch.randint(E, (n,), generator=g).to(dev)
torch.manual_seed(0)
m = km.create(name, E, R, d, device=dev)
opt = ko.Adagrad(m.parameters(), lr=0.1)
labels = torch.zeros(n, K + 1, device=dev); labels[:, 0] = 1
for it in range(int(os.environ.get("STEPS", "30"))):
    opt.zero_grad(set_to_none=True)
    for slot in (0, 2):
        neg = torch.randint(E, (n, K), generator=g).to(dev)
        scores = torch.empty(n, K + 1, device=dev)
        scores[:, 0] = m.score_spo(s, p, o)
        scores[:, 1:] = m.score_neg(s, p, o, slot, neg)
        loss = torch.nn.functional.kl_div(torch.log_softmax(scores, 1), torch.nn.functional.normalize(labels, p=1, dim=1),
                                          reduction="sum") / n
        loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done")
