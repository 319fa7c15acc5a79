import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kge_amd import engine
dev = torch.device("cuda", 0)
E, R, d, n = 14541, 237, 512, 512
g = torch.Generator().manual_seed(0)
ent = torch.empty(E, d).normal_(0, 0.1, generator=g).to(dev)
rel = torch.empty(R, d).normal_(0, 0.1, generator=g).to(dev)
s = torch.randint(E, (n,), generator=g).to(dev); p = torch.randint(R, (n,), generator=g).to(dev)
T = engine.Tables("complex", ent, rel)
gout = torch.randn(n, E, device=dev)
scores = engine.score_sp(T, s, p)
q = torch.randn(n, d, device=dev)
for _ in range(20):
    engine.score_pairs_bwd(T, "sp", s, p, None, gout, scores)
    a = gout @ ent
    b = gout.t() @ q
torch.cuda.synchronize()
