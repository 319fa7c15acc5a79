"""Stress of the asm-heavy scoring kernels with several launches in flight (the dropped-store bug of
pairs_bf16_v6 / v7 -- an asm output register re-used while an LDS read was still in flight -- showed only with
several launches on the chip at once and was found by a tool, not by a test).  >= 200 launches, 2-3 HIP streams, EVERY
element of every batch compared with the same batch scored alone; score buffers are poisoned with NaN between uses so
that a store that never happened is seen as well as a wrong one.

Reference semantics: KgeModel.score_sp / score_sp_po (kge/model/kge_model.py:682-702, 749-789)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E, R, D, N = 14541, 237, 512, 512


def _tables(eng, flags=0):
    g = torch.Generator().manual_seed(6)
    ent = (torch.randn(E, D, generator=g) * 0.3).bfloat16().to(DEV)
    rel = (torch.randn(R, D, generator=g) * 0.3).bfloat16().to(DEV)
    return eng.Tables("complex", ent, rel, flags=flags)


def _bad(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())


@pytest.mark.parametrize("combine,lanes,kernel", [("sp_", 3, "v7"), ("sp_po", 2, "v7"), ("sp_po", 2, "v6-split"),
                                                 ("sp_", 2, "v8")])
def test_pipelined_launches_in_flight(combine, lanes, kernel, monkeypatch, kge_switch):
    """ScorePipeline(streams = L): 7 batches x 30 rounds = 210 launches per case, L in flight; kernel "v8" forces the
    persistent kernel onto single batches (KGE_V8=1)."""
    from kge_amd import engine as eng
    if kernel == "v8":
        kge_switch.set("V8", "1")
    fl = eng.FLAG_SPLIT_QUERY if kernel == "v6-split" else None
    T = _tables(eng, fl or 0)
    nb = 7
    trip = []
    for k in range(nb):
        q = torch.Generator().manual_seed(20 + k)
        trip.append(torch.stack([torch.randint(hi, (N,), generator=q) for hi in (E, R, E)], 1).to(DEV))
    direct = (lambda t: eng.score_sp(T, t[:, 0], t[:, 1])) if combine == "sp_" else \
        (lambda t: eng.score_sp_po(T, t[:, 0], t[:, 1], t[:, 2]))
    want = [direct(t) for t in trip]
    torch.cuda.synchronize()
    bad = 0
    for it in range(30):
        pipe = eng.ScorePipeline(T, combine, N, flags=fl, streams=lanes)
        outs = [torch.full_like(want[0], float("nan")) for _ in range(lanes)]
        pipe.start(trip[:lanes])
        got = []
        for k in range(nb):
            pipe.step(next_batch=trip[k + lanes] if k + lanes < nb else None, out=outs[k % lanes])
            if k % lanes == lanes - 1 or k == nb - 1:
                pipe.join()
                for j in range(k - k % lanes, k + 1):
                    got.append(outs[j % lanes].clone())
                    outs[j % lanes].fill_(float("nan"))
                pipe.fork()
        torch.cuda.synchronize()
        bad += sum(_bad(got[k], want[k]) for k in range(nb))
    assert bad == 0, f"{bad} elements differ over 210 launches ({combine}, {lanes} streams, {kernel})"


@pytest.mark.parametrize("split", [0, 1])
def test_group_launches_on_two_streams(split):
    """kge_score_queries_multi on two streams at once: 2 x 40 group launches of 3 batches each (240 batches), each group
    building the next group's queries inside the launch; every block compared with its batch scored alone."""
    from kge_amd import engine as eng
    fl = eng.FLAG_SPLIT_QUERY if split else None
    T = _tables(eng, fl or 0)
    L, n = 3, 256
    P = eng.score_pitch(E)
    groups, want = [], []
    for k in range(4):
        q = torch.Generator().manual_seed(40 + k)
        tri = torch.stack([torch.randint(hi, (n * L,), generator=q) for hi in (E, R, E)], 1).to(DEV)
        groups.append(tri)
        want.append([eng.score_queries(T, eng.build_queries(T, "sp_po", tri[l * n:(l + 1) * n, 0], tri[l * n:(l + 1) * n, 1],
                                                            tri[l * n:(l + 1) * n, 2], flags=fl)) for l in range(L)])
    streams = [torch.cuda.Stream(DEV) for _ in range(2)]
    lanes = []
    for si, st in enumerate(streams):
        qs = [eng.QueriesGroup(T, "sp_po", n, L, flags=fl) for _ in range(2)]
        buf = torch.full((L, n, 2 * P), float("nan"), device=DEV)
        lanes.append({"qs": qs, "buf": buf, "out": buf.view(L, n, 2, P)[:, :, :, :E], "cur": 0, "g": si})
    torch.cuda.synchronize()
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            eng.build_queries_group(T, "sp_po", groups[lanes[si]["g"]], n, L, out=lanes[si]["qs"][0])
    bad = 0
    for it in range(40):
        for si, st in enumerate(streams):
            ln = lanes[si]
            nxt = (ln["g"] + 2) % 4
            with torch.cuda.stream(st):
                eng.score_queries_group(T, ln["qs"][ln["cur"]], ln["out"], next_batch=groups[nxt],
                                        next_queries=ln["qs"][1 - ln["cur"]])
        torch.cuda.synchronize()
        for si in range(2):
            ln = lanes[si]
            for l in range(L):
                bad += _bad(ln["out"][l].reshape(n, 2 * E), want[ln["g"]][l])
            assert int((~torch.isnan(ln["buf"])).sum()) == L * n * 2 * E
            ln["buf"].fill_(float("nan"))
            ln["g"] = (ln["g"] + 2) % 4
            ln["cur"] = 1 - ln["cur"]
    assert bad == 0, f"{bad} elements differ over 80 group launches (split={split})"
