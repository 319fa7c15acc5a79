"""kge_shard_gather / kge_shard_pick (the two row moves of ShardedEntityTable.exchange_rows, include/kge_amd.h) against
the torch-op form they replace: local = clamp(id - lo, 0, rows - 1), pick = owner * (k n) + position.  Byte moves:
bit-exact.  Four emulated ranks in one process (the all-gather = a cat of the ranks' blocks)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("itype", [torch.int32, torch.int64])
@pytest.mark.parametrize("k", [1, 2])
def test_exchange_row_moves(dtype, itype, k):
    from kge_amd import engine as eng
    E, R, d, dr, n, world = 1037, 9, 128, 64, 75, 4       # ragged last shard; relation rows of another length
    g = torch.Generator().manual_seed(7)
    ent = torch.randn(E, d, generator=g).to(dtype).to(DEV)
    rel = torch.randn(R, dr, generator=g).to(dtype).to(DEV)
    trip = torch.stack([torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g),
                        torch.randint(E, (n,), generator=g)], 1).to(itype).to(DEV)   # strided id columns
    ids = [trip[:, 0], trip[:, 2]][:k]
    shard = (E + world - 1) // world
    blocks = []
    for r in range(world):
        lo, hi = min(r * shard, E), min((r + 1) * shard, E)
        T = eng.Tables("rotate", ent[lo:hi].contiguous(), rel)   # "rotate": dim_rel = dim / 2
        send = torch.full((k * n, d), 7.0, dtype=dtype, device=DEV)
        rel_rows = torch.empty(n, dr, dtype=dtype, device=DEV)
        eng.shard_gather(T, lo, ids, trip[:, 1], send, rel_rows)
        gid = torch.cat([x.long() for x in ids])
        assert torch.equal(send, ent[lo:hi][(gid - lo).clamp(0, hi - lo - 1)])
        assert torch.equal(rel_rows, rel[trip[:, 1].long()])
        blocks.append(send)
    gath = torch.cat(blocks)
    rows = torch.empty(k * n, d, dtype=dtype, device=DEV)
    eng.shard_pick(gath, shard, world, ids, rows)
    assert torch.equal(rows, ent[torch.cat([x.long() for x in ids])])
    # without relation rows; n = 0
    send = torch.empty(k * n, d, dtype=dtype, device=DEV)
    T0 = eng.Tables("rotate", ent[:shard].contiguous(), rel)
    eng.shard_gather(T0, 0, ids, None, send, None)
    assert torch.equal(send, ent[:shard][torch.cat([x.long() for x in ids]).clamp(0, shard - 1)])
    eng.shard_gather(T0, 0, [x[:0] for x in ids], None, send[:0], None)


def test_lanes_replay_a_captured_step():
    """ShardedScoreLanes on one rank (no collectives): each lane's step is captured into a hipGraph on first use and
    replayed with the batch's ids copied into static index vectors; six steps over two alternating batches (strided
    id columns) = the direct calls, bit for bit; the call-by-call lanes (graph=False) the same."""
    from kge_amd import engine as eng
    from kge_amd.sharded import ShardedEntityTable, ShardedScoreLanes
    E, R, d, n = 14541, 11, 512, 128
    g = torch.Generator().manual_seed(3)
    ent = (torch.randn(E, d, generator=g) * 0.3).bfloat16().to(DEV)
    rel = (torch.randn(R, d, generator=g) * 0.3).bfloat16().to(DEV)
    sh = ShardedEntityTable("complex", ent, rel, E)
    trip = [torch.stack([torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g),
                         torch.randint(E, (n,), generator=g)], 1).to(DEV) for _ in range(2)]
    want = [tuple(x.clone() for x in sh.score_sp_po_blocks(t[:, 0], t[:, 1], t[:, 2])) for t in trip]
    for graph in (True, False):
        lanes = ShardedScoreLanes(sh, 2, graph=graph)
        lanes.fork()
        for k in range(6):
            t = trip[(k // 2) & 1]          # lane k % 2 sees batch 0, 1, 0: its static ids change between replays
            a_sp, a_po = lanes.score_sp_po_blocks(t[:, 0], t[:, 1], t[:, 2])
            lanes.join()
            w_sp, w_po = want[(k // 2) & 1]
            assert torch.equal(a_sp, w_sp) and torch.equal(a_po, w_po), (graph, k)
            lanes.fork()
        assert lanes.graph_replays == (4 if graph else 0), getattr(lanes, "graph_error", None)
