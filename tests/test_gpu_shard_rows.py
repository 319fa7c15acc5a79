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
