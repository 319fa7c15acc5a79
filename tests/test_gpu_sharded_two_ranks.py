"""ShardedEntityTable with the REAL engine in TWO processes (both on cuda:0; a 1-GPU box cannot host two RCCL
ranks, so the process group is gloo and, in these workers only, the three collectives the class uses are
staged through host memory).  What this adds over the other sharded tests: the gloo CPU tests run the
choreography on a fake backend, the one-rank GPU tests run the kernels with identity collectives -- here the
shard arithmetic (ragged shards, owner / pick indices of the exchange, cross-shard log-sum-exp, the all-reduce of
the query-row gradients, rank-count sums, top-k merge) meets the HIP kernels across two real ranks.
Checked against the unsharded engine on the same tables: rank / tie counts and top-k exactly, the sharded 1vsAll
loss to float32 rounding of the merged statistics, both table gradients."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"



def _quiet_teardown():
    """Every rank reaches this point before any rank closes its sockets: a rank that tears its gloo context down while the
    other is still inside its last collective aborts the straggler ("terminate called without an active exception": one
    run in six on this box before the barrier was here)."""
    try:
        if dist.is_initialized():
            import datetime
            dist.monitored_barrier(timeout=datetime.timedelta(seconds=30))  # (bounded: the other rank may have died)
    except Exception:
        pass
    if dist.is_initialized():
        dist.destroy_process_group()

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _host_staged_collectives():
    """gloo moves host tensors: stage the device tensors of the three collectives through the host."""
    agt, ar, ag = dist.all_gather_into_tensor, dist.all_reduce, dist.all_gather

    def all_gather_into_tensor(out, inp, group=None):
        o = torch.empty(out.shape, dtype=out.dtype if out.dtype != torch.bfloat16 else torch.float32)
        agt(o, inp.detach().to("cpu", dtype=o.dtype).contiguous(), group=group)
        out.copy_(o.to(out.dtype))

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        h = t.detach().cpu()
        ar(h, op=op, group=group)
        t.copy_(h)

    def all_gather(outs, t, group=None):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        ag(hs, t.detach().cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)

    dist.all_gather_into_tensor, dist.all_reduce, dist.all_gather = all_gather_into_tensor, all_reduce, all_gather


def _worker(rank, world, port, model, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _host_staged_collectives()
        from kge_amd.sharded import ShardedEntityTable
        E, R, d, n = 1000 + 37, 5, 256, 96   # odd E: ragged shards (519 + 518 rows)
        g = torch.Generator().manual_seed(3)
        ent = (torch.randn(E, d, generator=g) * 0.3)
        rel = (torch.randn(R, d, generator=g) * 0.3)
        tri = torch.stack([torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g),
                           torch.randint(E, (n,), generator=g)], 1).to(DEV)
        w = (torch.rand(2 * n, generator=g) + 0.5).to(DEV)
        lo, hi = ShardedEntityTable.partition(E, world, rank)
        ent_m = ent[lo:hi].clone().to(DEV).requires_grad_(True)
        rel_m = rel.clone().to(DEV).requires_grad_(True)
        sh = ShardedEntityTable(model, ent_m.detach().bfloat16(), rel_m.detach().bfloat16(), E)
        assert sh.world == 2 and sh.collectives
        s, p, o = tri[:, 0], tri[:, 1], tri[:, 2]
        ranks = [x.cpu().numpy() for x in sh.rank_batch(tri, None)]
        # raw + filtered counts out of the scoring kernel (kge_score_rank_emb_sp_po on each shard, one counter
        # all-reduce) against score slabs + rank_counts_multi, and against rank_batch's raw counts
        cnt = torch.randint(0, 6, (n,), generator=g)
        end = torch.cumsum(cnt, 0)
        beg = end - cnt
        def values():  # distinct ids within a row's range (the contract of the filter index)
            v = [torch.randperm(E, generator=g)[:int(c)] for c in cnt] + [torch.zeros(1, dtype=torch.int64)]
            return torch.cat(v).to(DEV)
        fo, fs = [(beg.to(DEV), end.to(DEV), values())], [(beg.to(DEV), end.to(DEV), values())]
        fused = sh._rank_batch_fused(s, p, o, fo, fs, 1e-5, 1e-4)
        assert fused is not None, "the sharded path did not count inside the scoring kernel"
        assert torch.equal(sh.rank_batch_multi(tri, fo, fs), fused)
        sh.fused_rank = False
        assert torch.equal(sh.rank_batch_multi(tri, fo, fs), fused)
        sh.fused_rank = True
        for a, b in zip((fused[1, 0, 0], fused[1, 1, 0], fused[0, 0, 0], fused[0, 1, 0]), ranks):
            assert np.array_equal(a.cpu().numpy(), b)
        # the whole evaluation loop over the sharded table (replicated filter index, shard-local counts inside the
        # scoring kernel, one counter all-reduce per batch): every rank ends with the same ranks and metrics
        from kge_amd.eval import EntityRankingEvaluator
        from kge_amd.synthetic import make_splits
        splits = make_splits(E, R, 3000, 250, 150, seed=11)
        ev_m, ev_r = EntityRankingEvaluator(sh, splits, E, R, eval_split="valid", batch_size=96).run(return_ranks=True)
        assert sh.fused_rank
        tv, ti = sh.topk(sh.score_sp(s, p), 7)
        blk_sp, blk_po = (x.clone() for x in sh.score_sp_po_blocks(s, p, o))  # the step bench.py --gpus N times
        small, sh.BIG_SLAB_BYTES = sh.BIG_SLAB_BYTES, 0                      # ... and its big-slab form (padded pitch,
        big_sp, big_po = (x.clone() for x in sh.score_sp_po_blocks(s, p, o))  #     one launch per direction)
        sh.BIG_SLAB_BYTES = small
        assert torch.equal(big_sp, blk_sp) and torch.equal(big_po, blk_po)
        # two batches in flight (ShardedScoreLanes: batch k's exchange + scoring on HIP stream k % 2, the
        # collectives of the lanes in one order on both ranks): five steps, results read after join()
        from kge_amd.sharded import ShardedScoreLanes
        lanes = ShardedScoreLanes(sh, 2)
        tri2 = torch.flip(tri, dims=[0]).contiguous()
        want2 = tuple(x.clone() for x in sh.score_sp_po_blocks(tri2[:, 0], tri2[:, 1], tri2[:, 2]))
        lanes.fork()
        res = [lanes.score_sp_po_blocks(*( (t[:, 0], t[:, 1], t[:, 2]) )) for t in (tri, tri2, tri, tri2, tri)]
        lanes.join()
        for k, (a_sp, a_po) in enumerate(res):
            w_sp, w_po = (blk_sp, blk_po) if k % 2 == 0 else want2
            assert torch.equal(a_sp, w_sp) and torch.equal(a_po, w_po), ("lanes", k)
        loss = torch.cat([sh.ce_loss("sp", s, p, o, ent_m, rel_m), sh.ce_loss("po", o, p, s, ent_m, rel_m)])
        (loss * w).sum().backward()
        torch.cuda.synchronize()
        q.put((rank, lo, hi, ranks, tv.cpu().numpy(), ti.cpu().numpy(), loss.detach().cpu().numpy(),
               ent_m.grad.cpu().numpy(), rel_m.grad.cpu().numpy(), ent.numpy(), rel.numpy(), tri.cpu().numpy(),
               w.cpu().numpy(), blk_sp.cpu().numpy(), blk_po.cpu().numpy(), ev_m, ev_r))
    finally:
        _quiet_teardown()


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_two_ranks_on_the_real_kernels(model):
    from kge_amd import engine as eng
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs, t0 = [], time.time()
    while len(outs) < world and time.time() - t0 < 240:  # a crashed worker must fail the test, not hang it
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(60)
        if pr.is_alive():
            pr.kill()
    assert len(outs) == world, [pr.exitcode for pr in procs]
    outs.sort(key=lambda x: x[0])
    ent, rel, tri, w = (torch.from_numpy(outs[0][k]) for k in (9, 10, 11, 12))
    E, n = ent.shape[0], tri.shape[0]
    e16, r16 = ent.bfloat16().to(DEV), rel.bfloat16().to(DEV)
    T = eng.Tables(model, e16, r16)
    s, p, o = (tri[:, k].to(DEV) for k in range(3))
    both = eng.score_sp_po(T, s, p, o)
    ar = torch.arange(n, device=DEV)
    r_o, t_o = eng.rank_counts(both[:, :E].contiguous(), both[ar, o], None, None, 0, o)
    r_s, t_s = eng.rank_counts(both[:, E:].contiguous(), both[ar, E + s], None, None, 0, s)
    want_ranks = [x.cpu().numpy() for x in (r_s, t_s, r_o, t_o)]
    tv, ti = torch.topk(both[:, :E], 7, dim=1)
    # the unsharded fused loss and its gradients on the same bf16 tables
    l_sp, z_sp = eng.ce_fwd(T, "sp", s, p, o)
    l_po, z_po = eng.ce_fwd(T, "po", o, p, s)
    ge, gr = torch.zeros(E, ent.shape[1], device=DEV), torch.zeros(rel.shape[0], rel.shape[1], device=DEV)
    wd = w.to(DEV)
    for direction, a, lab, lse, ww in (("sp", s, o, z_sp, wd[:n]), ("po", o, s, z_po, wd[n:])):
        g_a, g_p, g_t = eng.ce_bwd(T, direction, a, p, lab, lse, g_rows=ww.contiguous())
        ge += g_t
        ge.index_add_(0, a, g_a)
        gr.index_add_(0, p, g_p)
    want_loss = torch.cat([l_sp, l_po]).cpu().numpy()
    # the unsharded evaluator on the same bf16 tables and splits
    from kge_amd.eval import EntityRankingEvaluator
    from kge_amd.synthetic import make_splits
    splits = make_splits(E, rel.shape[0], 3000, 250, 150, seed=11)
    want_m, want_r = EntityRankingEvaluator(T, splits, E, rel.shape[0], eval_split="valid", batch_size=96).run(return_ranks=True)
    for rank, lo, hi, ranks, rtv, rti, loss, g_ent, g_rel, *rest in outs:
        assert rest[6] == want_m, (model, rank, "sharded evaluation metrics")
        for k in want_r:
            assert np.array_equal(rest[7][k], want_r[k]), (model, rank, k)
        # the shard's score blocks are the unsharded matrix's columns, bit for bit
        assert np.array_equal(rest[4], both[:, lo:hi].cpu().numpy()), (model, rank, "sp block")
        assert np.array_equal(rest[5], both[:, E + lo:E + hi].cpu().numpy()), (model, rank, "po block")
        for a_, b_ in zip(ranks, want_ranks):
            assert np.array_equal(a_, b_), (model, rank)
        assert np.array_equal(rtv, tv.cpu().numpy()), (model, rank)
        assert np.array_equal(np.take_along_axis(both[:, :E].cpu().numpy(), rti, 1), rtv), (model, rank)
        assert np.allclose(loss, want_loss, rtol=2e-5, atol=2e-5), (model, rank)
        we = ge[lo:hi].cpu().numpy()
        assert np.linalg.norm(g_ent - we) <= 2e-3 * np.linalg.norm(we), (model, rank)
        wr = gr.cpu().numpy()
        assert np.linalg.norm(g_rel - wr) <= 2e-3 * np.linalg.norm(wr), (model, rank)
