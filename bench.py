#!/usr/bin/env python3
"""Benchmark of the KGE scoring hot path on MI355X (contract: see the task prompt / DESIGN.md 7).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): scored triples / s, 1vsAll ComplEx d=512.
Workload at N=1 = BASELINE configs[1]: FB15k-237 shape (E=14,541, R=237), ComplEx d=512,
bf16 tables, batch n=512.  One STEP = the two score blocks of a 1vsAll batch
(kge/job/train_1vsAll.py:64,75): score_sp(s,p) and score_po(p,o), each an [n, E] f32 score
matrix -> 2*n*E scored triples per step, computed as the reference's KgeModel.score_sp_po
(kge_model.py:749-789) does: both blocks from ONE two-sided launch (kge_score_sp_po; bit-identical
to the two separate calls, whose per-launch figures are reported next to it as
roofline.one_sided_launch).  Inputs (tables, index vectors) are resident in HBM before the
timed region.  Data is synthetic (no datasets/network here): N(0, 0.1)
tables (examples/toy-complex-train.yaml:18-22), uniform random queries.

N > 1 (weak scaling): the entity table is row-sharded, every rank owns an FB15k-237-sized
shard (global E = N * 14,541) and scores the same n queries against its shard; the query
rows live on their owner shards and are exchanged with ONE all-gather per step (RCCL), then
scored by the same two-sided launch on the gathered dense rows (kge_score_emb_sp_po).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

E_FB, R_FB, DIM, BATCH = 14541, 237, 512, 512
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-sided", action="store_true",
                    help="skip the reference region of one-sided launches (profiling runs: the kernel "
                         "trace then holds two-sided launches only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_inputs(rank, device, n):
    g = torch.Generator().manual_seed(0 + rank)
    ent = torch.empty(E_FB, DIM).normal_(0, 0.1, generator=g)
    rel = torch.empty(R_FB, DIM).normal_(0, 0.1, generator=torch.Generator().manual_seed(1234))
    q = torch.Generator().manual_seed(1)
    s = torch.randint(E_FB, (n,), generator=q)
    p = torch.randint(R_FB, (n,), generator=q)
    o = torch.randint(E_FB, (n,), generator=q)
    return (ent.to(torch.bfloat16).to(device), rel.to(torch.bfloat16).to(device),
            s.to(device), p.to(device), o.to(device))


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of
    this same command (profiles/pmc_latest.json: FETCH_SIZE x2 per the gfx950 correction in
    MI355X_MICROARCH.md + WRITE_SIZE).  bench.py itself cannot collect PMC counters."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        # only a summary of the same launch shape (two-sided score_sp_po launches) applies
        return d["hbm_bytes_per_launch"] if d.get("launch") == "score_sp_po" else None
    except Exception:
        return None


def algorithmic_bytes(n, m, d, elt=2, sides=1):
    """target rows once + query rows (s and r) + f32 scores out + indices, per
    scoring call; a two-sided launch (sides=2: score_sp and score_po blocks of the batch) reads
    the target rows once for both."""
    return m * d * elt + sides * (n * (d + d) * elt + n * m * 4 + 2 * n * 8)


def cpu_baseline(n, seconds):
    """Reference CPU path restated op-for-op in torch (oracle/torch_port.py, bit-identical to
    the live reference in the build container), fp32, all host cores, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port as tp

    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E_FB, DIM).normal_(0, 0.1, generator=g)
    rel = torch.empty(R_FB, DIM).normal_(0, 0.1, generator=g)
    s = torch.randint(E_FB, (n,), generator=g)
    p = torch.randint(R_FB, (n,), generator=g)
    o = torch.randint(E_FB, (n,), generator=g)
    cores = torch.get_num_threads()
    with torch.no_grad():
        tp.score_sp("complex", ent, rel, s, p)  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            tp.score_sp("complex", ent, rel, s, p)
            tp.score_po("complex", ent, rel, p, o)
            reps += 1
            el = time.perf_counter() - t0
            if el > seconds or reps >= 200:
                break
    return {"value": 2.0 * n * E_FB * reps / el, "unit": "scored triples/s", "cores": cores,
            "kind": "port",
            "sample": f"{reps} 1vsAll steps (score_sp+score_po, n={n}, E={E_FB}, d={DIM}, fp32) "
                      f"of oracle/torch_port.py (reference torch op sequence) in {el:.1f}s"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # KGE_BENCH_FORCE_DIST=1: exercise the sharded step (RCCL init + all-gather) with one rank
    dist = world > 1 or os.environ.get("KGE_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=device)

    from kge_amd import engine

    n = a.batch
    ent, rel, s, p, o = make_inputs(rank, device, n)
    T = engine.Tables("complex", ent, rel)

    if dist:
        import torch.distributed as td
        # Query rows live on their owner shard: rank r owns rows [r*n/world, (r+1)*n/world).
        # Exchange of a step: one gather kernel for the s/o rows this rank owns, ONE all-gather
        # over RCCL, one gather of the relation rows; then the two scoring calls.
        per = (n + world - 1) // world
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        own = torch.zeros(per, dtype=torch.int64, device=device)
        own_o = torch.zeros(per, dtype=torch.int64, device=device)
        own[: hi - lo], own_o[: hi - lo] = s[lo:hi], o[lo:hi]
        so_idx = torch.stack([own, own_o], 1).reshape(-1)  # rows interleaved: s_0, o_0, s_1, o_1, ...
        buf = {
            "loc": torch.empty(per * 2, DIM, dtype=torch.bfloat16, device=device),
            "gath": torch.empty(world * per, 2 * DIM, dtype=torch.bfloat16, device=device),
            "pe": torch.empty(n, DIM, dtype=torch.bfloat16, device=device),
        }
        s_rows, o_rows = buf["gath"][:n, :DIM], buf["gath"][:n, DIM:]  # row i = [s row | o row] of query i

        def exchange():
            # the query rows this rank owns (s and o interleaved) and the relation rows: ONE
            # gather launch (kge_embed), then ONE all-gather
            engine.embed(T, so_idx, p, buf["loc"], buf["pe"])
            td.all_gather_into_tensor(buf["gath"].view(-1), buf["loc"].view(-1))

        def score():  # both score blocks from one two-sided launch on the gathered dense rows
            engine.score_emb_sp_po("complex", s_rows, buf["pe"], o_rows, ent)

        # The exchange (a gather kernel and the RCCL all-gather) can be captured once in a hipGraph.  Measured
        # alternatives on one rank (tools/dist_probe.py, profiles/): replaying it on a side stream
        # one step ahead of the scoring is SLOWER (53 vs 47 us per step): the persistent scoring
        # kernel needs whole CUs (160 KB LDS, all VGPRs), so the side stream's kernels and its
        # workgroups only take turns, and the event traffic adds host work.
        # Off by default: measured on one rank the graph saves 5 % (46.0 vs 48.5 us per step) and
        # capturing RCCL collectives is the one thing here that cannot be tried on more than one
        # rank in the build environment.  KGE_BENCH_EXCHANGE_GRAPH=1 turns it on.
        xg = None
        if os.environ.get("KGE_BENCH_EXCHANGE_GRAPH") == "1":
            try:
                exchange()  # warm up the kernels / the RCCL channel outside the capture
                torch.cuda.synchronize()
                td.barrier()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    exchange()
                g.replay()
                torch.cuda.synchronize()
                xg = g
            except Exception as e:  # RCCL / torch without capture support
                print(f"[bench] graph capture of the exchange unavailable ({type(e).__name__}: {e}); "
                      "eager exchange", file=sys.stderr)
                xg = None
                torch.cuda.synchronize()
        exchange_mode = "one all-gather per step, exchange replayed as a hipGraph" if xg else \
            "one all-gather per step, eager"

        def run_steps(k_steps):
            for _ in range(k_steps):
                if xg is not None:
                    xg.replay()
                else:
                    exchange()
                score()
    else:
        exchange_mode = None

        def run_steps(k):  # KgeModel.score_sp_po: the score_sp and score_po blocks of the batch, one launch
            for _ in range(k):
                engine.score_sp_po(T, s, p, o)

    def sync():
        if dist:
            import torch.distributed as td
            td.barrier()
        torch.cuda.synchronize()

    run_steps(a.warmup)
    sync()
    t0 = time.perf_counter()
    run_steps(a.steps)
    host_el = time.perf_counter() - t0  # host time to ISSUE the steps (no device wait)
    sync()
    el = time.perf_counter() - t0
    if dist:
        import torch.distributed as td
        tt = torch.tensor([el], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        el = float(tt.item())

    # Duration of one scoring call (= one launch of the dominant kernel pairs_bf16_v4_kernel):
    # HIP events on the launch stream bracketing a
    # second timed region of the same K steps, divided by the 2K calls.  Back-to-back calls
    # pipeline their launch overhead exactly as in the timed region above.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps):
        engine.score_sp_po(T, s, p, o)
    e1.record()
    torch.cuda.synchronize()
    avg_ms = e0.elapsed_time(e1) / a.steps
    # isolated calls (event pair around every call; includes un-hidden launch latency)
    ev = []
    for k in range(min(a.steps, 50)):
        x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x0.record()
        engine.score_sp_po(T, s, p, o)
        x1.record()
        ev.append((x0, x1))
    torch.cuda.synchronize()
    iso = sorted(x0.elapsed_time(x1) for x0, x1 in ev)
    # for reference: the same step as two one-sided launches (score_sp, then score_po), as
    # TrainingJob1vsAll issues them (north_star quotes its roofline target on score_sp)
    one_ms = None
    if not a.no_one_sided:
        e0.record()
        for _ in range(a.steps):
            engine.score_sp(T, s, p)
            engine.score_po(T, p, o)
        e1.record()
        torch.cuda.synchronize()
        one_ms = e0.elapsed_time(e1) / (2 * a.steps)

    if rank == 0:
        total = 2.0 * n * E_FB * world * a.steps
        ab = algorithmic_bytes(n, E_FB, DIM, sides=2)
        ab1 = algorithmic_bytes(n, E_FB, DIM)
        achieved = ab / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "scored triples/sec (1vsAll, ComplEx d=512)",
            "value": total / el,
            "unit": "scored triples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "host_issue_ms_per_step": host_el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": "FB15k-237 shape ComplEx d=512 1vsAll scoring, bf16 tables, f32 scores: the "
                            "score_sp and score_po blocks of the batch per step (KgeModel.score_sp_po, one "
                            "two-sided launch)",
                "num_entities_per_gpu": E_FB, "num_relations": R_FB, "dim": DIM, "batch": n,
                "parallelism": f"entity-shard x{world}" if world > 1 else "single GPU",
                **({"exchange": exchange_mode} if exchange_mode else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "pairs_bf16_v4_kernel<ComplEx,d=512>, two-sided (one score_sp_po call = one launch: "
                          "gather + cooperative query build + MFMA contraction + store of both score blocks)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": ab,
                "avg_launch_us": avg_ms * 1e3,
                "isolated_call_median_us": iso[len(iso) // 2] * 1e3,
                "traffic": pmc_traffic(),
                # the same kernel launched once per direction (score_sp, score_po), same run
                **({"one_sided_launch": {"avg_launch_us": one_ms * 1e3, "algorithmic_bytes_per_launch": ab1,
                                         "achieved": ab1 / (one_ms * 1e-3) / 1e9,
                                         "frac": ab1 / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
                   if one_ms is not None else {}),
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, a.cpu_seconds)
        print(json.dumps(out))
    if dist:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
