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

N > 1: the entity table is row-sharded over the ranks (kge_amd/sharded.py, the class the gloo
tests cover): every rank sees the same batch, gathers the query rows it owns, ONE all-gather over
RCCL brings every rank the s and o rows of the batch, and each rank scores the batch against its own
shard with the same two-sided launch (kge_score_emb_sp_po) -- no collective on the score data.
Default shape for N > 1 (--shape wikidata5m) = BASELINE configs[4], the one north_star names for
scaling: E = 4,594,485 entities split over the N ranks (574,311 rows per rank at N = 8), R = 822,
d = 256, bf16, n = 512: STRONG scaling (total work fixed).  The FB15k-237-shape weak-scaling step of
earlier rounds (every rank an E = 14,541, d = 512 shard) is measured in the same run and reported
beside it (`fb15k_weak`).
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
E_WD, R_WD, DIM_WD = 4594485, 822, 256  # Wikidata5M shape (K5)
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense f32 matrix peak
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="N = 1: do not spawn the two rocprofv3 --pmc passes that measure `roofline.traffic` live "
                         "(report the committed passes instead; runs that are themselves under rocprofv3 skip them anyway)")
    ap.add_argument("--no-one-sided", action="store_true",
                    help="skip the reference region of one-sided launches (profiling runs: the kernel "
                         "trace then holds two-sided launches only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--group", type=int, default=8,
                    help="N = 1: batches per persistent launch (kge_score_queries_multi); 1 = one launch per batch")
    ap.add_argument("--lanes", type=int, default=1,
                    help="N = 1: group launches in flight (group k on HIP stream k %% lanes).  2 pays in sustained issue "
                         "(16 launches per region: 148 -> 131 us per single-pass group launch, 236 -> 216 us split, "
                         "tools/group_lanes_probe.py: the next launch's cold start under the tail of the one before) and "
                         "not in a region of 20 steps = 3 launches (16.8 against 15.5 us per step), hence 1")
    ap.add_argument("--streams", type=int, default=None,
                    help="N > 1: batches in flight in the sharded step (batch k's exchange and scoring on HIP stream "
                         "k %% streams; default: 2 with the step as a hipGraph where that has checked out, else 1)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of K steps is run this many times; the median region is reported")
    ap.add_argument("--shape", choices=["wikidata5m", "fb15k"], default="wikidata5m",
                    help="N > 1 only: which sharded workload is the headline value (the other is reported beside it)")
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


def pmc_traffic(mode, group):
    """HBM bytes per launch of the dominant kernel (one GROUP launch of `group` two-sided batches) in query mode
    `mode` ("parity" / "training") from the committed rocprofv3 --pmc passes over the same launches
    (tools/v8_pmc_target.py, tools/gpu_r4prof.sh -> profiles/r4_rocprofv3_pmc_hbm.txt, profiles/pmc_latest.json:
    FETCH_SIZE x 2 per the gfx950 correction in MI355X_MICROARCH.md + WRITE_SIZE).  bench.py itself cannot collect
    counters; None when the committed passes are of another group size."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        return d[mode]["hbm_bytes_per_launch"] if d.get("group") == group else None
    except Exception:
        return None


_PMC_LIVE = {}


def pmc_traffic_live(group, timeout_s=90):
    """HBM bytes per GROUP launch of the dominant kernel, both query modes, from counters collected BY THIS RUN: two
    separate `rocprofv3 --pmc <counter>` passes (FETCH_SIZE, then WRITE_SIZE; counters only, no trace domain) over
    tools/v8_pmc_target.py -- 12 launches per query mode of exactly the launch the timed region issues -- spawned as
    child processes on the same GPU after the timed regions.  Units and corrections as MI355X_MICROARCH.md's HBM /
    rocprofv3 section prescribes: both counters in KiB, gfx950's FETCH_SIZE counts 128-byte requests as 64 bytes (x 2);
    per launch = the median over the dispatches behind the first two.  {} when rocprofv3 is not on the PATH, a pass
    fails or takes longer than `timeout_s` (the caller then reports the committed passes and says so)."""
    if _PMC_LIVE.get("group") == group:
        return _PMC_LIVE
    import glob
    import shutil
    import sqlite3
    import statistics
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return {}
    res = {}
    work = tempfile.mkdtemp(prefix="kge_pmc_", dir="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            env = dict(os.environ, TMPDIR="/tmp", GROUP=str(group))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            t0 = time.time()
            r = subprocess.run([rp, "--pmc", c, "-d", os.path.join(work, c), "-o", "v8", "--", sys.executable,
                                os.path.join(ROOT, "tools", "v8_pmc_target.py")], cwd="/tmp", env=env, timeout=timeout_s,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dbs = glob.glob(os.path.join(work, c, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return {}
            per = {}
            con = sqlite3.connect(dbs[0])
            for name, v in con.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)):
                per.setdefault(name, []).append(v)
            con.close()
            res[c] = per
            res[c + "_seconds"] = time.time() - t0
        out = {"group": group, "seconds": res["FETCH_SIZE_seconds"] + res["WRITE_SIZE_seconds"]}
        for mode, split in (("parity", 1), ("training", 0)):
            vals = {}
            for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
                ks = [k for k in res[c] if ("pairs_bf16_v8_kernel<0, %d," % split) in k.replace("(int)", "")]
                if not ks:
                    return {}
                v = res[c][ks[0]]
                vals[c] = statistics.median(v[2:] if len(v) > 4 else v) * 1024.0 * mul
                vals[c + "_dispatches"] = len(v)
            out[mode] = {"hbm_bytes_per_launch": vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
                         "fetch_bytes_corrected": vals["FETCH_SIZE"], "write_bytes": vals["WRITE_SIZE"],
                         "dispatches": vals["FETCH_SIZE_dispatches"]}
        _PMC_LIVE.clear()
        _PMC_LIVE.update(out)
        return out
    except Exception:  # a counter pass must never take the bench line down
        return {}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def traffic_of(mode, group, live):
    """(`traffic`, `traffic_source`, detail) of the roofline object of query mode `mode`."""
    if live and mode in live:
        d = live[mode]
        return d["hbm_bytes_per_launch"], (
            "live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate child processes of this run, after the "
            "timed regions) over tools/v8_pmc_target.py = 12 launches of this group launch per query mode; KiB, FETCH_SIZE "
            "x 2 (gfx950), median per dispatch behind the first two"), {
                "fetch_bytes_corrected": d["fetch_bytes_corrected"], "write_bytes": d["write_bytes"],
                "dispatches": d["dispatches"], "passes_seconds": live.get("seconds")}
    return pmc_traffic(mode, group), ("profiles/pmc_latest.json (COMMITTED passes of tools/gpu_r6prof.sh over the same group "
                                      "launch on the builder's box: no live counter pass in this run -- rocprofv3 absent, "
                                      "--no-pmc, or a pass failed)"), None


def algorithmic_bytes(n, m, d, elt=2, sides=1):
    """target rows once + query rows (s and r) + f32 scores out + indices, per
    scoring call; a two-sided launch (sides=2: score_sp and score_po blocks of the batch) reads
    the target rows once for both."""
    return m * d * elt + sides * (n * (d + d) * elt + n * m * 4 + 2 * n * 8)


def cpu_baseline(n, seconds):
    """Reference CPU path restated op-for-op in torch (oracle/torch_port.py, bit-identical to
    the live reference in the build container), fp32, torch.no_grad, on a bounded sample: for 8,
    32 and all physical cores (an oversubscribed MKL makes the reference look worse
    than it is) one warm-up step, then best of 5 steps (score_sp + score_po); the best thread
    count is the reported baseline, the others are listed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_port as tp
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()

    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E_FB, DIM).normal_(0, 0.1, generator=g)
    rel = torch.empty(R_FB, DIM).normal_(0, 0.1, generator=g)
    s = torch.randint(E_FB, (n,), generator=g)
    p = torch.randint(R_FB, (n,), generator=g)
    o = torch.randint(E_FB, (n,), generator=g)
    # The reference itself where its package is on the box: KgeModel.score_sp / score_po of a reference ComplEx model
    # holding the same tables (oracle/ref_harness.py; tools/gpu_plugin.sh places an untracked copy of the package
    # under oracle/_ref/ for the duration of one gpurun call -- the driver's box has none: `kind` is "port" there).
    step, kind, what = None, "port", "oracle/torch_port.py (the reference's torch op sequence)"
    if os.environ.get("KGE_BENCH_REFERENCE", "1") != "0":
        try:
            import ref_harness
            if ref_harness.available():
                ref = ref_harness.make_model("complex", E_FB, R_FB, DIM)
                ref_harness.set_tables(ref, ent, rel)

                def step():
                    ref.score_sp(s, p)
                    ref.score_po(p, o)
                kind, what = "reference", "the reference's KgeModel.score_sp + score_po (ComplExScorer, LookupEmbedder)"
        except Exception as exc:  # a reference that does not import here: the port
            print(f"bench: reference not usable for cpu_baseline ({type(exc).__name__}: {exc})", file=sys.stderr)
            step = None
    if step is None:
        def step():
            tp.score_sp("complex", ent, rel, s, p)
            tp.score_po("complex", ent, rel, p, o)
    before = torch.get_num_threads()
    runs, steps_run, t_all = {}, {}, time.perf_counter()
    with torch.no_grad():
        for threads in sorted({min(8, phys), min(32, phys), phys}):
            torch.set_num_threads(threads)
            step()
            best, t_cfg, k = float("inf"), time.perf_counter(), 0
            # at least 5 steps, then more until this thread count has had a third of the budget (~10 s in all)
            while k < 5 or (time.perf_counter() - t_cfg < seconds / 3.0 and k < 400):
                t0 = time.perf_counter()
                step()
                best = min(best, time.perf_counter() - t0)
                k += 1
                if time.perf_counter() - t_all > seconds:
                    break
            runs[threads] = 2.0 * n * E_FB / best
            steps_run[threads] = k
    torch.set_num_threads(before)
    cores = max(runs, key=runs.get)
    return {"value": runs[cores], "unit": "scored triples/s", "cores": cores, "kind": kind,
            "by_threads": {str(k): v for k, v in runs.items()},
            "sample": f"best of {steps_run} 1vsAll steps per thread count (score_sp + score_po, n={n}, E={E_FB}, "
                      f"d={DIM}, fp32, no_grad) of {what} ({phys} physical cores), "
                      f"{time.perf_counter() - t_all:.1f}s of CPU work in all"}


def timed_regions(run_steps, sync, steps, repeats, reduce_max=None):
    """`repeats` timed regions of exactly `steps` steps, each bracketed by barrier + synchronize on
    both sides (max over ranks); returns (median seconds, all seconds, host-issue seconds of the
    median region)."""
    regions = []
    for _ in range(max(1, repeats)):
        sync()
        t0 = time.perf_counter()
        run_steps(steps)
        host = time.perf_counter() - t0
        sync()
        el = time.perf_counter() - t0
        if reduce_max is not None:
            el = reduce_max(el)
        regions.append((el, host))
    order = sorted(regions)
    med = order[len(order) // 2]
    return med[0], [r[0] for r in regions], med[1]


def event_avg_ms(fn, steps, repeats=1):
    """Mean duration of one call: HIP events on the launch stream around `steps` back-to-back calls; the median of
    `repeats` such regions (the timed region's K steps are short at the driver's K = 20: one region is noise)."""
    got = []
    for _ in range(max(1, repeats)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        got.append(e0.elapsed_time(e1) / steps)
    got.sort()
    return got[len(got) // 2]


def matrix_pipe_probe(device):
    """What the matrix pipe sustains on THIS box with nothing beside it: kge_debug_mfma_rate (include/kge_amd_debug.h) --
    the grid and wave layout of the persistent kernels, bare v_mfma_f32_32x32x16_bf16 on random operands, ~0.3 ms per
    launch so that the clock settles where it does under matrix load.  The MFMA-bound legs quote their fraction of the
    NOMINAL dense peak; this is the fraction a kernel of pure MFMAs gets."""
    import ctypes
    from kge_amd import _lib
    L = _lib.lib()
    fn = L.kge_debug_mfma_rate
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    g = torch.Generator(device=device).manual_seed(3)
    sink = torch.zeros(4, device=device)
    iters = 400
    st = torch.cuda.current_stream(device).cuda_stream
    # 8 waves x 4 operands x 64 lanes x 8 values.  The pipe's power draw -- and with it the clock the chip holds -- depends
    # on the DATA: random bf16 values toggle every multiplier input, zeros toggle nothing.  Both are measured; the
    # random-operand rate is what a scoring kernel's matrix work can be compared with, the zero-operand rate is what the
    # guide's 2.5 PFLOP/s (/opt/skills/guides/MI355X_MICROARCH.md) corresponds to.
    kinds = {"random": torch.randn(8 * 4 * 512, generator=g, device=device).bfloat16(),
             "zeros": torch.zeros(8 * 4 * 512, device=device).bfloat16(),
             "ones": torch.ones(8 * 4 * 512, device=device).bfloat16()}
    out = {}
    for kind, ops in kinds.items():
        def launch(ops=ops):
            fl = fn(ops.data_ptr(), iters, sink.data_ptr(), st)
            assert fl > 0, fl
            return fl
        flops = launch()
        ms = event_avg_ms(launch, 20)
        out[kind] = (ms, flops / (ms * 1e-3) / 1e12)
    ms, tf = out["random"]
    return {"kernel": "mfma_rate_kernel (kge_debug_mfma_rate: 2 waves per SIMD, two independent accumulators each, no "
                      "memory traffic)", "operands": "random bf16 values (N(0,1))", "launch_us": ms * 1e3, "achieved": tf,
            "unit": "TFLOP/s", "frac_of_nominal_peak": tf / BF16_MFMA_PEAK_TF,
            "by_operands_tflops": {k: v[1] for k, v in out.items()},
            "note": "the same instruction stream on zeros / ones / random values: the difference is the clock the chip "
                    "holds under the data's switching activity"}


def rank_legs(engine, device, n, steps):
    """One entity-ranking evaluation batch (raw + filtered + filtered-with-test counts, both directions) with the
    counts taken inside the scoring kernel (kge_score_rank_sp_po: no score matrix) and as score_sp_po + two
    rank_counts_multi scans, at the FB15k-237 shape and at one of eight Wikidata5M shards, in BOTH query modes:
    `parity` = split queries (the default of hip_entity_ranking with score_dtype bfloat16: ranks equal to float32
    arithmetic on the bf16 tables up to its summation noise; twice the matrix work per score) and
    `training_tolerance` = single-pass queries.  MFMA-bound: ALGORITHMIC flops = 2 directions * 2 n E d (the split
    mode executes twice that; `frac_executed`).  `frac` is against the nominal dense bf16 peak; `matrix_pipe_probe`
    (measured in this run) is what a kernel of bare v_mfma_f32_32x32x16_bf16 chains holds of it on this box -- 0.55 on
    the boxes profiled (1.3-1.4 PFLOP/s: the clock drops to ~1.3 GHz under matrix load;
    profiles/r4_rank8_stamps_probes.txt, probe 31)."""
    import numpy as np
    out = {"bound": "mfma", "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
           "kernel": "pairs_bf16_v8_rank_kernel<ComplEx, d/2, SPLIT> (kge_score_rank_sp_po = ONE launch that builds the "
                     "query fragments and sets the filter bits + the persistent counting kernel, which clears them)",
           "matrix_pipe_probe": matrix_pipe_probe(device)}
    rng = np.random.default_rng(0)
    for tag, E, R, d in (("fb15k-237", E_FB, R_FB, DIM), ("wikidata5m_shard", (E_WD + 7) // 8, R_WD, DIM_WD)):
        g = torch.Generator(device=device).manual_seed(7)
        ent = (torch.randn(E, d, generator=g, device=device) * 0.3).bfloat16()
        rel = (torch.randn(R, d, generator=g, device=device) * 0.3).bfloat16()
        s, p, o = (torch.from_numpy(rng.integers(0, hi, n)).to(device) for hi in (E, R, E))
        lists = []
        for tc in (o.cpu().numpy(), s.cpu().numpy()):
            per = [np.unique(np.append(rng.integers(0, E, 4), c)) for c in tc]
            end = np.cumsum([len(x) for x in per])
            beg = end - np.array([len(x) for x in per])
            one = tuple(torch.from_numpy(np.asarray(x, np.int64)).to(device) for x in (beg, end, np.concatenate(per)))
            lists.append([one, one])
        oc, sc = o.contiguous(), s.contiguous()
        flops = 2.0 * 2.0 * n * E * d
        leg = {"num_entities": E, "dim": d, "batch": n, "flops_per_batch": flops}
        for mode, flags in (("parity", engine.FLAG_SPLIT_QUERY), ("training_tolerance", 0)):
            T = engine.Tables("complex", ent, rel, flags=flags)
            t_sp = engine.score_sp(T, s, p, o).diagonal().contiguous()
            t_po = engine.score_po(T, p, o, s).diagonal().contiguous()
            cnt = torch.zeros(2, 2, 3, n, dtype=torch.int64, device=device)

            def fused():
                ok = engine.score_rank_sp_po(T, s, p, o, t_sp, t_po, lists[0], lists[1], 1e-5, 1e-4, cnt[0, 0],
                                             cnt[0, 1], cnt[1, 0], cnt[1, 1])
                assert ok

            def two_step():
                sc2 = engine.score_sp_po(T, s, p, o)
                engine.rank_counts_multi(sc2[:, :E], t_sp, lists[0], 0, oc, 1e-5, 1e-4, cnt[0, 0], cnt[0, 1])
                engine.rank_counts_multi(sc2[:, E:], t_po, lists[1], 0, sc, 1e-5, 1e-4, cnt[1, 0], cnt[1, 1])

            for fn in (fused, two_step):
                for _ in range(3):
                    fn()
            f_ms, t_ms = event_avg_ms(fused, steps), event_avg_ms(two_step, steps)
            # the way the evaluator issues it (kge_amd/eval.py): the batch captured into a hipGraph per lane, three
            # lanes in flight on three streams -- wall clock per batch over `steps` replays
            lanes = []
            cur = torch.cuda.current_stream(device)
            for _ in range(3):
                st, c = torch.cuda.Stream(device), torch.zeros_like(cnt)

                def call(c=c):
                    engine.score_rank_sp_po(T, s, p, o, t_sp, t_po, lists[0], lists[1], 1e-5, 1e-4, c[0, 0], c[0, 1],
                                            c[1, 0], c[1, 1])
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    call()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=st):
                        call()
                lanes.append((st, gr, c))

            def in_flight(k):
                for i in range(k):
                    st, gr, _ = lanes[i % 3]
                    with torch.cuda.stream(st):
                        gr.replay()
            in_flight(6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            in_flight(steps)
            torch.cuda.synchronize()
            l_ms = (time.perf_counter() - t0) / steps * 1e3
            del lanes
            ex = 2.0 if flags else 1.0
            leg[mode] = {"fused_us": f_ms * 1e3, "two_step_us": t_ms * 1e3,
                         "achieved": flops / (f_ms * 1e-3) / 1e12,
                         "frac": flops / (f_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
                         "frac_executed": ex * flops / (f_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
                         "three_in_flight": {"us_per_batch": l_ms * 1e3, "achieved": flops / (l_ms * 1e-3) / 1e12,
                                             "frac": flops / (l_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF}}
            del T
        # ---- band-and-rescore (kge_score_rank_sp_po_band) on triples that RANK HIGH -- the true object
        # is the k-th best of (s, p), which puts the same score in the tail of (p, o)'s row too: a trained model's
        # evaluation batch.  (With the true score drawn at random, as in the legs above, every tile holds a score
        # inside the band and the evaluator keeps the split kernel: engine.RankBand / EntityRankingEvaluator's probe.)
        # The split kernel, the single-pass kernel and the band form on the SAME fixture; the band's counts are checked
        # against the split kernel's in this run.
        kth = max(2, E // 10000)
        Tsp = engine.Tables("complex", ent, rel, flags=engine.FLAG_SPLIT_QUERY)
        T1 = engine.Tables("complex", ent, rel, flags=0)
        o_pl = torch.empty_like(o)
        for i0 in range(0, n, 64):
            o_pl[i0:i0 + 64] = engine.score_sp(Tsp, s[i0:i0 + 64], p[i0:i0 + 64]).topk(kth, dim=1).indices[:, -1]
        tp_sp = engine.score_sp(Tsp, s, p, o_pl).diagonal().contiguous()
        tp_po = engine.score_po(Tsp, p, o_pl, s).diagonal().contiguous()
        t1_sp = engine.score_sp(T1, s, p, o_pl).diagonal().contiguous()
        t1_po = engine.score_po(T1, p, o_pl, s).diagonal().contiguous()
        band = engine.RankBand(Tsp, n)

        def counts_of(T_, a_sp, a_po, b=None):
            c = torch.zeros(2, 2, 3, n, dtype=torch.int64, device=device)
            assert engine.score_rank_sp_po(T_, s, p, o_pl, a_sp, a_po, lists[0], lists[1], 1e-5, 1e-4, c[0, 0], c[0, 1],
                                           c[1, 0], c[1, 1], band=b)
            return c
        want, got = counts_of(Tsp, tp_sp, tp_po), counts_of(Tsp, tp_sp, tp_po, band)
        listed, dropped = band.status()
        if dropped != 0 or not torch.equal(want, got):
            raise RuntimeError("bench: band-and-rescore counts differ from the split kernel's")
        us = {}
        for key, (T_, a_sp, a_po, b) in (("split_us", (Tsp, tp_sp, tp_po, None)), ("single_pass_us", (T1, t1_sp, t1_po, None)),
                                         ("band_us", (Tsp, tp_sp, tp_po, band))):
            fn = lambda: engine.score_rank_sp_po(T_, s, p, o_pl, a_sp, a_po, lists[0], lists[1], 1e-5, 1e-4, cnt[0, 0],
                                                 cnt[0, 1], cnt[1, 0], cnt[1, 1], band=b)
            for _ in range(3):
                fn()
            us[key] = event_avg_ms(fn, steps) * 1e3
        leg["planted_true_scores"] = dict(
            us, true_object="the k-th best entity of (s, p)", kth_best=kth, pairs_listed=listed,
            pairs=band.pairs_of(n), listed_share=listed / band.pairs_of(n),
            band_over_single_pass=us["band_us"] / us["single_pass_us"], band_over_split=us["band_us"] / us["split_us"],
            counts_equal_the_split_kernels=True,
            frac=flops / (us["band_us"] * 1e-6) / 1e12 / BF16_MFMA_PEAK_TF)
        del Tsp, T1, band
        # (round-3 readers: the single-pass figures under their old keys)
        leg.update({k: leg["training_tolerance"][k] for k in ("fused_us", "two_step_us", "achieved", "frac")})
        out[tag] = leg
        del ent, rel
        torch.cuda.empty_cache()
    return out


def neg_legs(engine, device, steps):
    """BASELINE configs[2]: negative-sampling scores, WN18RR shape (E = 40,943, R = 11, d = 512; RotatE relations 256),
    512 positives x 1,000 uniform negatives per slot (kge/util/sampler.py:291-306, 592-595), float32 tables:
    kge_score_neg (the "triple" implementation without the [n K, 3] index tensor) -- one gathered entity row per
    scored triple, HBM-gather bound.  Algorithmic bytes per launch: n K (d 4 + 4 + 8) + the fixed rows.
    The 84 MB table sits in the 256 MB Infinity Cache, so the same launch is also measured on a table beyond it
    (E = 2,000,000: 4.1 GB): THAT figure is an HBM figure.  `traffic` = FETCH_SIZE x 2 + WRITE_SIZE of the committed
    rocprofv3 --pmc passes of tools/neg_pmc.py (profiles/pmc_neg_latest.json)."""
    n, K, d = 512, 1000, 512
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "kernel": "neg_kernel<scorer> (kge_score_neg: fixed side in registers, corrupted rows stream)",
           "batch": n, "num_negatives": K, "dim": d, "dtype": "f32"}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_neg_latest.json")) as f:
            pmc = json.load(f)
    except Exception:
        pmc = {}
    for tag, E in (("wn18rr", 40943), ("beyond_infinity_cache", 2000000)):
        g = torch.Generator(device=device).manual_seed(5)
        ent = torch.empty(E, d, device=device).normal_(0, 0.1, generator=g)
        q = torch.Generator().manual_seed(6)
        s, o = (torch.randint(E, (n,), generator=q).to(device) for _ in range(2))
        p = torch.randint(11, (n,), generator=q).to(device)
        neg = torch.randint(E, (n, K), generator=q).to(device)
        leg = {"num_entities": E, "table_bytes": E * d * 4}
        for model in ("rotate", "transe"):
            dr = d // 2 if model == "rotate" else d
            rel = torch.empty(11, dr, device=device).uniform_(-3.14, 3.14, generator=g)
            T = engine.Tables(model, ent, rel)
            for _ in range(3):
                engine.score_neg(T, s, p, o, 2, neg)
            ms = event_avg_ms(lambda: engine.score_neg(T, s, p, o, 2, neg), steps)
            ab = n * K * (d * 4 + 4 + 8) + n * (d + dr + d) * 4 + 3 * n * 8
            leg[model] = {"avg_launch_us": ms * 1e3, "algorithmic_bytes_per_launch": ab,
                          "achieved": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "scored_triples_per_s": n * K / (ms * 1e-3),
                          "traffic": pmc.get(f"{tag}_{model}")}
            del T, rel
        out[tag] = leg
        del ent, neg
        torch.cuda.empty_cache()
    return out


def eval_leg(engine, device):
    """BASELINE configs[3] on one GPU: one EntityRankingJob-equivalent pass (kge_amd.eval.EntityRankingEvaluator, the
    mirror of eval_entity_ranking.py:103-481) at the FB15k-237 shape -- DistMult d = 512, 17,535 validation triples,
    filters from a 272,115-triple Zipf train split + valid + test, raw / filtered / filtered-with-test rankings of
    both directions, batch 512 -- in the three scoring settings: float32 tables (the reference's precision), bf16
    tables with split queries (rank parity with float32 arithmetic on them) and bf16 tables with the counts taken
    inside the scoring kernel.  Wall clock of the whole pass (second run: indexes and graphs built), per batch,
    and the share of it the scoring (+ counting) kernels account for (their back-to-back HIP-event time per batch)."""
    import numpy as np
    from kge_amd import eval as kev
    from kge_amd.synthetic import SHAPES, make_splits
    E, R, ntr, nva, nte = SHAPES["fb15k-237"]
    d, bs = 512, 512
    splits = {k: v.astype(np.int64) for k, v in make_splits(E, R, ntr, nva, nte, seed=0).items()}
    g = torch.Generator().manual_seed(0)
    ent = torch.empty(E, d).normal_(0, 0.1, generator=g)
    rel = torch.empty(R, d).normal_(0, 0.1, generator=g)
    nb = (nva + bs - 1) // bs
    q = torch.Generator().manual_seed(1)
    s, p, o = (torch.randint(hi, (bs,), generator=q).to(device) for hi in (E, R, E))
    out = {"model": "distmult", "num_entities": E, "dim": d, "triples": nva, "batch": bs, "batches": nb,
           "rankings": ["raw", "filtered", "filtered_with_test"]}
    for tag, e_, r_, flags in (("f32_tables", ent, rel, 0),
                               ("bf16_split_queries", ent.bfloat16(), rel.bfloat16(), engine.FLAG_SPLIT_QUERY),
                               ("bf16_counting_kernel", ent.bfloat16(), rel.bfloat16(), 0)):
        T = engine.Tables("distmult", e_.to(device), r_.to(device), flags=flags)
        ev = kev.EntityRankingEvaluator(T, splits, E, R, batch_size=bs)
        ev.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = ev.run()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if tag == "bf16_counting_kernel":
            t_sp = engine.score_sp(T, s, p, o).diagonal().contiguous()
            t_po = engine.score_po(T, p, o, s).diagonal().contiguous()
            cnt = torch.zeros(2, 2, 1, bs, dtype=torch.int64, device=device)
            k_ms = event_avg_ms(lambda: engine.score_rank_sp_po(T, s, p, o, t_sp, t_po, [], [], 1e-5, 1e-4, cnt[0, 0],
                                                                cnt[0, 1], cnt[1, 0], cnt[1, 1]), 50)
        else:
            for _ in range(3):
                engine.score_sp_po(T, s, p, o)
            k_ms = event_avg_ms(lambda: engine.score_sp_po(T, s, p, o), 50)
        out[tag] = {"pass_ms": wall * 1e3, "ms_per_batch": wall * 1e3 / nb, "triples_per_s": nva / wall,
                    "scored_triples_per_s": 2.0 * nva * E / wall, "scoring_kernel_ms_per_batch": k_ms,
                    "scoring_kernel_share": k_ms * nb / (wall * 1e3), "graph_batches": ev.graph_batches,
                    "mrr_filtered": m["mean_reciprocal_rank_filtered"]}
        del T, ev
        torch.cuda.empty_cache()
    return out


def sharded_workload(shape, world, rank, device, n, engine):
    """The rank's shard of the named shape + the replicated batch: (ShardedEntityTable, s, p, o, E, d)."""
    from kge_amd.sharded import ShardedEntityTable
    if shape == "wikidata5m":
        E, R, d = E_WD, R_WD, DIM_WD
    else:  # every rank an FB15k-237-sized shard: weak scaling
        E, R, d = E_FB * world, R_FB, DIM
    lo, hi = ShardedEntityTable.partition(E, world, rank)
    g = torch.Generator(device=device).manual_seed(100 + rank)
    ent = torch.empty(hi - lo, d, device=device, dtype=torch.bfloat16).normal_(0, 0.1, generator=g)
    rel = torch.empty(R, d, dtype=torch.float32).normal_(0, 0.1, generator=torch.Generator().manual_seed(1234))
    q = torch.Generator().manual_seed(1)  # the same batch on every rank
    s = torch.randint(E, (n,), generator=q).to(device)
    p = torch.randint(R, (n,), generator=q).to(device)
    o = torch.randint(E, (n,), generator=q).to(device)
    # one forced rank (KGE_BENCH_FORCE_DIST=1): still run every collective, so a 1-GPU box exercises RCCL
    sh = ShardedEntityTable("complex", ent, rel.to(torch.bfloat16).to(device), E, backend=engine,
                            force_collectives=os.environ.get("KGE_BENCH_FORCE_DIST") == "1")
    return sh, s, p, o, E, d


def main_sharded(a, world, rank, device):
    import torch.distributed as td
    from kge_amd import engine
    from kge_amd.sharded import ShardedScoreLanes
    # This RCCL build prints a version banner (five lines) on STDOUT when its communicator comes up; the contract is
    # ONE JSON line there.  File descriptor 1 points at stderr until the communicator exists.
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        td.init_process_group("nccl", device_id=device)
        warm = torch.zeros(1, device=device)
        td.all_reduce(warm)
        torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if td.get_world_size() != world or (a.gpus != world and os.environ.get("KGE_BENCH_FORCE_DIST") != "1"):
        raise SystemExit(f"bench: RCCL reports {td.get_world_size()} ranks, --gpus {a.gpus}, WORLD_SIZE {world}")
    n = a.batch

    def sync():
        td.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        t = torch.tensor([x], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    results = {}
    for shape in ("wikidata5m", "fb15k"):
        sh, s, p, o, E, d = sharded_workload(shape, world, rank, device, n, engine)

        # `--streams` batches in flight (kge_amd.sharded.ShardedScoreLanes: batch k's exchange and scoring on HIP
        # stream k % L; the exchange of batch k + 1 runs under the scoring launch of batch k); slabs that outgrow
        # the Infinity Cache (the Wikidata5M shards) go one batch at a time -- two of them in flight would be
        # 2 x 2.4 GB of scores written at once for an exchange that is 1 % of the step
        big_slab = n * (sh.hi - sh.lo) * 4 > sh.BIG_SLAB_BYTES
        # Default: ONE batch at a time, issued call by call -- the configuration every earlier round ran.  Two lanes pay
        # only with the step captured into a hipGraph (call by call the step is bound by the host: 60 -> 84 us with
        # two lanes, 60 -> 47 us with two lanes + graph, one rank); the capture of a multi-rank RCCL all-gather has
        # not run on this code, so N > 1 takes it only when asked (--streams 2 with ShardedScoreLanes.GRAPH = True).
        lanes = ShardedScoreLanes(sh, 1 if big_slab else max(1, a.streams if a.streams is not None else 2))
        tri3 = torch.stack([s.long(), p.long(), o.long()], 1).contiguous()

        def run_steps(k, lanes=lanes, tri3=tri3):
            lanes.fork()
            for i in range(k):
                lanes.score_sp_po_blocks(tri3)  # exchange (gather, all-gather, gather) + the scoring launch(es)
                if (i + 1) % lanes.L == 0:
                    lanes.join()  # the consumer's wait; the slabs of the two batches are released here
            lanes.join()

        run_steps(max(a.warmup, 2 * lanes.L))  # (the first step of a lane captures its hipGraph and checks it)
        if lanes.L > 1 and not lanes.use_graph and a.streams is None:
            # the capture did not pass its self-check (ShardedScoreLanes warned): two lanes call by call are bound by
            # the host (86 against 63 us per step, one rank) -- one batch at a time then
            if rank == 0:
                print(f"bench: sharded step not captured ({lanes.graph_error}); one batch at a time", file=sys.stderr)
            lanes = ShardedScoreLanes(sh, 1, graph=False)

            def run_steps(k, lanes=lanes, tri3=tri3):
                for _ in range(k):
                    lanes.score_sp_po_blocks(tri3)
            run_steps(a.warmup)
        el, regions, host = timed_regions(run_steps, sync, a.steps, a.repeats, reduce_max)
        # the scoring launch alone, on rows already exchanged (HIP events on the launch stream)
        rows, rel_rows = sh.exchange_rows([s, o], p)
        s_rows, o_rows = rows[:n].clone(), rows[n:].clone()
        rel_rows = rel_rows.clone()
        big = n * (sh.hi - sh.lo) * 4 > sh.BIG_SLAB_BYTES

        def score_only():  # what ShardedEntityTable.score_sp_po_blocks launches after the exchange
            if big:
                engine.score_emb("complex", s_rows, rel_rows, sh.ent_local, "sp_", pad_pitch=True)
                engine.score_emb("complex", sh.ent_local, rel_rows, o_rows, "_po", pad_pitch=True)
            else:
                engine.score_emb_sp_po("complex", s_rows, rel_rows, o_rows, sh.ent_local, pad_pitch=True)
        # (at least 40 launches after two untimed ones, median of three regions: five launches right behind a
        # synchronize measured the first launch's wake-up, 41 us for a 23 us launch)
        for _ in range(2):
            score_only()
        k_ms = event_avg_ms(score_only, max(40, a.steps), repeats=3)
        x_ms = event_avg_ms(lambda: sh.exchange_rows([s, o], p), max(20, a.steps), repeats=3)
        m = sh.hi - sh.lo
        # one 1vsAll TRAINING step on the same shard shapes (kge_amd.sharded_train.ShardedTrainingJob1vsAll: fused score
        # + loss per shard, statistics exchanged, backward, this rank's Adagrad step, tables re-cast): the job-level
        # number beside the scoring step
        train_ms = None
        try:
            from kge_amd.sharded_train import ShardedTrainingJob1vsAll
            import gc
            del rows
            job = ShardedTrainingJob1vsAll("complex", E, sh.rel.shape[0], d, state_dict=None, seed=3, lr=0.1,
                                           optimizer="Adagrad", device=device, backend=engine) if E <= 200000 else None
            if job is not None:
                tri = torch.stack([s, p, o], 1)
                for _ in range(3):
                    job.step(tri)
                sync()
                t0 = time.perf_counter()
                for _ in range(max(3, a.steps // 4)):
                    job.step(tri)
                sync()
                train_ms = reduce_max((time.perf_counter() - t0) / max(3, a.steps // 4) * 1e3)
                del job
                gc.collect()
        except Exception as exc:  # a training leg that fails must not take the scoring line with it
            train_ms = f"failed: {type(exc).__name__}: {exc}"
        rows = None
        results[shape] = {
            "train_step_ms": train_ms,
            "value": 2.0 * n * E * a.steps / el, "ms_per_step": el / a.steps * 1e3,
            "host_issue_ms_per_step": host / a.steps * 1e3, "regions_ms_per_step": [r / a.steps * 1e3 for r in regions],
            "num_entities": E, "rows_per_rank": m, "dim": d, "batch": n,
            "scaling": "strong" if shape == "wikidata5m" else "weak",
            "scoring_launch_ms": k_ms, "exchange_ms": x_ms, "launches_per_step": 2 if big else 1,
            "batches_in_flight": lanes.L, "step_as_hipgraph": bool(lanes.use_graph and lanes.graph_replays > 0),
            "algorithmic_bytes_per_launch": algorithmic_bytes(n, m, d, sides=2) if not big else
            2 * algorithmic_bytes(n, m, d, sides=1),
        }
        del sh, s_rows, o_rows
        torch.cuda.empty_cache()
    if rank == 0:
        main_shape = a.shape
        r = results[main_shape]
        ach = r["algorithmic_bytes_per_launch"] / (r["scoring_launch_ms"] * 1e-3) / 1e9
        out = {
            "metric": "scored triples/sec (1vsAll, ComplEx d=512)",
            "value": r["value"], "unit": "scored triples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "repeats": a.repeats, "ms_per_step": r["ms_per_step"], "host_issue_ms_per_step": r["host_issue_ms_per_step"],
            "higher_is_better": True, "scaling": r["scaling"], "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": ("Wikidata5M shape (E=4,594,485, R=822) ComplEx d=256 1vsAll scoring, entity table "
                             "row-sharded over the ranks, bf16 tables, f32 scores: score_sp and score_po blocks of "
                             "the batch against every shard per step" if main_shape == "wikidata5m" else
                             "FB15k-237 shape ComplEx d=512 1vsAll scoring, one E=14,541 shard per rank"),
                "num_entities": r["num_entities"], "rows_per_rank": r["rows_per_rank"], "dim": r["dim"], "batch": n,
                "parallelism": f"entity-shard x{world} (kge_amd.sharded.ShardedEntityTable)",
                "exchange": "kge_embed gather -> ONE all_gather_into_tensor (RCCL) -> kge_embed pick, per step",
            },
            "rccl_ranks": td.get_world_size(),
            # what to compare this line with: the N = 1 line of `python bench.py --gpus 1` is another workload (the
            # FB15k-237 shape on one GPU, no sharding, the parity query mode)
            "scaling_note": ("strong scaling of the Wikidata5M shape (north_star: 'near-linear triples/s to 8 GPUs on "
                             "Wikidata5M-scale entity shards'): the one-GPU point of THIS curve is the whole table on one "
                             "rank -- `KGE_BENCH_FORCE_DIST=1` under torch.distributed.run with one process "
                             "(profiles/r4_bench_dist1rank.json: 9.1-9.6e11 scored triples/s, 4.9-5.2 ms per step, box to box) -- not the "
                             "`--gpus 1` line; the FB15k-237-shape weak-scaling step (one shard per rank, single-pass "
                             "bf16 queries) is `fb15k_weak`, whose one-rank point is 3.05-3.08e11 in the same file"
                             if main_shape == "wikidata5m" else
                             "weak scaling of the FB15k-237 shape (one E=14,541 shard per rank, single-pass bf16 queries); "
                             "one-rank point of this curve: profiles/r4_bench_dist1rank.json (3.05-3.08e11 scored triples/s)"),
            "roofline": {"bound": "hbm", "kernel": "the scoring launch(es) of one step on this rank's shard after the "
                                                   "exchange (kge_score_emb_sp_po_blocks: query build on the exchanged rows + "
                                                   "pairs_bf16_v7_kernel on the prepared fragments, two-sided, both blocks on "
                                                   "256-byte lines; "
                                                   "slabs beyond the Infinity Cache: one one-sided launch per direction)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                         "avg_launch_us": r["scoring_launch_ms"] * 1e3, "traffic": None},
            "exchange_ms": r["exchange_ms"],
            # one whole 1vsAll training step on the same shards (ShardedTrainingJob1vsAll; max over ranks)
            "train_step_ms": r.get("train_step_ms"),
            ("fb15k_weak" if main_shape == "wikidata5m" else "wikidata5m_strong"):
                results["fb15k" if main_shape == "wikidata5m" else "wikidata5m"],
        }
        print(json.dumps(out))
    try:  # every rank reaches the end before any rank tears its communicator down
        td.barrier()
    except Exception:
        pass
    td.destroy_process_group()


def train_leg(device, n, steps):
    """One whole 1vsAll TRAINING step at the bench shape (kge/job/train_1vsAll.py:48-82 + train.py:471-474): the fused
    cross-entropy forward of both directions (no score matrix), its backward (d loss / d score recomputed in the
    scoring kernel, two gradient products) and a one-pass Adagrad step over both tables -- kge_amd.model.loss_sp_po_sum +
    kge_amd.optim.Adagrad, the path `train.type: hip_1vsAll` drives.  bf16 scoring copies of float32 master tables
    (mixed precision) and float32 scoring.  Wall clock over `steps` steps, synchronised at both ends."""
    from kge_amd import model as km, optim as kopt
    out = {"batch": n, "num_entities": E_FB, "dim": DIM, "loss": "kl (1vsAll cross entropy, both directions)",
           "optimizer": "Adagrad (one pass: kge_adagrad_step)"}
    q = torch.Generator().manual_seed(3)
    s, p, o = (torch.randint(hi, (n,), generator=q).to(device) for hi in (E_FB, R_FB, E_FB))
    for tag, sd in (("bf16_scoring", torch.bfloat16), ("f32_scoring", torch.float32)):
        torch.manual_seed(0)
        m = km.create("complex", E_FB, R_FB, DIM, device=device, score_dtype=sd)
        opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=(sd == torch.bfloat16))

        def step():
            opt.zero_grad(set_to_none=True)
            m.loss_sp_po_sum(s, p, o).backward()
            opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        # forward alone (the scoring launch with the loss epilogue), HIP events
        with torch.no_grad():
            f_ms = event_avg_ms(lambda: m.loss_sp_po(s, p, o), max(10, steps))
        # the same step as ONE hipGraph replay (kge_amd.train_graph.GraphedStep: the eager step is issued by ~0.3 ms
        # of Python for ~0.17 ms of kernels)
        g_ms = None
        if sd == torch.bfloat16:
            from kge_amd.train_graph import GraphedStep
            tri = torch.stack([s, p, o], 1)  # the batch as LibKGE hands it over: one [n, 3] tensor, one copy per replay
            gs = GraphedStep(lambda t_: m.loss_sp_po_sum(t_[:, 0], t_[:, 1], t_[:, 2]), opt, warmup=1)
            for _ in range(4):
                gs(tri)
            if gs.replays > 0:
                # >= 100 replays per timed region: the region's fixed cost (the first launch's latency, the final
                # synchronize: ~0.2 ms) over the 10 steps of a default run read as 0.02-0.03 ms per step
                g_steps = max(100, steps)
                g_copy_ms = None
                # (a) the batch handed over as a device tensor: one device-to-device copy kernel in front of every replay;
                # (b) the batch written INTO the captured step's input buffer (GraphedStep.static_inputs: what the LibKGE
                # plugin does with the loader's host batch -- one host-to-device copy, no device-side hop): the replay alone
                for which in ("copy", "in_place"):
                    arg = tri if which == "copy" else gs.static_inputs[0]
                    if which == "in_place":
                        arg.copy_(tri)
                    for _ in range(5):
                        gs(arg)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(g_steps):
                        gs(arg)
                    torch.cuda.synchronize()
                    t_ms = (time.perf_counter() - t0) / g_steps * 1e3
                    if which == "copy":
                        g_copy_ms = t_ms
                    else:
                        g_ms = t_ms
            del gs
        flops = 3 * 2.0 * 2.0 * n * DIM * E_FB  # forward + two gradient products, both directions
        peak = BF16_MFMA_PEAK_TF if sd == torch.bfloat16 else F32_MFMA_PEAK_TF
        out[tag] = {"ms_per_step": ms, "forward_ms": f_ms, "scored_triples_per_s": 2.0 * n * E_FB / (ms * 1e-3),
                    "flops_per_step": flops, "achieved_tflops": flops / (ms * 1e-3) / 1e12,
                    "frac_of_mfma_peak": flops / (ms * 1e-3) / 1e12 / peak}
        if g_ms is not None:
            out[tag]["graph_replay"] = {"ms_per_step": g_ms, "replays_timed": max(100, steps),
                                        "inputs": "written into GraphedStep.static_inputs (no copy kernel per replay)",
                                        "ms_per_step_with_device_copy_of_the_batch": g_copy_ms,
                                        "scored_triples_per_s": 2.0 * n * E_FB / (g_ms * 1e-3),
                                        "achieved_tflops": flops / (g_ms * 1e-3) / 1e12,
                                        "frac_of_mfma_peak": flops / (g_ms * 1e-3) / 1e12 / peak}
        del m, opt
        torch.cuda.empty_cache()
    return out


def ns_step_leg(device, n, steps):
    """One whole NEGATIVE-SAMPLING training step at BASELINE configs[2] (WN18RR shape, RotatE d = 512, 1000 negatives per
    slot, float32): TrainingJobNegativeSampling._process_subbatch (kge/job/train_negative_sampling.py:103-164) -- per slot
    the positives (score_spo), the [n, K] negative block (kge_score_neg: fixed side in registers, corrupted rows stream),
    the kl loss on [n, 1 + K], backward (kge_score_neg_bwd_accum) -- + one-pass Adagrad over both tables, through
    kge_amd.model + kge_amd.optim: what `train.type: hip_negative_sampling` drives.  Wall clock per step issued call by
    call and as ONE hipGraph replay (kge_amd.train_graph.GraphedStep; hip_negative_sampling.graph_step).  The negatives'
    gather is the algorithmic traffic: 2 slots x n x K rows of d floats forward, the same rows read again + their
    gradient rows scattered in the backward."""
    from kge_amd import engine, model as km, optim as kopt
    from kge_amd.train_graph import GraphedStep
    E, R, d, K = 40943, 11, DIM, 1000
    q = torch.Generator().manual_seed(5)
    s, p, o = (torch.randint(hi, (n,), generator=q).to(device) for hi in (E, R, E))
    negs = [torch.randint(E, (n, K), generator=q).to(device) for _ in range(2)]
    torch.manual_seed(0)
    m = km.create("rotate", E, R, d, device=device)
    opt = kopt.Adagrad(m.parameters(), lr=0.1)
    labels = torch.zeros(n, K + 1, device=device)
    labels[:, 0] = 1
    target = torch.nn.functional.normalize(labels, p=1, dim=1)

    def loss_fn(s_, p_, o_, ns_, no_):
        total = None
        pos, sc_s, sc_o = m.score_neg_blocks(s_, p_, o_, ns_, no_)  # one autograd node: one pair of table gradients
        for sc in (sc_s, sc_o):
            scores = torch.cat([pos.view(-1, 1), sc], dim=1)
            part = torch.nn.functional.kl_div(torch.log_softmax(scores, 1), target, reduction="sum") / n
            total = part if total is None else total + part
        return total

    def step():
        opt.zero_grad(set_to_none=True)
        loss_fn(s, p, o, negs[0], negs[1]).backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"model": "rotate", "num_entities": E, "dim": d, "batch": n, "num_negatives_per_slot": K, "dtype": "f32",
           "loss": "kl on [n, 1 + K] per slot", "optimizer": "Adagrad (one pass: kge_adagrad_step)",
           "eager": {"ms_per_step": ms, "scored_triples_per_s": 2.0 * n * (K + 1) / (ms * 1e-3)}}
    # the same eager step with the backward's scatter as one float atomic per element and occurrence
    # (kge_score_neg_bwd_accum) instead of sorted by entity (kge_score_neg_bwd_accum_sorted: the default at this shape)
    engine.NEG_BWD_SORTED = False
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out["eager_with_atomic_scatter"] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3}
    finally:
        engine.NEG_BWD_SORTED = None
    gs = GraphedStep(loss_fn, opt, warmup=1)
    for _ in range(4):
        gs(s, p, o, negs[0], negs[1])
    if gs.replays > 0:
        g_steps = max(50, steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(g_steps):
            gs(s, p, o, negs[0], negs[1])
        torch.cuda.synchronize()
        g_ms = (time.perf_counter() - t0) / g_steps * 1e3
        # forward gather of both slots + the same rows again and their gradient rows in the backward (f32 rows of d)
        abytes = 2 * n * K * d * 4 * 3.0
        out["graph_replay"] = {"ms_per_step": g_ms, "scored_triples_per_s": 2.0 * n * (K + 1) / (g_ms * 1e-3),
                               "gather_bytes_per_step": abytes, "frac_of_hbm_peak": abytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    else:
        out["graph_replay"] = {"disabled": gs.disabled_reason}
    return out


def kvsall_step_leg(device, n, steps):
    """One whole KvsAll TRAINING step at BASELINE configs[3] (FB15k-237 shape, DistMult d = 512, bf16 scoring copies of
    float32 masters): TrainingJobKvsAll._process_subbatch (kge/job/train_KvsAll.py:216-294) -- n sp_ queries and n _po
    queries, each with its multi-hot labels as a CSR (1-8 known answers per query, as the KvsAll index hands them over),
    the kl loss fused into the scoring kernel (kge_kl_fwd: no [n, E] score or label matrix), ONE backward for both query
    types (kge_multilabel2_bwd_accum; `ms_per_step_one_backward_per_type`: kge_kl_bwd per type, autograd assembling the
    table gradients) -- + one-pass Adagrad, through kge_amd.model + kge_amd.optim: what `train.type: hip_KvsAll` drives.
    Wall clock per step issued call by call, and as one hipGraph replay."""
    from kge_amd import model as km, optim as kopt
    q = torch.Generator().manual_seed(9)
    m = km.create("distmult", E_FB, R_FB, DIM, device=device, score_dtype=torch.bfloat16)
    opt = kopt.Adagrad(m.parameters(), lr=0.1, bf16_copies=True)
    a, b = (torch.randint(hi, (n,), generator=q).to(device) for hi in (E_FB, R_FB))
    c = torch.randint(E_FB, (n,), generator=q).to(device)
    csr = []
    for _ in range(2):
        cnt = torch.randint(1, 9, (n,), generator=q)
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(cnt, 0)
        col = torch.cat([torch.randperm(E_FB, generator=q)[:int(k)].sort().values for k in cnt])
        csr.append((rowptr.to(device), col.to(device)))

    def both(a_, b_, c_, rp0, cl0, rp1, cl1):  # both query types, one backward (kge_multilabel2_bwd_accum)
        return m.multilabel_loss_sp_po("kl", a_, b_, rp0, cl0, c_, b_, rp1, cl1, sum_scale=1.0 / (2 * n))

    def step():
        opt.zero_grad(set_to_none=True)
        both(a, b, c, *csr[0], *csr[1]).backward()
        opt.step()

    def step_per_type():  # the reference's order: each type's loss back-propagated on its own
        opt.zero_grad(set_to_none=True)
        (m.kl_loss_sp(a, b, *csr[0]).sum() / (2 * n)).backward()
        (m.kl_loss_po(b, c, *csr[1]).sum() / (2 * n)).backward()
        opt.step()
    for _ in range(3):
        step_per_type()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_per_type()
    torch.cuda.synchronize()
    ms_per_type = (time.perf_counter() - t0) / steps * 1e3
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    flops = 3 * 2.0 * 2.0 * n * DIM * E_FB
    out = {"model": "distmult", "num_entities": E_FB, "dim": DIM, "queries_per_type": n, "labels_per_query": "1-8",
           "loss": "kl on multi-hot labels (CSR), fused into the scoring kernel", "ms_per_step": ms,
           "ms_per_step_one_backward_per_type": ms_per_type,
           "scored_triples_per_s": 2.0 * n * E_FB / (ms * 1e-3), "flops_per_step": flops,
           "frac_of_mfma_peak": flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF}
    # The same step as ONE hipGraph replay: what the GPU needs for it once the host is out of the way.  (The kernels
    # follow the row pointers, so a label buffer of fixed CAPACITY replays for any batch; `hip_KvsAll` does not do
    # this yet -- its batches also differ in how many sp_ and _po queries they hold: DESIGN 11.7.)
    from kge_amd.train_graph import GraphedStep

    gs = GraphedStep(both, opt, warmup=1)
    args = (a, b, c, csr[0][0], csr[0][1], csr[1][0], csr[1][1])
    for _ in range(4):
        gs(*args)
    if gs.replays > 0:
        g_steps = max(100, steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(g_steps):
            gs(*args)
        torch.cuda.synchronize()
        g_ms = (time.perf_counter() - t0) / g_steps * 1e3
        out["graph_replay"] = {"ms_per_step": g_ms, "replays_timed": g_steps,
                               "scored_triples_per_s": 2.0 * n * E_FB / (g_ms * 1e-3),
                               "frac_of_mfma_peak": flops / (g_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF}
    else:
        out["graph_replay"] = {"disabled": gs.disabled_reason}
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node
    (the contract's own command line), so that the line never reports n_gpus = 1 for a request of N."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, KGE_BENCH_SPAWNED="1")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        if os.environ.get("KGE_BENCH_SPAWNED") == "1":
            raise SystemExit("bench: re-executed under torch.distributed.run but WORLD_SIZE is not set")
        raise SystemExit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not (world == 1 and os.environ.get("KGE_BENCH_FORCE_DIST") == "1"):
        raise SystemExit(f"bench: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to "
                         f"report a line for another world size")
    if os.environ.get("KGE_BENCH_DEVICE_CHECK") == "1":  # tests (no GPU): everything up to the device selection
        print(json.dumps({"rank": rank, "local_rank": local, "world_size": world, "gpus": a.gpus}), flush=True)
        return
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # KGE_BENCH_FORCE_DIST=1: exercise the sharded step (RCCL init + all-gather) with one rank
    if world > 1 or os.environ.get("KGE_BENCH_FORCE_DIST") == "1":
        return main_sharded(a, world, rank, device)

    from kge_amd import engine

    n = a.batch
    L = max(1, a.group)
    ent, rel, s, p, o = make_inputs(rank, device, n)
    PITCH = engine.score_pitch(E_FB)  # 14,656 floats: whole 256-byte lines, an odd number of them per row
    ab = algorithmic_bytes(n, E_FB, DIM, sides=2)  # per batch: table once + both sides' query rows, scores, indices

    # Batches are known ahead in every caller of this path (a DataLoader over the split: eval_entity_ranking.py:158-170,
    # train_1vsAll.py:31-41), so the loop is issued the way such a caller can issue it: GROUPS of `--group` batches, one
    # persistent launch per group (kge_score_queries_multi, pairs_bf16_v8_kernel: the table streams through every
    # compute unit once per batch; cold start, launch gap and the compute units a single batch's grid cannot fill are
    # paid once per group), which also builds the NEXT group's query vectors behind its last unit.  One STEP = one batch
    # = KgeModel.score_sp_po of that batch (both score blocks); K steps = K batches in ceil(K / group) launches (a
    # remainder of K % group batches goes as one smaller group), all of a step's work inside the timed region.
    # Score rows on a 256-byte pitch (the C ABI's `ldo`; engine.score_pitch), the po block on a column of its own.
    # `--lanes` L > 1 (default 1, see its help): group k is issued on HIP stream k % L (each lane its own query fragments
    # and score buffer; a lane's launch builds the queries of THAT lane's next group), so the next persistent launch
    # moves into the compute units the one before is leaving -- its tail, the drain of its stores, the cold start.  The
    # results of a region belong to the caller after the join at its end, as with engine.ScorePipeline(streams=2).
    # `roofline` stays on ONE stream's back-to-back launches (the kernel's own duration); `timed_region` is the
    # chip-level rate.
    class Mode:
        def __init__(self, flags, tag, lanes):
            self.tag, self.flags = tag, flags
            self.T = engine.Tables("complex", ent, rel, flags=flags or 0)
            self.by_size = {}
            self.lanes = max(1, int(lanes))
            self.streams = [torch.cuda.Stream(device=device) for _ in range(self.lanes)] if self.lanes > 1 else None
            self.last = None  # (group size, lane) of the last launch

        def group(self, g):
            st = self.by_size.get(g)
            if st is None:
                q = torch.Generator().manual_seed(100 + g)
                st = self.by_size[g] = []
                for _ in range(self.lanes):
                    tri = [torch.stack([torch.randint(hi, (n * g,), generator=q) for hi in (E_FB, R_FB, E_FB)], 1).to(device)
                           for _ in range(2)]
                    qs = [engine.QueriesGroup(self.T, "sp_po", n, g, flags=self.flags) for _ in range(2)]
                    engine.build_queries_group(self.T, "sp_po", tri[0], n, g, out=qs[0])
                    buf = torch.empty(g, n, 2 * PITCH, device=device)
                    st.append({"tri": tri, "qs": qs, "buf": buf, "out": buf.view(g, n, 2, PITCH)[:, :, :, :E_FB], "cur": 0})
                torch.cuda.synchronize()  # (built on the current stream, used on the lanes' streams)
            return st

        def launch(self, g, lane=0, on_lane=False):
            """One group launch from lane `lane`'s buffers, on that lane's stream (on_lane) or on the current stream."""
            st = self.group(g)[lane]
            c = st["cur"]
            engine.score_queries_group(self.T, st["qs"][c], st["out"], next_batch=st["tri"][1 - c],
                                       next_queries=st["qs"][1 - c],
                                       stream=self.streams[lane].cuda_stream if on_lane and self.streams else None)
            st["cur"] = 1 - c
            self.last = (g, lane)

        def run_steps(self, k):
            if self.streams is None:
                while k > 0:
                    g = min(L, k)
                    self.launch(g)
                    k -= g
                return
            for g in {L, k % L} - {0}:
                self.group(g)
            cur = torch.cuda.current_stream(device)
            ev = torch.cuda.Event()
            ev.record(cur)
            for stm in self.streams:  # fork: the lanes wait for what the current stream has issued
                stm.wait_event(ev)
            i = 0
            while k > 0:
                g = min(L, k)
                self.launch(g, i % self.lanes, on_lane=True)
                i += 1
                k -= g
            for stm in self.streams:  # join: the current stream waits for every lane
                ev = torch.cuda.Event()
                ev.record(stm)
                cur.wait_event(ev)

        def check(self):
            """The last group launched, bit for bit against one launch per batch on the round-3 kernels."""
            g, lane = self.last
            st = self.group(g)[lane]
            c = 1 - st["cur"]  # the queries the last launch scored
            tri = st["tri"][c]
            torch.cuda.synchronize()
            for l in (0, g - 1):
                t = tri[l * n:(l + 1) * n]
                want = engine.score_queries(self.T, engine.build_queries(self.T, "sp_po", t[:, 0], t[:, 1], t[:, 2],
                                                                         flags=self.flags))
                if not torch.equal(st["out"][l].reshape(n, 2 * E_FB), want):
                    return f"batch {l} of the {self.tag} group launch (lane {lane}) differs from the single launch"
            return None

    # `value`: the PARITY-COMPLIANT mode -- split queries (KGE_FLAG_SPLIT_QUERY: q = q_hi + q_lo, f32-level parity on the
    # bf16 tables: ranks equal to float32 arithmetic's up to its own summation noise, tests/test_gpu_bshape_ranks.py).
    # The single-pass mode (query vector rounded to ONE bf16: training tolerance, 4 % of the ranks move against that
    # bar) is timed the same way and reported beside it.
    def measure(flags, tag, lanes):
        md = Mode(flags, tag, lanes)
        md.run_steps(a.warmup)
        md.run_steps(L)
        el, regions, host_el = timed_regions(md.run_steps, torch.cuda.synchronize, a.steps, a.repeats)
        bad = None
        md.run_steps(L * md.lanes)  # one full group per lane, in flight together: every lane's scores are checked
        for lane in range(md.lanes):
            md.last = (L, lane)
            bad = bad or md.check()
        if bad:
            return md, None, bad
        # the dominant kernel: HIP events on the launch stream around back-to-back full-group launches of ONE stream
        # SETTLED rate: >= 40 group launches back to back on one stream, no idle gap -- the same launch runs 25-45 %
        # faster at the start of a burst than sustained (profiles/r4_rocprofv3_kernel_stats.txt, header), and a timed
        # region of `steps` = 20 batches is 3 launches: a burst.  `roofline` uses this number.
        launches = max(40, (a.steps + L - 1) // L)
        k_ms = event_avg_ms(lambda: md.launch(L), launches, min(a.repeats, 3))
        return md, {"el": el, "regions": regions, "host": host_el, "launch_ms": k_ms, "lanes": md.lanes,
                    "settled_launches": launches}, None

    modes, res = {}, {}
    for key, flags, tag in (("parity", engine.FLAG_SPLIT_QUERY, "split-query"), ("training", None, "single-pass")):
        r = why = None
        if a.lanes > 1:
            try:
                md, r, why = measure(flags, tag, a.lanes)
            except Exception as exc:  # lanes are an optimisation of the issue order: never lose the line to them
                why = f"{type(exc).__name__}: {exc}"
                torch.cuda.synchronize()
            if r is None:
                print(f"bench: {tag}: {a.lanes} lanes failed ({why}); one lane", file=sys.stderr)
        if r is None:
            md, r, why = measure(flags, tag, 1)
            if r is None:
                raise SystemExit(f"bench: {why}")
        modes[key], res[key] = md, r
    if os.environ.get("KGE_BENCH_MAIN_ONLY") == "1":  # a quick check of the timed region alone
        total = 2.0 * n * E_FB * a.steps
        print(json.dumps({k: {"value": total / v["el"], "us_per_step": v["el"] / a.steps * 1e6, "lanes": v["lanes"],
                              "launch_us": v["launch_ms"] * 1e3, "host_us_per_step": v["host"] / a.steps * 1e6}
                          for k, v in res.items()}))
        return

    def roofline_of(key, kernel):
        r = res[key]
        ach = L * ab / (r["launch_ms"] * 1e-3) / 1e9
        flops = 2.0 * 2.0 * n * DIM * E_FB * L * (2 if key == "parity" else 1)  # executed on the matrix cores
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "batches_per_launch": L, "algorithmic_bytes_per_launch": L * ab,
                "algorithmic_bytes_per_batch": ab, "avg_launch_us": r["launch_ms"] * 1e3,
                "us_per_batch": r["launch_ms"] * 1e3 / L,
                # the table counted once per LAUNCH instead of once per batch (SURVEY.md 8d counts it per call)
                "frac_table_once_per_launch": (L * ab - (L - 1) * E_FB * DIM * 2) / (r["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "measured_over": f"{r['settled_launches']} back-to-back group launches on one stream (settled; HIP events)",
                # what holds the kernel: the split-query mode executes 2 x the algorithmic 2 n d m flops and is bound by
                # the matrix pipe (profiles/r5_split_store_probe.txt: without its store instructions it is 13 % faster,
                # no more); the single-pass mode by the HBM write path.  `frac` stays the HBM fraction north_star
                # states its target in; the matrix-pipe view beside it: executed flops against the nominal dense peak
                # and against what the bare pipe holds on this box on random operands (roofline_rank.matrix_pipe_probe)
                "limited_by": "matrix pipe (2 x algorithmic flops executed)" if key == "parity" else "HBM write path",
                "mfma_algorithmic_tflops": flops / (2 if key == "parity" else 1) / (r["launch_ms"] * 1e-3) / 1e12,
                "mfma_executed_tflops": flops / (r["launch_ms"] * 1e-3) / 1e12,
                "mfma_executed_frac": flops / (r["launch_ms"] * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
                "score_row_pitch_floats": 2 * PITCH, "block2_offset_floats": PITCH,
                "score_bytes_per_launch": L * n * 2 * E_FB * 4}

    total = 2.0 * n * E_FB * a.steps
    extra = {}
    if not a.no_one_sided:
        # ---- one launch per batch (the round-3 step: kge_score_queries, pairs_bf16_v7_kernel, the next batch's queries
        # built in the same launch), two- and one-sided, and the one-call entry points
        T = modes["training"].T
        q2 = torch.Generator().manual_seed(2)
        batch_b = tuple(torch.randint(hi, (n,), generator=q2).to(device) for hi in (E_FB, R_FB, E_FB))
        batches = [(s, p, o), batch_b]
        tri = [torch.stack(b, 1).contiguous() for b in batches]
        out_pad = torch.empty(n, 2 * PITCH, device=device)
        out_buf = out_pad.view(n, 2, PITCH)[:, :, :E_FB]
        pipe = engine.ScorePipeline(T, "sp_po", n)
        pipe.start(*batches[0])
        step_no = [0]

        def one_step():
            step_no[0] += 1
            pipe.step(next_batch=tri[step_no[0] & 1], out=out_buf)
        for _ in range(5):
            one_step()
        one2_ms = event_avg_ms(one_step, max(a.steps, 40), a.repeats)
        for _ in range(5):
            engine.score_sp_po(T, s, p, o)
        coop_ms = event_avg_ms(lambda: engine.score_sp_po(T, s, p, o), max(a.steps, 40), 3)
        out1 = torch.empty(n, PITCH, device=device)[:, :E_FB]
        pipe1 = engine.ScorePipeline(T, "sp_", n)
        pipe1.start(*batches[0])
        k1 = [0]

        def one_sp():
            k1[0] += 1
            pipe1.step(next_batch=tri[k1[0] & 1], out=out1)
        for _ in range(5):
            one_sp()
        one_ms = event_avg_ms(one_sp, max(a.steps, 40), a.repeats)
        for _ in range(3):
            engine.score_sp(T, s, p)
        one_coop_ms = event_avg_ms(lambda: engine.score_sp(T, s, p), max(a.steps, 40), 3)
        ab1 = algorithmic_bytes(n, E_FB, DIM)
        extra["one_launch_per_batch"] = {
            "kernel": "pairs_bf16_v7_kernel (kge_score_queries: one batch per launch, the next batch's queries built "
                      "inside it) -- single-pass queries, HIP events over back-to-back launches on one stream",
            "two_sided": {"avg_launch_us": one2_ms * 1e3, "frac": ab / (one2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "one_call_entry_us": coop_ms * 1e3,
                          "one_call_entry_frac": ab / (coop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "one_sided": {"avg_launch_us": one_ms * 1e3, "algorithmic_bytes_per_launch": ab1,
                          "frac": ab1 / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "one_call_entry_us": one_coop_ms * 1e3,
                          "one_call_entry_frac": ab1 / (one_coop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        del pipe, pipe1, out_pad, out1
        # ---- the PRODUCT path: what KgeModel.score_sp / score_sp_po of the plugin executes -- engine.score_sp /
        # score_sp_po = ONE call of kge_score_sp / kge_score_sp_po with index vectors, a fresh contiguous [n, E] /
        # [n, 2E] block from torch's allocator (rows NOT sector-aligned: E = 14,541), both query modes.  n >= 1024 runs
        # as groups of 512-row batches of pairs_bf16_v8_kernel behind one query-build launch (api.hip one_call_v8);
        # n = 512: query-build launch + the single-batch kernel.  Settled: HIP events over back-to-back calls.
        one_call = {"entry": "engine.score_sp / engine.score_sp_po (kge_amd._C -> kge_score_sp / kge_score_sp_po), "
                             "contiguous output rows, allocation inside the timed calls"}
        for mode_key, md in modes.items():
            leg = {}
            # (100 / 128 / 256: LibKGE's default train.batch_size / eval.batch_size is 100, config-default.yaml:214,417)
            for nn in (100, 128, 256, 512, 2048, 4096):
                q = torch.Generator().manual_seed(1000 + nn)
                sn, pn, on = (torch.randint(hi, (nn,), generator=q).to(device) for hi in (E_FB, R_FB, E_FB))
                row = {}
                # score_sp_padded: what the plugin's models return without a recorded gradient (`padded_scores`, default
                # true): the [:, :E] view of rows on the pitch engine.score_pitch(E) -- the same C entry with that ldo
                for name, call, sides in (("score_sp", lambda: engine.score_sp(md.T, sn, pn), 1),
                                          ("score_sp_padded", lambda: engine.score_sp(md.T, sn, pn, padded=True), 1),
                                          ("score_sp_po", lambda: engine.score_sp_po(md.T, sn, pn, on), 2)):
                    for _ in range(3):
                        call()
                    ms = event_avg_ms(call, max(10, min(a.steps, 40)), 3)
                    abn = algorithmic_bytes(nn, E_FB, DIM, sides=sides)
                    row[name] = {"us_per_call": ms * 1e3, "frac": abn / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "scored_triples_per_s": sides * nn * E_FB / (ms * 1e-3)}
                leg[str(nn)] = row
                torch.cuda.empty_cache()
            one_call[mode_key] = leg
        extra["one_call_entry"] = one_call
        # ---- groups of one-sided batches (score_sp alone: north_star quotes its target on score_sp)
        g1 = {}
        for key, md in modes.items():
            q = torch.Generator().manual_seed(7)
            tri1 = [torch.stack([torch.randint(hi, (n * L,), generator=q) for hi in (E_FB, R_FB, E_FB)], 1).to(device)
                    for _ in range(2)]
            qs = [engine.QueriesGroup(md.T, "sp_", n, L, flags=md.flags) for _ in range(2)]
            engine.build_queries_group(md.T, "sp_", tri1[0], n, L, out=qs[0])
            b1 = torch.empty(L, n, PITCH, device=device)
            c1 = [0]

            def launch1():
                c = c1[0]
                engine.score_queries_group(md.T, qs[c], b1[:, :, :E_FB], next_batch=tri1[1 - c], next_queries=qs[1 - c])
                c1[0] = 1 - c
            for _ in range(3):
                launch1()
            ms1 = event_avg_ms(launch1, max(3, (max(a.steps, 40) + L - 1) // L), a.repeats)
            g1[key] = {"avg_launch_us": ms1 * 1e3, "us_per_batch": ms1 * 1e3 / L, "batches_per_launch": L,
                       "algorithmic_bytes_per_launch": L * ab1, "frac": L * ab1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "scored_triples_per_s": L * n * E_FB / (ms1 * 1e-3)}
            del b1, qs
        extra["score_sp_groups"] = g1
        by_n = {}
        for nn in (128, 1024, 2048):
            q = torch.Generator().manual_seed(nn)
            trin = torch.stack([torch.randint(hi, (nn * 4,), generator=q) for hi in (E_FB, R_FB, E_FB)], 1).to(device)
            qn = engine.build_queries_group(T, "sp_", trin, nn, 4)
            outn = torch.empty(4, nn, PITCH, device=device)
            for _ in range(3):
                engine.score_queries_group(T, qn, outn[:, :, :E_FB])
            ms = event_avg_ms(lambda: engine.score_queries_group(T, qn, outn[:, :, :E_FB]), max(10, a.steps // 4))
            abn = 4 * algorithmic_bytes(nn, E_FB, DIM)
            by_n[str(nn)] = {"batches_per_launch": 4, "avg_launch_us": ms * 1e3, "frac": abn / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "scored_triples_per_s": 4 * nn * E_FB / (ms * 1e-3)}
            del qn, outn, trin
        extra["score_sp_groups_by_batch"] = by_n
        # d = 256 -- BASELINE configs[4]'s dimension (Wikidata5M ComplEx d = 256).  Until round 6 the persistent store
        # kernel existed for d = 512 only and these groups ran one launch per batch on the round-3 kernels; now
        # pairs_bf16_v8_ce_kernel<128, V3_STORE> (the parametric persistent structure with a score-store epilogue:
        # 16-byte stores, whole sectors per instruction).  Single-pass queries; one-sided groups on the padded pitch.
        by_d = {}
        for tag, Ed, Ld in (("fb15k-237_shape", E_FB, 8), ("wikidata5m_shard", (E_WD + 7) // 8, 2)):
            gd = torch.Generator(device=device).manual_seed(256)
            e256 = (torch.randn(Ed, 256, generator=gd, device=device) * 0.1).bfloat16()
            r256 = (torch.randn(R_FB, 256, generator=gd, device=device) * 0.1).bfloat16()
            T256 = engine.Tables("complex", e256, r256)
            q = torch.Generator().manual_seed(257)
            trid = torch.stack([torch.randint(hi, (n * Ld,), generator=q) for hi in (Ed, R_FB, Ed)], 1).to(device)
            qd = engine.build_queries_group(T256, "sp_", trid, n, Ld)
            pd = engine.score_pitch(Ed)
            outd = torch.empty(Ld, n, pd, device=device)
            for _ in range(3):
                engine.score_queries_group(T256, qd, outd[:, :, :Ed])
            ms = event_avg_ms(lambda: engine.score_queries_group(T256, qd, outd[:, :, :Ed]), max(5, a.steps // 4))
            abd = Ld * algorithmic_bytes(n, Ed, 256)
            by_d[tag] = {"num_entities": Ed, "dim": 256, "batch": n, "batches_per_launch": Ld, "avg_launch_us": ms * 1e3,
                         "us_per_batch": ms * 1e3 / Ld, "algorithmic_bytes_per_launch": abd,
                         "frac": abd / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "scored_triples_per_s": Ld * n * Ed / (ms * 1e-3)}
            # the parity mode of the same groups (split queries: q_hi + q_lo, twice the matrix work per score; the
            # SPLIT instantiation of the same kernel), beside one launch per batch on the single-batch split kernel
            Ts = engine.Tables("complex", e256, r256, flags=engine.FLAG_SPLIT_QUERY)
            qs_ = engine.build_queries_group(Ts, "sp_", trid, n, Ld, flags=engine.FLAG_SPLIT_QUERY)
            for _ in range(3):
                engine.score_queries_group(Ts, qs_, outd[:, :, :Ed])
            ms_s = event_avg_ms(lambda: engine.score_queries_group(Ts, qs_, outd[:, :, :Ed]), max(5, a.steps // 4))
            q1 = engine.build_queries(Ts, "sp_", trid[:n, 0], trid[:n, 1], None, flags=engine.FLAG_SPLIT_QUERY)
            for _ in range(3):
                engine.score_queries(Ts, q1, out=outd[0, :, :Ed])
            ms_1 = event_avg_ms(lambda: engine.score_queries(Ts, q1, out=outd[0, :, :Ed]), max(5, a.steps // 4))
            by_d[tag]["parity"] = {"avg_launch_us": ms_s * 1e3, "us_per_batch": ms_s * 1e3 / Ld,
                                   "frac": abd / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "scored_triples_per_s": Ld * n * Ed / (ms_s * 1e-3),
                                   "one_launch_per_batch_us": ms_1 * 1e3}
            del T256, qd, outd, e256, r256, trid, Ts, qs_, q1
            torch.cuda.empty_cache()
        extra["score_sp_groups_d256"] = by_d
        # float32 tables (the dtype of an unmodified LibKGE config; the reference's own precision):
        # the exact f32 chain on v_mfma_f32_32x32x2_f32 -- MFMA-bound, flops = 2 n d m
        T32 = engine.Tables("complex", ent.float(), rel.float())
        for _ in range(3):
            engine.score_sp(T32, s, p)
        f_ms = event_avg_ms(lambda: engine.score_sp(T32, s, p), max(30, a.steps // 4))
        tf = 2.0 * n * DIM * E_FB / (f_ms * 1e-3) / 1e12
        extra_f32 = {"bound": "mfma", "kernel": "pairs_f32_kernel<ComplEx> (score_sp, float32 tables, exact f32 chain)",
                     "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF,
                     "avg_launch_us": f_ms * 1e3, "flops_per_launch": 2.0 * n * DIM * E_FB,
                     "scored_triples_per_s": n * E_FB / (f_ms * 1e-3)}
        del T32
        # The distance scorers on the same score_sp call (float32 tables; no matrix-core form: VALU-bound).  Vector
        # operations per scored coordinate: TransE (l_norm 1) ONE subtract + ONE add with the |x| source modifier = 2 per
        # REAL coordinate (round 6: pairs_transe_kernel issues exactly those, the subtracts of two rows packed; until
        # round 5 this leg counted 3 and the kernel spent ~2.7 issue slots); RotatE per
        # COMPLEX coordinate 4 for the rotation (2 mul + 2 fma), 2 sub, 2 for re^2 + im^2, 11 issue slots for the
        # correctly rounded sqrt (common.hpp sqrt_rn_core: the quarter-rate v_rsq_f32 + 7), 2 for the range check's
        # min / max, 1 add: ~22 (with the compiler's IEEE sqrt sequence it was ~30).  Peak: 256 CUs x
        # 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-operations/s.
        VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
        exact = {"bound": "valu", "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s",
                 "kernel": "pairs_transe_kernel (TransE, l_norm 1) / pairs_kernel<RotatE> (kge_score_sp, float32 tables: the "
                           "reference's arithmetic chain)"}
        for name, rdim, ops in (("transe", DIM, 2.0 * DIM), ("rotate", DIM // 2, 22.0 * (DIM // 2))):
            gq = torch.Generator().manual_seed(11)
            Tx = engine.Tables(name, ent.float(), torch.empty(R_FB, rdim).normal_(0, 0.1, generator=gq).to(device),
                               l_norm=1.0)
            for _ in range(3):
                engine.score_sp(Tx, s, p)
            x_ms = event_avg_ms(lambda: engine.score_sp(Tx, s, p), max(10, a.steps // 8))
            tops = ops * n * E_FB / (x_ms * 1e-3) / 1e12
            exact[name] = {"avg_launch_us": x_ms * 1e3, "scored_triples_per_s": n * E_FB / (x_ms * 1e-3),
                           "lane_ops_per_score": ops, "achieved": tops, "frac": tops / VALU_PEAK_TOPS}
            del Tx
        extra["roofline_exact"] = exact
        for md in modes.values():
            md.by_size.clear()
        torch.cuda.empty_cache()
        extra_rank = rank_legs(engine, device, n, max(25, min(a.steps, 400) // 4))
        extra_neg = neg_legs(engine, device, max(20, min(a.steps, 400) // 8))
        extra_eval = eval_leg(engine, device)
        extra_train = train_leg(device, n, max(10, min(a.steps, 200) // 4))
        for key, fn in (("negative_sampling_step", ns_step_leg), ("kvsall_step", kvsall_step_leg)):
            try:
                extra_train[key] = fn(device, n, max(10, min(a.steps, 200) // 4))
            except Exception as exc:  # (a secondary leg never costs the line)
                extra_train[key] = {"error": f"{type(exc).__name__}: {exc}"}
    else:
        extra_f32 = extra_rank = extra_neg = extra_eval = extra_train = None

    rp = res["parity"]
    rt = res["training"]
    # roofline.traffic, measured by this run where it can be: two rocprofv3 --pmc child passes (pmc_traffic_live).  Not
    # when this process is itself being profiled (its own kernel trace would then hold the children's launches too).
    profiled = any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)
    live = {} if (a.no_pmc or profiled) else pmc_traffic_live(L)
    tr_par, tr_trn = traffic_of("parity", L, live), traffic_of("training", L, live)
    out = {
        "metric": "scored triples/sec (1vsAll, ComplEx d=512)",
        "value": total / rp["el"],
        "unit": "scored triples/s",
        # `value` / `ms_per_step` come from the contract's timed region: K = `steps` batches = ceil(K / group) launches
        # behind an idle gap -- a BURST (the first launches of a burst run 25-45 % faster than sustained issue).
        # `value_settled` is the same step at the settled rate: the average of >= 40 back-to-back group launches.
        "value_kind": (f"settled: {a.steps} steps = {(a.steps + L - 1) // L} back-to-back group launches per timed region"
                       if (a.steps + L - 1) // L >= 40 else
                       f"burst: {a.steps} steps = {(a.steps + L - 1) // L} group launch(es) per timed region"),
        "value_settled": 2.0 * n * E_FB * L / (rp["launch_ms"] * 1e-3),
        "ms_per_step_settled": rp["launch_ms"] / L,
        "value_vs_reference_ranks": "ranks are bit-exact against the oracle (float32 arithmetic on the bf16 tables in the "
                                    "kernels' summation order); against the live reference (torch's summation order) <= 12 "
                                    "of 12,000 ranks differ by one position for ComplEx / DistMult at this shape "
                                    "(tests/test_oracle_golden.py:131-190, tests/test_gpu_bshape_ranks.py)",
        "value_mode": "parity-compliant: split queries (q = q_hi + q_lo on the matrix cores; ranks equal to float32 "
                      "arithmetic on the bf16 tables up to its summation noise).  The single-pass mode is "
                      "`training_tolerance` below",
        "n_gpus": 1,
        "steps": a.steps,
        "warmup": a.warmup,
        "repeats": a.repeats,
        "ms_per_step": rp["el"] / a.steps * 1e3,
        "regions_ms_per_step": [r / a.steps * 1e3 for r in rp["regions"]],
        "host_issue_ms_per_step": rp["host"] / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": "FB15k-237 shape ComplEx d=512 1vsAll scoring, bf16 tables, f32 scores: the score_sp and "
                        "score_po blocks of a batch per step (KgeModel.score_sp_po), batches issued in groups of "
                        "`group` -- one persistent launch per group (kge_score_queries_multi), which also builds the "
                        "next group's query vectors; group k on HIP stream k % `group_launches_in_flight`; score rows "
                        "on a 256-byte pitch",
            "num_entities_per_gpu": E_FB, "num_relations": R_FB, "dim": DIM, "batch": n, "group": L,
            "group_launches_in_flight": rp["lanes"],
            "parallelism": "single GPU", "queries": "split (q_hi + q_lo)",
        },
        "roofline": {**roofline_of("parity", "pairs_bf16_v8_kernel<ComplEx, SPLIT> (kge_score_queries_multi: one "
                                             "persistent launch = `group` two-sided batches, split queries: twice the "
                                             "matrix-core work per score)"),
                     "traffic": tr_par[0],
                     # what in this object is measured in THIS run and what is a committed constant
                     "traffic_source": tr_par[1], "traffic_detail": tr_par[2],
                     "achieved_source": "live: HIP events around back-to-back group launches in this run",
                     "timed_region": {"us_per_step": rp["el"] / a.steps * 1e6,
                                      "frac": ab / (rp["el"] / a.steps) / 1e9 / HBM_PEAK_GBS}},
        # the same step with the query vector rounded to ONE bf16 (what 1vsAll TRAINING needs; 4 % of the ranks of an
        # evaluation would move against float32 arithmetic on the same tables)
        "training_tolerance": {
            "value": total / rt["el"], "unit": "scored triples/s", "ms_per_step": rt["el"] / a.steps * 1e3,
            "value_kind": "as value_kind above", "value_settled": 2.0 * n * E_FB * L / (rt["launch_ms"] * 1e-3),
            "ms_per_step_settled": rt["launch_ms"] / L,
            "regions_ms_per_step": [r / a.steps * 1e3 for r in rt["regions"]],
            "host_issue_ms_per_step": rt["host"] / a.steps * 1e3, "group_launches_in_flight": rt["lanes"],
            "roofline": {**roofline_of("training", "pairs_bf16_v8_kernel<ComplEx> (kge_score_queries_multi: one "
                                                   "persistent launch = `group` two-sided batches, single-pass queries)"),
                         "traffic": tr_trn[0], "traffic_source": tr_trn[1], "traffic_detail": tr_trn[2],
                         "timed_region": {"us_per_step": rt["el"] / a.steps * 1e6,
                                          "frac": ab / (rt["el"] / a.steps) / 1e9 / HBM_PEAK_GBS}}},
        **extra,
    }
    if extra_f32 is not None:
        out["roofline_f32"] = extra_f32
    if extra_rank is not None:
        out["roofline_rank"] = extra_rank
        # the matrix-pipe view of the headline kernel against what the BARE pipe holds on this box on random operands
        # (measured in this run: roofline_rank.matrix_pipe_probe) -- the denominator VERDICT r5 asked to see in the line
        probe_tf = extra_rank["matrix_pipe_probe"]["achieved"]
        for ro in (out["roofline"], out["training_tolerance"]["roofline"]):
            ro["mfma_executed_frac_of_probe"] = ro["mfma_executed_tflops"] / probe_tf
            ro["mfma_probe_tflops"] = probe_tf
    if extra_neg is not None:
        out["roofline_neg"] = extra_neg
    if extra_eval is not None:
        out["roofline_eval"] = extra_eval
    if extra_train is not None:
        out["roofline_train"] = extra_train
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, a.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
