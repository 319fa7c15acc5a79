"""Host-side logic that needs no GPU: the CSR filter index (replacement of the reference's
KvsAllIndex + label coordinates), metrics, tie policy, synthetic dataset writer."""
import json
import os

import numpy as np
import pytest
import torch

import oracle as ko
from conftest import GOLDEN
from kge_amd.eval import FilterIndex, compute_metrics, get_ranks
from kge_amd.synthetic import make_splits, write_libkge_dataset


def test_filter_index_equals_dict_index():
    E, R = 50, 4
    splits = make_splits(E, R, 600, 80, 80, seed=3)
    fi = FilterIndex([splits["train"], splits["valid"]], E, R)
    ix_sp = [ko.build_index(splits[s], (0, 1), 2) for s in ("train", "valid")]
    ix_po = [ko.build_index(splits[s], (1, 2), 0) for s in ("train", "valid")]
    batch = np.concatenate([splits["valid"][:40], [[E - 1, R - 1, E - 1], [0, 0, 0]]])
    sp_rp, sp_col, po_rp, po_col = fi.labels(batch)
    w_rp, w_col = ko.labels_csr(batch[:, [0, 1]], ix_sp)
    assert np.array_equal(sp_rp, w_rp) and np.array_equal(sp_col, w_col)
    w_rp, w_col = ko.labels_csr(batch[:, [1, 2]], ix_po)
    assert np.array_equal(po_rp, w_rp) and np.array_equal(po_col, w_col)


def test_filter_index_empty_hits():
    fi = FilterIndex([np.array([[1, 0, 2]])], 5, 2)
    rp, col, rp2, col2 = fi.labels(np.array([[3, 1, 4], [1, 0, 2]]))
    assert rp.tolist() == [0, 0, 1] and col.tolist() == [2]
    assert rp2.tolist() == [0, 0, 1] and col2.tolist() == [1]


@pytest.mark.parametrize("model", ["complex", "transe"])
def test_metrics_from_golden_ranks(model):
    """compute_metrics / hist arithmetic reproduce the reference's final metrics from the
    reference's own per-example ranks."""
    g = np.load(os.path.join(GOLDEN, f"eval_{model}.npz"))
    E = int(g["num_entities"])
    ref = json.loads(str(g["metrics_full"]))
    for key, suf in (("", ""), ("_filt", "_filtered"), ("_filt_test", "_filtered_with_test")):
        hist = torch.zeros(E)
        for r in (g[f"o_rank{key}_full"], g[f"s_rank{key}_full"]):
            u, cnt = torch.unique(torch.from_numpy(r), return_counts=True)
            hist.index_add_(0, u, cnt.float())
        m = compute_metrics(hist, [1, 3, 10, 50], suf)
        for name in ("mean_reciprocal_rank", "mean_rank", "hits_at_1", "hits_at_10"):
            assert abs(m[name + suf] - ref[name + suf]) < 1e-6, (name, key)


def test_tie_policies():
    r, t = torch.tensor([3, 0]), torch.tensor([4, 1])
    assert get_ranks(r, t, "rounded_mean_rank").tolist() == [5, 0]
    assert get_ranks(r, t, "best_rank").tolist() == [3, 0]
    assert get_ranks(r, t, "worst_rank").tolist() == [6, 0]
    with pytest.raises(NotImplementedError):
        get_ranks(r, t, "x")


def test_synthetic_dataset_roundtrip(tmp_path):
    splits = make_splits(30, 3, 100, 10, 10, seed=1)
    folder = write_libkge_dataset(str(tmp_path / "ds"), "ds", 30, 3, splits)
    back = np.loadtxt(os.path.join(folder, "train.del"), dtype=np.int64, delimiter="\t")
    assert np.array_equal(back, splits["train"])
    y = open(os.path.join(folder, "dataset.yaml")).read()
    assert "num_entities: 30" in y and "files.valid.size: 10" in y


# ---- bench.py --gpus N launches N ranks itself (`a.gpus` was parsed and never read) ----------------------
def _bench(args, env_extra, timeout=240):
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([_sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus 2` with no launcher re-executes itself under torch.distributed.run: two ranks reach the
    device selection, each seeing WORLD_SIZE = 2 (KGE_BENCH_DEVICE_CHECK=1 stops there: no GPU here)."""
    import json
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"KGE_BENCH_DEVICE_CHECK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert sorted(x["rank"] for x in lines) == [0, 1]
    assert all(x["world_size"] == 2 and x["gpus"] == 2 for x in lines)
    assert sorted(x["local_rank"] for x in lines) == [0, 1]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Launched with WORLD_SIZE = 1 but asked for --gpus 2 (or the other way round): no line, a non-zero exit."""
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"],
               {"KGE_BENCH_DEVICE_CHECK": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1"],
               {"KGE_BENCH_DEVICE_CHECK": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_traffic_falls_back_to_the_committed_counter_passes():
    """`roofline.traffic`: bench.py spawns two rocprofv3 --pmc passes itself (pmc_traffic_live: round 6) and says so in
    `traffic_source`; where it cannot (no rocprofv3, --no-pmc, a failed pass) the figure is read from
    profiles/pmc_latest.json, the summary of the committed passes over the SAME group launch -- per query mode, only
    for the group size the passes were taken at, and labelled COMMITTED."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = json.load(open(os.path.join(root, "profiles", "pmc_latest.json")))
    g = d["group"]
    alg = g * bench.algorithmic_bytes(512, bench.E_FB, bench.DIM, sides=2)
    assert d["algorithmic_bytes_per_launch"] == alg
    for mode in ("parity", "training"):
        t = bench.pmc_traffic(mode, g)
        assert t == d[mode]["hbm_bytes_per_launch"] and 0.5 * alg < t < 1.5 * alg
        assert abs(d[mode]["fetch_bytes_corrected"] + d[mode]["write_bytes"] - t) < 1.0
        assert bench.pmc_traffic(mode, g + 1) is None     # another group size: no figure rather than a wrong one
    assert os.path.exists(os.path.join(root, d["source"]))
    # which of the two a line reports, it says
    live = {"group": g, "seconds": 30.0, "parity": {"hbm_bytes_per_launch": 6.6e8, "fetch_bytes_corrected": 1.7e8,
                                                     "write_bytes": 4.9e8, "dispatches": 12}}
    t, src, detail = bench.traffic_of("parity", g, live)
    assert t == 6.6e8 and src.startswith("live:") and detail["write_bytes"] == 4.9e8
    t, src, detail = bench.traffic_of("training", g, live)        # a mode the live passes did not deliver
    assert t == d["training"]["hbm_bytes_per_launch"] and "COMMITTED" in src and detail is None
    t, src, _ = bench.traffic_of("parity", g, {})
    assert t == d["parity"]["hbm_bytes_per_launch"] and "COMMITTED" in src


def test_graphed_step_gate_refuses_step_count_dependent_optimizers():
    """GraphedStep may capture only optimizers whose step() launch arguments do not depend on the
    step count.  kge_amd.optim.Adagrad declares itself capturable, kge_amd.optim.Adam (host-side bias correction) does
    not, of torch's own optimizers only SGD is admitted.  (Decided in the constructor: no GPU needed.)"""
    import torch
    from kge_amd import optim as kopt
    from kge_amd.train_graph import GraphedStep
    w = [torch.nn.Parameter(torch.zeros(4, 4))]
    f = lambda *a: None
    assert GraphedStep(f, kopt.Adagrad(w, lr=0.1)).enabled
    assert GraphedStep(f, torch.optim.SGD(w, lr=0.1)).enabled
    for opt in (kopt.Adam(w, lr=0.1), torch.optim.Adam(w, lr=0.1), torch.optim.Adagrad(w, lr=0.1)):
        st = GraphedStep(f, opt)
        assert not st.enabled and "step count" in st.disabled_reason, type(opt)
    st = GraphedStep(f, kopt.Adagrad(w, lr=0.1, lr_decay=0.1))
    assert not st.enabled and "lr_decay" in st.disabled_reason
