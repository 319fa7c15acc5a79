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
