"""Golden EntityRanking ranks at the WN18RR SHAPE (E=40,943, R=11, d=512; RotatE relations d/2 = 256:
rotate.py:88-93) from the LIVE reference, for all four scorers (SURVEY.md 8c gate 2: ranks exactly equal
"at dataset_test, B and W shapes"; the B shape is make_golden_bshape.py).

Run inside the build container only (needs /root/reference; ~15 minutes of CPU, most of it RotatE's
[n, E, d/2] temporaries):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_wshape.py [model ...]

As at the B shape, tables and the Zipf train split are regenerated from seeds (wshape_tables,
wshape_base_train), the fixture holds the planted triples and what the reference computed:

  wshape_<model>.npz   valid / test / planted train triples; per-example ranks (raw, filtered,
                       filtered_with_test; both directions) and the final metrics of the reference's
                       EntityRankingJob._evaluate on CPU with float32 tables ("f32"); for ComplEx and
                       DistMult also on the bf16-rounded tables in float32 arithmetic ("bf16t").

Evaluation triples: 1,000 for DistMult / ComplEx, 200 for TransE / RotatE (the reference's RotatE sp_
scoring materialises n x E x 256 complex temporaries: 25 triples per batch is what fits comfortably).
Planting as at the B shape: the object of a random (s, p) is one of its 10 best-scoring objects, four
more of them become train / test triples of the same (s, p), so that the filters change the ranks.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import torch  # noqa: E402

E, R, D = 40943, 11, 512
N_TRAIN = 86835
MODELS = ("distmult", "complex", "transe", "rotate")
N_VALID = {"distmult": 1000, "complex": 1000, "transe": 200, "rotate": 200}
EVAL_BATCH = {"distmult": 100, "complex": 100, "transe": 50, "rotate": 25}
SEED = {"distmult": 303, "complex": 404, "transe": 505, "rotate": 606}


def wshape_tables(model: str):
    """The float32 tables of the case (CPU generator: the same bits on every box).  TransE / RotatE
    entities at a scale where distances separate; RotatE phases uniform in (-pi, pi) (rotate.yaml:22-26)."""
    g = torch.Generator().manual_seed(SEED[model])
    if model in ("distmult", "complex"):
        return torch.randn(E, D, generator=g) * 0.35, torch.randn(R, D, generator=g) * 0.35
    ent = torch.randn(E, D, generator=g) * 0.5
    if model == "transe":
        return ent, torch.randn(R, D, generator=g) * 0.5
    return ent, (torch.rand(R, D // 2, generator=g) * 2.0 - 1.0) * 3.141592653589793


def wshape_base_train():
    from kge_amd.synthetic import make_splits
    return make_splits(E, R, N_TRAIN, 0, 0, seed=78)["train"]


def wshape_splits(fixture):
    train = np.concatenate([wshape_base_train(), fixture["planted_train"].astype(np.int32)])
    return {"train": train, "valid": fixture["valid"].astype(np.int32), "test": fixture["test"].astype(np.int32)}


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _plant(model, ent, rel, rng):
    import torch_port as tp
    n = N_VALID[model]
    s = torch.from_numpy(rng.integers(0, E, n))
    p = torch.from_numpy(rng.integers(0, R, n))
    tops = []
    with torch.no_grad():
        for b0 in range(0, n, EVAL_BATCH[model]):
            sl = slice(b0, b0 + EVAL_BATCH[model])
            tops.append(tp.score_sp(model, ent, rel, s[sl], p[sl]).topk(10, dim=1).indices.numpy())
    top = np.concatenate(tops)
    valid, train, test = [], [], []
    for i in range(n):
        pick = rng.permutation(10)
        valid.append((int(s[i]), int(p[i]), int(top[i, pick[0]])))
        for j in pick[1:4]:
            train.append((int(s[i]), int(p[i]), int(top[i, j])))
        test.append((int(s[i]), int(p[i]), int(top[i, pick[4]])))
    return (np.asarray(valid, np.int32), np.asarray(train, np.int32), np.asarray(test, np.int32))


def _reference_eval(folder, model, ent, rel):
    import ref_harness as rh
    rh.import_reference()
    from kge import Config, Dataset
    from kge.job import EvaluationJob
    from kge.model import KgeModel
    config = Config()
    config.folder = tempfile.mkdtemp(prefix="kge_wshape_out_")
    config.set("console.quiet", True)
    config.set("model", model)
    config._import(model)
    config.set("dataset.name", "wshape")
    config.set("job.device", "cpu")
    config.set("job.type", "eval")
    config.set_all({"lookup_embedder.dim": D})
    config.set("eval.split", "valid")
    config.set("eval.batch_size", EVAL_BATCH[model])
    config.set("eval.trace_level", "example")
    config.init_folder()
    dataset = Dataset.create(config, folder=folder)
    m = KgeModel.create(config, dataset)
    rh.set_tables(m, ent, rel)
    job = EvaluationJob.create(config, dataset, parent_job=None, model=m)
    examples = []
    orig = job.trace

    def capture(**kw):
        if kw.get("event") == "example_rank":
            examples.append(dict(kw))
        return orig(**kw)

    job.trace = capture
    result = job.run()
    sp = [e for e in examples if e["task"] == "sp"]
    po = [e for e in examples if e["task"] == "po"]
    out = dict(
        o_rank=np.array([e["rank"] for e in sp]) - 1,
        o_rank_filt=np.array([e["rank_filtered"] for e in sp]) - 1,
        o_rank_filt_test=np.array([e["rank_filtered_with_test"] for e in sp]) - 1,
        s_rank=np.array([e["rank"] for e in po]) - 1,
        s_rank_filt=np.array([e["rank_filtered"] for e in po]) - 1,
        s_rank_filt_test=np.array([e["rank_filtered_with_test"] for e in po]) - 1)
    metrics = {k: float(v) for k, v in result.items()
               if isinstance(v, (int, float)) and (k.startswith("mean_") or k.startswith("hits_at_"))}
    shutil.rmtree(config.folder, ignore_errors=True)
    return out, metrics, np.array([[e["s"], e["p"], e["o"]] for e in sp])


def main(models):
    import time
    from kge_amd.synthetic import write_libkge_dataset
    for model in models:
        t0 = time.time()
        ent, rel = wshape_tables(model)
        rng = np.random.default_rng(SEED[model] + 1)
        valid, planted, test = _plant(model, ent, rel, rng)
        fixture = {"valid": valid, "planted_train": planted, "test": test}
        splits = wshape_splits(fixture)
        tmp = tempfile.mkdtemp(prefix="kge_wshape_ds_")
        res = {}
        try:
            folder = write_libkge_dataset(os.path.join(tmp, "wshape"), "wshape", E, R, splits)
            cases = [("f32", (ent, rel))]
            if model in ("distmult", "complex"):
                cases.append(("bf16t", (bf16_round(ent), bf16_round(rel))))
            for tag, (e_, r_) in cases:
                ranks, metrics, triples = _reference_eval(folder, model, e_, r_)
                assert np.array_equal(triples, valid)
                res[tag] = (ranks, metrics)
                print(model, tag, "MRR filt/test", metrics["mean_reciprocal_rank_filtered"],
                      metrics["mean_reciprocal_rank_filtered_with_test"], "raw", metrics["mean_reciprocal_rank"],
                      f"({time.time() - t0:.0f} s)", flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        arrays = {}
        for tag, (ranks, metrics) in res.items():
            arrays[f"metrics_{tag}"] = json.dumps(metrics, sort_keys=True)
            arrays.update({f"{k}_{tag}": v.astype(np.int32) for k, v in ranks.items()})
        np.savez_compressed(os.path.join(HERE, f"wshape_{model}.npz"), model=model, valid=valid,
                            planted_train=planted, test=test, **arrays)


if __name__ == "__main__":
    main(sys.argv[1:] or MODELS)
