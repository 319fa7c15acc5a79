"""Generate the golden vectors under tests/golden/ from the LIVE reference.

Run inside the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's own tests hold no golden vectors for this path (SURVEY.md 8c:
only tolerance-level self-consistency), so parity is pinned by running the
reference's *own code* -- KgeModel.score_spo/score_sp/score_po/score_sp_po
(kge/model/kge_model.py:663-789) and EntityRankingJob (kge/job/
eval_entity_ranking.py) -- on seeded inputs and committing inputs + outputs:

  scores_<case>.npz   tables, index vectors, reference score outputs
  rankcore.npz        crafted score matrices (ties, NaN, +-inf) and the output of
                      EntityRankingJob._get_ranks_and_num_ties / _filter_and_rank
  eval_<model>.npz    a synthetic LibKGE dataset, model tables, per-example ranks
                      (raw / filtered / filtered_with_test) and final metrics of
                      EntityRankingJob._evaluate, with and without chunking
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
sys.dont_write_bytecode = True

import ref_harness as rh  # noqa: E402

rh.import_reference()
import torch  # noqa: E402

from kge_amd.synthetic import make_splits, write_libkge_dataset  # noqa: E402

SCORE_CASES = [
    # name, model, options, E, R, d, n
    ("complex_d32", "complex", {}, 37, 5, 32, 11),
    ("distmult_d32", "distmult", {}, 37, 5, 32, 11),
    ("distmult_d33", "distmult", {}, 29, 4, 33, 9),
    ("transe_l1_d32", "transe", {"transe.l_norm": 1.0}, 37, 5, 32, 11),
    ("transe_l2_d32", "transe", {"transe.l_norm": 2.0}, 37, 5, 32, 11),
    ("transe_l1_d33", "transe", {"transe.l_norm": 1.0}, 29, 4, 33, 9),
    ("rotate_l1_d32", "rotate", {"rotate.l_norm": 1.0}, 37, 5, 32, 11),
    ("rotate_l2_d32", "rotate", {"rotate.l_norm": 2.0}, 37, 5, 32, 11),
    ("complex_d128", "complex", {}, 300, 7, 128, 16),
    ("rotate_l1_d128", "rotate", {"rotate.l_norm": 1.0}, 300, 7, 128, 16),
    ("distmult_d512", "distmult", {}, 130, 7, 512, 8),
    ("complex_d512", "complex", {}, 130, 7, 512, 8),
]


def gen_scores():
    for name, model, opts, E, R, d, n in SCORE_CASES:
        torch.manual_seed(1234)
        m = rh.make_model(model, E, R, d, options=opts)
        ent, rel = rh.get_tables(m)
        g = torch.Generator().manual_seed(99)
        s = torch.randint(E, (n,), generator=g)
        p = torch.randint(R, (n,), generator=g)
        o = torch.randint(E, (n,), generator=g)
        sub = torch.randperm(E, generator=g)[: max(3, E // 3)]
        K = 6
        neg = torch.randint(E, (n, K), generator=g)
        with torch.no_grad():
            out = {
                "spo": m.score_spo(s, p, o, "o"),
                "sp": m.score_sp(s, p),
                "po": m.score_po(p, o),
                "sp_sub": m.score_sp(s, p, sub),
                "po_sub": m.score_po(p, o, sub),
                "sp_po_sub": m.score_sp_po(s, p, o, sub),
                "sp_po_all": m.score_sp_po(s, p, o, None),
            }
            # negative sampling, implementation "triple" (kge/util/sampler.py:291-306):
            # repeat the positives K times, overwrite the slot, score_spo.
            for slot, key in ((0, "neg_s"), (2, "neg_o")):
                tr = torch.stack([s, p, o], dim=1).repeat(1, K).view(-1, 3).clone()
                tr[:, slot] = neg.contiguous().view(-1)
                out[key] = m.score_spo(tr[:, 0], tr[:, 1], tr[:, 2]).view(n, K)
        l_norm = float(opts.get(f"{model}.l_norm", 1.0))
        np.savez_compressed(
            os.path.join(HERE, f"scores_{name}.npz"),
            model=model, l_norm=np.float32(l_norm), ent=ent.numpy(), rel=rel.numpy(),
            s=s.numpy(), p=p.numpy(), o=o.numpy(), sub=sub.numpy(), neg=neg.numpy(),
            **{k: v.numpy() for k, v in out.items()})
        print("scores", name, {k: tuple(v.shape) for k, v in out.items()})


def _make_job(folder, model, dim, extra=None, E=None):
    from kge import Config, Dataset
    from kge.job import EvaluationJob
    from kge.model import KgeModel

    config = Config()
    config.folder = tempfile.mkdtemp(prefix="kge_golden_out_")
    config.set("console.quiet", True)
    config.set("model", model)
    config._import(model)
    config.set("dataset.name", "synthetic")
    config.set("job.device", "cpu")
    config.set("job.type", "eval")
    config.set_all({"lookup_embedder.dim": dim})
    config.set("eval.split", "valid")
    config.set("eval.batch_size", 16)
    config.set("eval.trace_level", "example")
    if extra:
        config.set_all(extra)
    config.init_folder()
    dataset = Dataset.create(config, folder=folder)
    torch.manual_seed(4321)
    m = KgeModel.create(config, dataset)
    job = EvaluationJob.create(config, dataset, parent_job=None, model=m)
    return config, dataset, m, job


def gen_rankcore(folder):
    config, dataset, m, job = _make_job(folder, "distmult", 8)
    rng = np.random.default_rng(7)
    n, c = 9, 40
    sc = rng.standard_normal((n, c)).astype(np.float32)
    true = sc[np.arange(n), rng.integers(0, c, n)].copy()
    # crafted rows: exact ties, values straddling the isclose boundary, NaN, +-inf
    sc[0, :5] = true[0]
    sc[1, 3] = true[1] + np.float32(1e-5)
    sc[1, 4] = true[1] + np.float32(1.2e-5) + np.float32(1e-4) * abs(true[1])
    sc[1, 5] = true[1] - np.float32(0.9e-5)
    sc[2, 7] = np.nan
    sc[2, 8] = np.inf
    sc[2, 9] = -np.inf
    true[3] = np.nan
    sc[3, 2] = np.nan
    true[4] = np.inf
    sc[4, 1] = np.inf
    true[5] = -np.inf
    sc[5, 6] = -np.inf
    sc[6, :] = 0.0
    true[6] = 0.0
    sc[7, :] = np.float32(1000.0) + rng.integers(-3, 4, c).astype(np.float32) * np.float32(0.05)
    true[7] = np.float32(1000.0)
    labels = np.zeros((n, 2 * c), dtype=np.float32)
    lab_mask = rng.random((n, 2 * c)) < 0.15
    labels[lab_mask] = np.inf
    with torch.no_grad():
        rk, ti = job._get_ranks_and_num_ties(torch.from_numpy(sc), torch.from_numpy(true))
        sc_po = rng.standard_normal((n, c)).astype(np.float32)
        true_po = sc_po[np.arange(n), rng.integers(0, c, n)].copy()
        s_rank, s_ties, o_rank, o_ties, sp_f, po_f = job._filter_and_rank(
            torch.from_numpy(sc), torch.from_numpy(sc_po), torch.from_numpy(labels),
            torch.from_numpy(true), torch.from_numpy(true_po))
    np.savez_compressed(
        os.path.join(HERE, "rankcore.npz"), scores=sc, true=true, rank=rk.numpy(),
        ties=ti.numpy(), scores_po=sc_po, true_po=true_po, labels=labels,
        filt_s_rank=s_rank.numpy(), filt_s_ties=s_ties.numpy(), filt_o_rank=o_rank.numpy(),
        filt_o_ties=o_ties.numpy(), atol=np.float32(job.tie_atol), rtol=np.float32(job.tie_rtol))
    print("rankcore", rk.tolist(), ti.tolist())
    shutil.rmtree(config.folder, ignore_errors=True)


def gen_eval(folder, splits, E, R):
    for model, dim, extra in [
        ("complex", 16, {}),
        ("distmult", 16, {}),
        ("transe", 16, {}),
        ("rotate", 16, {}),
    ]:
        res = {}
        for tag, chunk in (("full", -1), ("chunk17", 17)):
            ex = dict(extra)
            ex["entity_ranking.chunk_size"] = chunk
            config, dataset, m, job = _make_job(folder, model, dim, ex)
            ent, rel = rh.get_tables(m)
            examples = []
            orig_trace = job.trace

            def capture(**kw):
                if kw.get("event") == "example_rank":
                    examples.append(dict(kw))
                return orig_trace(**kw)

            job.trace = capture
            result = job.run()
            sp = [e for e in examples if e["task"] == "sp"]
            po = [e for e in examples if e["task"] == "po"]
            assert len(sp) == len(splits["valid"]) == len(po)
            res[tag] = dict(
                o_rank=np.array([e["rank"] for e in sp]) - 1,
                o_rank_filt=np.array([e["rank_filtered"] for e in sp]) - 1,
                o_rank_filt_test=np.array([e["rank_filtered_with_test"] for e in sp]) - 1,
                s_rank=np.array([e["rank"] for e in po]) - 1,
                s_rank_filt=np.array([e["rank_filtered"] for e in po]) - 1,
                s_rank_filt_test=np.array([e["rank_filtered_with_test"] for e in po]) - 1,
                triples=np.array([[e["s"], e["p"], e["o"]] for e in sp]),
            )
            metrics = {k: float(v) for k, v in result.items()
                       if isinstance(v, (int, float)) and
                       (k.startswith("mean_") or k.startswith("hits_at_"))}
            res[tag]["metrics"] = json.dumps(metrics, sort_keys=True)
            shutil.rmtree(config.folder, ignore_errors=True)
        assert np.array_equal(res["full"]["triples"], splits["valid"])
        for k in ("o_rank", "o_rank_filt", "o_rank_filt_test", "s_rank", "s_rank_filt", "s_rank_filt_test"):
            if not np.array_equal(res["full"][k], res["chunk17"][k]):
                print("NOTE: reference ranks differ between chunked/unchunked for", model, k)
        l_norm = 1.0
        np.savez_compressed(
            os.path.join(HERE, f"eval_{model}.npz"), model=model, l_norm=np.float32(l_norm),
            num_entities=E, num_relations=R, ent=ent.numpy(), rel=rel.numpy(),
            train=splits["train"], valid=splits["valid"], test=splits["test"],
            metrics_full=res["full"]["metrics"], metrics_chunk17=res["chunk17"]["metrics"],
            **{f"{k}_full": v for k, v in res["full"].items() if k not in ("metrics", "triples")},
            **{f"{k}_chunk17": v for k, v in res["chunk17"].items() if k not in ("metrics", "triples")})
        print("eval", model, json.loads(res["full"]["metrics"])["mean_reciprocal_rank_filtered_with_test"])


def main():
    gen_scores()
    E, R = 60, 4
    splits = make_splits(E, R, 400, 50, 50, seed=5)
    tmp = tempfile.mkdtemp(prefix="kge_golden_ds_")
    folder = write_libkge_dataset(os.path.join(tmp, "synthetic"), "synthetic", E, R, splits)
    try:
        gen_rankcore(folder)
        gen_eval(folder, splits, E, R)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
