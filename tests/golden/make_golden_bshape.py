"""Golden EntityRanking ranks at the FB15k-237 SHAPE (E=14,541, R=237, d=512) from the LIVE reference.

Run inside the build container only (needs /root/reference; ~3 minutes of CPU):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bshape.py

BASELINE.json's config 4 asks for filtered-MRR parity at this shape; the eval_<model>.npz fixtures of
make_golden.py pin it at E=60 only (missing #4).  The tables are far too large to commit
(2 x 29.8 MB), so everything big is REGENERATED from seeds by `bshape_case()` below -- on the GPU box
too -- and only what the reference computed is stored:

  bshape_<model>.npz   valid / test triples and the planted train triples (see below), the reference's
                       per-example ranks (raw / filtered / filtered_with_test, both directions) and
                       final metrics for
                         f32   : float32 tables (the reference's own precision),
                         bf16t : the same tables rounded to bfloat16, float32 arithmetic
                                 (SURVEY.md 8c gate 4: "the fp32 reference run on bf16-rounded tables"),
                       both produced by the reference's EntityRankingJob._evaluate on CPU, and
                         bf16q : ranks of bf16-OPERAND scoring -- tables AND the query vector
                                 q = s (x) r rounded to bfloat16, float32 accumulation: what a bf16
                                 matrix-core GEMM computes -- scores from the reference's op sequence
                                 (oracle/torch_port.py) on the rounded operands, ranks from the C oracle's
                                 rank core (the reference has no bf16 path to run).

Dataset: entity popularity ~ Zipf (kge_amd/synthetic.py).  So that true answers rank near the top and
the filters matter (otherwise every rank is ~E/2 and MRR parity is vacuous), the evaluation triples
are PLANTED: for a random (s, p) the object is drawn from the 10 best-scoring objects under the
float32 tables, and four more of those 10 become train / test triples of the same (s, p).
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

import torch  # noqa: E402

E, R, D = 14541, 237, 512
N_TRAIN, N_VALID = 272115, 2000


def bshape_tables(model: str):
    """The float32 tables of the case (CPU generator: the same bits on every box)."""
    g = torch.Generator().manual_seed({"distmult": 101, "complex": 202}[model])
    ent = torch.randn(E, D, generator=g) * 0.35
    rel = torch.randn(R, D, generator=g) * 0.35
    return ent, rel


def bshape_base_train():
    from kge_amd.synthetic import make_splits
    return make_splits(E, R, N_TRAIN, 0, 0, seed=77)["train"]


def bshape_splits(fixture):
    """train / valid / test of the case: Zipf train from its seed + the planted triples stored in the
    fixture."""
    train = np.concatenate([bshape_base_train(), fixture["planted_train"].astype(np.int32)])
    return {"train": train, "valid": fixture["valid"].astype(np.int32), "test": fixture["test"].astype(np.int32)}


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _plant(model, ent, rel, rng):
    import torch_port as tp
    s = torch.from_numpy(rng.integers(0, E, N_VALID))
    p = torch.from_numpy(rng.integers(0, R, N_VALID))
    with torch.no_grad():
        top = tp.score_sp(model, ent, rel, s, p).topk(10, dim=1).indices.numpy()
    valid, train, test = [], [], []
    for i in range(N_VALID):
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
    config.folder = tempfile.mkdtemp(prefix="kge_bshape_out_")
    config.set("console.quiet", True)
    config.set("model", model)
    config._import(model)
    config.set("dataset.name", "bshape")
    config.set("job.device", "cpu")
    config.set("job.type", "eval")
    config.set_all({"lookup_embedder.dim": D})
    config.set("eval.split", "valid")
    config.set("eval.batch_size", 100)
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


def bf16q_scores(model, ent16, rel16, s, p, o):
    """[n, 2E] scores with bf16 OPERANDS (tables already rounded; the query vector rounded here, each
    product rounded on its own as in DESIGN.md section 4) and float32 accumulation, from the
    reference's contraction `q.mm(T^t)` (complex.py:36-39, distmult.py:17-21)."""
    h = D // 2

    def q_of(a, r, po):
        if model == "distmult":
            return bf16_round(a * r)
        a_re, a_im, r_re, r_im = a[:, :h], a[:, h:], r[:, :h], r[:, h:]
        if not po:  # Re<s, r, conj(o)>: q = s * r
            return bf16_round(torch.cat([a_re * r_re - a_im * r_im, a_re * r_im + a_im * r_re], 1))
        # _po: q = conj(r) * o, scored against the subjects
        return bf16_round(torch.cat([a_re * r_re + a_im * r_im, a_im * r_re - a_re * r_im], 1))

    sp = q_of(ent16[s], rel16[p], False).mm(ent16.t())
    po = q_of(ent16[o], rel16[p], True).mm(ent16.t())
    return torch.cat([sp, po], 1)


def _bf16q_ranks(model, ent16, rel16, splits):
    import oracle as ko
    valid = splits["valid"].astype(np.int64)
    fs = [splits["train"], splits["valid"]]
    idx_sp = [ko.build_index(t, (0, 1), 2) for t in fs]
    idx_po = [ko.build_index(t, (1, 2), 0) for t in fs]
    test_sp, test_po = ko.build_index(splits["test"], (0, 1), 2), ko.build_index(splits["test"], (1, 2), 0)
    out = {k: [] for k in ("o_rank", "o_rank_filt", "o_rank_filt_test", "s_rank", "s_rank_filt", "s_rank_filt_test")}
    for b0 in range(0, len(valid), 100):
        b = valid[b0:b0 + 100]
        s, p, o = (torch.from_numpy(b[:, i]) for i in range(3))
        with torch.no_grad():
            sc = bf16q_scores(model, ent16, rel16, s, p, o).numpy()
        n = len(b)
        o_true, s_true = sc[np.arange(n), b[:, 2]], sc[np.arange(n), E + b[:, 0]]
        for key, isp, ipo in (("", None, None), ("_filt", idx_sp, idx_po),
                              ("_filt_test", idx_sp + [test_sp], idx_po + [test_po])):
            rp_o = cl_o = rp_s = cl_s = None
            if isp is not None:
                rp_o, cl_o = ko.labels_csr(b[:, [0, 1]], isp)
                rp_s, cl_s = ko.labels_csr(b[:, [1, 2]], ipo)
            r_o, t_o = ko.rank_counts(np.ascontiguousarray(sc[:, :E]), o_true, rp_o, cl_o, 0, b[:, 2])
            r_s, t_s = ko.rank_counts(np.ascontiguousarray(sc[:, E:]), s_true, rp_s, cl_s, 0, b[:, 0])
            out["o_rank" + key].append(ko.get_ranks(r_o, t_o))
            out["s_rank" + key].append(ko.get_ranks(r_s, t_s))
    return {k: np.concatenate(v) for k, v in out.items()}


def main():
    from kge_amd.synthetic import write_libkge_dataset
    for model in ("distmult", "complex"):
        ent, rel = bshape_tables(model)
        rng = np.random.default_rng({"distmult": 5, "complex": 6}[model])
        valid, planted, test = _plant(model, ent, rel, rng)
        fixture = {"valid": valid, "planted_train": planted, "test": test}
        splits = bshape_splits(fixture)
        tmp = tempfile.mkdtemp(prefix="kge_bshape_ds_")
        try:
            folder = write_libkge_dataset(os.path.join(tmp, "bshape"), "bshape", E, R, splits)
            res = {}
            for tag, (e_, r_) in (("f32", (ent, rel)), ("bf16t", (bf16_round(ent), bf16_round(rel)))):
                ranks, metrics, triples = _reference_eval(folder, model, e_, r_)
                assert np.array_equal(triples, valid)
                res[tag] = (ranks, metrics)
                print(model, tag, "MRR filt/test", metrics["mean_reciprocal_rank_filtered"],
                      metrics["mean_reciprocal_rank_filtered_with_test"], "raw", metrics["mean_reciprocal_rank"])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        q = _bf16q_ranks(model, bf16_round(ent), bf16_round(rel), splits)
        for k in q:
            print(model, "bf16q vs bf16t:", k, int((q[k] != res["bf16t"][0][k]).sum()), "of", len(q[k]), "differ")
        np.savez_compressed(
            os.path.join(HERE, f"bshape_{model}.npz"), model=model, valid=valid, planted_train=planted, test=test,
            metrics_f32=json.dumps(res["f32"][1], sort_keys=True),
            metrics_bf16t=json.dumps(res["bf16t"][1], sort_keys=True),
            **{f"{k}_f32": v.astype(np.int32) for k, v in res["f32"][0].items()},
            **{f"{k}_bf16t": v.astype(np.int32) for k, v in res["bf16t"][0].items()},
            **{f"{k}_bf16q": v.astype(np.int32) for k, v in q.items()})


if __name__ == "__main__":
    main()
