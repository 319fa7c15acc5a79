"""The LibKGE plugin driven by an UNMODIFIED LibKGE on the MI355X (VERDICT r1 task 1).

Needs the reference package `kge` importable on the GPU box.  It is never committed: the runner
`tools/gpu_plugin.sh` copies /root/reference/kge into the git-ignored `oracle/_ref/libkge/` for the
duration of one gpurun call and removes it afterwards (`oracle/ref_harness.py` finds it there).
Without it every test here is skipped.  The log of a run is kept under `profiles/`.

What runs, all through the reference's own factories (`TrainingJob.create`, `EvaluationJob.create`,
`KgeModel.create`; config.modules = [..., kge_amd.libkge_plugin]) on a synthetic FB15k-237-SHAPE
dataset (E=14,541, R=237; kge_amd/synthetic.py) written in LibKGE's on-disk format:

  (a) `model: hip_complex` under `train.type: 1vsAll` (f32 kernels) against the reference's
      `complex` on the same GPU, seed and batches: epoch loss within 1e-4 relative
      (train_1vsAll.py:48-82); the fused configuration (hip_1vsAll + score_dtype bfloat16 +
      HipAdagrad with bf16 copies) within the bf16 bound stated at the assert;
  (b) `hip_rotate` / `hip_transe` under `negative_sampling` and `hip_negative_sampling`
      (train_negative_sampling.py:103-164) against `rotate` / `transe`;
  (c) `eval.type: hip_entity_ranking` against `entity_ranking` (eval_entity_ranking.py:103-481):
      per-example ranks identical, metrics equal, chunked and unchunked;
  (d) `reciprocal_relations_model` on top of `hip_distmult` against the same on `distmult`
      (reciprocal_relations_model.py:74-124).
"""
import json
import os
import shutil
import sys

import pytest
import torch

import ref_harness as rh

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not rh.available(), reason="reference package `kge` not on this box")]

E, R = 14541, 237
DEVICE = os.environ.get("KGE_PLUGIN_DEVICE", "cuda")  # "cpu" only for dry runs of the reference-model halves
MODULES = ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"]
LOG = []


def _log(**kw):
    LOG.append(kw)
    print("PLUGIN_GPU " + json.dumps(kw, sort_keys=True), file=sys.stderr)


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    from kge_amd.synthetic import make_splits, write_libkge_dataset
    root = tmp_path_factory.mktemp("libkge_gpu")
    splits = make_splits(E, R, 51200, 1500, 1500, seed=3)
    folder = write_libkge_dataset(str(root / "fbshape"), "fbshape", E, R, splits)
    yield str(root), folder
    out = os.environ.get("KGE_PLUGIN_LOG")
    if out:
        with open(out, "w") as f:
            for rec in LOG:
                f.write(json.dumps(rec, sort_keys=True) + "\n")


def _config(root, tag, model, train_type="1vsAll", dim=512, opts=None):
    rh.import_reference()
    from kge import Config
    config = Config()
    config.folder = os.path.join(root, tag)
    shutil.rmtree(config.folder, ignore_errors=True)
    os.makedirs(config.folder)
    config.set("console.quiet", True)
    config.set("modules", MODULES)
    base = None
    if isinstance(model, tuple):  # ("reciprocal_relations_model", base)
        model, base = model
    config.set("model", model)
    config._import(model)
    if base is not None:
        config._import(base)
        config.set(f"{model}.base_model.type", base)
    config.set("dataset.name", "fbshape")
    config.set("job.device", DEVICE)
    config.set("train.max_epochs", 1)
    config.set("train.batch_size", 512)
    config.set("train.num_workers", 0)
    config.set("lookup_embedder.dim", dim)
    config.set("random_seed.default", 17)
    config.set("random_seed.torch", 17)
    config.set("random_seed.numpy", 17)
    config.set("random_seed.python", 17)
    config.set("valid.every", 0)
    for t in {train_type, "hip_entity_ranking", (opts or {}).get("eval.type", "")}:
        if t.startswith("hip_"):
            config._import(t)
    config.set("train.type", train_type)
    for k, v in (opts or {}).items():
        config.set(k, v, create=True)
    return config


def _train_epoch(root, folder, tag, model, train_type="1vsAll", dim=512, opts=None, init_from=None, before_epoch=None):
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    from kge.util.seed import seed_from_config
    config = _config(root, tag, model, train_type, dim, opts)
    seed_from_config(config)
    torch.manual_seed(17)
    dataset = Dataset.create(config, folder=folder)
    job = TrainingJob.create(config, dataset)
    if init_from is not None:
        job.model.load_state_dict(init_from)
    state0 = {k: v.detach().clone() for k, v in job.model.state_dict().items()}
    torch.manual_seed(23)  # batch order / negative samples
    job._prepare()
    job._is_prepared = True
    if before_epoch is not None:
        before_epoch(job)
    trace = job.run_epoch()
    job.last_epoch_trace = trace
    if DEVICE != "cpu":
        torch.cuda.synchronize()
    job.first_epoch_seconds = trace.get("epoch_time", float("nan"))
    return job, trace["avg_loss"], state0


def _second_epoch_seconds(job):
    """Wall time of one more epoch of an already warm job (job-level number, .item() syncs of the
    unmodified trainers included: SURVEY.md 8d asks for these separately from the kernel numbers)."""
    import time
    if DEVICE != "cpu":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.run_epoch()
    if DEVICE != "cpu":
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def _rel(a, b):
    return abs(a - b) / max(1.0, abs(b))


def _param_diff(job_a, job_b):
    out = 0.0
    for (ka, a), (kb, b) in zip(job_a.model.state_dict().items(), job_b.model.state_dict().items()):
        assert ka == kb and a.shape == b.shape
        out = max(out, float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)))
    return out


def test_a_hip_complex_under_1vsAll_matches_the_reference_model(data):
    root, folder = data
    ref, l_ref, st = _train_epoch(root, folder, "a_ref", "complex")
    hip, l_hip, _ = _train_epoch(root, folder, "a_hip", "hip_complex", init_from=st)
    assert type(hip.model).__name__ == "HipComplEx" and type(hip).__name__ == "TrainingJob1vsAll"
    d = _param_diff(hip, ref)
    _log(case="a: hip_complex + 1vsAll (f32 kernels) vs complex", loss_ref=l_ref, loss_hip=l_hip,
         rel=_rel(l_hip, l_ref), param_rel_diff=d)
    assert _rel(l_hip, l_ref) <= 1e-4
    assert d <= 1e-3
    # the whole fused configuration: fused kl loss (kge_ce_sp_po_*), bf16 scoring copies kept fresh
    # by the one-pass Adagrad.  bf16 rounds the query vectors and both tables to 8 bits of mantissa:
    # the bound is the bf16 score error (2^-8 relative per operand), not an f32 one.
    fus, l_fus, _ = _train_epoch(
        root, folder, "a_fused", "hip_complex", "hip_1vsAll", init_from=st,
        opts={"hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
              "train.optimizer.default.args.bf16_copies": True})
    assert type(fus).__name__ == "HipTrainingJob1vsAll" and type(fus.optimizer).__module__ == "kge_amd.optim"
    d16 = _param_diff(fus, ref)
    _log(case="a: hip_complex + hip_1vsAll + bf16 scoring + HipAdagrad vs complex", loss_ref=l_ref,
         loss_hip=l_fus, rel=_rel(l_fus, l_ref), param_rel_diff=d16)
    assert _rel(l_fus, l_ref) <= 1e-2
    assert d16 <= 5e-2
    # hip_1vsAll.graph_step (default true): every full batch behind the two warm-up batches was ONE hipGraph replay
    # (forward + backward + HipAdagrad).  Switched off, the same kernels issued from Python take the same steps: the
    # first epoch's loss to 1e-6, a SECOND epoch over the same batch order to 1e-5 (that one failed until round 4
    # replaced hipMemsetAsync inside the library: a memset captured into a hipGraph is replayed from a pattern the graph
    # does not own, and after ~100 replays the zeroed relation-gradient accumulator was not zero)
    gs = fus._graph_step
    assert gs is not None and gs.disabled_reason is None and gs.replays >= 90 and gs.captures == 1, vars(gs)
    gra = fus
    eag, l_eag, _ = _train_epoch(
        root, folder, "a_fused_eager", "hip_complex", "hip_1vsAll", init_from=st,
        opts={"hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
              "train.optimizer.default.args.bf16_copies": True, "hip_1vsAll.graph_step": False})
    assert eag._graph_step is None
    assert _rel(l_fus, l_eag) <= 1e-6, (l_fus, l_eag)
    second = {}
    for tag, job in (("graph", gra), ("eager", eag)):
        torch.manual_seed(29)
        second[tag] = job.run_epoch()["avg_loss"]
    _log(case="a: hip_1vsAll.graph_step true / false, loss of the first and of the second epoch",
         first_graph=l_fus, first_eager=l_eag, second=second, replays=gs.replays, param_rel_diff=_param_diff(gra, eag))
    assert _rel(second["graph"], second["eager"]) <= 1e-5, second
    # job-level wall time of one more (warm) epoch: 100 batches of 512, E = 14,541, d = 512
    _log(case="a: seconds per epoch (100 batches of 512), TrainingJob1vsAll through LibKGE on the GPU",
         reference_complex=_second_epoch_seconds(ref), hip_complex_f32=_second_epoch_seconds(hip),
         hip_complex_fused_bf16=_second_epoch_seconds(fus),
         hip_complex_fused_bf16_without_graph_step=_second_epoch_seconds(eag))


@pytest.mark.parametrize("loss", ["bce_mean", "bce_self_adversarial"])
def test_b2_negative_sampling_with_the_fused_bce_losses(data, loss):
    """`train.loss: bce_mean | bce_self_adversarial` (kge/util/loss.py:160-186) under hip_negative_sampling: the job's
    loss object is the one-kernel stand-in (loss + gradient of a slot's score block in one launch, no torch.nonzero);
    one epoch against the reference model + job + loss from the same initial parameters."""
    root, folder = data
    opts = {"negative_sampling.num_samples.s": 64, "negative_sampling.num_samples.o": 64,
            "negative_sampling.implementation": "triple", "train.loss": loss}
    ref, l_ref, st = _train_epoch(root, folder, f"b2_ref_{loss}", "rotate", "negative_sampling", 128, opts)
    fus, l_fus, _ = _train_epoch(root, folder, f"b2_fus_{loss}", "hip_rotate", "hip_negative_sampling", 128, opts,
                                 init_from=st)
    assert type(fus.loss).__name__ == "_HipNsBceLoss" and fus.loss.fused_calls > 0
    d = _param_diff(fus, ref)
    _log(case=f"b2: hip_rotate + hip_negative_sampling, train.loss {loss}, vs rotate + negative_sampling",
         loss_ref=l_ref, loss_fused=l_fus, rel_fused=_rel(l_fus, l_ref), param_rel_diff_fused=d,
         seconds_reference=_second_epoch_seconds(ref), seconds_fused=_second_epoch_seconds(fus))
    assert _rel(l_fus, l_ref) <= 1e-4 and d <= 1e-3


@pytest.mark.parametrize("model", ["rotate", "transe"])
def test_b_negative_sampling_jobs(data, model):
    root, folder = data
    opts = {"negative_sampling.num_samples.s": 100, "negative_sampling.num_samples.o": 100,
            "negative_sampling.implementation": "triple"}
    ref, l_ref, st = _train_epoch(root, folder, f"b_ref_{model}", model, "negative_sampling", 128, opts)
    hip, l_hip, _ = _train_epoch(root, folder, f"b_hip_{model}", "hip_" + model, "negative_sampling", 128, opts,
                                 init_from=st)
    fus, l_fus, _ = _train_epoch(root, folder, f"b_fus_{model}", "hip_" + model, "hip_negative_sampling", 128,
                                 opts, init_from=st)
    assert type(fus).__name__ == "HipTrainingJobNegativeSampling"
    d1, d2 = _param_diff(hip, ref), _param_diff(fus, ref)
    _log(case=f"b: hip_{model} + negative_sampling / hip_negative_sampling vs {model}", loss_ref=l_ref,
         loss_hip=l_hip, loss_fused=l_fus, rel=_rel(l_hip, l_ref), rel_fused=_rel(l_fus, l_ref),
         param_rel_diff=d1, param_rel_diff_fused=d2)
    _log(case=f"b: seconds per epoch (100 batches of 512, 2 x 100 negatives), {model}, through LibKGE on the GPU",
         reference=_second_epoch_seconds(ref), hip_model=_second_epoch_seconds(hip),
         hip_negative_sampling=_second_epoch_seconds(fus))
    assert _rel(l_hip, l_ref) <= 1e-4 and _rel(l_fus, l_ref) <= 1e-4
    # TransE's default L1 norm has a sign() gradient and Adagrad's first step is +-lr whatever the
    # gradient's size: a coordinate whose |s + p - o| is within rounding of 0 steps the other way --
    # the parameter bound for it is wider than for the smooth models, the loss bound is not
    bound = 5e-3 if model == "transe" else 1e-3
    assert d1 <= bound and d2 <= bound


def test_b4_negative_sampling_over_the_hip_reciprocal_wrapper(data):
    """Negative sampling over `hip_reciprocal_relations_model(hip_distmult)` (round 6): the wrapper's score_spo(direction)
    and its score_neg hook -- a corrupted subject is the corrupted OBJECT of the reversed triple (o, p + R, s') -- against
    the reference wrapper over the reference model, under the reference's job and under hip_negative_sampling (whose
    whole-step graph capture needs score_neg_blocks, which a reciprocal model cannot offer: ONE positives vector for both
    slots does not exist there; the fused per-slot scoring is what is taken)."""
    root, folder = data
    rr, hrr = "reciprocal_relations_model", "hip_reciprocal_relations_model"
    opts = {"negative_sampling.num_samples.s": 100, "negative_sampling.num_samples.o": 100,
            "negative_sampling.implementation": "triple"}
    ref, l_ref, st = _train_epoch(root, folder, "b4_ref", (rr, "distmult"), "negative_sampling", 128, opts)
    hip, l_hip, _ = _train_epoch(root, folder, "b4_hip", (hrr, "hip_distmult"), "negative_sampling", 128, opts, init_from=st)
    from kge_amd import engine
    calls = {"neg": 0}
    orig = engine.score_neg

    def spy(*a, **k):
        calls["neg"] += 1
        return orig(*a, **k)

    engine.score_neg = spy
    try:
        fus, l_fus, _ = _train_epoch(root, folder, "b4_fus", (hrr, "hip_distmult"), "hip_negative_sampling", 128, opts,
                                     init_from=st)
    finally:
        engine.score_neg = orig
    assert type(fus).__name__ == "HipTrainingJobNegativeSampling" and calls["neg"] >= 100, calls
    d1, d2 = _param_diff(hip, ref), _param_diff(fus, ref)
    _log(case="b4: negative sampling over the hip reciprocal wrapper vs the reference wrapper (distmult, 2 x 100 negatives)",
         loss_ref=l_ref, loss_hip=l_hip, loss_fused=l_fus, rel=_rel(l_hip, l_ref), rel_fused=_rel(l_fus, l_ref),
         param_rel_diff=d1, param_rel_diff_fused=d2, score_neg_calls=calls["neg"],
         seconds_reference=_second_epoch_seconds(ref), seconds_hip_negative_sampling=_second_epoch_seconds(fus))
    assert _rel(l_hip, l_ref) <= 1e-4 and _rel(l_fus, l_ref) <= 1e-4 and d1 <= 1e-3 and d2 <= 1e-3


def _eval(root, folder, tag, model, eval_type, state, chunk=-1, dim=512, opts=None):
    rh.import_reference()
    from kge import Dataset
    from kge.job import EvaluationJob
    from kge.model import KgeModel
    o = {"eval.type": eval_type, "eval.batch_size": 512, "eval.trace_level": "example",
         "entity_ranking.chunk_size": chunk}
    o.update(opts or {})
    config = _config(root, tag, model, "1vsAll", dim, o)
    dataset = Dataset.create(config, folder=folder)
    m = KgeModel.create(config, dataset)
    m.load_state_dict(state)
    job = EvaluationJob.create(config, dataset, parent_job=None, model=m)
    examples = []
    orig = job.trace

    def capture(**kw):
        if kw.get("event") == "example_rank":
            examples.append((kw["task"], kw["s"], kw["p"], kw["o"], kw["rank"], kw["rank_filtered"],
                             kw.get("rank_filtered_with_test")))
        return orig(**kw)

    job.trace = capture
    result = job.run()
    job.eval_seconds = result.get("epoch_time", float("nan"))
    metrics = {k: v for k, v in result.items() if k.startswith("mean_") or k.startswith("hits_at_")}
    return job, examples, metrics


@pytest.mark.parametrize("model", ["distmult", "complex"])
def test_c_hip_entity_ranking_matches_entity_ranking(data, model):
    root, folder = data
    torch.manual_seed(5)
    d = 512
    state = {"_entity_embedder._embeddings.weight": torch.randn(E, d, device=DEVICE),
             "_relation_embedder._embeddings.weight": torch.randn(R, d, device=DEVICE)}
    j0, ex_ref, m_ref = _eval(root, folder, f"c_ref_{model}", model, "entity_ranking", state)
    for chunk in (-1, 5000):
        # the reference's job over the hip model: the kernels score, the reference ranks
        j1, ex_1, m_1 = _eval(root, folder, f"c_mid_{model}", "hip_" + model, "entity_ranking", state, chunk)
        # the plugin's job over the hip model: everything on the device
        j2, ex_2, m_2 = _eval(root, folder, f"c_hip_{model}", "hip_" + model, "hip_entity_ranking", state, chunk)
        assert type(j2).__name__ == "HipEntityRankingJob" and j2._hip_fast
        assert len(ex_1) == len(ex_2) == len(ex_ref) == 2 * 1500
        assert ex_1 == ex_2, "per-example ranks of hip_entity_ranking differ from entity_ranking on the same scores"
        flips = sum(a != b for a, b in zip(ex_ref, ex_2))
        for k in m_1:
            assert m_1[k] == m_2[k], k
        dm = abs(m_ref["mean_reciprocal_rank_filtered_with_test"] - m_2["mean_reciprocal_rank_filtered_with_test"])
        _log(case=f"c: hip_entity_ranking vs entity_ranking, hip_{model}, chunk {chunk}", examples=len(ex_2),
             identical_to_reference_job_on_same_scores=True, examples_differing_from_reference_model=flips,
             mrr_ref_model=m_ref["mean_reciprocal_rank_filtered_with_test"],
             mrr_hip=m_2["mean_reciprocal_rank_filtered_with_test"], abs_mrr_diff=dm,
             eval_seconds_reference_model_and_job=j0.eval_seconds, eval_seconds_hip_model_reference_job=j1.eval_seconds,
             eval_seconds_hip_model_hip_job=j2.eval_seconds)
        # against the reference MODEL the f32 kernel's summation order differs from hipBLASLt's (the
        # reference's mm on this GPU): a rank moves where a score lies within rounding of the tie band's
        # edge (measured: 6-7 of 3000 examples on random N(0,1) tables at d=512, |dMRR| ~ 1e-10); bound
        # it at 0.5 % of the examples and the metric at north_star's 1e-5
        assert flips <= 15 and dm <= 1e-5
    # job-level wall time without per-example tracing (the 3,000 trace entries above dominate otherwise)
    quiet = {"eval.trace_level": "epoch"}
    ja, _, ma = _eval(root, folder, f"c_tref_{model}", model, "entity_ranking", state, opts=quiet)
    jb, _, mb = _eval(root, folder, f"c_thip_{model}", "hip_" + model, "hip_entity_ranking", state, opts=quiet)
    _log(case=f"c: seconds per evaluation of 1,500 triples (3 batches of 512, raw + filtered + filtered-with-test), {model}",
         reference_model_and_job=ja.eval_seconds, hip_model_and_job=jb.eval_seconds)
    assert abs(ma["mean_reciprocal_rank_filtered"] - mb["mean_reciprocal_rank_filtered"]) <= 1e-5


@pytest.mark.parametrize("model", ["distmult", "complex"])
def test_c2_fused_counting_equals_the_reference_job_on_the_same_bf16_scores(data, model):
    """`score_dtype: bfloat16`: HipEntityRankingJob counts inside the scoring kernel (kge_score_rank_sp_po, no
    score matrix).  The reference's EntityRankingJob over the same hip model ranks the score matrices the
    kernel writes: per-example ranks and metrics must be identical, unchunked and with entity chunks."""
    from kge_amd import engine
    root, folder = data
    torch.manual_seed(6)
    d = 512
    state = {"_entity_embedder._embeddings.weight": torch.randn(E, d, device=DEVICE) * 0.3,
             "_relation_embedder._embeddings.weight": torch.randn(R, d, device=DEVICE) * 0.3}
    bf = {f"hip_{model}.score_dtype": "bfloat16"}
    # both jobs on single rounded query vectors (the defaults -- split queries on both sides -- are test_c3's / c4's)
    single = dict(bf, **{"hip_entity_ranking.bf16_queries": "single"})
    bf_single = dict(bf, **{f"hip_{model}.no_grad_queries": "single"})
    calls = {"n": 0}
    orig = engine.score_rank_sp_po

    def counting(*a, **k):
        calls["n"] += 1
        ok = orig(*a, **k)
        assert ok
        return ok

    engine.score_rank_sp_po = counting
    try:
        for chunk in (-1, 5000):
            j1, ex_1, m_1 = _eval(root, folder, f"c2_mid_{model}", "hip_" + model, "entity_ranking", state, chunk, opts=bf_single)
            assert calls["n"] == 0
            j2, ex_2, m_2 = _eval(root, folder, f"c2_hip_{model}", "hip_" + model, "hip_entity_ranking", state, chunk,
                                  opts=single)
            assert calls["n"] > 0, "the fused entry was not used"
            calls["n"] = 0
            assert len(ex_1) == len(ex_2) == 2 * 1500
            assert ex_1 == ex_2
            for k in m_1:
                assert m_1[k] == m_2[k], k
            _log(case=f"c2: fused counting (score_dtype bfloat16), hip_{model}, chunk {chunk}", examples=len(ex_2),
                 identical_to_reference_job_on_same_scores=True, eval_seconds_hip_model_reference_job=j1.eval_seconds,
                 eval_seconds_hip_model_hip_job=j2.eval_seconds)
    finally:
        engine.score_rank_sp_po = orig


@pytest.mark.parametrize("model", ["distmult", "complex"])
def test_c3_split_queries_are_the_default_of_bf16_evaluation(data, model):
    """`score_dtype: bfloat16` + `eval.type: hip_entity_ranking` with its default `bf16_queries: split`: the ranks of
    the REFERENCE model and job on the bf16-rounded tables in float32 arithmetic (the reference's own precision on
    those table values) -- at most a handful of the 3,000 examples differ, every MRR within 1e-5.  Since round 4 the
    split scores are COUNTED inside the scoring kernel (pairs_bf16_v8_rank_kernel: no [n, 2E] score matrix): every
    batch goes through kge_score_rank_sp_po with KGE_FLAG_SPLIT_QUERY."""
    from kge_amd import engine
    root, folder = data
    torch.manual_seed(7)
    d = 512
    ent = (torch.randn(E, d, device=DEVICE) * 0.3).bfloat16().float()
    rel = (torch.randn(R, d, device=DEVICE) * 0.3).bfloat16().float()
    state = {"_entity_embedder._embeddings.weight": ent, "_relation_embedder._embeddings.weight": rel}
    bf = {f"hip_{model}.score_dtype": "bfloat16"}
    calls = {"n": 0, "flags": set()}
    orig = engine.score_rank_sp_po

    def counting(*a, **k):
        calls["n"] += 1
        calls["flags"].add(k.get("flags"))
        return orig(*a, **k)

    engine.score_rank_sp_po = counting
    try:
        for chunk in (-1, 5000):
            _, ex_ref, m_ref = _eval(root, folder, f"c3_ref_{model}", model, "entity_ranking", state, chunk)
            _, ex_hip, m_hip = _eval(root, folder, f"c3_hip_{model}", "hip_" + model, "hip_entity_ranking", state, chunk,
                                     opts=bf)
            assert calls["n"] > 0 and calls["flags"] == {engine.FLAG_SPLIT_QUERY}, calls
            flips = sum(a != b for a, b in zip(ex_ref, ex_hip))
            dm = max(abs(m_ref[k] - m_hip[k]) for k in m_ref if k.startswith("mean_reciprocal_rank"))
            _log(case=f"c3: split queries counted in the kernel (default of score_dtype bfloat16), hip_{model}, chunk {chunk}",
                 examples=len(ex_hip), examples_differing=flips, abs_mrr_diff=dm)
            # random tables, unplanted answers: the ranks are deep (hundreds of neighbours per unit of score), where a
            # neighbour within the ~1e-6 float32 summation noise of the tie band's edge falls on either side -- the
            # reference model on this GPU (hipBLASLt's order) against itself on the CPU differs as often
            assert flips <= 0.02 * len(ex_hip) and dm <= 1e-5
    finally:
        engine.score_rank_sp_po = orig


def test_d2_hip_reciprocal_relations_model_takes_the_fused_paths(data):
    """`model: hip_reciprocal_relations_model` (round 6): the reference's reciprocal wrapper with score_po / score_sp_po
    routed to the base model's fused index-level score_sp (relation p + R) and the fused-loss hooks of hip_1vsAll /
    hip_KvsAll -- against `reciprocal_relations_model` over the reference's distmult: (i) float32 kernels under the
    reference's own 1vsAll job, (ii) hip_1vsAll with bf16 scoring + HipAdagrad: ONE fused loss launch per batch over the
    2 n sp_ queries, the step replayed as a hipGraph, (iii) hip_KvsAll (kl), (iv) both evaluation jobs on the trained
    state; the reference wrapper's checkpoints load (same parameter names)."""
    root, folder = data
    rr, hrr = "reciprocal_relations_model", "hip_reciprocal_relations_model"
    ref, l_ref, st = _train_epoch(root, folder, "d2_ref", (rr, "distmult"), dim=256)
    hip, l_hip, _ = _train_epoch(root, folder, "d2_hip", (hrr, "hip_distmult"), dim=256, init_from=st)
    assert type(hip.model).__name__ == "HipReciprocalRelationsModel" and type(hip.model._base_model).__name__ == "HipDistMult"
    assert list(hip.model.state_dict().keys()) == list(ref.model.state_dict().keys())
    d = _param_diff(hip, ref)
    _log(case="d2: hip_reciprocal_relations_model(hip_distmult) + 1vsAll (f32 kernels) vs the reference wrapper",
         loss_ref=l_ref, loss_hip=l_hip, rel=_rel(l_hip, l_ref), param_rel_diff=d)
    assert _rel(l_hip, l_ref) <= 1e-4 and d <= 1e-3
    from kge_amd import engine
    calls = {"ce": 0}
    orig = engine.ce_fwd

    def spy(*a, **k):
        calls["ce"] += 1
        return orig(*a, **k)

    engine.ce_fwd = spy
    try:
        fus, l_fus, _ = _train_epoch(
            root, folder, "d2_fused", (hrr, "hip_distmult"), "hip_1vsAll", dim=256, init_from=st,
            opts={"hip_distmult.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
                  "train.optimizer.default.args.bf16_copies": True})
    finally:
        engine.ce_fwd = orig
    assert type(fus).__name__ == "HipTrainingJob1vsAll"
    gs = fus._graph_step
    assert calls["ce"] >= 1, "the fused loss of the base model was not reached"
    assert gs is not None and gs.disabled_reason is None and gs.replays >= 90, vars(gs) if gs is not None else None
    d16 = _param_diff(fus, ref)
    _log(case="d2: ... + hip_1vsAll + bf16 scoring + HipAdagrad (one fused launch over 2n sp_ queries, graph step)",
         loss_ref=l_ref, loss_hip=l_fus, rel=_rel(l_fus, l_ref), param_rel_diff=d16, replays=gs.replays,
         seconds_per_epoch_reference=_second_epoch_seconds(ref), seconds_per_epoch_fused=_second_epoch_seconds(fus))
    assert _rel(l_fus, l_ref) <= 1e-2 and d16 <= 5e-2
    # a weighted penalty (what the tuned configurations carry): the wrapper's penalty() -- the base model's terms + the
    # reciprocal relation rows' (reciprocal_relations_model.py:59-72) -- is the reference's code next to the fused loss
    popts = {"lookup_embedder.regularize_args.weighted": True, "distmult.entity_embedder.regularize_weight": 1e-2,
             "distmult.relation_embedder.regularize_weight": 1e-2}
    hpopts = {k.replace("distmult.", "hip_distmult."): v for k, v in popts.items()}
    pref, lp_ref, _ = _train_epoch(root, folder, "d2_pref", (rr, "distmult"), dim=256, init_from=st, opts=popts)
    phip, lp_hip, _ = _train_epoch(root, folder, "d2_phip", (hrr, "hip_distmult"), "hip_1vsAll", dim=256, init_from=st,
                                   opts=dict(hpopts, **{"hip_distmult.score_dtype": "bfloat16"}))
    pen_ref = pref.last_epoch_trace["avg_penalty"]
    pen_hip = phip.last_epoch_trace["avg_penalty"]
    _log(case="d2: hip_1vsAll over the hip reciprocal wrapper with weighted L2 penalties vs the reference", loss_ref=lp_ref,
         loss_hip=lp_hip, rel=_rel(lp_hip, lp_ref), avg_penalty_ref=pen_ref, avg_penalty_hip=pen_hip,
         param_rel_diff=_param_diff(phip, pref))
    assert pen_ref > 0 and abs(pen_hip - pen_ref) <= 2e-2 * pen_ref and _rel(lp_hip, lp_ref) <= 1e-2
    # KvsAll, kl, with label smoothing
    kopts = {"KvsAll.label_smoothing": 0.1}
    kref, lk_ref, _ = _train_epoch(root, folder, "d2_kref", (rr, "distmult"), "KvsAll", dim=256, init_from=st, opts=kopts)
    khip, lk_hip, _ = _train_epoch(root, folder, "d2_khip", (hrr, "hip_distmult"), "hip_KvsAll", dim=256, init_from=st,
                                   opts=dict(kopts, **{"hip_distmult.score_dtype": "bfloat16"}))
    _log(case="d2: hip_KvsAll (kl, label smoothing 0.1) over the hip reciprocal wrapper vs the reference",
         loss_ref=lk_ref, loss_hip=lk_hip, rel=_rel(lk_hip, lk_ref), param_rel_diff=_param_diff(khip, kref))
    assert _rel(lk_hip, lk_ref) <= 1e-2
    # evaluation on the reference's trained state: the reference job and hip_entity_ranking over the hip wrapper
    state = {k: v.detach().clone() for k, v in ref.model.state_dict().items()}
    _, ex_ref, m_ref = _eval(root, folder, "d2_eval_ref", (rr, "distmult"), "entity_ranking", state, dim=256)
    key = "mean_reciprocal_rank_filtered_with_test"
    for job_type in ("entity_ranking", "hip_entity_ranking"):
        for chunk in (-1, 3000):
            _, ex_hip, m_hip = _eval(root, folder, "d2_eval_hip", (hrr, "hip_distmult"), job_type, state, chunk, dim=256)
            flips = sum(a != b for a, b in zip(ex_ref, ex_hip))
            _log(case=f"d2: {job_type} over the hip reciprocal wrapper (f32 tables), chunk {chunk}", examples=len(ex_hip),
                 examples_differing=flips, abs_mrr_diff=abs(m_ref[key] - m_hip[key]))
            assert flips <= 15 and abs(m_ref[key] - m_hip[key]) <= 1e-5


@pytest.mark.parametrize("model", ["complex", "distmult"])
def test_c4_the_reference_job_over_a_hip_model_keeps_the_ranks(data, model):
    """The drop-in a user makes first: ONLY the model's name changes (`hip_complex`, `score_dtype: bfloat16`), the job
    stays the reference's own `eval.type: entity_ranking` (kge/job/eval_entity_ranking.py:143-229).  That job scores
    under torch.no_grad() through score_sp_po(s, p, o, torch.arange(chunk_start, chunk_end)) and score_sp / score_po
    against the batch's unique answers.  Round 6: (i) those calls carry split queries (`no_grad_queries: split`) -- the
    ranks of float32 arithmetic on the bf16 tables, as `hip_entity_ranking` counts them; with `single` the same job
    moves percent of the ranks --, (ii) the arange is recognised as a contiguous chunk of the table and scored by the
    all-entities kernels (no listed-target launch, no f32 chain)."""
    from kge_amd import engine
    root, folder = data
    torch.manual_seed(7)
    d = 512
    ent = (torch.randn(E, d, device=DEVICE) * 0.3).bfloat16().float()
    rel = (torch.randn(R, d, device=DEVICE) * 0.3).bfloat16().float()
    state = {"_entity_embedder._embeddings.weight": ent, "_relation_embedder._embeddings.weight": rel}
    bf = {f"hip_{model}.score_dtype": "bfloat16"}
    seen = {"targets": [], "flags": set()}
    orig = engine.score_sp_po

    def spy(t, s, p, o, entity_subset=None, flags=None):
        seen["targets"].append(type(entity_subset).__name__ if entity_subset is not None else "all")
        seen["flags"].add(flags)
        return orig(t, s, p, o, entity_subset, flags=flags)

    engine.score_sp_po = spy
    try:
        for chunk in (-1, 2000):
            _, ex_ref, m_ref = _eval(root, folder, f"c4_ref_{model}", model, "entity_ranking", state, chunk)
            seen["targets"], seen["flags"] = [], set()
            _, ex_two, m_two = _eval(root, folder, f"c4_two_{model}", "hip_" + model, "entity_ranking", state, chunk, opts=bf)
            # (a last chunk below the model's RANGE_MIN = 1,024 entities stays a listed subset)
            want = {"all"} if chunk < 0 else {"range", "Tensor"}
            assert seen["targets"] and set(seen["targets"]) <= want and seen["targets"][0] != "Tensor", \
                (chunk, set(seen["targets"]))
            assert all(f is not None and f & engine.FLAG_SPLIT_QUERY for f in seen["flags"]), seen["flags"]
            _, ex_hip, m_hip = _eval(root, folder, f"c4_hip_{model}", "hip_" + model, "hip_entity_ranking", state, chunk, opts=bf)
            flips_ref = sum(a != b for a, b in zip(ex_ref, ex_two))
            flips_hip = sum(a != b for a, b in zip(ex_hip, ex_two))
            dm = max(abs(m_ref[k] - m_two[k]) for k in m_ref if k.startswith("mean_reciprocal_rank"))
            _, ex_one, m_one = _eval(root, folder, f"c4_one_{model}", "hip_" + model, "entity_ranking", state, chunk,
                                     opts=dict(bf, **{f"hip_{model}.no_grad_queries": "single"}))
            flips_one = sum(a != b for a, b in zip(ex_ref, ex_one))
            _log(case=f"c4: eval.type entity_ranking (the reference's job) over hip_{model} bf16, chunk {chunk}",
                 examples=len(ex_two), differing_from_reference_model=flips_ref, differing_from_hip_entity_ranking=flips_hip,
                 differing_with_single_queries=flips_one, abs_mrr_diff=dm)
            assert flips_hip == 0, "two-step scores and the counting kernel disagree on split queries"
            assert flips_ref <= 0.02 * len(ex_two) and dm <= 1e-5
    finally:
        engine.score_sp_po = orig


def test_d_reciprocal_relations_model_over_hip_distmult(data):
    root, folder = data
    rr = "reciprocal_relations_model"
    ref, l_ref, st = _train_epoch(root, folder, "d_ref", (rr, "distmult"), dim=256)
    hip, l_hip, _ = _train_epoch(root, folder, "d_hip", (rr, "hip_distmult"), dim=256, init_from=st)
    assert type(hip.model).__name__ == "ReciprocalRelationsModel"
    assert type(hip.model._base_model).__name__ == "HipDistMult"
    d = _param_diff(hip, ref)
    _log(case="d: reciprocal_relations_model(hip_distmult) vs reciprocal_relations_model(distmult), 1vsAll",
         loss_ref=l_ref, loss_hip=l_hip, rel=_rel(l_hip, l_ref), param_rel_diff=d)
    assert _rel(l_hip, l_ref) <= 1e-4 and d <= 1e-3
    # and the ranking evaluation on top of it
    state = {k: v.detach().clone() for k, v in ref.model.state_dict().items()}
    _, ex_ref, m_ref = _eval(root, folder, "d_eval_ref", (rr, "distmult"), "entity_ranking", state, dim=256)
    _, ex_hip, m_hip = _eval(root, folder, "d_eval_hip", (rr, "hip_distmult"), "hip_entity_ranking", state, dim=256)
    flips = sum(a != b for a, b in zip(ex_ref, ex_hip))
    dm = abs(m_ref["mean_reciprocal_rank_filtered_with_test"] - m_hip["mean_reciprocal_rank_filtered_with_test"])
    _log(case="d: evaluation of the reciprocal model", examples=len(ex_hip), examples_differing=flips,
         abs_mrr_diff=dm)
    assert flips <= 15 and dm <= 1e-5


@pytest.mark.parametrize("loss,smoothing", [("kl", 0.0), ("kl", 0.1), ("bce", 0.1), ("kl_s_o", 0.1)])
def test_e_kvsall_jobs_with_label_smoothing(data, loss, smoothing):
    """TrainingJobKvsAll (reference model, reference job) vs HipTrainingJobKvsAll over hip_complex with
    bf16 scoring: the fused kl / bce losses (kge_kl_fwd, kge_kl_weighted_fwd, kge_bce_fwd), with
    KvsAll.label_smoothing (train_KvsAll.py:260-266) handled by the per-row label weight plus the linear
    column-sum score (kge_amd.model.kl_fused / bce_fused)."""
    root, folder = data
    tag = f"e_{loss}_{int(smoothing * 10)}"
    s_o = loss.endswith("_s_o")   # KvsAll.query_types.s_o (round 6): relation-target queries beside the fused losses
    loss = loss[:-4] if s_o else loss
    opts = {"train.loss": loss, "KvsAll.label_smoothing": smoothing}
    if s_o:
        opts["KvsAll.query_types.s_o"] = True
    ref, l_ref, st = _train_epoch(root, folder, tag + "_ref", "complex", "KvsAll", opts=opts)
    fus, l_fus, _ = _train_epoch(
        root, folder, tag + "_fused", "hip_complex", "hip_KvsAll", init_from=st,
        opts={**opts, "hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
              "train.optimizer.default.args.bf16_copies": True})
    assert type(fus).__name__ == "HipTrainingJobKvsAll" and fus._fused_ok()
    d16 = _param_diff(fus, ref)
    _log(case=f"e: hip_complex + hip_KvsAll ({loss}, label_smoothing {smoothing}{', with s_o queries' if s_o else ''}) + bf16 scoring vs complex + KvsAll",
         loss_ref=l_ref, loss_hip=l_fus, rel=_rel(l_fus, l_ref), param_rel_diff=d16,
         seconds_per_epoch_reference=_second_epoch_seconds(ref), seconds_per_epoch_fused=_second_epoch_seconds(fus))
    assert _rel(l_fus, l_ref) <= 1e-2
    assert d16 <= 5e-2


def test_f_embedder_dropout_keeps_the_fused_loss(data, monkeypatch):
    """A tuned-config shape: entity / relation dropout > 0 (lookup_embedder.py:64-69, 102-105) under hip_1vsAll with
    bfloat16 scoring.  VERDICT r3 (missing 6): the fused path declined dropout and such configs trained on the
    unfused path.  Now the masks are applied inside kge_amd.model.ce_fused_dropout and the fused kernels run on the
    dropped-out rows: every subbatch goes through it (counted), the epoch loss is finite and close to the reference
    model's with the same dropout rates (different random masks: a statistical bar)."""
    root, folder = data
    import kge_amd.libkge_plugin.models as pm
    calls = {"n": 0}
    orig = pm.ce_fused_dropout

    def counted(*a, **kw):
        calls["n"] += 1
        return orig(*a, **kw)
    monkeypatch.setattr(pm, "ce_fused_dropout", counted)
    drop = {"hip_complex.entity_embedder.dropout": 0.2, "hip_complex.relation_embedder.dropout": 0.1}
    fus, l_fus, st = _train_epoch(
        root, folder, "f_fused", "hip_complex", "hip_1vsAll",
        opts=dict(drop, **{"hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
                           "train.optimizer.default.args.bf16_copies": True}))
    assert type(fus).__name__ == "HipTrainingJob1vsAll"
    assert calls["n"] == 2 * 100, calls   # two directions x 100 batches: nothing fell back to the unfused path
    ref, l_ref, _ = _train_epoch(root, folder, "f_ref", "complex", init_from=st,
                                 opts={"complex.entity_embedder.dropout": 0.2, "complex.relation_embedder.dropout": 0.1})
    _log(case="f: hip_1vsAll with embedder dropout 0.2 / 0.1 (fused loss, masks inside) vs complex with the same rates",
         loss_ref=l_ref, loss_hip=l_fus, rel=_rel(l_fus, l_ref), fused_calls=calls["n"])
    assert l_fus == l_fus and _rel(l_fus, l_ref) <= 3e-2


def test_g_hip_entity_ranking_counts_split_queries_inside_the_kernel(data, monkeypatch):
    """The default evaluation of bf16 tables (hip_entity_ranking.bf16_queries: split) issues the counting kernel
    (kge_score_rank_sp_po with KGE_FLAG_SPLIT_QUERY: pairs_bf16_v8_rank_kernel) for every batch and never the
    two-step path's score matrix (missing 2): counted calls."""
    root, folder = data
    from kge_amd import engine
    fused, stored = {"n": 0, "flags": set()}, {"n": 0}
    o_rank, o_sp_po = engine.score_rank_sp_po, engine.score_sp_po

    def c_rank(*a, **kw):
        fused["n"] += 1
        fused["flags"].add(kw.get("flags"))
        return o_rank(*a, **kw)

    def c_sp_po(t, s, p, o, sub=None, **kw):
        if sub is None or sub.numel() > 2 * s.numel():  # (the true scores use the batch's own 2n targets)
            stored["n"] += 1
        return o_sp_po(t, s, p, o, sub, **kw)
    monkeypatch.setattr(engine, "score_rank_sp_po", c_rank)
    monkeypatch.setattr(engine, "score_sp_po", c_sp_po)
    job, _, st = _train_epoch(root, folder, "g_train", "hip_complex", "hip_1vsAll",
                              opts={"hip_complex.score_dtype": "bfloat16"})
    _, _, m = _eval(root, folder, "g_eval", "hip_complex", "hip_entity_ranking", job.model.state_dict(),
                    opts={"hip_complex.score_dtype": "bfloat16"})
    assert fused["n"] >= 3 and fused["flags"] == {engine.FLAG_SPLIT_QUERY}, fused
    assert stored["n"] == 0, stored
    _log(case="g: hip_entity_ranking, bf16 tables, split queries counted inside the kernel", batches=fused["n"],
         mrr=m["mean_reciprocal_rank_filtered"])


@pytest.fixture
def no_reference_scoring(monkeypatch):
    """VERDICT r4 (weak 7, next 3): while a hip_* model on a CUDA device scores / trains / evaluates, the REFERENCE's
    scoring arithmetic must never be reached -- the four scorers' `score_emb` (complex.py:18-43, distmult.py:13-25,
    transe.py:15-37, rotate.py:20-69) and `LookupEmbedder.embed_all` (lookup_embedder.py:107-112: the full-table copy
    the fused gather removes) raise for the duration of the test."""
    rh.import_reference()
    from kge.model.complex import ComplExScorer
    from kge.model.distmult import DistMultScorer
    from kge.model.rotate import RotatEScorer
    from kge.model.transe import TransEScorer
    from kge.model import LookupEmbedder
    hit = []

    def trap(name):
        def raiser(*a, **kw):
            hit.append(name)
            raise AssertionError(f"the reference's {name} was reached while a hip_* model ran on a CUDA device")
        return raiser
    for cls in (ComplExScorer, DistMultScorer, TransEScorer, RotatEScorer):
        monkeypatch.setattr(cls, "score_emb", trap(cls.__name__ + ".score_emb"))
    monkeypatch.setattr(LookupEmbedder, "embed_all", trap("LookupEmbedder.embed_all"))
    return hit


@pytest.mark.parametrize("model,dim", [("complex", 512), ("distmult", 512), ("transe", 128), ("rotate", 128)])
def test_h_hip_models_on_cuda_never_reach_the_reference_scorers(data, no_reference_scoring, model, dim):
    """Every entry of the index-level API (kge_model.py:663-789), a training epoch of the job type the model is
    configured for in BASELINE.json and an evaluation -- with the reference scorers and embed_all trapped.  Also the
    libraries that did the work are the in-tree ones (mapped into this process)."""
    if DEVICE == "cpu":
        pytest.skip("job.device cpu is the reference's arithmetic by design (configs[0] plumbing)")
    root, folder = data
    from kge_amd import _lib
    if model in ("complex", "distmult"):
        ttype = "hip_1vsAll" if model == "complex" else "hip_KvsAll"
        opts = {f"hip_{model}.score_dtype": "bfloat16"}
    else:
        ttype = "hip_negative_sampling"
        opts = {"negative_sampling.num_samples.s": 32, "negative_sampling.num_samples.o": 32,
                "negative_sampling.implementation": "triple"}
    job, loss, _ = _train_epoch(root, folder, f"h_{model}", "hip_" + model, ttype, dim, opts)
    assert loss == loss and type(job.model).__name__.startswith("Hip")
    m = job.model
    m.eval()
    g = torch.Generator().manual_seed(2)
    s = torch.randint(E, (64,), generator=g).to(DEVICE)
    p = torch.randint(R, (64,), generator=g).to(DEVICE)
    o = torch.randint(E, (64,), generator=g).to(DEVICE)
    sub = torch.randint(E, (100,), generator=g).to(DEVICE)
    with torch.no_grad():
        shapes = [tuple(m.score_spo(s, p, o).shape), tuple(m.score_sp(s, p).shape), tuple(m.score_po(p, o).shape),
                  tuple(m.score_sp(s, p, sub).shape), tuple(m.score_sp_po(s, p, o).shape),
                  tuple(m.score_sp_po(s, p, o, sub).shape)]
    assert shapes == [(64,), (64, E), (64, E), (64, 100), (64, 2 * E), (64, 200)], shapes
    m.train()
    m.score_sp(s, p).sum().backward()        # autograd through the kernels' own backward
    _, _, metrics = _eval(root, folder, f"h_eval_{model}", "hip_" + model, "hip_entity_ranking", m.state_dict(), dim=dim,
                          opts={k: v for k, v in opts.items() if k.startswith("hip_")})
    assert metrics["mean_reciprocal_rank_filtered"] > 0
    assert no_reference_scoring == [], no_reference_scoring
    with open("/proc/self/maps") as f:
        mapped = f.read()
    assert _lib.LIB_PATH in mapped, "libkge_amd.so is not mapped into the process that just scored"
    _log(case=f"h: hip_{model} on cuda with the reference scorers and embed_all trapped", train_type=ttype,
         avg_loss=loss, mrr=metrics["mean_reciprocal_rank_filtered"], reference_scorer_calls=len(no_reference_scoring))


def test_i_subbatch_auto_tune_fires_on_rocm(data):
    """train.subbatch_auto_tune (kge/job/train.py:384-413) halves `train.subbatch_size` when a batch raises a
    RuntimeError containing "CUDA out of memory".  torch-ROCm says "HIP out of memory", so under an unmodified LibKGE
    the tuner is dead on this GPU; the hip_* jobs re-raise under the text the trainer matches.
    Here: hip_complex (float32 kernels, the [n, E] score matrices of the unfused 1vsAll path) with a batch of 16,384
    under a per-process limit of ~1 GiB above what is already reserved: 953 MB per score matrix cannot fit, the tuner
    must halve until the batch goes through, and the epoch's loss equals an untuned run's with that sub-batch size."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU allocator")
    import gc
    root, folder = data
    dev = torch.device("cuda", 0)
    gc.collect()
    torch.cuda.empty_cache()
    total = torch.cuda.get_device_properties(dev).total_memory
    limit = torch.cuda.memory_reserved(dev) + (1 << 30)
    opts = {"train.batch_size": 16384, "train.subbatch_auto_tune": True}
    torch.cuda.set_per_process_memory_fraction(min(1.0, limit / total), dev)
    try:
        job, loss, st = _train_epoch(root, folder, "i_tuned", "hip_complex", "hip_1vsAll", opts=opts)
    finally:
        torch.cuda.set_per_process_memory_fraction(1.0, dev)
    sub = job.config.get("train.subbatch_size")
    _log(case="i: train.subbatch_auto_tune under a 1 GiB limit, hip_complex + hip_1vsAll, batch 16,384", avg_loss=loss,
         subbatch_size_after=sub)
    assert loss == loss and 256 <= sub <= 8192, sub
    gc.collect()
    torch.cuda.empty_cache()
    ref, l_ref, _ = _train_epoch(root, folder, "i_fixed", "hip_complex", "hip_1vsAll", init_from=st,
                                 opts={"train.batch_size": 16384, "train.subbatch_size": int(sub)})
    assert _rel(loss, l_ref) <= 1e-5, (loss, l_ref)


@pytest.mark.parametrize("case", ["1vsAll", "KvsAll", "negative_sampling", "1vsAll-reciprocal", "KvsAll-smoothed"])
def test_j_sharded_jobs_behind_the_plugin_api_on_the_gpu(data, monkeypatch, case):
    """train.type: hip_sharded_* + eval.type: hip_sharded_entity_ranking through TrainingJob.create / EvaluationJob.create
    of an unmodified LibKGE on the MI355X, as ONE rank of an RCCL group (torchrun's environment for a world of one,
    ShardedEntityTable.FORCE_COLLECTIVES: every all-gather / all-reduce of the N > 1 path is issued) with the engine's
    kernels -- against the unsharded hip_* job of the same config.  (Two ranks: tests/test_libkge_sharded_plugin_cpu.py
    on gloo; the driver's box has one GPU.)"""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    import socket
    import torch.distributed as dist
    root, folder = data
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port)), ("RANK", "0"), ("WORLD_SIZE", "1"),
                 ("LOCAL_RANK", "0")):
        monkeypatch.setenv(k, v)
    from kge_amd.sharded import ShardedEntityTable
    monkeypatch.setattr(ShardedEntityTable, "FORCE_COLLECTIVES", True)
    if case == "negative_sampling":
        model, dim, plain = "hip_rotate", 128, "hip_negative_sampling"
        opts = {"negative_sampling.num_samples.s": 64, "negative_sampling.num_samples.o": 64,
                "negative_sampling.implementation": "triple"}
        bound = 1e-4
    elif case == "KvsAll-smoothed":
        # KvsAll.label_smoothing under kl (round 6): the uniform part of the smoothed labels as the gradient kernel's
        # per-row bias + one score per query against the shard's column sum
        model, dim, plain = "hip_distmult", 512, "hip_KvsAll"
        opts = {"hip_distmult.score_dtype": "bfloat16", "KvsAll.label_smoothing": 0.1}
        bound, case = 2e-3, "KvsAll"
    elif case == "1vsAll-reciprocal":
        # the reciprocal wrapper (round 6): the sharded table scores the subject direction as an sp_ query with relation
        # p + R (kge_amd.sharded: _recip) -- against hip_1vsAll over the same wrapper (one fused launch over 2n sp_ queries)
        model, dim, plain = ("hip_reciprocal_relations_model", "hip_complex"), 512, "hip_1vsAll"
        opts = {"hip_complex.score_dtype": "bfloat16"}
        bound, case = 2e-3, "1vsAll"
    else:
        model, dim, plain = ("hip_complex", 512, "hip_1vsAll") if case == "1vsAll" else ("hip_distmult", 512, "hip_KvsAll")
        opts = {f"{model}.score_dtype": "bfloat16"}
        bound = 2e-3   # both runs score on bf16 copies of the same float32 masters; the kernels' summation orders differ
    try:
        # the sharded jobs seed the process-wide generators per epoch from a base number + the epoch
        # (sharded_job._seed_epoch): the plain job draws the same batches / negatives when seeded the same way
        import random
        import numpy as np
        rh.import_reference()
        from kge_amd.libkge_plugin.sharded_job import epoch_seed

        def seed_like_sharded(job):
            v = epoch_seed(4242, job.epoch)
            torch.manual_seed(v)
            np.random.seed(v % (2 ** 32))
            random.seed(v)
        tagm = "recip_" if isinstance(model, tuple) else ("smoothed_" if "KvsAll.label_smoothing" in opts else "")
        ref, l_ref, st = _train_epoch(root, folder, f"j_plain_{tagm}{case}", model, plain, dim, opts, before_epoch=seed_like_sharded)
        sopts = dict(opts)
        sopts["eval.type"] = "hip_sharded_entity_ranking"
        shd, l_shd, _ = _train_epoch(root, folder, f"j_sharded_{tagm}{case}", model, "hip_sharded_" + case, dim, sopts, init_from=st,
                                     before_epoch=lambda job: setattr(job, "_seed_base", 4242))
        assert type(shd).__name__.startswith("HipShardedTrainingJob") and dist.is_initialized()
        assert dist.get_backend() == "nccl" and shd._sh.table.collectives
        d = _param_diff(shd, ref) if case != "negative_sampling" else None
        if tagm == "recip_":
            assert shd._sh.table.reciprocal_R == R
        _log(case=f"j: hip_sharded_{case} {tagm}(one RCCL rank, collectives forced) vs {plain}", loss_plain=l_ref,
             loss_sharded=l_shd, rel=_rel(l_shd, l_ref), param_rel_diff_own_rows=d)
        assert _rel(l_shd, l_ref) <= bound, (l_shd, l_ref)
        # validation through the job's own valid_job = the sharded evaluation on the table being trained
        shd.valid_job.epoch = 1
        tr = shd.valid_job.run()
        assert type(shd.valid_job).__name__ == "HipShardedEntityRankingJob"
        ref.valid_job.epoch = 1
        tr_ref = ref.valid_job.run()
        k = "mean_reciprocal_rank_filtered_with_test"
        _log(case=f"j: validation of hip_sharded_{case} {tagm}by hip_sharded_entity_ranking vs the plain job's entity_ranking",
             mrr_sharded=tr[k], mrr_plain=tr_ref[k])
        assert abs(tr[k] - tr_ref[k]) <= 2e-3 * max(tr_ref[k], 1e-3) + 1e-4
        # the checkpoint: the reference's layout, ONE [E, d] parameter
        ck = shd.save_to({})
        assert ck["model"][0]["_entity_embedder._embeddings.weight"].shape == (E, dim) and ck["type"] == "train"
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_k_a_large_batch_through_the_plugin_model_reaches_the_persistent_kernel(data):
    """VERDICT r4 (next 2): `HipComplEx.score_sp(s, p)` with n = 4096 -- KgeModel.score_sp of an unmodified LibKGE model
    object (kge_model.py:682-702), bf16 scoring copies -- launches pairs_bf16_v8_kernel (kge_debug_launch_count moves by
    one per call), as do score_po and score_sp_po; the scores equal the same rows scored 512 at a time."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    import ctypes
    from kge_amd import _lib
    rh.import_reference()
    from kge import Dataset
    from kge.model import KgeModel
    root, folder = data
    config = _config(root, "k_model", "hip_complex", opts={"hip_complex.score_dtype": "bfloat16"})
    m = KgeModel.create(config, Dataset.create(config, folder=folder)).to(DEVICE)
    m.eval()
    count = _lib.lib().kge_debug_launch_count
    count.restype, count.argtypes = ctypes.c_int, [ctypes.c_int]
    g = torch.Generator().manual_seed(4)
    n = 4096
    s, p, o = (torch.randint(hi, (n,), generator=g).to(DEVICE) for hi in (E, R, E))
    with torch.no_grad():
        for name, call, small in (
                ("score_sp", lambda: m.score_sp(s, p), lambda a, b: m.score_sp(s[a:b], p[a:b])),
                ("score_po", lambda: m.score_po(p, o), lambda a, b: m.score_po(p[a:b], o[a:b])),
                ("score_sp_po", lambda: m.score_sp_po(s, p, o), lambda a, b: m.score_sp_po(s[a:b], p[a:b], o[a:b]))):
            before = count(0)
            big = call()
            torch.cuda.synchronize()
            assert count(0) == before + 1, f"HipComplEx.{name}(n = {n}) did not launch pairs_bf16_v8_kernel"
            for a in range(0, n, 512):
                assert torch.equal(big[a:a + 512], small(a, a + 512)), (name, a)
    _log(case="k: HipComplEx.score_sp / score_po / score_sp_po with n = 4096 run on pairs_bf16_v8_kernel", launches=count(0))


@pytest.mark.parametrize("model,loss", [("rotate", "kl"), ("transe", "bce_self_adversarial")])
def test_b3_negative_sampling_step_as_one_hipgraph(data, model, loss):
    """hip_negative_sampling.graph_step (VERDICT r4 missing 6): with an optimizer whose step is kernels only (HipAdagrad)
    every full batch behind the two warm-up batches is ONE hipGraph replay -- positives, both slots' negative blocks,
    the loss, backward, the optimizer's step.  Switched off, the same kernels issued from Python take the same steps:
    epoch loss to 1e-5 (the replay orders the float atomics of the gradient scatter differently), and both agree with
    the reference model + job + torch Adagrad from the same initial parameters at the bar of test b."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    root, folder = data
    opts = {"negative_sampling.num_samples.s": 64, "negative_sampling.num_samples.o": 48,
            "negative_sampling.implementation": "triple", "train.loss": loss}
    ref, l_ref, st = _train_epoch(root, folder, f"b3_ref_{model}", model, "negative_sampling", 128, opts)
    hopts = dict(opts, **{"train.optimizer.default.type": "HipAdagrad"})
    gra, l_gra, _ = _train_epoch(root, folder, f"b3_graph_{model}", "hip_" + model, "hip_negative_sampling", 128, hopts,
                                 init_from=st)
    eag, l_eag, _ = _train_epoch(root, folder, f"b3_eager_{model}", "hip_" + model, "hip_negative_sampling", 128,
                                 dict(hopts, **{"hip_negative_sampling.graph_step": False}), init_from=st)
    gs = gra._graph_step
    assert gs is not None and gs.disabled_reason is None and gs.replays >= 90 and gs.captures == 1, vars(gs)
    assert eag._graph_step is None
    _log(case=f"b3: hip_{model} + hip_negative_sampling ({loss}) + HipAdagrad, graph_step true / false vs {model} + "
              "negative_sampling + Adagrad", loss_ref=l_ref, loss_graph=l_gra, loss_eager=l_eag, replays=gs.replays,
         rel_graph_vs_eager=_rel(l_gra, l_eag), rel_vs_ref=_rel(l_gra, l_ref),
         seconds_reference=_second_epoch_seconds(ref), seconds_graph=_second_epoch_seconds(gra),
         seconds_eager=_second_epoch_seconds(eag))
    assert _rel(l_gra, l_eag) <= 1e-5, (l_gra, l_eag)
    assert _rel(l_gra, l_ref) <= 1e-4, (l_gra, l_ref)


@pytest.mark.parametrize("regularize,space,weights", [("lp", "euclidean", (0.8e-7, 0.8e-7)), ("lp", "euclidean", (2e-4, 5e-3)),
                                                      ("lp3", "euclidean", (2e-4, 5e-3)), ("n3", "complex", (2e-4, 5e-3))])
def test_l_unweighted_penalty_terms_stay_on_the_captured_step(data, regularize, space, weights):
    """BASELINE configs[0] (examples/toy-complex-train.yaml) sets lookup_embedder.regularize_weight 0.8e-7: an unweighted
    L2 term over both tables, back-propagated by TrainingJob.run_epoch between the batch and optimizer.step()
    (kge/job/train.py:417-436, lookup_embedder.py:122-147, kge_model.py:603-649).  Until round 6 any penalty switched
    hip_1vsAll.graph_step off.  Now HipAdagrad folds the terms into its pass (kge_adagrad_step_multi_penalty) and the
    captured step carries them: the job replays, the trace shows the reference's penalty values, and the parameters
    follow the reference job (complex + 1vsAll + torch Adagrad, the terms through autograd) at the bf16-scoring bar of
    test a and the eager hip_1vsAll job (the reference's autograd terms on the same kernels) at float rounding; also at
    weights where the terms move the parameters, L3, and N3 over complex coordinates (lookup_embedder.space complex)."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    root, folder = data
    tag = f"{regularize}_{space}_{weights[0]}"
    pen = {"lookup_embedder.regularize": regularize[:2], "lookup_embedder.space": space,
           "lookup_embedder.regularize_args.p": 3 if regularize == "lp3" else 2, "complex.entity_embedder.regularize_weight": weights[0],
           "complex.relation_embedder.regularize_weight": weights[1]}
    ref, l_ref, st = _train_epoch(root, folder, f"l_ref_{tag}", "complex", opts=pen)
    hpen = {k.replace("complex.", "hip_complex."): v for k, v in pen.items()}
    hpen.update({"train.optimizer.default.type": "HipAdagrad", "hip_complex.score_dtype": "bfloat16",
                 "train.optimizer.default.args.bf16_copies": True})
    gra, l_gra, _ = _train_epoch(root, folder, f"l_graph_{tag}", "hip_complex", "hip_1vsAll", opts=hpen, init_from=st)
    eag, l_eag, _ = _train_epoch(root, folder, f"l_eager_{tag}", "hip_complex", "hip_1vsAll", init_from=st,
                                 opts=dict(hpen, **{"hip_1vsAll.graph_step": False}))
    gs = gra._graph_step
    assert gs is not None and gs.disabled_reason is None and gs.replays >= 90 and gs.captures == 1, vars(gs)
    assert gra.optimizer.has_penalties() and len(gra._folded_penalties) == 2
    assert eag._graph_step is None and not eag.optimizer.has_penalties()   # the eager job: the reference's autograd terms
    tr_ref, tr_gra, tr_eag = (j.last_epoch_trace for j in (ref, gra, eag))
    # (the keys carry the model's configuration key: complex.entity_embedder.L2_penalty / hip_complex.entity_...)
    tr_ref, tr_gra, tr_eag = (dict(tr, avg_penalties={k.split(".", 1)[1]: v for k, v in tr["avg_penalties"].items()})
                              for tr in (tr_ref, tr_gra, tr_eag))
    keys = sorted(tr_ref["avg_penalties"])
    assert keys == sorted(tr_gra["avg_penalties"]) == sorted(tr_eag["avg_penalties"]) and len(keys) == 2, (keys, tr_gra["avg_penalties"])
    assert [k_[2] for k_ in gra._folded_penalties] == ["n3_complex" if (regularize, space) == ("n3", "complex") else "lp"] * 2
    _log(case=f"l: hip_1vsAll.graph_step with unweighted {regularize} ({space}) penalties {weights} folded into HipAdagrad",
         loss_ref=l_ref, loss_graph=l_gra, loss_eager=l_eag, penalties_ref=tr_ref["avg_penalties"],
         penalties_graph=tr_gra["avg_penalties"], penalties_eager=tr_eag["avg_penalties"], replays=gs.replays,
         param_rel_diff_graph_vs_eager=_param_diff(gra, eag), param_rel_diff_graph_vs_ref=_param_diff(gra, ref))
    for k in keys:   # the trace's penalty values: the folded ones against autograd on the same kernels, and the reference
        assert _rel(tr_gra["avg_penalties"][k], tr_eag["avg_penalties"][k]) <= 1e-5 * max(1.0, abs(tr_eag["avg_penalties"][k]))
        assert abs(tr_gra["avg_penalties"][k] - tr_ref["avg_penalties"][k]) <= 2e-2 * abs(tr_ref["avg_penalties"][k])
    assert _rel(l_gra, l_eag) <= 1e-5, (l_gra, l_eag)
    assert _param_diff(gra, eag) <= 1e-4
    assert _rel(l_gra, l_ref) <= 1e-2 and _param_diff(gra, ref) <= 5e-2   # (bf16 scoring: the bar of test a)


@pytest.mark.parametrize("model,dim", [("distmult", 256), ("complex", 512)])
def test_m_band_and_rescore_behind_hip_entity_ranking(data, model, dim):
    """hip_entity_ranking.band_rescore: on a model that ranks its validation triples high -- here the
    true object's row nudged along the query vector of its triple, so that score(s, p, o) sits ~5 sigma out in (s, p)'s
    row and in (p, o)'s -- the split-query counts come from a single-pass counting launch + a gather launch over the few
    undecided pairs (kge_score_rank_sp_po_band).  Per-example ranks and every metric equal to the same job with
    `band_rescore: never` (the split kernel), batches counted; the same on untrained tables (1.5 % of the pairs listed)."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    from kge_amd import engine
    root, folder = data
    rh.import_reference()
    from kge import Dataset
    torch.manual_seed(11)
    ent = (torch.randn(E, dim, device=DEVICE) * 0.3)
    rel = (torch.randn(R, dim, device=DEVICE) * 0.3)
    cfg = _config(root, f"m_probe_{model}", "hip_" + model, "1vsAll", dim, {})
    valid = Dataset.create(cfg, folder=folder).split("valid").to(DEVICE).long()
    s, p, o = valid[:, 0], valid[:, 1], valid[:, 2]
    if model == "distmult":
        q = ent[s] * rel[p]
    else:
        h = dim // 2
        sr, si, rr, ri = ent[s, :h], ent[s, h:], rel[p, :h], rel[p, h:]
        q = torch.cat([sr * rr - si * ri, sr * ri + si * rr], 1)   # Re<s r, conj(o)> = <q, o> on the [re | im] layout
    ent.index_add_(0, o, 1.5 * q / q.norm(dim=1, keepdim=True))
    state = {"_entity_embedder._embeddings.weight": ent.bfloat16().float(),
             "_relation_embedder._embeddings.weight": rel.bfloat16().float()}
    bf = {f"hip_{model}.score_dtype": "bfloat16"}
    banded = {"n": 0}
    orig = engine.score_rank_sp_po

    def counting(*a, **k):
        banded["n"] += 1 if k.get("band") is not None else 0
        return orig(*a, **k)
    engine.score_rank_sp_po = counting
    try:
        j_ref, ex_ref, m_ref = _eval(root, folder, f"m_split_{model}", "hip_" + model, "hip_entity_ranking", state, dim=dim,
                                     opts=dict(bf, **{"hip_entity_ranking.band_rescore": "never"}))
        assert banded["n"] == 0
        j_band, ex_band, m_band = _eval(root, folder, f"m_band_{model}", "hip_" + model, "hip_entity_ranking", state, dim=dim,
                                        opts=dict(bf, **{"hip_entity_ranking.band_rescore": "always"}))
        assert banded["n"] >= 3 and j_band._ev["band_batches"] == banded["n"], (banded, j_band._ev["band_batches"])
        assert ex_band == ex_ref and m_band == m_ref
        assert m_ref["mean_reciprocal_rank_filtered"] > 0.3      # (the nudged triples do rank high)
        # "auto" stays off at this entity count; untrained tables drop pairs and fall back, batch by batch
        j_auto, ex_auto, _ = _eval(root, folder, f"m_auto_{model}", "hip_" + model, "hip_entity_ranking", state, dim=dim,
                                   opts=bf)
        assert j_auto._ev["band"] is None and ex_auto == ex_ref
        raw = {"_entity_embedder._embeddings.weight": (torch.randn(E, dim, device=DEVICE) * 0.3).bfloat16().float(),
               "_relation_embedder._embeddings.weight": state["_relation_embedder._embeddings.weight"]}
        _, ex_r0, m_r0 = _eval(root, folder, f"m_raw_split_{model}", "hip_" + model, "hip_entity_ranking", raw, dim=dim,
                               opts=dict(bf, **{"hip_entity_ranking.band_rescore": "never"}))
        j_r1, ex_r1, m_r1 = _eval(root, folder, f"m_raw_band_{model}", "hip_" + model, "hip_entity_ranking", raw, dim=dim,
                                  opts=dict(bf, **{"hip_entity_ranking.band_rescore": "always"}))
        # (1.5 % of the pairs are listed there: at this shape -- 64 lists per 32 rows -- they still fit; on a large table
        # they do not, tests/test_gpu_score_rank.py::test_band_on_random_triples_reports_the_pairs_it_drops)
        assert ex_r1 == ex_r0 and m_r1 == m_r0
    finally:
        engine.score_rank_sp_po = orig
    _log(case=f"m: hip_entity_ranking.band_rescore on hip_{model} d={dim}: ranks and metrics == the split kernel's",
         batches_through_the_band=j_band._ev["band_batches"], mrr_filtered=m_ref["mean_reciprocal_rank_filtered"],
         seconds_split=j_ref.eval_seconds, seconds_band=j_band.eval_seconds)


@pytest.mark.parametrize("loss", ["kl", "bce"])
def test_n_kvsall_step_as_one_hipgraph_on_padded_inputs(data, loss):
    """hip_KvsAll.graph_step (VERDICT r5 missing 5; train_KvsAll.py:216-294): with an optimizer whose step is kernels only
    the whole step of a batch -- both query types' fused losses, ONE backward for both, HipAdagrad -- is captured once
    on inputs padded to a capacity of rows and label entries per query type and replayed per batch (padding rows: weight
    0 in the batch loss, gradient rows exactly zero).  Switched off, the same kernels issued from Python take the same
    steps: epoch losses to 1e-5 (the replay orders the float atomics of the gradient scatter differently), parameters
    1e-4; both agree with the reference model + job at the bf16 bar of test e.  A penalty term (unweighted L2, folded
    into HipAdagrad) rides along."""
    if DEVICE == "cpu":
        pytest.skip("needs the GPU")
    root, folder = data
    opts = {"train.loss": loss, "lookup_embedder.regularize_weight": 1e-5}
    ref, l_ref, st = _train_epoch(root, folder, f"n_ref_{loss}", "complex", "KvsAll", opts=opts)
    hopts = {**opts, "hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
             "train.optimizer.default.args.bf16_copies": True}
    gra, l_gra, _ = _train_epoch(root, folder, f"n_graph_{loss}", "hip_complex", "hip_KvsAll", init_from=st, opts=hopts)
    eag, l_eag, _ = _train_epoch(root, folder, f"n_eager_{loss}", "hip_complex", "hip_KvsAll", init_from=st,
                                 opts=dict(hopts, **{"hip_KvsAll.graph_step": False}))
    gs = gra._graph_step
    batches = len(gra.loader)
    assert gs is not None and gs.disabled_reason is None and gs.replays >= 0.8 * batches - 6, (vars(gs), gra._graph_caps)
    assert gra.graph_batches >= 0.9 * batches and eag._graph_step is None and eag.graph_batches == 0
    assert gra.optimizer.has_penalties()
    second = {}
    for tag, job in (("graph", gra), ("eager", eag)):
        torch.manual_seed(29)
        second[tag] = job.run_epoch()["avg_loss"]
    _log(case=f"n: hip_KvsAll.graph_step true / false ({loss}), first and second epoch", loss_ref=l_ref, first_graph=l_gra,
         first_eager=l_eag, second=second, replays=gs.replays, captures=gs.captures, capacities=gra._graph_caps,
         batches_through_the_step=gra.graph_batches, param_rel_diff=_param_diff(gra, eag),
         seconds_graph=_second_epoch_seconds(gra), seconds_eager=_second_epoch_seconds(eag),
         seconds_reference=_second_epoch_seconds(ref))
    assert _rel(l_gra, l_eag) <= 1e-5, (l_gra, l_eag)
    assert _rel(second["graph"], second["eager"]) <= 1e-4, second
    assert _param_diff(gra, eag) <= 3e-3   # (after two epochs of float-atomic gradient scatters in different orders)
    assert _rel(l_gra, l_ref) <= 1e-2 and _param_diff(gra, ref) <= 5e-2
