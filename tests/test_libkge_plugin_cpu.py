"""LibKGE plugin plumbing (build container only: needs the reference tree).  The plugin is
discovered through the reference's own mechanism (modules: + <name>.yaml + class_name),
keeps its parameter names, and refuses to compute on job.device=cpu (no CPU fallback)."""
import pytest
import torch

import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")

MODELS = ["hip_complex", "hip_distmult", "hip_transe", "hip_rotate"]


def _config(model, dim=16):
    rh.import_reference()
    from kge import Config
    config = Config()
    config.folder = None
    config.set("console.quiet", True)
    config.set("modules", ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"])
    config.set("model", model)
    config._import(model)
    config.set("job.device", "cpu")
    config.set_all({"lookup_embedder.dim": dim})
    config.set("dataset.num_entities", 30)
    config.set("dataset.num_relations", 4)
    return config


@pytest.mark.parametrize("model", MODELS)
def test_plugin_is_discovered_and_keeps_parameter_names(model):
    config = _config(model)
    from kge import Dataset
    from kge.model import KgeModel
    import kge_amd.libkge_plugin as plugin
    m = KgeModel.create(config, Dataset(config, folder=None))
    assert type(m).__name__ == config.get(f"{model}.class_name")
    assert isinstance(m, KgeModel) and isinstance(m, plugin.models._FusedScoring)
    assert list(m.state_dict().keys()) == ["_entity_embedder._embeddings.weight",
                                           "_relation_embedder._embeddings.weight"]
    e, r = m.state_dict().values()
    assert e.shape == (30, 16) and r.shape == (4, 8 if model == "hip_rotate" else 16)
    # job.device: cpu (BASELINE configs[0]): no HIP device -- the plugin classes subclass the reference's KgeModel,
    # and on CPU tensors they ARE the reference: KgeModel.score_* with the reference scorer's own score_emb
    assert not m._fused()
    ref_config = _config(model[4:])
    ref = KgeModel.create(ref_config, Dataset(ref_config, folder=None))
    ref.load_state_dict(m.state_dict())
    s, p, o = torch.tensor([1, 2]), torch.tensor([0, 3]), torch.tensor([5, 6])
    for call in (lambda mm: mm.score_sp(s, p), lambda mm: mm.score_po(p, o), lambda mm: mm.score_spo(s, p, o),
                 lambda mm: mm.score_sp_po(s, p, o), lambda mm: mm.score_so(s, o)):
        assert torch.equal(call(m), call(ref))
    # the engine itself has no CPU path (and the oracle is never one)
    from kge_amd import engine
    with pytest.raises(RuntimeError, match="no CPU path"):
        engine.Tables(model[4:], e, r)


@pytest.mark.parametrize("model", MODELS)
def test_penalty_shortcut_equals_the_reference_penalty(model):
    """The plugin models skip KgeModel.penalty's per-batch host-to-device copy when no penalty is configured
    (the default: regularize_weight 0) -- the reference then returns the empty list too -- and hand everything
    else to the reference's code: same names, same values as the reference model with the same weights."""
    config = _config(model)  # (imports the reference)
    from kge import Dataset
    from kge.model import KgeModel
    batch = {"triples": torch.tensor([[1, 0, 5], [2, 3, 6], [1, 1, 7]])}
    kw = dict(epoch=1, batch_index=0, num_batches=1, batch=batch)
    m = KgeModel.create(config, Dataset(config, folder=None))
    ref_config = _config(model[4:])
    ref = KgeModel.create(ref_config, Dataset(ref_config, folder=None))
    assert m.penalty(**kw) == [] and ref.penalty(**kw) == []
    for weighted in (False, True):
        cfgs = []
        for name in (model, model[4:]):
            c = _config(name)
            c.set("lookup_embedder.regularize_weight", 0.3)
            c.set("lookup_embedder.regularize_args.weighted", weighted)
            cfgs.append(c)
        m = KgeModel.create(cfgs[0], Dataset(cfgs[0], folder=None))
        ref = KgeModel.create(cfgs[1], Dataset(cfgs[1], folder=None))
        ref.load_state_dict(m.state_dict())
        got, want = m.penalty(**kw), ref.penalty(**kw)
        assert len(got) == len(want) > 0
        for (gk, gv), (wk, wv) in zip(got, want):
            assert gk.replace("hip_", "") == wk
            torch.testing.assert_close(gv, wv)


@pytest.mark.parametrize("model", ["hip_complex", "hip_distmult"])
def test_mixed_precision_option(model):
    """`score_dtype` defaults to float32 (no bf16 copies); bfloat16 selects the bf16 copies, which
    only exist on a GPU -- on job.device=cpu the request fails loudly like every scoring call."""
    config = _config(model)
    assert config.get(f"{model}.score_dtype") == "float32"
    from kge import Dataset
    from kge.model import KgeModel
    m = KgeModel.create(config, Dataset(config, folder=None))
    assert m._fwd_tables() is None
    config.set(f"{model}.score_dtype", "bfloat16")
    m = KgeModel.create(config, Dataset(config, folder=None))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m._fwd_tables()


def test_reference_checkpoint_state_loads_into_plugin_model():
    """Same shapes/names: a reference model's state_dict loads into the plugin class."""
    rh.import_reference()
    ref = rh.make_model("complex", 30, 4, 16)
    config = _config("hip_complex")
    from kge import Dataset
    from kge.model import KgeModel
    m = KgeModel.create(config, Dataset(config, folder=None))
    m.load_state_dict(ref.state_dict())
    assert torch.equal(m.get_s_embedder()._embeddings.weight, ref.get_s_embedder()._embeddings.weight)


def test_eval_job_plugin_resolves():
    config = _config("hip_distmult")
    config.set("eval.type", "hip_entity_ranking", create=True)
    config._import("hip_entity_ranking")
    assert config.get_default("hip_entity_ranking.class_name") == "HipEntityRankingJob"
    from kge.misc import init_from  # noqa: F401
    import kge_amd.libkge_plugin as plugin
    from kge.job import EntityRankingJob
    assert issubclass(plugin.HipEntityRankingJob, EntityRankingJob)


def test_training_job_plugin_resolves_and_declines_without_gpu():
    """train.type: hip_1vsAll resolves through the reference's factory lookup
    (train.py:127-137); without a GPU the models decline the fused loss (loss_sp -> None), so
    the job runs the reference's _process_subbatch -- whose scoring call then fails loudly."""
    config = _config("hip_complex")
    config.set("train.type", "hip_1vsAll")
    config._import("hip_1vsAll")
    assert config.get_default("hip_1vsAll.class_name") == "HipTrainingJob1vsAll"
    import kge_amd.libkge_plugin as plugin
    from kge.job.train_1vsAll import TrainingJob1vsAll
    assert issubclass(plugin.HipTrainingJob1vsAll, TrainingJob1vsAll)
    from kge import Dataset
    from kge.model import KgeModel
    m = KgeModel.create(config, Dataset(config, folder=None))
    s, p, o = torch.tensor([1, 2]), torch.tensor([0, 3]), torch.tensor([5, 6])
    assert m.loss_sp(s, p, o) is None and m.loss_po(p, o, s) is None


def _job_config(tmp, model, train_type, seed=7):
    rh.import_reference()
    import os
    from kge import Config
    config = Config()
    config.folder = os.path.join(tmp, f"run_{train_type}_{model}")
    os.makedirs(config.folder, exist_ok=True)
    config.set("console.quiet", True)
    config.set("modules", ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"])
    config.set("model", model)
    config._import(model)
    config.set("dataset.name", "dataset_test")
    config.set("job.device", "cpu")
    config.set("train.max_epochs", 1)
    config.set("train.batch_size", 32)
    config.set("train.num_workers", 0)
    config.set("lookup_embedder.dim", 16)
    config.set("random_seed.default", seed)
    if train_type.startswith("hip_"):
        config._import(train_type)
    config.set("train.type", train_type)
    return config


@pytest.mark.parametrize("loss", ["kl", "bce", "kl_smoothed", "bce_smoothed", "kl_s_o", "bce_smoothed_s_o"])
@pytest.mark.parametrize("ref_type,hip_type", [("1vsAll", "hip_1vsAll"), ("KvsAll", "hip_KvsAll")])
def test_fused_loss_jobs_follow_the_reference_jobs(tmp_path, ref_type, hip_type, loss):
    """Control flow of the plugin training jobs on CPU: with a model whose loss_sp / loss_po
    (kl_loss_sp / kl_loss_po) are the reference's own ops, one epoch of HipTrainingJob* must give
    the reference job's avg_loss and parameters -- batching, label CSR cut-out (KvsAll),
    loss scaling by the batch size, backward and optimizer steps are then the same."""
    import os
    import shutil
    import types
    import torch.nn.functional as F
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    from kge_amd.model import KgeModel as Mirror
    data = os.path.join(str(tmp_path), "dataset_test")  # the jobs write index caches next to the data
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), data)
    smoothing = 0.0
    s_o = loss.endswith("_s_o")   # KvsAll.query_types.s_o (round 6): relation targets beside the fused entity-target losses
    if s_o:
        if ref_type != "KvsAll":
            pytest.skip("query types are a KvsAll option")
        loss = loss[:-4]
    if loss.endswith("_smoothed"):  # KvsAll.label_smoothing (train_KvsAll.py:260-266): kl_/bce_loss_*'s last argument
        if ref_type != "KvsAll":
            pytest.skip("label smoothing is a KvsAll option")
        loss, smoothing = loss[:-9], 0.4  # dataset_test has 4 entities; the job wants > 1/E
    results = {}
    for train_type in (ref_type, hip_type):
        config = _job_config(str(tmp_path), "complex", train_type)
        config.set("train.loss", loss)
        config.set("KvsAll.label_smoothing", smoothing)
        if s_o:
            config.set("KvsAll.query_types.s_o", True)
        if loss == "bce":
            config.set("train.loss_arg", -0.5)  # score offset
        torch.manual_seed(11)  # same initialisation and batch order for both jobs
        job = TrainingJob.create(config, Dataset.create(config, folder=data))
        assert type(job).__name__ == ("TrainingJob" if not train_type.startswith("hip_") else "HipTrainingJob") + ref_type
        m = job.model
        if train_type.startswith("hip_"):
            m._ce_tables = lambda: object()  # "the fused loss applies", asked once per subbatch
            m.loss_sp = types.MethodType(
                lambda self, s, p, o: F.cross_entropy(self.score_sp(s, p), o.long(), reduction="none"), m)
            m.loss_po = types.MethodType(
                lambda self, p, o, s: F.cross_entropy(self.score_po(p, o), s.long(), reduction="none"), m)
            if train_type == "hip_1vsAll" and os.environ.get("KGE_TEST_TWO_SIDED", "1") == "1":
                m.loss_sp_po = types.MethodType(
                    lambda self, s, p, o: torch.cat([self.loss_sp(s, p, o), self.loss_po(p, o, s)]), m)
            m.kl_loss_sp = types.MethodType(
                lambda self, s, p, rp, col, ls=0.0: Mirror._kl_composed(self.score_sp(s, p), rp, col, ls), m)
            m.kl_loss_po = types.MethodType(
                lambda self, p, o, rp, col, ls=0.0: Mirror._kl_composed(self.score_po(p, o), rp, col, ls), m)
            m.bce_loss_sp = types.MethodType(
                lambda self, s, p, rp, col, off, ls=0.0: Mirror._bce_composed(self.score_sp(s, p), rp, col, off, ls), m)
            m.bce_loss_po = types.MethodType(
                lambda self, p, o, rp, col, off, ls=0.0: Mirror._bce_composed(self.score_po(p, o), rp, col, off, ls), m)
        job._prepare()
        trace = job.run_epoch()
        results[train_type] = (trace["avg_loss"], [x.detach().clone() for x in m.parameters()])
        if train_type == "hip_1vsAll" and loss == "kl":
            # hip_1vsAll.graph_step (default true) is for a HIP device only: on CPU the job decided against it at its
            # first batch and every optimizer step was the trainer's own
            assert config.get_default("hip_1vsAll.graph_step") is True
            assert job._graph_step is None and job._graph_step_ok is False and not job._skip_optimizer_step
    (l_ref, p_ref), (l_hip, p_hip) = results[ref_type], results[hip_type]
    assert abs(l_ref - l_hip) <= 1e-5 * max(1.0, abs(l_ref)), (l_ref, l_hip)
    for a, b in zip(p_ref, p_hip):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_optimizer_plugin_resolves_through_the_reference_factory():
    """train.optimizer.default.type: HipAdagrad -> KgeOptimizer.create (optimizer.py:15-20) finds the
    class the plugin registered on torch.optim; on CPU parameters it steps like torch's Adagrad."""
    config = _config("hip_distmult")
    import kge_amd.libkge_plugin  # noqa: F401  (registers torch.optim.HipAdagrad)
    from kge import Dataset
    from kge.model import KgeModel
    from kge.util import KgeOptimizer
    config.set("train.optimizer.default.type", "HipAdagrad")
    config.set("train.optimizer.default.args.lr", 0.1, create=True)
    m = KgeModel.create(config, Dataset(config, folder=None))
    opt = KgeOptimizer.create(config, m)
    assert type(opt).__name__ == "Adagrad" and type(opt).__module__ == "kge_amd.optim"
    ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    o_ref = torch.optim.Adagrad(ref, lr=0.1)
    for p, r in zip(m.parameters(), ref):
        g = torch.randn_like(p)
        p.grad, r.grad = g.clone(), g.clone()
    opt.step()
    o_ref.step()
    for p, r in zip(m.parameters(), ref):
        torch.testing.assert_close(p.detach(), r.detach())


def test_full_plugin_configuration_runs_an_epoch_and_a_validation_on_cpu(tmp_path):
    """Everything the plugin registers, switched on at once in one LibKGE config:
    train.type hip_1vsAll, optimizer HipAdagrad with bf16_copies, eval.type hip_entity_ranking.
    With the reference's own `complex` model on job.device=cpu every fused path declines or falls
    back, so this checks the plumbing: the factories resolve, an epoch trains, a validation
    produces the reference's metric keys."""
    import os
    import shutil
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    data = os.path.join(str(tmp_path), "dataset_test")
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), data)
    config = _job_config(str(tmp_path), "complex", "hip_1vsAll")
    config._import("hip_entity_ranking")
    config.set("eval.type", "hip_entity_ranking")
    config.set("train.optimizer.default.type", "HipAdagrad")
    config.set("train.optimizer.default.args.lr", 0.1, create=True)
    config.set("train.optimizer.default.args.bf16_copies", True, create=True)
    config.set("valid.every", 1)
    config.set("valid.metric", "mean_reciprocal_rank_filtered")
    torch.manual_seed(3)
    job = TrainingJob.create(config, Dataset.create(config, folder=data))
    assert type(job).__name__ == "HipTrainingJob1vsAll"
    assert type(job.optimizer).__module__ == "kge_amd.optim"
    assert type(job.valid_job).__name__ == "HipEntityRankingJob"
    job.run()
    assert job.epoch == 1 and len(job.valid_trace) == 1
    metrics = job.valid_trace[0]
    for key in ("mean_reciprocal_rank", "mean_reciprocal_rank_filtered", "hits_at_1_filtered"):
        assert key in metrics and 0.0 <= metrics[key] <= 1.0


@pytest.mark.parametrize("model", ["transe", "distmult"])
def test_negative_sampling_job_follows_the_reference_job(tmp_path, model):
    """Control flow of HipTrainingJobNegativeSampling on CPU: with a model whose `score_neg` is the
    reference's own composition (expanded triples -> score_spo, sampler.py:291-306), one epoch must
    give the reference job's avg_loss and parameters -- labels, slot loop, loss scaling, the
    positives' column and the relation slot (which stays with the sampler) are then the same."""
    import os
    import shutil
    import types
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    data = os.path.join(str(tmp_path), "dataset_test")
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), data)
    calls = []

    def score_neg(self, s, p, o, slot, neg):
        calls.append(slot)
        K = neg.shape[1]
        tr = [x.reshape(-1).long().repeat_interleave(K) for x in (s, p, o)]
        tr[slot] = neg.reshape(-1).long()
        return self.score_spo(tr[0], tr[1], tr[2]).view(-1, K)

    results = {}
    for train_type in ("negative_sampling", "hip_negative_sampling"):
        config = _job_config(str(tmp_path), model, train_type)
        config.set("negative_sampling.num_samples.s", 4)
        config.set("negative_sampling.num_samples.p", 2)
        config.set("negative_sampling.num_samples.o", 3)
        config.set("negative_sampling.implementation", "triple")
        torch.manual_seed(11)
        job = TrainingJob.create(config, Dataset.create(config, folder=data))
        assert type(job).__name__ == ("TrainingJobNegativeSampling" if train_type == "negative_sampling"
                                      else "HipTrainingJobNegativeSampling")
        if train_type.startswith("hip_"):
            job.model.score_neg = types.MethodType(score_neg, job.model)
        torch.manual_seed(12)
        job._prepare()
        trace = job.run_epoch()
        results[train_type] = (trace["avg_loss"], [x.detach().clone() for x in job.model.parameters()])
    assert set(calls) == {0, 2}  # subject and object slots through score_neg, the relation slot not
    (l_ref, p_ref), (l_hip, p_hip) = results["negative_sampling"], results["hip_negative_sampling"]
    assert abs(l_ref - l_hip) <= 1e-6 * max(1.0, abs(l_ref)), (l_ref, l_hip)
    for a, b in zip(p_ref, p_hip):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("shared_type", ["naive", "default"])
def test_negative_sampling_with_shared_samples_keeps_the_samplers_scoring(tmp_path, shared_type):
    """negative_sampling.shared: true -- the reference resolves implementation "auto" to "batch" and the shared
    sample objects score without ever materialising per-triple samples (NaiveSharedNegativeSample.samples even
    fails on a slice: sampler.py:412-426).  hip_negative_sampling must leave those objects alone: same loss and
    parameters as the reference job, score_neg never called."""
    import os
    import random
    import shutil
    import types
    import numpy as np
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    data = os.path.join(str(tmp_path), "dataset_test")
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), data)
    calls = []

    def score_neg(self, s, p, o, slot, neg):
        calls.append(slot)
        return None

    results = {}
    for train_type in ("negative_sampling", "hip_negative_sampling"):
        config = _job_config(str(tmp_path), "distmult", train_type)
        config.set("negative_sampling.num_samples.s", 4)
        config.set("negative_sampling.num_samples.o", 3)
        config.set("negative_sampling.shared", True)
        config.set("negative_sampling.shared_type", shared_type)
        torch.manual_seed(21)
        job = TrainingJob.create(config, Dataset.create(config, folder=data))
        if train_type.startswith("hip_"):
            job.model.score_neg = types.MethodType(score_neg, job.model)
        torch.manual_seed(22)
        np.random.seed(23)   # (the shared samplers draw with numpy / random: sampler.py:640-700)
        random.seed(24)
        job._prepare()
        trace = job.run_epoch()
        results[train_type] = (trace["avg_loss"], [x.detach().clone() for x in job.model.parameters()])
    assert calls == []
    (l_ref, p_ref), (l_hip, p_hip) = results["negative_sampling"], results["hip_negative_sampling"]
    assert abs(l_ref - l_hip) <= 1e-6 * max(1.0, abs(l_ref)), (l_ref, l_hip)
    for a, b in zip(p_ref, p_hip):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("model,train_type", [("complex", "1vsAll"), ("distmult", "KvsAll"),
                                              ("rotate", "negative_sampling"), ("transe", "negative_sampling")])
def test_hip_models_train_on_cpu_like_the_reference_models(tmp_path, model, train_type):
    """BASELINE.json configs[0] (examples/toy-complex-train.yaml with --job.device cpu; here on tests/data/dataset_test,
    the toy dataset is not shipped): `model: hip_<m>` trained and evaluated on CPU gives the losses, parameters and
    metrics of `model: <m>` -- without a HIP device the plugin classes are the reference classes they subclass."""
    import os
    import shutil
    rh.import_reference()
    from kge import Dataset
    from kge.job import TrainingJob
    data = os.path.join(str(tmp_path), "dataset_test")
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), data)
    results = {}
    for name in (model, "hip_" + model):
        config = _job_config(str(tmp_path), name, train_type)
        config.set("valid.every", 1)
        config.set("eval.batch_size", 16)
        torch.manual_seed(31)
        job = TrainingJob.create(config, Dataset.create(config, folder=data))
        torch.manual_seed(32)
        job.run()
        results[name] = (job.valid_trace[0], [x.detach().clone() for x in job.model.parameters()])
    (v_ref, p_ref), (v_hip, p_hip) = results[model], results["hip_" + model]
    for a, b in zip(p_ref, p_hip):
        assert torch.equal(a, b)
    for key in ("mean_reciprocal_rank", "mean_reciprocal_rank_filtered", "hits_at_1_filtered"):
        assert v_ref[key] == v_hip[key], key


@pytest.mark.parametrize("bce_type", [None, "mean", "self_adversarial"])
def test_ns_bce_stand_in_hands_everything_it_does_not_recognise_to_the_reference_loss(bce_type):
    """_HipNsBceLoss (the one-kernel stand-in hip_negative_sampling installs for the bce family on a GPU) wraps the
    job's BCEWithLogitsKgeLoss: CPU tensors and index labels reach the reference's object unchanged -- same value, same
    gradient -- and its attributes stay readable through the wrapper."""
    config = _config("hip_complex")   # (imports the reference and the plugin package)
    from kge.util.loss import BCEWithLogitsKgeLoss
    from kge_amd.libkge_plugin.train_job import _HipNsBceLoss, _fusable_ns_loss
    kw = {"temperature": 2.0} if bce_type == "self_adversarial" else {}
    ref = BCEWithLogitsKgeLoss(config, offset=0.25, bce_type=bce_type, **kw)
    assert _fusable_ns_loss(ref)
    w = _HipNsBceLoss(ref)
    assert w._offset == 0.25 and w.kind == {None: "bce", "mean": "bce_mean", "self_adversarial": "bce_self_adversarial"}[bce_type]
    g = torch.Generator().manual_seed(2)
    scores = torch.randn(9, 13, generator=g)
    labels = torch.zeros(9, 13)
    labels[:, 0] = 1
    a, b = scores.clone().requires_grad_(True), scores.clone().requires_grad_(True)
    la, lb = ref(a, labels, num_negatives=12), w(b, labels, num_negatives=12)   # CPU: the reference's loss
    la.backward()
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad) and w.fused_calls == 0
    if bce_type is not None:   # index labels (positions of the ones)
        idx = torch.zeros(9, dtype=torch.long)
        assert torch.equal(ref(scores, idx), w(scores, idx))
    weighted = BCEWithLogitsKgeLoss(config, offset=0.0, bce_type=bce_type, pos_weight=torch.tensor(2.0), **kw)
    assert not _fusable_ns_loss(weighted)


def test_an_arange_subset_is_recognised_as_a_range_of_the_table():
    """_FusedScoring._targets (round 6): the reference's EntityRankingJob hands an entity chunk over as
    torch.arange(chunk_start, chunk_end) (kge/job/eval_entity_ranking.py:216-229) -- recognised (one reduction, one host
    read) it becomes None (the whole table) or a Python range (kge_index.start: the all-entities kernels on those rows);
    anything else stays the listed subset it is."""
    rh.import_reference()
    import types
    from kge_amd.libkge_plugin.models import _FusedScoring
    E = 5000
    stub = types.SimpleNamespace(RANGE_MIN=_FusedScoring.RANGE_MIN, _w=lambda: (torch.zeros(E, 4), torch.zeros(3, 4)))
    t = lambda x: _FusedScoring._targets(stub, x)
    assert t(None) is None
    assert t(torch.arange(0, E)) is None
    assert t(torch.arange(0, E, dtype=torch.int32)) is None
    assert t(torch.arange(1000, 3000)) == range(1000, 3000)
    assert t(torch.arange(E - 1024, E)) == range(E - 1024, E)
    for keep in (torch.arange(10, 500),                                    # short: not worth a host read
                 torch.arange(0, 4000, 2),                                 # a stride
                 torch.arange(2000, 0, -1),                                # descending
                 torch.cat((torch.arange(0, 1500), torch.arange(1501, 3002))),   # one id missing: last - first != len - 1
                 torch.cat((torch.arange(0, 1500), torch.tensor([1499]), torch.arange(1500, 2999))),  # a duplicate
                 torch.arange(E - 1000, E + 500),                          # beyond the table
                 torch.arange(0, 2048).view(2, 1024)):                     # not a vector
        assert t(keep) is keep
    perm = torch.randperm(E)
    assert t(perm) is perm


def test_hip_reciprocal_wrapper_is_the_reference_wrapper_without_a_gpu():
    """`model: hip_reciprocal_relations_model` over hip_distmult on job.device cpu: the reference wrapper's parameter names
    (its checkpoints load), and -- without a HIP device the base model's fused path does not apply -- the reference
    wrapper's own scores bit for bit; the fused-loss hooks answer None (the hip_* jobs then run the reference's code)."""
    import os
    rh.import_reference()
    from kge import Config, Dataset
    from kge.model import KgeModel
    data = os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test")   # (the wrapper reads the relation ids)
    models = {}
    for name, base in (("reciprocal_relations_model", "distmult"), ("hip_reciprocal_relations_model", "hip_distmult")):
        config = Config()
        config.folder = None
        config.set("console.quiet", True)
        config.set("modules", ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"])
        config.set("model", name)
        config._import(name)
        config._import(base)
        config.set(f"{name}.base_model.type", base)
        config.set("dataset.name", "dataset_test")
        config.set("job.device", "cpu")
        config.set_all({"lookup_embedder.dim": 16})
        torch.manual_seed(5)
        models[name] = KgeModel.create(config, Dataset.create(config, folder=data))
    ref, hip = models["reciprocal_relations_model"], models["hip_reciprocal_relations_model"]
    assert type(hip).__name__ == "HipReciprocalRelationsModel" and type(hip._base_model).__name__ == "HipDistMult"
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys())
    hip.load_state_dict(ref.state_dict())
    E, R = ref.dataset.num_entities(), ref.dataset.num_relations()
    assert hip._base_model.get_p_embedder()._embeddings.weight.shape[0] == 2 * R   # p and p + R
    g = torch.Generator().manual_seed(6)
    s, p, o = (torch.randint(hi, (11,), generator=g) for hi in (E, R, E))
    sub = torch.tensor([E - 1, 1, 0])   # (the reference's test dataset: 4 entities, 3 relations)
    for call in (lambda m: m.score_sp(s, p), lambda m: m.score_po(p, o), lambda m: m.score_po(p, o, sub),
                 lambda m: m.score_sp_po(s, p, o), lambda m: m.score_sp_po(s, p, o, sub),
                 lambda m: m.score_spo(s, p, o, "s"), lambda m: m.score_spo(s, p, o, "o")):
        assert torch.equal(call(hip), call(ref))
    assert hip._ce_tables() is None and hip._dropout_only() is None
    assert hip.loss_sp(s, p, o) is None and hip.loss_po(p, o, s) is None and hip.loss_sp_po(s, p, o) is None
