"""The sharded engine behind LibKGE's own job factories (VERDICT r4 "e-plugin"): two gloo ranks run
`Job.create(config, dataset)` + `job.run()` of an UNMODIFIED LibKGE on the reference's own tests/data/dataset_test with
`train.type: hip_sharded_1vsAll | hip_sharded_KvsAll | hip_sharded_negative_sampling` and
`eval.type: hip_sharded_entity_ranking` (kge/job/train.py:118-137, kge/job/eval.py:35-48 resolve them by class_name from
`modules`), two epochs with a validation after each, and must equal the UNSHARDED run of the same config under
`hip_1vsAll / hip_KvsAll / hip_negative_sampling` + `hip_entity_ranking`: per-epoch loss, validation metrics, final
parameters; the checkpoint rank 0 wrote loads into an unsharded reference job and holds ONE [E, d] entity parameter
and the gathered optimizer state.

job.device cpu: the unsharded run is the reference's arithmetic (the plugin's CPU path IS the reference scorer); the
sharded ranks score with the test suite's stand-in backend (tests/test_sharded_gloo_cpu.OracleBackend, handed in through
sharded_job.SHARD_BACKEND -- on a GPU the default is kge_amd.engine).  Needs the reference package (/root/reference
here, oracle/_ref on a box where build() placed it)."""
import os
import shutil

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")

MODULES = ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"]
CASES = {
    # name: (model, dim, sharded train.type, unsharded train.type, extra options)
    "1vsAll-complex": ("hip_complex", 16, "hip_sharded_1vsAll", "hip_1vsAll", {}),
    "KvsAll-distmult-kl": ("hip_distmult", 16, "hip_sharded_KvsAll", "hip_KvsAll", {"train.loss": "kl"}),
    "KvsAll-complex-bce": ("complex", 16, "hip_sharded_KvsAll", "KvsAll", {"train.loss": "bce"}),
    # train.optimizer.<group> (kge/util/optimizer.py:28-95): the relation table in a group of its own (another learning
    # rate), which also puts it FIRST in torch's parameter numbering -- the checkpoint's optimizer state must follow
    "1vsAll-complex-optimizer-groups": ("complex", 16, "hip_sharded_1vsAll", "1vsAll", {
        "train.optimizer.relation.regex": ".*relation_embedder.*", "train.optimizer.relation.args.lr": 0.05}),
    # KvsAll.label_smoothing (train_KvsAll.py:260-266) under kl, against the REFERENCE model and job: the uniform part of
    # the smoothed labels is a per-row bias of the gradient kernel and ONE score per query against each shard's column sum
    "KvsAll-complex-kl-smoothed": ("complex", 16, "hip_sharded_KvsAll", "KvsAll", {
        "train.loss": "kl", "KvsAll.label_smoothing": 0.3, "lookup_embedder.initialize": "normal_",
        "lookup_embedder.initialize_args.normal_.mean": 0.0, "lookup_embedder.initialize_args.normal_.std": 0.1}),
    "negative_sampling-transe": ("hip_transe", 16, "hip_sharded_negative_sampling", "hip_negative_sampling",
                                 {"negative_sampling.num_samples.s": 7, "negative_sampling.num_samples.o": 5,
                                  "negative_sampling.implementation": "triple", "hip_transe.l_norm": 2.0}),
    "negative_sampling-rotate-bce": ("hip_rotate", 16, "hip_sharded_negative_sampling", "negative_sampling",
                                     {"negative_sampling.num_samples.s": 6, "negative_sampling.num_samples.o": 6,
                                      "negative_sampling.implementation": "triple", "train.loss": "bce"}),
    # ---- penalty terms (round 6; VERDICT r5 "e-plugin": BASELINE configs[0] sets regularize_weight 0.8e-7) ----
    # the options of the reference's examples/toy-complex-train.yaml (unweighted L2 at 0.8e-7, normal_ initialisation,
    # ReduceLROnPlateau) against the REFERENCE's own model and job
    "1vsAll-complex-toy-yaml": ("complex", 16, "hip_sharded_1vsAll", "1vsAll", {
        "lookup_embedder.regularize_weight": 0.8e-7, "lookup_embedder.initialize": "normal_",
        "lookup_embedder.initialize_args.normal_.mean": 0.0, "lookup_embedder.initialize_args.normal_.std": 0.1,
        "train.lr_scheduler": "ReduceLROnPlateau", "train.lr_scheduler_args.mode": "max",
        "train.lr_scheduler_args.patience": 4}),
    # unweighted L3 at weights where the terms move the parameters (entity term doubled: kge_model.py:620-625)
    "1vsAll-distmult-l3": ("distmult", 16, "hip_sharded_1vsAll", "1vsAll", {
        "lookup_embedder.regularize_args.p": 3, "distmult.entity_embedder.regularize_weight": 0.02,
        "distmult.relation_embedder.regularize_weight": 0.3}),
    # weighted N3 over complex coordinates (what LibKGE's tuned ComplEx configs use): the batch's unique rows x counts
    "KvsAll-complex-n3-weighted": ("complex", 16, "hip_sharded_KvsAll", "KvsAll", {
        "train.loss": "kl", "lookup_embedder.space": "complex", "lookup_embedder.regularize": "n3",
        "lookup_embedder.regularize_args.weighted": True, "complex.entity_embedder.regularize_weight": 0.5,
        "complex.relation_embedder.regularize_weight": 0.8}),
    # ---- the reference's reciprocal-relations wrapper (round 6): 2 R relation rows, the subject direction an sp_ query
    # with relation p + R; against the REFERENCE wrapper over the reference model under the reference's jobs
    # (normal_(0, 0.1) initialisation: with the default xavier_uniform_ on this 4-entity dataset a query's softmax
    # saturates, its relation row's gradient is ~1e-6 of rounding noise, and Adagrad's first step turns that noise's
    # SIGNS into +-lr: two correct implementations then differ by 2 lr in such a row)
    "1vsAll-reciprocal-complex": ("reciprocal_relations_model:complex", 16, "hip_sharded_1vsAll", "1vsAll", {
        "lookup_embedder.initialize": "normal_", "lookup_embedder.initialize_args.normal_.mean": 0.0,
        "lookup_embedder.initialize_args.normal_.std": 0.1}),
    "KvsAll-reciprocal-distmult-l2-weighted": ("reciprocal_relations_model:distmult", 16, "hip_sharded_KvsAll", "KvsAll", {
        "train.loss": "kl", "lookup_embedder.regularize_args.weighted": True,
        "distmult.entity_embedder.regularize_weight": 0.5, "distmult.relation_embedder.regularize_weight": 0.8,
        "lookup_embedder.initialize": "normal_", "lookup_embedder.initialize_args.normal_.mean": 0.0,
        "lookup_embedder.initialize_args.normal_.std": 0.1}),
    "negative_sampling-transe-l2-weighted": ("transe", 16, "hip_sharded_negative_sampling", "negative_sampling", {
        "negative_sampling.num_samples.s": 7, "negative_sampling.num_samples.o": 5,
        "negative_sampling.implementation": "triple", "lookup_embedder.regularize_args.weighted": True,
        "transe.entity_embedder.regularize_weight": 0.4, "transe.relation_embedder.regularize_weight": 0.1}),
}
PENALTY_CASES = [c for c in CASES if any("regularize" in k for k in CASES[c][4])]



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

def _dataset_dir(tmp):
    """A writable copy of the reference's tests/data/dataset_test (LibKGE drops .pckl caches beside the files)."""
    src = os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test")
    dst = os.path.join(tmp, "dataset_test")
    if not os.path.isdir(dst):
        shutil.copytree(src, dst)
    return dst


def _config(tmp, tag, model, dim, train_type, eval_type, extra):
    rh.import_reference()
    from kge import Config
    config = Config()
    config.folder = os.path.join(tmp, tag)
    shutil.rmtree(config.folder, ignore_errors=True)
    os.makedirs(config.folder)
    config.set("console.quiet", True)
    config.set("modules", MODULES)
    base = None
    if ":" in model:  # "reciprocal_relations_model:complex"
        model, base = model.split(":")
    config.set("model", model)
    config._import(model)
    if base is not None:
        config._import(base)
        config.set(f"{model}.base_model.type", base)
    for t in (train_type, eval_type):
        if t.startswith("hip_"):
            config._import(t)
    config.set("dataset.name", "dataset_test")
    config.set("job.device", "cpu")
    config.set("job.type", "train")
    config.set("train.type", train_type)
    config.set("eval.type", eval_type)
    config.set("train.max_epochs", 2)
    config.set("train.batch_size", 16)
    config.set("train.num_workers", 0)
    config.set("train.optimizer.default.type", "Adagrad")
    config.set("train.optimizer.default.args.lr", 0.2, create=True)
    config.set("eval.batch_size", 8)
    config.set("valid.every", 1)
    config.set("valid.metric", "mean_reciprocal_rank_filtered")
    config.set("lookup_embedder.dim", dim)
    for k in ("default", "torch", "numpy", "python"):
        config.set("random_seed." + k, 11)
    if train_type in ("hip_sharded_1vsAll", "hip_sharded_KvsAll"):
        config.set(train_type + ".score_dtype", "float32")
    for k, v in extra.items():
        config.set(k, v, create=True)
    return config


def _run(config, folder, like_sharded_seeding=False):
    """Job.create + run, as kge/cli.py:262-290 does; -> (per-epoch avg_loss, valid_trace, state_dict, job)."""
    import random
    from kge import Dataset
    from kge.job import Job
    from kge.util.seed import seed_from_config
    seed_from_config(config)
    dataset = Dataset.create(config, folder=folder)
    job = Job.create(config, dataset)
    losses = []
    job.post_epoch_hooks.append(lambda j: losses.append(j.current_trace["epoch"]["avg_loss"]))
    job.penalty_trace = []   # per epoch: {key: average value} as the trace shows it (train.py:417-436, 497-499)
    job.post_epoch_hooks.append(lambda j: j.penalty_trace.append(dict(j.current_trace["epoch"]["avg_penalties"])))
    if like_sharded_seeding:
        # the sharded jobs draw ONE number from rank 0's torch generator at job creation and seed the process-wide
        # generators at the start of every epoch from it and the epoch (sharded_job._seed_epoch): the same draw and
        # the same seeding here make the two runs see the same batches and the same negatives
        from kge_amd.libkge_plugin.sharded_job import epoch_seed
        base = int(torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64))
        orig = job.run_epoch

        def run_epoch():
            v = epoch_seed(base, job.epoch)
            torch.manual_seed(v)
            np.random.seed(v % (2 ** 32))
            random.seed(v)
            return orig()
        job.run_epoch = run_epoch
    job.run()
    return losses, job.valid_trace, {k: v.detach().clone() for k, v in job.model.state_dict().items()}, job


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        for pth in (here, os.path.join(os.path.dirname(here), "oracle"), os.path.dirname(here)):
            if pth not in sys.path:
                sys.path.insert(0, pth)
        rh.import_reference()
        from test_sharded_gloo_cpu import OracleBackend
        import kge_amd.libkge_plugin.sharded_job as sj
        sj.SHARD_BACKEND = OracleBackend
        model, dim, sharded_type, _, extra = CASES[case]
        config = _config(tmp, f"sharded_rank{rank}", model, dim, sharded_type, "hip_sharded_entity_ranking", extra)
        losses, valid, state, job = _run(config, _dataset_dir(tmp))  # (the job creates the gloo group from the env)
        assert dist.is_initialized() and dist.get_world_size() == world
        assert type(job).__name__.startswith("HipShardedTrainingJob")
        assert type(job.valid_job).__name__ == "HipShardedEntityRankingJob"
        sh = job._sh
        E = job.dataset.num_entities()
        assert (sh.lo, sh.hi) == ((0, (E + 1) // 2) if rank == 0 else ((E + 1) // 2, E))
        assert sh.ent_master.shape[0] == sh.hi - sh.lo                       # this rank trains its rows only
        for st in job.optimizer.state.values():                              # ... and holds their optimizer state only
            if torch.is_tensor(st.get("sum")) and st["sum"].dim() == 2 and st["sum"].shape[1] == dim:
                assert st["sum"].shape[0] in (sh.hi - sh.lo, job.dataset.num_relations() * (2 if ":" in model else 1))
        # stand-alone evaluation (kge valid / kge test of a checkpoint under the launcher): no parent training job -- the
        # job cuts this rank's rows out of the (gathered) model and must find the last validation's metrics
        from kge.job import EvaluationJob
        sh.sync_model()
        ev_conf = config.clone()
        ev_conf.set("job.type", "eval")
        ev_conf.set("eval.split", "valid")
        ev = EvaluationJob.create(ev_conf, job.dataset, parent_job=None, model=job.model)
        ev.epoch = 2
        alone = ev.run()
        assert type(ev).__name__ == "HipShardedEntityRankingJob" and getattr(ev, "_own_table", None) is not None
        for k in ("mean_reciprocal_rank_filtered", "mean_reciprocal_rank_filtered_with_test", "hits_at_1_filtered"):
            assert abs(alone[k] - valid[-1][k]) <= 1e-6 * max(1.0, abs(valid[-1][k])), (k, alone[k], valid[-1][k])
        # `kge resume` on the same two ranks: Job.create_from(rank 0's checkpoint) builds the sharded job again and
        # _load scatters the unsharded optimizer state onto this rank's rows
        from kge.job import Job
        dist.barrier()
        ck0 = os.path.join(tmp, "sharded_rank0", "checkpoint_00002.pt")
        re_conf = _config(tmp, f"resumed_rank{rank}", model, dim, sharded_type, "hip_sharded_entity_ranking", extra)
        resumed = Job.create_from(rh.load_checkpoint(ck0, "cpu"), new_config=re_conf, dataset=job.dataset)
        assert type(resumed).__name__ == type(job).__name__ and resumed.epoch == 2
        assert torch.equal(resumed._sh.ent_master.detach(), sh.ent_master.detach())
        for pa, pb in zip(resumed.optimizer.param_groups[0]["params"], job.optimizer.param_groups[0]["params"]):
            sa, sb = resumed.optimizer.state[pa], job.optimizer.state[pb]
            assert sa["sum"].shape == sb["sum"].shape and torch.allclose(sa["sum"], sb["sum"], rtol=0, atol=0)
        ck = os.path.join(config.folder, "checkpoint_00002.pt")
        q.put((rank, losses, [{k: v for k, v in t.items() if isinstance(v, (int, float))} for t in valid],
               {k: v.numpy() for k, v in state.items()}, os.path.exists(ck), ck, job.penalty_trace))
    finally:
        if dist.is_initialized():
            _quiet_teardown()


@pytest.mark.parametrize("case", list(CASES))
def test_sharded_jobs_through_job_create_equal_the_unsharded_plugin_jobs(case, tmp_path):
    tmp = str(tmp_path)
    _dataset_dir(tmp)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, case, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    # the unsharded run, in this process, while the ranks work
    model, dim, _, plain_type, extra = CASES[case]
    config = _config(tmp, "unsharded", model, dim, plain_type, "hip_entity_ranking", extra)
    l_ref, v_ref, s_ref, job_ref = _run(config, _dataset_dir(tmp), like_sharded_seeding=True)
    outs = []
    import time
    t0 = time.time()
    while len(outs) < world and time.time() - t0 < 600:
        if not q.empty():
            outs.append(q.get())
        elif any(pr.exitcode not in (None, 0) for pr in procs):
            break
        else:
            time.sleep(0.05)
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert len(outs) == world
    outs.sort(key=lambda x: x[0])
    assert len(l_ref) == 2 and len(v_ref) == 2
    for rank, losses, valid, state, has_ck, ck, pens in outs:
        np.testing.assert_allclose(losses, l_ref, rtol=2e-5, atol=1e-6)
        # the penalty terms of the trace: every shard's part summed == the reference's value over the whole table
        assert len(pens) == len(job_ref.penalty_trace) == 2
        for got, want in zip(pens, job_ref.penalty_trace):
            assert sorted(got) == sorted(want) and (len(want) >= 2) == (case in PENALTY_CASES), (got, want)
            for k in want:
                assert abs(got[k] - want[k]) <= 2e-5 * abs(want[k]) + 1e-12, (rank, k, got[k], want[k])
        assert len(valid) == 2
        for got, want in zip(valid, v_ref):
            keys = [k for k in want if k.startswith(("mean_", "hits_at_")) and isinstance(want[k], (int, float))]
            assert "mean_reciprocal_rank_filtered_with_test" in keys and "hits_at_1_filtered" in keys
            for k in keys:
                assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (rank, k, got[k], want[k])
        for k, v in s_ref.items():                          # the model every rank ends with = the unsharded model
            np.testing.assert_allclose(state[k], v.numpy(), rtol=2e-4, atol=5e-5)
        assert has_ck == (rank == 0)                        # rank 0 writes the checkpoints
    # rank 0's checkpoint is the reference's: an UNSHARDED job of an unmodified LibKGE resumes from it
    from kge.job import Job
    checkpoint = rh.load_checkpoint(outs[0][5], "cpu")
    E, R = job_ref.dataset.num_entities(), job_ref.dataset.num_relations()
    sd = checkpoint["model"][0]
    assert sd["_entity_embedder._embeddings.weight"].shape == (E, dim) and checkpoint["epoch"] == 2
    if ":" in model:  # the wrapper's relation table: p and p + R
        assert sd["_relation_embedder._embeddings.weight"].shape[0] == 2 * R
    opt_state = checkpoint["optimizer_state_dict"]["state"]
    ref_state = job_ref.optimizer.state_dict()["state"]
    ent_slots = [k for k, v in opt_state.items() if tuple(v["sum"].shape) == (E, dim)]
    assert len(ent_slots) == 1 and sorted(opt_state) == sorted(ref_state)
    for k in opt_state:
        np.testing.assert_allclose(opt_state[k]["sum"].numpy(), ref_state[k]["sum"].numpy(), rtol=2e-4, atol=1e-7)
    if case in PENALTY_CASES:
        assert all(v > 0 for v in job_ref.penalty_trace[-1].values())
    new = _config(tmp, "resumed", model, dim, plain_type, "hip_entity_ranking", extra)
    resumed = Job.create_from(checkpoint, new_config=new, dataset=job_ref.dataset)
    assert resumed.epoch == 2 and type(resumed).__name__ == type(job_ref).__name__
    for k, v in s_ref.items():
        np.testing.assert_allclose(resumed.model.state_dict()[k].numpy(), v.numpy(), rtol=2e-4, atol=5e-5)


def test_one_shard_job_with_embedder_dropout_equals_the_reference_job(tmp_path):
    """Embedder dropout under hip_sharded_1vsAll / hip_sharded_KvsAll (lookup_embedder.py:64-69, 102-105): with ONE
    shard the masks are drawn from the process's generator in the reference's order (kge_model.py:682-725: s rows, p
    rows, all entities for sp_; all entities, o rows, p rows for _po), so the job must take the reference job's steps
    -- `complex` + `1vsAll` / `KvsAll` with entity dropout 0.3 and relation dropout 0.2, plus a penalty term --: epoch
    losses, penalties, validation metrics, final parameters.  (Two shards draw their table masks per rank: the masks
    handed in, tests/test_sharded_gloo_cpu.py.)"""
    rh.import_reference()
    import kge_amd.libkge_plugin.sharded_job as sj
    from test_sharded_gloo_cpu import OracleBackend
    tmp = str(tmp_path)
    extra = {"complex.entity_embedder.dropout": 0.3, "complex.relation_embedder.dropout": 0.2,
             "lookup_embedder.regularize_weight": 0.01}
    for sharded_type, plain_type, more in (("hip_sharded_1vsAll", "1vsAll", {}),
                                           ("hip_sharded_KvsAll", "KvsAll", {"train.loss": "kl"}),
                                           ("hip_sharded_KvsAll", "KvsAll", {"train.loss": "kl", "KvsAll.label_smoothing": 0.3}),
                                           ("hip_sharded_KvsAll", "KvsAll", {"train.loss": "bce"})):
        opts = dict(extra, **more)
        config = _config(tmp, "dropout_ref", "complex", 16, plain_type, "entity_ranking", opts)
        l_ref, v_ref, s_ref, job_ref = _run(config, _dataset_dir(tmp), like_sharded_seeding=True)
        assert job_ref.model.get_s_embedder().dropout.p == 0.3 and job_ref.model.get_p_embedder().dropout.p == 0.2
        sj.SHARD_BACKEND = OracleBackend
        try:
            config = _config(tmp, "dropout_sharded", "complex", 16, sharded_type, "hip_sharded_entity_ranking", opts)
            l_got, v_got, s_got, job = _run(config, _dataset_dir(tmp))
        finally:
            sj.SHARD_BACKEND = None
        assert type(job).__name__.startswith("HipShardedTrainingJob") and job._sh.world == 1 and job._dropout == (0.3, 0.2)
        np.testing.assert_allclose(l_got, l_ref, rtol=2e-5, atol=1e-6)
        for got, want in zip(job.penalty_trace, job_ref.penalty_trace):
            for k in want:
                assert abs(got[k] - want[k]) <= 2e-5 * abs(want[k]), (k, got[k], want[k])
        for got, want in zip(v_got, v_ref):
            for k in ("mean_reciprocal_rank_filtered", "hits_at_1_filtered"):
                assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])
        for k, v in s_ref.items():
            np.testing.assert_allclose(s_got[k].numpy(), v.numpy(), rtol=2e-4, atol=5e-5)


def test_launcher_runs_kge_start_on_two_ranks(tmp_path):
    """`torchrun --nproc-per-node 2 -m kge_amd.libkge_plugin.launch start cfg.yaml --folder F --job.device cpu` (through
    tests/_launch_with_oracle_backend.py, which hands the CPU stand-in backend to the plugin and calls that launcher): an
    unmodified kge.cli underneath, rank 0 in F, rank 1 in F-rank1, checkpoints written by rank 0 only, the training
    trace of both ranks carrying the same losses -- and the same as the in-process run of the case above."""
    import subprocess
    import sys
    import yaml
    tmp = str(tmp_path)
    # a dataset the CLI can find by name: <base dir of a root module in `modules`>/data/<name> (kge/dataset.py:103-112)
    os.makedirs(os.path.join(tmp, "dsmod"))
    open(os.path.join(tmp, "dsmod", "__init__.py"), "w").close()
    shutil.copytree(os.path.join(rh.REFERENCE_ROOT, "tests", "data", "dataset_test"), os.path.join(tmp, "data", "dataset_test"))
    cfg = {
        "modules": MODULES + ["dsmod"], "model": "hip_complex", "import": ["hip_complex", "hip_sharded_1vsAll",
                                                                           "hip_sharded_entity_ranking"],
        "dataset": {"name": "dataset_test"}, "job": {"type": "train"},
        "train": {"type": "hip_sharded_1vsAll", "max_epochs": 2, "batch_size": 16, "num_workers": 0,
                  "optimizer": {"default": {"type": "Adagrad", "args": {"lr": 0.2}}}},
        "hip_sharded_1vsAll": {"score_dtype": "float32"},
        "eval": {"type": "hip_sharded_entity_ranking", "batch_size": 8},
        "valid": {"every": 1, "metric": "mean_reciprocal_rank_filtered"},
        "lookup_embedder": {"dim": 16}, "random_seed": {"default": 11, "torch": 11, "numpy": 11, "python": 11},
    }
    path = os.path.join(tmp, "cfg.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle", "ref_stubs"), rh.REFERENCE_ROOT, root, here,
                                           os.path.join(root, "oracle"), tmp, os.environ.get("PYTHONPATH", "")]))
    folder = os.path.join(tmp, "run")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(here, "_launch_with_oracle_backend.py"), "start", path,
           "--folder", folder, "--job.device", "cpu", "--console.quiet", "True"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert os.path.isfile(os.path.join(folder, "checkpoint_00002.pt")) and os.path.isfile(os.path.join(folder, "checkpoint_best.pt"))
    assert os.path.isdir(folder + "-rank1") and not os.path.exists(os.path.join(folder + "-rank1", "checkpoint_00002.pt"))

    def epoch_losses(trace_file):
        import re
        out = []
        with open(trace_file) as f:
            for line in f:  # (one flow-style yaml mapping per line; some carry python tags safe_load refuses)
                if "scope: epoch" in line and "job: train" in line:
                    m = re.search(r"avg_loss: ([-+0-9.eE]+)", line)
                    if m:
                        out.append(float(m.group(1)))
        return out
    l0 = epoch_losses(os.path.join(folder, "trace.yaml"))
    l1 = epoch_losses(os.path.join(folder + "-rank1", "trace.yaml"))
    assert len(l0) == 2 and l0 == l1, (l0, l1)
    ck = rh.load_checkpoint(os.path.join(folder, "checkpoint_00002.pt"), "cpu")
    assert ck["model"][0]["_entity_embedder._embeddings.weight"].shape[1] == 16 and ck["epoch"] == 2


def test_the_plugin_imports_nothing_named_by_the_environment():
    """until round 6 `sharded_job._backend_for` imported `module:attr` from KGE_AMD_TEST_SHARD_BACKEND.
    The package reads the environment for switches of its own behaviour only -- never for a module, attribute or path
    to import: no importlib / __import__ / exec / eval next to an environment read anywhere under kge_amd/."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kge_amd")
    offenders = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            if "KGE_AMD_TEST_SHARD_BACKEND" in src:
                offenders.append((f, "KGE_AMD_TEST_SHARD_BACKEND"))
            if re.search(r"importlib\.import_module|__import__\(|\bexec\(|\beval\(", src) and "os.environ" in src:
                for m in re.finditer(r"importlib\.import_module\(([^)]*)\)|__import__\(([^)]*)\)", src):
                    arg = (m.group(1) or m.group(2) or "")
                    if "environ" in arg or "getenv" in arg or "spec" in arg:
                        offenders.append((f, m.group(0)))
    assert offenders == [], offenders
    import kge_amd.libkge_plugin.sharded_job as sj
    os.environ["KGE_AMD_TEST_SHARD_BACKEND"] = "os:path"
    try:
        assert sj._backend_for("cpu") is sj.SHARD_BACKEND
    finally:
        del os.environ["KGE_AMD_TEST_SHARD_BACKEND"]


def test_sharded_negative_sampling_refuses_the_reciprocal_wrapper(tmp_path):
    """A reciprocal-relations wrapper's embedders ARE its base model's plain lookup tables: an embedder check alone would
    pass and the job would score the subject direction with relation p instead of p + R -- another model, silently.
    hip_sharded_1vsAll / _KvsAll / _entity_ranking translate the direction (the cases above); negative sampling, which
    does not, refuses the wrapper by type."""
    rh.import_reference()
    from kge import Dataset
    from kge.job import Job
    tmp = str(tmp_path)
    for wrapper, base in (("reciprocal_relations_model", "transe"), ("hip_reciprocal_relations_model", "hip_transe")):
        config = _config(tmp, "recip_" + wrapper, wrapper + ":" + base, 16, "hip_sharded_negative_sampling",
                         "hip_sharded_entity_ranking", {"negative_sampling.implementation": "triple"})
        with pytest.raises(ValueError, match="reciprocal-relations wrapper"):
            Job.create(config, Dataset.create(config, folder=_dataset_dir(tmp)))
