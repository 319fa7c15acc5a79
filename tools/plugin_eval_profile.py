#!/usr/bin/env python3
"""Where a batch of `eval.type: hip_entity_ranking` goes, through an unmodified LibKGE on the GPU (needs the reference
package on the box: tools/gpu_plugin.sh).  A 17,535-triple validation split (35 batches of 512) at the FB15k-237 shape,
DistMult d=512, trace_level epoch; cProfile of the second run of the job."""
import cProfile, os, pstats, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ref_harness as rh
rh.import_reference()
from kge import Config, Dataset
from kge.job import EvaluationJob
from kge.model import KgeModel
from kge_amd.synthetic import make_splits, write_libkge_dataset

E, R = 14541, 237
root = tempfile.mkdtemp(prefix="kge_evalprof_")
splits = make_splits(E, R, 272115, 17535, 2000, seed=3)
folder = write_libkge_dataset(os.path.join(root, "fbshape"), "fbshape", E, R, splits)
for eval_type, model in (("entity_ranking", "distmult"), ("hip_entity_ranking", "hip_distmult")):
    config = Config()
    config.folder = os.path.join(root, eval_type)
    os.makedirs(config.folder)
    config.set("console.quiet", True)
    config.set("modules", ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"])
    config.set("model", model); config._import(model)
    config._import("hip_entity_ranking")
    config.set("dataset.name", "fbshape"); config.set("job.device", "cuda")
    config.set("lookup_embedder.dim", 512)
    config.set("eval.type", eval_type); config.set("eval.batch_size", 512); config.set("eval.trace_level", "epoch")
    dataset = Dataset.create(config, folder=folder)
    m = KgeModel.create(config, dataset)
    job = EvaluationJob.create(config, dataset, parent_job=None, model=m)
    job.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); job.run(); torch.cuda.synchronize()
    print(f"{eval_type}: {time.perf_counter() - t0:.4f} s for 17,535 triples (second run)")
    if eval_type.startswith("hip"):
        pr = cProfile.Profile(); pr.enable(); job.run(); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
shutil.rmtree(root, ignore_errors=True)
