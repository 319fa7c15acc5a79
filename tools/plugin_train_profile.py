#!/usr/bin/env python3
"""One warm epoch of TrainingJob1vsAll through an unmodified LibKGE on the GPU at the FB15k-237 size (272,115 train
triples = 532 batches of 512, ComplEx d=512): the reference model + job against the fully fused plugin configuration
(hip_complex, hip_1vsAll, bf16 scoring, HipAdagrad with bf16 copies), then a cProfile of the latter.  Needs the reference
package on the box (tools/gpu_plugin.sh)."""
import cProfile, os, pstats, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ref_harness as rh
rh.import_reference()
from kge import Config, Dataset
from kge.job import TrainingJob
from kge_amd.synthetic import make_splits, write_libkge_dataset

E, R = 14541, 237
root = tempfile.mkdtemp(prefix="kge_trainprof_")
splits = make_splits(E, R, 272115, 2000, 2000, seed=3)
folder = write_libkge_dataset(os.path.join(root, "fbshape"), "fbshape", E, R, splits)
FUSED = {"train.optimizer.default.type": "HipAdagrad", "train.optimizer.default.args.bf16_copies": True}
NS = {"negative_sampling.num_samples.s": 100, "negative_sampling.num_samples.o": 100, "lookup_embedder.dim": 256}
MODE = sys.argv[1] if len(sys.argv) > 1 else "1vsAll"
CASES = [("reference: complex + 1vsAll", "complex", "1vsAll", {}),
         ("plugin, f32 kernels: hip_complex + 1vsAll", "hip_complex", "1vsAll", {}),
         ("plugin, fused: hip_complex + hip_1vsAll + bf16 scoring + HipAdagrad", "hip_complex", "hip_1vsAll",
          {"hip_complex.score_dtype": "bfloat16", "train.optimizer.default.type": "HipAdagrad",
           "train.optimizer.default.args.bf16_copies": True})]
if MODE == "KvsAll":
    CASES = [("reference: complex + KvsAll", "complex", "KvsAll", {}),
             ("plugin, fused: hip_complex + hip_KvsAll + bf16 scoring + HipAdagrad", "hip_complex", "hip_KvsAll",
              {"hip_complex.score_dtype": "bfloat16", **FUSED}),
             ("plugin, fused, label_smoothing 0.1", "hip_complex", "hip_KvsAll",
              {"hip_complex.score_dtype": "bfloat16", "KvsAll.label_smoothing": 0.1, **FUSED})]
elif MODE == "ns":
    CASES = [("reference: rotate + negative_sampling (2 x 100 negatives, d = 256)", "rotate", "negative_sampling", NS),
             ("plugin: hip_rotate + negative_sampling", "hip_rotate", "negative_sampling", NS),
             ("plugin, fused: hip_rotate + hip_negative_sampling", "hip_rotate", "hip_negative_sampling", NS),
             ("reference: transe + negative_sampling", "transe", "negative_sampling", NS),
             ("plugin, fused: hip_transe + hip_negative_sampling", "hip_transe", "hip_negative_sampling", NS)]
if os.environ.get("KGE_PROFILE_SKIP_REF") == "1":  # (the reference RotatE epoch alone takes 28 s)
    CASES = [c for c in CASES if not c[0].startswith("reference")]
for name, model, ttype, opts in CASES:
    config = Config()
    config.folder = os.path.join(root, ttype + "_" + model + str(len(opts)))
    os.makedirs(config.folder)
    config.set("console.quiet", True)
    config.set("modules", ["kge.job", "kge.model", "kge.model.embedder", "kge_amd.libkge_plugin"])
    config.set("model", model); config._import(model)
    if ttype.startswith("hip_"):
        config._import(ttype)
    config.set("dataset.name", "fbshape"); config.set("job.device", "cuda")
    config.set("lookup_embedder.dim", 512)
    config.set("train.type", ttype); config.set("train.batch_size", 512); config.set("train.num_workers", 0)
    config.set("train.max_epochs", 1); config.set("valid.every", 0)
    for k, v in opts.items():
        config.set(k, v, create=True)
    torch.manual_seed(1)
    dataset = Dataset.create(config, folder=folder)
    job = TrainingJob.create(config, dataset)
    job._prepare(); job._is_prepared = True
    job.run_epoch(); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr = job.run_epoch(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nb = tr.get("batches", 532) if isinstance(tr.get("batches"), int) else 532
    print(f"{name}: {el:.3f} s per epoch = {el / max(1, len(job.loader)) * 1e3:.3f} ms per batch ({len(job.loader)} batches); avg_loss {tr['avg_loss']:.4f}; "
          f"forward {tr.get('forward_time', 0):.3f} backward {tr.get('backward_time', 0):.3f} "
          f"optimizer {tr.get('optimizer_time', 0):.3f} prepare {tr.get('prepare_time', 0):.3f} s", flush=True)
    if "fused" in name and "smoothing" not in name:
        pr = cProfile.Profile(); pr.enable(); job.run_epoch(); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(24)
shutil.rmtree(root, ignore_errors=True)
