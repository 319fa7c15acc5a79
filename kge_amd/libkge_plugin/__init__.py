"""LibKGE plugin: the fused gfx950 kernels behind the reference's own plugin API.

Usage inside an unmodified LibKGE installation (config-default.yaml:137, README.md:556-563):

    modules: [kge.job, kge.model, kge.model.embedder, kge_amd.libkge_plugin]
    model: hip_complex            # or hip_distmult / hip_transe / hip_rotate
    #   (or model: hip_reciprocal_relations_model with hip_reciprocal_relations_model.base_model.type: hip_complex)
    # optional: eval.type: hip_entity_ranking
    # optional: train.type: hip_1vsAll / hip_KvsAll  (kl / bce loss fused into the scoring kernel)
    #           train.type: hip_negative_sampling  (negatives through the fused gather + score kernel)
    # optional: train.type: hip_sharded_1vsAll / hip_sharded_KvsAll / hip_sharded_negative_sampling and
    #           eval.type: hip_sharded_entity_ranking  (entity table sharded over the ranks of a torch.distributed job:
    #           torchrun ... -m kge_amd.libkge_plugin.launch start cfg.yaml; sharded_job.py)
    # optional: train.optimizer.default.type: HipAdagrad / HipAdam  (one-pass update; args as for Adagrad / Adam,
    #           plus bf16_copies: true to keep the bf16 scoring tables fresh without a cast)

`Config._import("hip_complex")` finds hip_complex.yaml in this package (config.py:280-325)
and `init_from(class_name, modules)` (misc.py:13-42) resolves the classes below.  They
subclass the reference's KgeModel / RelationalScorer / EntityRankingJob, keep its parameter
names (`_entity_embedder._embeddings.weight`, ...) so optimizers, checkpoints and
ReciprocalRelationsModel work unchanged, and override only the score_* hot path.

This module needs the reference package `kge` to be importable; it is not used on the GPU
box of this repo's CI (no reference tree there) -- kge_amd.model mirrors the same API
stand-alone.
"""
try:
    import kge  # noqa: F401
except ImportError as e:  # pragma: no cover
    raise ImportError("kge_amd.libkge_plugin needs LibKGE (`kge`) to be importable") from e

from .models import (HipComplEx, HipComplExScorer, HipDistMult, HipDistMultScorer,  # noqa: F401
                     HipReciprocalRelationsModel, HipRotatE, HipRotatEScorer, HipTransE, HipTransEScorer)
from .eval_job import HipEntityRankingJob  # noqa: F401
from .train_job import (HipTrainingJob1vsAll, HipTrainingJobKvsAll,  # noqa: F401
                        HipTrainingJobNegativeSampling)
from .sharded_job import (HipShardedEntityRankingJob, HipShardedTrainingJob1vsAll,  # noqa: F401
                          HipShardedTrainingJobKvsAll, HipShardedTrainingJobNegativeSampling)

# kge/util/optimizer.py:15-20 resolves train.optimizer.default.type with getattr(torch.optim, ...)
import torch.optim as _torch_optim
from ..optim import Adagrad as HipAdagrad  # noqa: E402
from ..optim import Adam as HipAdam  # noqa: E402

_torch_optim.HipAdagrad = HipAdagrad
_torch_optim.HipAdam = HipAdam
