"""Import the LibKGE reference (read-only, /root/reference) as a CPU oracle.

TEST INFRASTRUCTURE ONLY.  Nothing under `kge_amd/` may import this module.
It only works inside the build container (the GPU box has no /root/reference);
it is used by `tests/golden/make_golden.py` to produce the committed golden
vectors and by `tests/test_oracle_vs_reference.py` (skipped when the reference
tree is absent) to pin the oracle restatement (`oracle/kge_oracle.c`,
`oracle/oracle.py`) to the live reference.

What is stubbed (oracle/ref_stubs/): numba, path, igraph, ConfigSpace, ax,
hpbandster -- none of them is on the score arithmetic path.
`torch.load` is patched to `weights_only=False` for the reference's checkpoint
loader (kge/util/io.py:41).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_STUBS = os.path.join(_HERE, "ref_stubs")


def _find_root():
    """KGE_REFERENCE_ROOT, else /root/reference (build container), else oracle/_ref/libkge: an
    UNTRACKED, git-ignored copy of the reference's `kge` package that tools/gpu_plugin.sh places
    there for the duration of one gpurun call, so that tests/test_gpu_libkge_plugin.py can drive
    the plugin through an unmodified LibKGE on the MI355X (reference sources are never committed)."""
    cands = [os.environ.get("KGE_REFERENCE_ROOT"), "/root/reference", os.path.join(_HERE, "_ref", "libkge")]
    for c in cands:
        if c and os.path.isdir(os.path.join(c, "kge")):
            return c
    return "/root/reference"


REFERENCE_ROOT = _find_root()


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "kge"))


def import_reference():
    """Put the reference + stubs on sys.path and import `kge`."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree
    for p in (_STUBS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import kge  # noqa: F401

    return kge


def make_config(model: str, dim: int, options=None):
    """Reference Config for `model` on CPU (mirrors reference tests/util.py:6-14)."""
    import_reference()
    from kge import Config

    config = Config()
    config.folder = None
    config.set("console.quiet", True)
    config.set("model", model)
    config._import(model)
    config.set("job.device", "cpu")
    config.set_all({"lookup_embedder.dim": dim})
    if options:
        config.set_all(options)
    return config


def make_model(model: str, num_entities: int, num_relations: int, dim: int,
               options=None, dataset=None):
    """Build a reference KgeModel (kge_model.py:472-503) without dataset files."""
    import_reference()
    from kge import Dataset
    from kge.model import KgeModel

    config = make_config(model, dim, options)
    if dataset is None:
        config.set("dataset.num_entities", num_entities)
        config.set("dataset.num_relations", num_relations)
        dataset = Dataset(config, folder=None)
    m = KgeModel.create(config, dataset)
    m.eval()
    return m


def set_tables(model, ent, rel):
    """Copy numpy/torch tables into the reference model's lookup embedders."""
    import torch

    with torch.no_grad():
        model.get_s_embedder()._embeddings.weight.copy_(torch.as_tensor(ent))
        model.get_p_embedder()._embeddings.weight.copy_(torch.as_tensor(rel))


def get_tables(model):
    return (
        model.get_s_embedder()._embeddings.weight.detach().clone(),
        model.get_p_embedder()._embeddings.weight.detach().clone(),
    )


def load_checkpoint(path: str, device: str = "cpu"):
    """kge.util.io.load_checkpoint (kge/util/io.py:29-45) under torch >= 2.6, whose torch.load defaults to
    weights_only=True and refuses the Config object a LibKGE checkpoint carries: the default is switched back for the
    duration of the call (the reference was written for torch 1.x)."""
    import_reference()
    import torch
    from kge.util.io import load_checkpoint as _load
    orig = torch.load

    def load(*a, **kw):
        kw.setdefault("weights_only", False)
        return orig(*a, **kw)
    torch.load = load
    try:
        return _load(path, device)
    finally:
        torch.load = orig
