"""Synthetic knowledge graphs in LibKGE's on-disk format.

There is no network in the build/bench environment and the reference does not
vendor its datasets (data/download_all.sh), so benches and parity tests run on
synthetic graphs of the same *shape* as the named datasets.  The
files written here follow the reference's format (dataset.yaml keys:
kge/config-default.yaml:71-116; example: tests/data/dataset_test/dataset.yaml;
tab separated int triples: kge/dataset.py:186-203) so the unmodified reference
jobs can load them too.
"""
import os

import numpy as np

# (num_entities, num_relations, train, valid, test) -- external facts, SURVEY.md 8
SHAPES = {
    "fb15k-237": (14541, 237, 272115, 17535, 20466),
    "wnrr": (40943, 11, 86835, 3034, 3134),
    "wikidata5m": (4594485, 822, 20614279, 5163, 5133),
}


def zipf_triples(num_entities, num_relations, n, rng, exponent=1.0):
    """n (s,p,o) triples; entity popularity ~ Zipf(exponent) so filters are non-trivial."""
    ranks = np.arange(1, num_entities + 1, dtype=np.float64)
    pe = ranks ** (-exponent)
    pe /= pe.sum()
    perm = rng.permutation(num_entities)
    s = perm[rng.choice(num_entities, size=n, p=pe)]
    o = perm[rng.choice(num_entities, size=n, p=pe)]
    p = rng.integers(0, num_relations, size=n)
    return np.stack([s, p, o], axis=1).astype(np.int32)


def make_splits(num_entities, num_relations, n_train, n_valid, n_test, seed=0):
    rng = np.random.default_rng(seed)
    t = zipf_triples(num_entities, num_relations, n_train + n_valid + n_test, rng)
    return {"train": t[:n_train], "valid": t[n_train:n_train + n_valid],
            "test": t[n_train + n_valid:]}


def write_libkge_dataset(folder, name, num_entities, num_relations, splits):
    """Write dataset.yaml + *.del files readable by the reference's Dataset.create."""
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "entity_ids.del"), "w") as f:
        for i in range(num_entities):
            f.write(f"{i}\te{i}\n")
    with open(os.path.join(folder, "relation_ids.del"), "w") as f:
        for i in range(num_relations):
            f.write(f"{i}\tr{i}\n")
    lines = ["dataset:",
             "  files.entity_ids.filename: entity_ids.del",
             "  files.entity_ids.type: map",
             "  files.relation_ids.filename: relation_ids.del",
             "  files.relation_ids.type: map"]
    for split, triples in splits.items():
        np.savetxt(os.path.join(folder, f"{split}.del"), triples, fmt="%d", delimiter="\t")
        lines += [f"  files.{split}.filename: {split}.del",
                  f"  files.{split}.size: {len(triples)}",
                  f"  files.{split}.type: triples"]
    lines += [f"  name: {name}", f"  num_entities: {num_entities}",
              f"  num_relations: {num_relations}"]
    with open(os.path.join(folder, "dataset.yaml"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return folder
