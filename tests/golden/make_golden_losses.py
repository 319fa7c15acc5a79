"""Golden vectors of the TRAINING LOSSES from the live reference (build container only; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_losses.py

The reference's own loss objects -- KLDivWithSoftmaxKgeLoss and BCEWithLogitsKgeLoss (kge/util/loss.py:137-213), with the
label forms TrainingJob1vsAll (indexes), TrainingJobKvsAll (multi-hot matrix, smoothed or not: train_KvsAll.py:260-266)
and TrainingJobNegativeSampling (column 0 positive; bce, bce_mean, bce_self_adversarial) hand them -- on seeded scores:
losses.npz holds the inputs, the loss values and the gradients w.r.t. the scores.  tests/test_oracle_golden.py checks
oracle/torch_port.kl_loss / bce_loss / ns_bce_loss against it wherever the tests run (the reference does not travel)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.dont_write_bytecode = True

import ref_harness as rh  # noqa: E402

rh.import_reference()
import torch  # noqa: E402
from kge.util.loss import BCEWithLogitsKgeLoss, KLDivWithSoftmaxKgeLoss  # noqa: E402


def main():
    config = rh.make_config("complex", 16)
    g = torch.Generator().manual_seed(20260925)
    n, E, K = 23, 61, 40
    scores = (torch.randn(n, E, generator=g) * 3.0).float()
    idx = torch.randint(E, (n,), generator=g)
    multi = (torch.rand(n, E, generator=g) < 0.08).float()
    multi[2] = 0.0
    multi[7, :30] = 1.0
    smoothed = (1.0 - 0.1) * multi + 1.0 / multi.size(1)          # train_KvsAll.py:260-266 with label_smoothing 0.1
    block = (torch.randn(n, 1 + K, generator=g) * 4.0).float()    # a negative-sampling slot: column 0 = the positive
    ns_labels = torch.zeros(n, 1 + K)
    ns_labels[:, 0] = 1
    out = {"scores": scores.numpy(), "idx": idx.numpy(), "multi": multi.numpy(), "smoothed": smoothed.numpy(),
           "block": block.numpy()}

    def run(name, loss, x, labels, **kw):
        a = x.clone().requires_grad_(True)
        v = loss(a, labels, **kw)
        v.backward()
        out[name + "_value"] = np.float64(v.item())
        out[name + "_value_f32"] = v.detach().numpy()
        out[name + "_grad"] = a.grad.numpy()

    kl = KLDivWithSoftmaxKgeLoss(config)
    run("kl_index", kl, scores, idx)
    run("kl_multi", kl, scores, multi)
    run("kl_smoothed", kl, scores, smoothed)
    for off in (0.0, -0.75):
        bce = BCEWithLogitsKgeLoss(config, offset=off)
        tag = "bce" if off == 0.0 else "bce_off"
        run(tag + "_index", bce, scores, idx)
        run(tag + "_multi", bce, scores, multi)
        run(tag + "_smoothed", bce, scores, smoothed)
    for kind, bt, off, temp in (("bce", None, 0.25, 1.0), ("bce_mean", "mean", -1.0, 1.0),
                                ("bce_self_adversarial", "self_adversarial", 0.25, 3.0)):
        kw = {"temperature": temp} if bt == "self_adversarial" else {}
        run("ns_" + kind, BCEWithLogitsKgeLoss(config, offset=off, bce_type=bt, **kw), block, ns_labels, num_negatives=K)
    out["ns_params"] = np.array([[0.25, 1.0], [-1.0, 1.0], [0.25, 3.0]])
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("wrote", os.path.join(HERE, "losses.npz"), {k: float(v) for k, v in out.items() if k.endswith("_value")})


if __name__ == "__main__":
    main()
