"""Live check of the oracles against the reference tree (build container only; skipped on
the GPU box where /root/reference does not exist)."""
import numpy as np
import pytest
import torch

import oracle as ko
import ref_harness as rh
import torch_port as tp

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")


@pytest.mark.parametrize("model,l_norm", [("complex", 1.0), ("distmult", 1.0), ("transe", 1.0),
                                          ("transe", 2.0), ("rotate", 1.0), ("rotate", 2.0)])
def test_torch_port_is_bit_identical_to_reference(model, l_norm):
    E, R, d, n = 211, 6, 64, 19
    torch.manual_seed(5)
    opts = {f"{model}.l_norm": l_norm} if model in ("transe", "rotate") else {}
    m = rh.make_model(model, E, R, d, options=opts)
    ent, rel = rh.get_tables(m)
    g = torch.Generator().manual_seed(1)
    s, p, o = torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g), torch.randint(E, (n,), generator=g)
    sub = torch.randperm(E, generator=g)[:50]
    with torch.no_grad():
        assert torch.equal(m.score_spo(s, p, o, "o"), tp.score_spo(model, ent, rel, s, p, o, l_norm))
        assert torch.equal(m.score_sp(s, p), tp.score_sp(model, ent, rel, s, p, None, l_norm))
        assert torch.equal(m.score_po(p, o, sub), tp.score_po(model, ent, rel, p, o, sub, l_norm))


@pytest.mark.parametrize("model", ["complex", "distmult", "transe", "rotate"])
def test_c_oracle_vs_live_reference_fb15k237_shape_slice(model):
    """A slice at the BASELINE shape (E=14541, d=512): scores within reference tolerance
    (atol relative to the score scale) and rank counts of the two implementations agree
    except where a score sits within f32 noise of the tie boundary."""
    E, R, d, n = 14541, 237, 512, 4
    torch.manual_seed(0)
    m = rh.make_model(model, E, R, d)
    ent, rel = rh.get_tables(m)
    g = torch.Generator().manual_seed(1)
    s, p, o = torch.randint(E, (n,), generator=g), torch.randint(R, (n,), generator=g), torch.randint(E, (n,), generator=g)
    with torch.no_grad():
        ref = m.score_sp(s, p).numpy()
    t = ko.Tables(model, ent.numpy(), rel.numpy(), 1.0)
    got = ko.score_sp(t, s.numpy(), p.numpy())
    scale = max(1.0, float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))
    np.testing.assert_allclose(got, ref, atol=1e-5 * scale, rtol=1e-4)
    true_ref = ref[np.arange(n), o.numpy()]
    true_got = got[np.arange(n), o.numpy()]
    r1, t1 = ko.rank_counts(ref, true_ref)
    r2, t2 = ko.rank_counts(got, true_got)
    assert np.abs((r1 + t1 // 2) - (r2 + t2 // 2)).max() <= 1


@pytest.mark.parametrize("kind,bce_type,offset,temperature", [("bce", None, 0.0, 1.0), ("bce", None, 0.5, 1.0),
                                                               ("bce_mean", "mean", 0.0, 1.0), ("bce_mean", "mean", -1.0, 1.0),
                                                               ("bce_self_adversarial", "self_adversarial", 0.0, 1.0),
                                                               ("bce_self_adversarial", "self_adversarial", 0.25, 3.0)])
def test_ns_bce_port_is_bit_identical_to_the_reference_loss(kind, bce_type, offset, temperature):
    """oracle/torch_port.ns_bce_loss against the reference's own BCEWithLogitsKgeLoss object (kge/util/loss.py:136-189)
    on the label matrix TrainingJobNegativeSampling builds: loss and gradient, bit for bit (the same torch ops)."""
    rh.import_reference()
    from kge.util.loss import BCEWithLogitsKgeLoss
    config = rh.make_config("complex", 16)
    kw = {"temperature": temperature} if bce_type == "self_adversarial" else {}
    ref = BCEWithLogitsKgeLoss(config, offset=offset, bce_type=bce_type, **kw)
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(37, 65, generator=g) * 5.0
    labels = torch.zeros(37, 65)
    labels[:, 0] = 1
    a = scores.clone().requires_grad_(True)
    b = scores.clone().requires_grad_(True)
    la, lb = ref(a, labels, num_negatives=64), tp.ns_bce_loss(b, kind, offset, temperature)
    la.backward()
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("loss,offset", [("kl", 0.0), ("bce", 0.0), ("bce", -0.75)])
@pytest.mark.parametrize("labels_as", ["index", "matrix", "smoothed"])
def test_training_loss_ports_are_bit_identical_to_the_reference_losses(loss, offset, labels_as):
    """oracle/torch_port.kl_loss / bce_loss (+ smooth_labels) against the reference's own loss objects
    (KLDivWithSoftmaxKgeLoss, BCEWithLogitsKgeLoss: kge/util/loss.py:137-159, 192-213) on the labels the 1vsAll job
    (indexes) and the KvsAll job (multi-hot matrix, smoothed or not: train_KvsAll.py:260-266) hand them: value and
    gradient bit for bit (the same torch ops), and the per-row form sums to the reduced one."""
    rh.import_reference()
    from kge.util.loss import BCEWithLogitsKgeLoss, KLDivWithSoftmaxKgeLoss
    config = rh.make_config("complex", 16)  # (job.device: cpu -- _labels_as_matrix allocates there)
    ref = KLDivWithSoftmaxKgeLoss(config) if loss == "kl" else BCEWithLogitsKgeLoss(config, offset=offset)
    g = torch.Generator().manual_seed(11)
    n, E = 29, 83
    scores = torch.randn(n, E, generator=g) * 4.0
    if labels_as == "index":
        labels = torch.randint(E, (n,), generator=g)
    else:
        labels = (torch.rand(n, E, generator=g) < 0.06).float()
        labels[3] = 0.0                      # a row without labels
        labels[5, :40] = 1.0                 # a row with many
        if labels_as == "smoothed":
            labels = tp.smooth_labels(labels, 0.1)
    a = scores.clone().requires_grad_(True)
    b = scores.clone().requires_grad_(True)
    la = ref(a, labels)
    lb = tp.kl_loss(b, labels) if loss == "kl" else tp.bce_loss(b, labels, offset)
    la.backward()
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad)
    rows = tp.kl_loss(scores, labels, "rows") if loss == "kl" else tp.bce_loss(scores, labels, offset, "rows")
    assert rows.shape == (n,)
    torch.testing.assert_close(rows.double().sum(), la.detach().double(), rtol=2e-6, atol=1e-6)

