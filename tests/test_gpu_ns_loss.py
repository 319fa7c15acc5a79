"""kge_ns_bce_loss (BCEWithLogitsKgeLoss over a negative-sampling score block, loss + gradient in one kernel) against
the oracle's restatement of the reference's op sequence (oracle/torch_port.ns_bce_loss <- kge/util/loss.py:153-186:
offset, BCEWithLogitsLoss elements, `bce_mean`'s positive / negative averaging, the detached softmax weights of
`bce_self_adversarial`) in float32 on the same scores, and torch autograd through it for the gradient.  Floating point, another summation order: loss within
2e-6 relative, gradient within 1e-6 + 1e-5 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(scores, kind, offset, temperature):
    """oracle/torch_port.ns_bce_loss: the reference's op sequence (pinned to the reference's own loss object, bit for bit,
    by tests/test_oracle_vs_reference.py in the build container)."""
    import torch_port as tp
    return tp.ns_bce_loss(scores, kind, offset, temperature)


@pytest.mark.parametrize("kind,offset,temperature", [("bce", 0.0, 1.0), ("bce", 0.7, 1.0), ("bce_mean", 0.0, 1.0),
                                                      ("bce_mean", -1.5, 1.0), ("bce_self_adversarial", 0.0, 1.0),
                                                      ("bce_self_adversarial", 0.3, 0.5), ("bce_self_adversarial", 0.0, 4.0)])
@pytest.mark.parametrize("n,c", [(512, 1001), (77, 2), (1, 65), (300, 130)])
def test_ns_bce_loss_and_gradient(kind, offset, temperature, n, c):
    from kge_amd import engine as eng
    g = torch.Generator().manual_seed(n * 1000 + c)
    scores = (torch.randn(n, c, generator=g) * 6.0).to(DEV)          # saturating logits on both sides
    scores[0, :min(c, 3)] = torch.tensor([40.0, -40.0, 90.0])[:min(c, 3)].to(DEV)
    x = scores.clone().requires_grad_(True)
    want = _reference(x, kind, offset, temperature)
    want.backward()
    rows, grad = eng.ns_bce_loss(scores, kind, offset, temperature)
    got = float(rows.double().sum())
    assert abs(got - float(want)) <= 2e-6 * abs(float(want)) + 1e-6, (got, float(want))
    torch.testing.assert_close(grad, x.grad, rtol=1e-5, atol=1e-6)
    # a strided block (the job's `scores` is contiguous; a slice of a wider matrix here), forward only
    wide = torch.zeros(n, c + 5, device=DEV)
    wide[:, 2:2 + c] = scores
    rows2, none = eng.ns_bce_loss(wide[:, 2:2 + c], kind, offset, temperature, want_grad=False)
    assert none is None and torch.equal(rows2, rows)
