"""GPU: DCMHT.object_function / our_loss / similarity_loss / soft_argmax_hash_loss (xmh_loss.hip) against goldens produced by
the reference's own our_loss, and against the oracle on other shapes."""
import numpy as np
import pytest
import torch

from test_oracle_losses import CASES, ORDER, load

pytestmark = pytest.mark.gpu


def _model(K, sim, vartheta=0.75, threshold=0.1, alpha=0.001):
    from xmh.models.dcmht import DCMHT
    m = DCMHT.__new__(DCMHT)                                   # the loss methods read attributes only; no backbone needed here
    torch.nn.Module.__init__(m)
    m.output_dim, m.vartheta, m.threshold, m.similarity_function, m.quan_alpha = K, vartheta, threshold, sim, alpha
    return m


def _vec(loss, d):
    return np.array([float(loss), float(d["Intra"]["Positive"]), float(d["Intra"]["Negative"]), float(d["Inter"]["Positive"]["i2t"]),
                     float(d["Inter"]["Negative"]["i2t"]), float(d["Inter"]["Positive"]["t2i"]), float(d["Inter"]["Negative"]["t2i"]),
                     float(d["Quan"]["Image"]), float(d["Quan"]["Text"])])


@pytest.mark.parametrize("name", CASES)
def test_loss_forward_matches_the_reference(name):
    img, txt, labels, K, sim, vartheta, threshold, alpha, ref = load(name)
    m = _model(K, sim, vartheta, threshold, alpha)
    loss, d = m.object_function(img.cuda(), txt.cuda(), labels=None if labels is None else labels.cuda())
    assert loss.is_cuda and loss.dim() == 0
    assert np.allclose(_vec(loss, d), ref, rtol=2e-5, atol=1e-6), (name, _vec(loss, d), ref)


def test_loss_forward_against_the_oracle_on_other_shapes():
    from oracle import losses as OL
    g = torch.Generator().manual_seed(7)
    for B, K, C, sim in ((128, 128, 80, "euclidean"), (33, 16, 21, "cosine"), (128, 2048, 24, "euclidean"), (5, 64, 3, "euclidean")):
        img = torch.softmax(torch.randn(B, K, 2, generator=g), -1).reshape(B, 2 * K)
        txt = torch.softmax(torch.randn(B, K, 2, generator=g), -1).reshape(B, 2 * K)
        labels = (torch.rand(B, C, generator=g) < 0.15).float()
        labels[torch.arange(B), torch.randint(0, C, (B,), generator=g)] = 1.0
        m = _model(K, sim)
        loss, d = m.our_loss(img.cuda(), txt.cuda(), labels.cuda())
        want = OL.our_loss(img, txt, labels, K, similarity_function=sim)
        assert np.allclose(_vec(loss, d), np.array([float(want[k]) for k in ORDER]), rtol=2e-5, atol=1e-6), (B, K, sim)


def test_loss_rejects_cpu_tensors_and_bad_shapes():
    m = _model(16, "euclidean")
    with pytest.raises(RuntimeError):
        m.soft_argmax_hash_loss(torch.rand(4, 32))
    with pytest.raises(ValueError):
        m.similarity_loss(torch.rand(4, 32).cuda(), torch.rand(5, 32).cuda(), torch.ones(4, 3).cuda())
