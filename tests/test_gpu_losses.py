"""GPU: DCMHT.object_function / our_loss / similarity_loss / soft_argmax_hash_loss (xmh_loss.hip) against goldens produced by
the reference's own our_loss, and against the oracle on other shapes; loss.backward() through the gradient kernels against the
gradients the reference's own backward produced (same goldens) and against the oracle's float64 autograd."""
import numpy as np
import pytest
import torch

from test_oracle_losses import CASES, ORDER, grads_close, load, load_grads

pytestmark = pytest.mark.gpu


def _model(K, sim, vartheta=0.75, threshold=0.1, alpha=0.001):
    from xmh.models.dcmht import DCMHT
    m = DCMHT.__new__(DCMHT)                                   # the loss methods read attributes only; no backbone needed here
    torch.nn.Module.__init__(m)
    m.output_dim, m.vartheta, m.threshold, m.similarity_function, m.quan_alpha = K, vartheta, threshold, sim, alpha
    return m


def _vec(loss, d):
    return np.array([float(loss.detach()), float(d["Intra"]["Positive"]), float(d["Intra"]["Negative"]), float(d["Inter"]["Positive"]["i2t"]),
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


@pytest.mark.parametrize("name", CASES)
def test_loss_backward_matches_the_reference(name):
    """runners/DCMHT/runner.py:124 loss.backward(): d loss / d img_hash and d loss / d txt_hash"""
    img, txt, labels, K, sim, vartheta, threshold, alpha, ref = load(name)
    m = _model(K, sim, vartheta, threshold, alpha)
    gi, gt = img.cuda().requires_grad_(True), txt.cuda().requires_grad_(True)
    loss, d = m.object_function(gi, gt, labels=None if labels is None else labels.cuda())
    assert loss.requires_grad and not d["All loss"].requires_grad and not d["Intra"]["Positive"].requires_grad
    assert np.allclose(_vec(loss, d), ref, rtol=2e-5, atol=1e-6)
    loss.backward()
    ri, rt = load_grads(name)
    assert gi.grad.shape == img.shape and gt.grad.shape == txt.shape
    assert grads_close(gi.grad.cpu().numpy(), ri), (name, np.abs(gi.grad.cpu().numpy() - ri).max(), np.abs(ri).max())
    assert grads_close(gt.grad.cpu().numpy(), rt), (name, np.abs(gt.grad.cpu().numpy() - rt).max(), np.abs(rt).max())


def test_loss_backward_against_the_oracle_on_other_shapes_and_through_a_graph():
    from oracle import losses as OL
    g = torch.Generator().manual_seed(11)
    for B, K, C, sim in ((128, 64, 80, "euclidean"), (33, 16, 21, "cosine"), (128, 1024, 24, "euclidean"), (5, 64, 3, "cosine")):
        logit_i, logit_t = torch.randn(B, K, 2, generator=g), torch.randn(B, K, 2, generator=g)
        labels = (torch.rand(B, C, generator=g) < 0.15).float()
        labels[torch.arange(B), torch.randint(0, C, (B,), generator=g)] = 1.0
        m = _model(K, sim)
        # the codes come out of a softmax in the graph ([B, K, 2], as the soft-argmax head emits them) and the loss is scaled
        # afterwards: the upstream gradient and the input shape both go through the Function
        li, lt = logit_i.cuda().requires_grad_(True), logit_t.cuda().requires_grad_(True)
        loss, _ = m.our_loss(torch.softmax(li, -1), torch.softmax(lt, -1), labels.cuda())
        (3.0 * loss).backward()
        di, dt = logit_i.double().requires_grad_(True), logit_t.double().requires_grad_(True)
        want = OL.our_loss(torch.softmax(di, -1).reshape(B, 2 * K), torch.softmax(dt, -1).reshape(B, 2 * K), labels, K, similarity_function=sim)
        (3.0 * want["loss"]).backward()
        assert grads_close(li.grad.cpu().numpy(), di.grad.numpy()), (B, K, sim, float((li.grad.cpu().double() - di.grad).abs().max()))
        assert grads_close(lt.grad.cpu().numpy(), dt.grad.numpy()), (B, K, sim)
    # only one side asks for a gradient
    a = torch.rand(6, 32, generator=g).cuda().requires_grad_(True)
    b = torch.rand(6, 32, generator=g).cuda()
    loss, _ = _model(16, "euclidean").object_function(a, b)
    loss.backward()
    assert a.grad is not None and b.grad is None
    # no gradient requested: same numbers, no graph
    loss2, _ = _model(16, "euclidean").object_function(a.detach(), b)
    assert not loss2.requires_grad and float(loss2) == float(loss)


def test_gradient_entry_points_write_and_accumulate():
    from xmh import retrieval as R
    from xmh._lib import check, current_stream, lib, ptr
    g = torch.Generator().manual_seed(3)
    B, D, C = 24, 48, 10
    a, b = torch.rand(B, D, generator=g).cuda(), torch.rand(B, D, generator=g).cuda()
    labels = (torch.rand(B, C, generator=g) < 0.3).float()
    lab = R.pack_labels(labels.cuda())
    up = torch.tensor([0.5], device="cuda")
    g1, g2 = torch.empty(B, D, device="cuda"), torch.full((B, D), 7.0, device="cuda")
    args = (ptr(a), ptr(b), B, D, ptr(lab), C, 0, 6.0, 0.1)
    check(lib.xmh_pair_similarity_loss_grad(*args, 1.0, None, ptr(g1), 0, current_stream()), "grad")
    check(lib.xmh_pair_similarity_loss_grad(*args, 2.0, ptr(up), ptr(g2), 1, current_stream()), "grad")
    assert torch.allclose(g2, 7.0 + g1, rtol=1e-6, atol=1e-7)                  # scale 2 x upstream 0.5, accumulated
    q1, q2 = torch.empty(B, D, device="cuda"), torch.ones(B, D, device="cuda")
    check(lib.xmh_quant_loss_grad(ptr(a), a.numel(), 1.0, None, ptr(q1), 0, current_stream()), "qgrad")
    check(lib.xmh_quant_loss_grad(ptr(a), a.numel(), 2.0, ptr(up), ptr(q2), 1, current_stream()), "qgrad")
    assert torch.allclose(q1, -4.0 * (2 * a - 1) / a.numel(), rtol=1e-6, atol=1e-9) and torch.allclose(q2, 1.0 + q1, rtol=1e-6, atol=1e-7)
    assert lib.xmh_pair_similarity_loss_grad(ptr(a), ptr(b), B, 20000, ptr(lab), C, 0, 6.0, 0.1, 1.0, None, ptr(g1), 0, current_stream()) != 0


def test_trainer_compute_loss_is_differentiable():
    """runners/DCMHT/runner.py:97-105, :121-125: compute_loss -> loss.backward()"""
    from xmh.runners.methods import DCMHTTrainer
    t = DCMHTTrainer.__new__(DCMHTTrainer)
    t.model, t.display_step, t.hash_scale = _model(16, "euclidean"), 20, 2
    g = torch.Generator().manual_seed(5)
    img = torch.rand(12, 32, generator=g).cuda().requires_grad_(True)
    txt = torch.rand(12, 32, generator=g).cuda().requires_grad_(True)
    label = (torch.rand(12, 7, generator=g) < 0.3).float()                     # the loader hands labels over on the host (:117)
    loss = t.compute_loss(img_hash=img, txt_hash=txt, label=label, index=None, epoch=0, times=1, global_step=1)
    loss.backward()
    assert img.grad.abs().sum() > 0 and txt.grad.abs().sum() > 0
