"""CPU: the oracle's restatement of the DCMHT loss, and of its gradient with respect to the codes, against goldens produced by
the reference's own our_loss and loss.backward() (oracle/make_golden_loss.py)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import losses as OL

CASES = ["b40_k16_euclid", "b40_k64_cos", "b96_k64_euclid", "b17_k32_euclid_nolabels"]
ORDER = ["loss", "intra_pos", "intra_neg", "inter_pos_i", "inter_neg_i", "inter_pos_t", "inter_neg_t", "quan_i", "quan_t"]


def load(name):
    g = np.load(os.path.join(GOLDEN, "loss_dcmht.npz"))
    img, txt = torch.from_numpy(g[name + "_img"]), torch.from_numpy(g[name + "_txt"])
    labels = torch.from_numpy(g[name + "_labels"]) if name + "_labels" in g.files else None
    K, cos, vartheta, threshold, alpha = g[name + "_meta"]
    return img, txt, labels, int(K), "cosine" if cos else "euclidean", float(vartheta), float(threshold), float(alpha), g[name + "_ref"]


def test_oracle_loss_matches_the_reference():
    for name in CASES:
        img, txt, labels, K, sim, vartheta, threshold, alpha, ref = load(name)
        if labels is None:
            labels = torch.eye(img.shape[0])                   # reference object_function :152-154
        got = OL.our_loss(img, txt, labels, K, vartheta, threshold, alpha, sim)
        vec = np.array([float(got[k]) for k in ORDER])
        # fp32 reference (and its cdist may take the matmul route) against a float64 restatement
        assert np.allclose(vec, ref, rtol=2e-5, atol=1e-6), (name, vec, ref)


def load_grads(name):
    g = np.load(os.path.join(GOLDEN, "loss_dcmht.npz"))
    return g[name + "_gimg"], g[name + "_gtxt"]


def grads_close(got, ref):
    """the reference differentiates in fp32 through cdist's matmul route: compare relative to the largest entry of the matrix"""
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max()) <= 2e-5 * float(np.abs(ref).max()) + 1e-9


def test_oracle_gradient_matches_the_reference_backward():
    for name in CASES:
        img, txt, labels, K, sim, vartheta, threshold, alpha, _ = load(name)
        if labels is None:
            labels = torch.eye(img.shape[0])
        gi, gt = OL.our_loss_grad(img, txt, labels, K, vartheta=vartheta, threshold=threshold, quan_alpha=alpha, similarity_function=sim)
        ri, rt = load_grads(name)
        assert grads_close(gi.numpy(), ri) and grads_close(gt.numpy(), rt), (name, np.abs(gi.numpy() - ri).max(), np.abs(ri).max())
