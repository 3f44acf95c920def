"""CPU: the oracle's restatement of the DCMHT loss forward against goldens produced by the reference's own our_loss
(oracle/make_golden_loss.py)."""
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
