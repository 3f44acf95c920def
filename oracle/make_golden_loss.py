#!/usr/bin/env python3
"""Golden vectors of the DCMHT loss and of its gradient with respect to the codes (loss.backward()), produced by the
UNMODIFIED reference (models/DCMHT/DCMHT.py our_loss) in this container: python oracle/make_golden_loss.py -> tests/golden/loss_dcmht.npz.  Build-container only (/root/reference)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import _ref_import  # noqa: E402

_ref_import.setup()
from models.DCMHT.DCMHT import DCMHT  # noqa: E402  (the reference class)


def ref_model(K, sim, vartheta=0.75, threshold=0.1, quan_alpha=0.001):
    m = DCMHT.__new__(DCMHT)                       # the loss methods only read these attributes; no backbone is built
    torch.nn.Module.__init__(m)
    m.output_dim, m.vartheta, m.threshold, m.similarity_function, m.quan_alpha = K, vartheta, threshold, sim, quan_alpha
    return m


out = {}
cases = [("b40_k16_euclid", 40, 16, 24, "euclidean", True), ("b40_k64_cos", 40, 64, 24, "cosine", True),
         ("b96_k64_euclid", 96, 64, 80, "euclidean", True), ("b17_k32_euclid_nolabels", 17, 32, 0, "euclidean", False)]
for name, B, K, C, sim, has_labels in cases:
    g = torch.Generator().manual_seed(1814 + B + K)
    img = torch.softmax(torch.randn(B, K, 2, generator=g) * 2.0, dim=-1).reshape(B, 2 * K)     # what the softmax hash emits
    txt = torch.softmax(torch.randn(B, K, 2, generator=g) * 2.0, dim=-1).reshape(B, 2 * K)
    labels = None
    if has_labels:
        labels = (torch.rand(B, C, generator=g) < 0.1).float()
        labels[torch.arange(B), torch.randint(0, C, (B,), generator=g)] = 1.0
    m = ref_model(K, sim)
    img.requires_grad_(True)
    txt.requires_grad_(True)
    loss, d = m.object_function(img, txt, labels=labels)
    loss.backward()                                 # runners/DCMHT/runner.py:124
    out[name + "_gimg"], out[name + "_gtxt"] = img.grad.numpy().copy(), txt.grad.numpy().copy()
    img, txt = img.detach(), txt.detach()
    out[name + "_img"], out[name + "_txt"] = img.numpy(), txt.numpy()
    if has_labels:
        out[name + "_labels"] = labels.numpy()
    out[name + "_meta"] = np.array([K, 1.0 if sim == "cosine" else 0.0, 0.75, 0.1, 0.001])
    out[name + "_ref"] = np.array([float(loss), float(d["Intra"]["Positive"]), float(d["Intra"]["Negative"]), float(d["Inter"]["Positive"]["i2t"]),
                                   float(d["Inter"]["Negative"]["i2t"]), float(d["Inter"]["Positive"]["t2i"]), float(d["Inter"]["Negative"]["t2i"]),
                                   float(d["Quan"]["Image"]), float(d["Quan"]["Text"])], dtype=np.float64)
    print(name, out[name + "_ref"])
np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", "loss_dcmht.npz"), **out)
