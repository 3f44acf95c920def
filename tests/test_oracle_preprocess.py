"""CPU: the preprocessing oracle (restated Pillow resample + ToTensor/Normalize) against golden vectors produced by
Pillow itself (oracle/make_golden_preprocess.py), and the product's host-side coefficient tables against the oracle's."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "preprocess.npz"))


def test_oracle_resize_is_pillow_bit_exact(gold):
    from oracle import preprocess as O
    n = 0
    while "img%d" % n in gold:
        assert np.array_equal(O.resize_bicubic_u8(gold["img%d" % n], 224, 224), gold["resized%d" % n]), n
        n += 1
    assert n >= 5


def test_oracle_float_stage_matches_torch_formula(gold):
    from oracle import preprocess as O
    assert np.array_equal(O.eval_transform(gold["img0"]), gold["tensor0"])
    assert np.array_equal(O.to_tensor_normalize(gold["resized0"]), gold["tensor0"])


@pytest.mark.parametrize("size", [1, 2, 45, 100, 223, 224, 225, 301, 500, 640, 1000, 4000])
def test_product_tables_equal_oracle_tables(size):
    from oracle import preprocess as O
    from xmh.dataset import preprocess as P
    b1, k1 = O.resample_coeffs(size, 224)
    b2, k2 = P.resample_tables(size, 224)
    assert np.array_equal(b1, b2) and np.array_equal(k1, k2)
    assert (b2[:, 0] >= 0).all() and (b2[:, 0] + b2[:, 1] <= size).all() and (b2[:, 1] >= 1).all()
    # the taps of one output pixel sum to 1.0 in 22-bit fixed point up to rounding of each tap
    assert np.abs(k2.sum(1) - (1 << 22)).max() <= k2.shape[1]
