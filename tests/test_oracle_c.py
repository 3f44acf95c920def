"""Pin the plain-C integer oracle against the numpy oracle (which is pinned against the reference goldens)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle as co
from oracle import retrieval as orc


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "calc_utils_K*.npz"))))
def test_c_oracle_matches_goldens(path):
    g = np.load(path)
    K = g["qB"].shape[1]
    qb, _ = orc.pack_bits(g["qB"])
    rb, _ = orc.pack_bits(g["rB"])
    ql, rl = orc.pack_labels(g["qL"]), orc.pack_labels(g["rL"])
    assert np.array_equal(co.hamming(qb, rb).astype(np.int16), g["dist"])
    dist, rel = orc.hamming_packed(qb, rb), orc.relevance_packed(ql, rl)
    ha, hr = co.hist(qb, ql, rb, rl, K + 1)
    wa, wr = orc.bucket_histograms(dist, rel, K + 1)
    assert np.array_equal(ha, wa) and np.array_equal(hr, wr)
    for k, tag in ((None, "all"), (1, "1"), (2, "2"), (50, "50"), (5000, "5000")):
        s, cap = co.ap(qb, ql, rb, rl, K + 1, k)
        assert np.allclose(s, orc.ap_from_ranking(dist, rel, k=k), rtol=1e-12)
        assert abs(float(np.mean(s / cap)) - float(g["map_stable_" + tag])) < 1e-6


def test_c_oracle_topk_is_stable_sort_prefix():
    rng = np.random.default_rng(5)
    qb = rng.integers(0, 2**32, size=(9, 2), dtype=np.uint32)
    rb = rng.integers(0, 2**32, size=(800, 2), dtype=np.uint32)
    d, i = co.topk(qb, rb, 65, 37, base_index=1000)
    full = orc.hamming_packed(qb, rb)
    order = np.argsort(full, axis=1, kind="stable")[:, :37]
    assert np.array_equal(i, order + 1000)
    assert np.array_equal(d, np.take_along_axis(full, order, 1))
    d2, i2 = co.topk(qb, rb[:5], 65, 8)
    assert (i2[:, 5:] == -1).all() and (d2[:, 5:] == 0xFFFF).all()
