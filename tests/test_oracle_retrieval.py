"""Pin the retrieval oracle against golden vectors generated from the reference (CPU only)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import retrieval as orc


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_kat1_k_semantics():
    g = _load("calc_utils_kat.npz")
    q, r = torch.from_numpy(g["kat1_q"]), torch.from_numpy(g["kat1_r"])
    qL, rL = torch.from_numpy(g["kat1_qL"]), torch.from_numpy(g["kat1_rL"])
    for k, key in ((None, "kat1_map_all"), (1, "kat1_map_1"), (2, "kat1_map_2")):
        assert abs(float(orc.map_k(q, r, qL, rL, k)) - float(g[key])) < 1e-7
        assert abs(orc.map_k_ranked(q, r, qL, rL, k) - float(g[key])) < 1e-6
    assert abs(float(g["kat1_map_1"]) - 0.75) < 1e-7 and abs(float(g["kat1_map_2"]) - 2 / 3) < 1e-6


def test_kat2_nan_when_no_relevant():
    g = _load("calc_utils_kat.npz")
    q, r = torch.from_numpy(g["kat1_q"]), torch.from_numpy(g["kat1_r"])
    qL, rL = torch.from_numpy(g["kat1_qL"]), torch.from_numpy(g["kat2_rL"])
    assert np.isnan(g["kat2_map"])
    assert torch.isnan(orc.map_k(q, r, qL, rL))
    assert np.isnan(orc.map_k_ranked(q, r, qL, rL))


def test_kat3_single_query_raises():
    g = _load("calc_utils_kat.npz")
    q, r = torch.from_numpy(g["kat1_q"])[:1], torch.from_numpy(g["kat1_r"])
    with pytest.raises(IndexError):
        orc.map_k(q, r, torch.from_numpy(g["kat1_qL"])[:1], torch.from_numpy(g["kat1_rL"]))


def test_kat4_nonbinary_distance():
    g = _load("calc_utils_kat.npz")
    a, b = torch.from_numpy(g["kat4_a"]), torch.from_numpy(g["kat4_b"])
    assert torch.equal(orc.hamming_dist(a, b), torch.from_numpy(g["kat4_dist"]))
    assert torch.equal(orc.hamming_dist(a[0], b), torch.from_numpy(g["kat4_dist_1d"]))
    assert float(g["kat4_dist"].reshape(-1)[0]) == 1.0


def test_make_hash_code_variants():
    g = _load("make_hash_code.npz")
    assert np.array_equal(orc.hash_code_sign(torch.from_numpy(g["sign_in"]).clone()).numpy(), g["sign_out"])
    assert np.array_equal(orc.hash_code_pair_argmax(torch.from_numpy(g["pair_in"]).clone()).numpy(), g["pair_out"])
    assert g["pair_out"][1, 4] == -1.0            # KAT-5: exact tie -> -1
    assert (g["sign_out"][5] == 0).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "calc_utils_K*.npz"))))
def test_seeded_cases(path):
    g = np.load(path)
    qB, rB = torch.from_numpy(g["qB"]).float(), torch.from_numpy(g["rB"]).float()
    qL, rL = torch.from_numpy(g["qL"]).long(), torch.from_numpy(g["rL"]).long()
    assert np.array_equal(orc.hamming_dist(qB, rB).numpy().astype(np.int16), g["dist"])
    assert np.array_equal(orc.label_sim(qL.float(), rL.float()).numpy().astype(np.int8), g["label_sim"])
    qb, qz = orc.pack_bits(g["qB"])
    rb, rz = orc.pack_bits(g["rB"])
    assert not qz.any() and not rz.any()
    assert np.array_equal(orc.hamming_packed(qb, rb).astype(np.int16), g["dist"])
    assert np.array_equal(orc.relevance_packed(orc.pack_labels(g["qL"]), orc.pack_labels(g["rL"])).astype(np.int8),
                          g["label_sim"])
    for k, tag in ((None, "all"), (1, "1"), (2, "2"), (50, "50"), (5000, "5000")):
        want = float(g["map_stable_" + tag])
        assert abs(float(orc.map_k(qB, rB, qL, rL, k, stable=True)) - want) < 1e-6
        assert abs(orc.map_k_ranked(qB, rB, qL, rL, k) - want) < 2e-6
    # the reference's own (unstable-sort) value differs only by tie order: recorded, loosely bounded
    assert abs(float(g["map_default_all"]) - float(g["map_stable_all"])) < 5e-2


def test_ternary_and_float_similarities():
    g = _load("calc_utils_ternary_float.npz")
    qB, rB = torch.from_numpy(g["qB"]).float(), torch.from_numpy(g["rB"]).float()
    qL, rL = torch.from_numpy(g["qL"]).long(), torch.from_numpy(g["rL"]).long()
    assert np.array_equal(orc.hamming_dist(qB, rB).numpy(), g["dist"])
    qb, qz = orc.pack_bits(g["qB"])
    rb, rz = orc.pack_bits(g["rB"])
    assert np.array_equal(orc.hamming2_ternary(qb, qz, rb, rz, 32), (2 * g["dist"]).astype(np.int32))
    assert abs(float(orc.map_k(qB, rB, qL, rL)) - float(g["map_stable_all"])) < 1e-6
    assert abs(float(orc.map_k(qB, rB, qL, rL, 50)) - float(g["map_stable_50"])) < 1e-6
    fa, fb = torch.from_numpy(g["fa"]), torch.from_numpy(g["fb"])
    assert np.allclose(orc.cosine_sim(fa, fb).numpy(), g["cos"], atol=1e-6)
    assert np.allclose(orc.cosine_sim(g["fa"], g["fb"]), g["cos_np"], atol=1e-6)
    assert np.allclose(orc.euclid_sim(fa, fb).numpy(), g["euc"], atol=1e-5)
    assert np.allclose(orc.euclid_sim(g["fa"], g["fb"]), g["euc_np"], atol=1e-4)
    with pytest.raises(ValueError):
        orc.cosine_sim(fa, g["fb"])
    fq, fr = torch.from_numpy(g["fq"]), torch.from_numpy(g["fr"])
    assert np.allclose(orc.hamming_dist(fq, fr).numpy(), g["float_dist"], atol=1e-5)
    got = float(orc.map_k(fq, fr, torch.from_numpy(g["fqL"]).long(), torch.from_numpy(g["frL"]).long()))
    assert abs(got - float(g["float_map_stable"])) < 1e-6


def test_oracle_reproduces_the_reference_runner_log_on_its_own_codes():
    """runner.npz holds the code buffers and the logged mAPs of the REFERENCE's DCMHTTrainer / MITHTrainer / TwDHTrainer.valid()
    (oracle/make_golden_runner.py).  The oracle's port with the reference's default sort reproduces every logged value on those
    codes, and the canonical-order value the HIP path is held to lies in the same tie-order envelope."""
    import sys
    from oracle import retrieval as orc
    from oracle import runner_fixture as RF
    g = np.load(os.path.join(GOLDEN, "runner.npz"))
    q, r = RF.datasets()
    qL, rL = q.get_all_label(), r.get_all_label()
    pairs = [("q_img", "r_txt"), ("q_txt", "r_img"), ("q_txt", "r_txt"), ("q_img", "r_img")]          # i2t, t2i, t2t, i2i (log order)
    streams = ["DCMHT", "MITH", "TwDH_long"] + ["TwDH_%d" % int(s) for s in g["TwDH_short_dims"]]
    for st in streams:
        logged = g[st + "_maps_i2t_t2i_t2t_i2i"]
        for ref, (a, b) in zip(logged, pairs):
            qc, rc = torch.from_numpy(g["%s_%s" % (st, a)]), torch.from_numpy(g["%s_%s" % (st, b)])
            lo, hi = orc.map_k_tie_bounds(qc, rc, qL, rL)
            stable = float(orc.map_k(qc, rc, qL, rL, None, stable=True))
            assert lo - 1e-6 <= ref <= hi + 1e-6 and lo - 1e-6 <= stable <= hi + 1e-6, (st, a, b)
            if sys.version_info[:2] == (3, 10) and torch.__version__.startswith("2.10"):     # tie order of the default sort is a torch detail
                assert abs(float(orc.map_k(qc, rc, qL, rL, None, stable=False)) - ref) < 1e-6, (st, a, b)
