"""Generate tests/golden/calc_utils_*.npz by RUNNING the reference's own functions.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference).
Inputs come from seeded generators, outputs from the unmodified reference
(``common/calc_utils.py`` and the two ``make_hash_code`` classmethods).  The
stable-order mAP values are produced by the reference's own ``calc_map_k`` with
``torch.sort`` wrapped to pass ``stable=True`` (reference code unchanged,
SURVEY.md H1); the unpatched values are recorded next to them.

    python oracle/make_golden_retrieval.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def synth_codes(gen, n, K, labels=None, structured=False):
    """i.i.d. +-1 codes, or label-correlated ones so mAP is non-trivial."""
    if not structured:
        return torch.randn(n, K, generator=gen).sign()
    C = labels.shape[1]
    Wm = torch.randn(C, K, generator=gen)
    x = labels.float() @ Wm + 0.8 * torch.randn(n, K, generator=gen)
    s = x.sign()
    s[s == 0] = 1.0
    return s


def synth_labels(gen, n, C, p):
    L = (torch.rand(n, C, generator=gen) > 1 - p).long()
    empty = L.sum(1) == 0
    L[empty, torch.randint(0, C, (int(empty.sum()),), generator=gen)] = 1
    return L


def main():
    _ref_import.setup()
    import common.calc_utils as ref                                   # the reference, unmodified
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)

    real_sort = torch.sort

    def with_stable(fn, *a, **kw):
        def stable_sort(x, dim=-1, descending=False, **k2):
            return real_sort(x, dim=dim, descending=descending, stable=True)
        ref.torch.sort = stable_sort
        try:
            return fn(*a, **kw)
        finally:
            ref.torch.sort = real_sort

    # ---- known-answer tests ---------------------------------
    kat = {}
    q = torch.ones(2, 4)
    r = torch.tensor([[1, 1, 1, 1], [1, 1, 1, -1], [1, 1, -1, -1], [1, -1, -1, -1], [-1, -1, -1, -1]], dtype=torch.float32)
    qL = torch.tensor([[1, 0], [0, 1]])
    rL = torch.tensor([[1, 0], [0, 1], [1, 0], [0, 1], [0, 1]])
    kat["kat1_q"], kat["kat1_r"], kat["kat1_qL"], kat["kat1_rL"] = q.numpy(), r.numpy(), qL.numpy(), rL.numpy()
    kat["kat1_map_all"] = ref.calc_map_k(q, r, qL, rL).numpy()
    kat["kat1_map_1"] = ref.calc_map_k(q, r, qL, rL, 1).numpy()
    kat["kat1_map_2"] = ref.calc_map_k(q, r, qL, rL, 2).numpy()
    rL2 = rL.clone()
    rL2[:, 0] = 0                                                         # query 0 has no relevant item
    kat["kat2_rL"] = rL2.numpy()
    kat["kat2_map"] = ref.calc_map_k(q, r, qL, rL2).numpy()              # NaN
    a = torch.tensor([[1.0, 0.0, 1.0, 1.0]])
    b = torch.tensor([[2.0, 1.0, 1.0, -1.0]])
    kat["kat4_a"], kat["kat4_b"] = a.numpy(), b.numpy()
    kat["kat4_dist"] = ref.calc_hammingDist(a, b).numpy()
    kat["kat4_dist_1d"] = ref.calc_hammingDist(a[0], b).numpy()
    np.savez(os.path.join(OUT, "calc_utils_kat.npz"), **kat)

    # ---- make_hash_code, both variants (runners/base.py:407-410, runners/DCMHT/runner.py:82-95)
    from runners.base import BaseTrainer
    from runners.DCMHT.runner import DCMHTTrainer
    g = torch.Generator().manual_seed(1814)
    x = torch.randn(6, 16, generator=g)
    x[0, 3] = 0.0
    x[2, 7] = -0.0
    x[5, :] = 0.0
    p = torch.softmax(torch.relu(torch.randn(6, 16, 2, generator=g)), dim=-1)
    p[1, 4] = 0.5                                                          # exact tie -> -1 (KAT-5)
    p = p.reshape(6, 32)
    np.savez(os.path.join(OUT, "make_hash_code.npz"),
             sign_in=x.numpy(), sign_out=BaseTrainer.make_hash_code(x.clone()).numpy(),
             pair_in=p.numpy(), pair_out=DCMHTTrainer.make_hash_code(p.clone()).numpy())

    # ---- seeded retrieval cases -------------------------------------------------
    for K in (16, 64, 128, 256):
        for structured in (False, True):
            g = torch.Generator().manual_seed(1814 + K + (7 if structured else 0))
            Q, R, C = 24, 700, 24 if K != 64 else 80
            qL, rL = synth_labels(g, Q, C, 0.1), synth_labels(g, R, C, 0.1)
            qB = synth_codes(g, Q, K, qL, structured)
            rB = synth_codes(g, R, K, rL, structured)
            rec = dict(qB=qB.numpy().astype(np.int8), rB=rB.numpy().astype(np.int8),
                       qL=qL.numpy().astype(np.int8), rL=rL.numpy().astype(np.int8),
                       dist=ref.calc_hammingDist(qB, rB).numpy().astype(np.int16),
                       label_sim=ref.calc_label_sim(qL.float(), rL.float()).numpy().astype(np.int8))
            for k in (None, 1, 2, 50, 5000):
                tag = "all" if k is None else str(k)
                rec["map_default_" + tag] = ref.calc_map_k(qB, rB, qL, rL, k).numpy()
                rec["map_stable_" + tag] = with_stable(ref.calc_map_k, qB, rB, qL, rL, k).numpy()
            name = "calc_utils_K%d_%s.npz" % (K, "struct" if structured else "iid")
            np.savez_compressed(os.path.join(OUT, name), **rec)
            print(name, {k: float(v) for k, v in rec.items() if k.startswith("map_stable")})

    # ---- ternary codes (sign(0) = 0) and float similarities --------------------
    g = torch.Generator().manual_seed(99)
    Q, R, K, C = 16, 300, 32, 21
    qL, rL = synth_labels(g, Q, C, 0.1), synth_labels(g, R, C, 0.1)
    qB = torch.randint(-1, 2, (Q, K), generator=g).float()
    rB = torch.randint(-1, 2, (R, K), generator=g).float()
    fa = torch.randn(12, 32, generator=g)
    fb = torch.randn(9, 32, generator=g)
    fq, fr = torch.tanh(torch.randn(12, 32, generator=g)), torch.tanh(torch.randn(40, 32, generator=g))
    fqL, frL = synth_labels(g, 12, 5, 0.3), synth_labels(g, 40, 5, 0.3)
    np.savez_compressed(os.path.join(OUT, "calc_utils_ternary_float.npz"),
                        qB=qB.numpy().astype(np.int8), rB=rB.numpy().astype(np.int8),
                        qL=qL.numpy().astype(np.int8), rL=rL.numpy().astype(np.int8),
                        dist=ref.calc_hammingDist(qB, rB).numpy(),
                        map_stable_all=with_stable(ref.calc_map_k, qB, rB, qL, rL).numpy(),
                        map_stable_50=with_stable(ref.calc_map_k, qB, rB, qL, rL, 50).numpy(),
                        fa=fa.numpy(), fb=fb.numpy(),
                        cos=ref.cosine_similarity(fa, fb).numpy(),
                        cos_np=ref.cosine_similarity(fa.numpy(), fb.numpy()),
                        euc=ref.euclidean_similarity(fa, fb).numpy(),
                        euc_np=ref.euclidean_similarity(fa.numpy(), fb.numpy()),
                        fq=fq.numpy(), fr=fr.numpy(), fqL=fqL.numpy().astype(np.int8), frL=frL.numpy().astype(np.int8),
                        float_dist=ref.calc_hammingDist(fq, fr).numpy(),
                        float_map_stable=with_stable(ref.calc_map_k, fq, fr, fqL, frL).numpy())
    print("done ->", OUT)


if __name__ == "__main__":
    main()
